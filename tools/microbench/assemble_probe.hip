// Why does k_assemble take ~29 us for dim = 259, 64 datasets?  Times the library's launcher on synthetic tables
// and three ablations of the same access pattern.  build: hipcc -O3 --offload-arch=gfx950 -I../../gadfit_amd/csrc
//   assemble_probe.hip ../../gadfit_amd/csrc/kernels.hip -o assemble_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace gfh;
__global__ void k_store_only(double* p, int dim) { const int row = blockIdx.x * 256 + threadIdx.x; if (row < dim) p[(i64)blockIdx.y * dim + row] = 1.0; }
__global__ void k_load_inv(const int* inv, const int* owner, double* p, int dim) {
  const int row = blockIdx.x * 256 + threadIdx.x; const int col = blockIdx.y;
  if (row < dim && col < dim) { const int d = owner[row] >= 0 ? owner[row] : 0; p[(i64)col * dim + row] = inv[d * dim + row] + inv[d * dim + col]; }
}
template <class F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 50; i++) f(); hipEventRecord(a); for (int i = 0; i < 200; i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 200 * 1e3f; }
int main() {
  const int nd = 64, nl = 4, ng = 3, na = 7, dim = ng + nl * nd, T = 1, gw = gram_partial_stride(T);
  std::vector<int> inv((size_t)nd * dim, -1), owner(dim, -1);
  // reference-style map: dataset 0 holds columns 0..6 (locals 0-3, globals 4-6), dataset d>0 its locals after them
  for (int d = 0; d < nd; d++) for (int k = 0; k < na; k++) { int col = k < nl ? (d == 0 ? k : na + (d - 1) * nl + k) : k; inv[(size_t)d * dim + col] = k; if (k < nl) owner[col] = d; }
  int *dinv, *down; double *G, *packed;
  hipMalloc(&dinv, inv.size() * 4); hipMalloc(&down, dim * 4); hipMalloc(&G, (size_t)nd * gw * 8); hipMalloc(&packed, ((size_t)dim * dim + dim + 1) * 8);
  hipMemcpy(dinv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(down, owner.data(), dim * 4, hipMemcpyHostToDevice);
  hipMemset(G, 0, (size_t)nd * gw * 8);
  printf("k_assemble (library)      : %.1f us\n", timeit([&] { launch_assemble(0, G, gw, T, nd, dim, dinv, down, packed); }));
  dim3 grid((dim + 256) / 256, dim + 1);
  printf("store only, same grid     : %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_store_only, grid, dim3(256), 0, 0, packed, dim); }));
  printf("inv loads + store         : %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_load_inv, grid, dim3(256), 0, 0, dinv, down, packed, dim); }));
  dim3 grid1((dim * dim + 255) / 256);
  printf("store only, flat 1-D grid : %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_store_only, dim3((dim * dim + 255) / 256, 1), dim3(256), 0, 0, packed, dim * dim); }));
  return 0;
}
