// Where do k_assemble_sparse's ~24 us at config 3 (dim 259, 64 datasets, 1.4 k pattern entries) go?  Times the library's launcher on a
// synthetic block-arrow pattern, the same with the dataset loop cut to one dataset (the entries that walk all datasets then cost one
// term), and an empty kernel of the same grid.  build: hipcc -O3 --offload-arch=gfx950 -I../../gadfit_amd/csrc
//   assemble_sparse_probe.hip ../../gadfit_amd/csrc/kernels.hip -o assemble_sparse_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace gfh;
__global__ void k_empty(double* p, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = 1.0; }
template <class F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 50; i++) f(); hipEventRecord(a); for (int i = 0; i < 200; i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 200 * 1e3f; }
int main() {
  const int nd = 64, nl = 4, ng = 3, na = 7, dim = ng + nl * nd, T = 1, gw = gram_partial_stride(T);
  std::vector<int> inv((size_t)nd * dim, -1), owner(dim, -1);
  for (int d = 0; d < nd; d++) for (int k = 0; k < na; k++) { int col = k < nl ? (d == 0 ? k : na + (d - 1) * nl + k) : k; inv[(size_t)d * dim + col] = k; if (k < nl) owner[col] = d; }
  std::vector<int> nr, nc;
  for (int c = 0; c < dim; c++) for (int r = 0; r <= c; r++) {
    bool touched = false;
    for (int d = 0; d < nd && !touched; d++) touched = inv[(size_t)d * dim + r] >= 0 && inv[(size_t)d * dim + c] >= 0;
    if (touched) { nr.push_back(r); nc.push_back(c); }
  }
  const int nnz = (int)nr.size();
  int *dinv, *down, *dnr, *dnc; double *G, *packed;
  hipMalloc(&dinv, inv.size() * 4); hipMalloc(&down, dim * 4); hipMalloc(&dnr, nnz * 4); hipMalloc(&dnc, nnz * 4);
  hipMalloc(&G, (size_t)nd * gw * 8); hipMalloc(&packed, ((size_t)nnz + dim + 1) * 8);
  hipMemcpy(dinv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(down, owner.data(), dim * 4, hipMemcpyHostToDevice);
  hipMemcpy(dnr, nr.data(), nnz * 4, hipMemcpyHostToDevice); hipMemcpy(dnc, nc.data(), nnz * 4, hipMemcpyHostToDevice);
  hipMemset(G, 0, (size_t)nd * gw * 8);
  printf("nnz = %d\n", nnz);
  printf("k_assemble_sparse (library)            : %.1f us\n", timeit([&] { launch_assemble_sparse(0, G, gw, T, nd, dim, dinv, down, dnr, dnc, nnz, packed); }));
  printf("same, dataset loop cut to one dataset  : %.1f us\n", timeit([&] { launch_assemble_sparse(0, G, gw, T, 1, dim, dinv, down, dnr, dnc, nnz, packed); }));
  printf("store only, same grid                  : %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3((nnz + dim + 1 + 255) / 256), dim3(256), 0, 0, packed, nnz + dim + 1); }));
  return 0;
}
