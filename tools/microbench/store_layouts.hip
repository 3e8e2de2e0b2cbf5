// Does the LAYOUT of the Jacobian decide how sensitive its store stream is to the physical placement of the buffer?
// For each of several allocations of the same size (held at once, so they are different pages):
//   A  parameter-major  J[a][ldj]            : a wave's pass = 32 stores of 512 B, columns ldj*8 bytes (80 MB) apart
//   D  block-major      J[pass][a][512]      : the same 32 stores of 512 B land inside one 128 KiB window per workgroup pass
// Same work split as gfh_k_sweep_gram: workgroups of 8 waves on contiguous slot ranges, non-temporal stores.
// build: hipcc -O3 --offload-arch=gfx950 store_layouts.hip -o store_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef long long i64;
constexpr int NA = 32;

__global__ __launch_bounds__(512) void k_param_major(double* __restrict__ J, i64 ldj, i64 per, i64 n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = (i64)blockIdx.x * per, e = s0 + per < n ? s0 + per : n;
  for (i64 iw = s0 + 64 * wv; iw < e; iw += 512) {
    const double v = (double)(iw + lane);
#pragma unroll
    for (int a = 0; a < NA; a++) __builtin_nontemporal_store(v + a, J + (i64)a * ldj + iw + lane);
  }
}
__global__ __launch_bounds__(512) void k_block_major(double* __restrict__ J, i64 per, i64 n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = (i64)blockIdx.x * per, e = s0 + per < n ? s0 + per : n;
  for (i64 iw = s0; iw < e; iw += 512) {                      // iw: first slot of the workgroup's pass
    double* __restrict__ B = J + (iw >> 9) * (i64)(NA * 512) + 64 * wv + lane;
    const double v = (double)(iw + lane);
#pragma unroll
    for (int a = 0; a < NA; a++) __builtin_nontemporal_store(v + a, B + a * 512);
  }
}
template <class F> static float timeit(F launch, int warm, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < warm; i++) launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main(int argc, char** argv) {
  const i64 n = 10000384;
  const int nbuf = argc > 1 ? atoi(argv[1]) : 8;
  const int nwg = 512;
  i64 per = (n + nwg - 1) / nwg; per = (per + 511) / 512 * 512;
  const int grid = (int)((n + per - 1) / per);
  std::vector<double*> bufs;
  // some ballast allocations of odd sizes in between, freed at once, so that the candidates come from a used heap
  std::vector<void*> ballast;
  const bool contiguous = argc > 2 && atoi(argv[2]) != 0;      // second argument 1: hipExtMallocWithFlags(hipDeviceMallocContiguous)
  for (int k = 0; k < nbuf; k++) {
    double* J;
    if (contiguous) { if (hipExtMallocWithFlags((void**)&J, sizeof(double) * NA * n, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous allocation %d failed\n", k); (void)hipGetLastError(); break; } }
    else if (hipMalloc(&J, sizeof(double) * NA * n) != hipSuccess) break;
    bufs.push_back(J);
    void* b; if (hipMalloc(&b, (size_t)(37 + 11 * k) << 20) == hipSuccess) ballast.push_back(b);
  }
  for (void* b : ballast) hipFree(b);
  if (bufs.empty()) return 1;
  timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, bufs[0], n, per, n); }, 40, 1);
  for (int round = 0; round < 1; round++)
    for (size_t k = 0; k < bufs.size(); k++) {
      float a = timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, bufs[k], n, per, n); }, 4, 20);
      float d = timeit([&] { hipLaunchKernelGGL(k_block_major, dim3(grid), dim3(512), 0, 0, bufs[k], per, n); }, 4, 20);
      printf("buffer %zu (%p): parameter-major %.3f ms (%.0f GB/s)   block-major %.3f ms (%.0f GB/s)\n", k, (void*)bufs[k], a, 2.56 * n / 1e7 / a * 1e3,
             d, 2.56 * n / 1e7 / d * 1e3);
    }
  return 0;
}
