// On a buffer where the parameter-major store stream is slow (physically contiguous allocation: 0.48-0.49 ms), does it help to let
// the workgroups run out of step -- workgroup b starting at pass (b * K) mod passes of its own range and wrapping around -- or to
// change the workgroups' spacing (slots per workgroup)?
// build: hipcc -O3 --offload-arch=gfx950 store_stagger.hip -o store_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef long long i64;
constexpr int NA = 32;
__global__ __launch_bounds__(512) void k_stag(double* __restrict__ J, i64 ldj, i64 per, i64 n, int K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = (i64)blockIdx.x * per, e = s0 + per < n ? s0 + per : n;
  const int np = (int)((e - s0) / 512);
  int k = np ? (int)(((i64)blockIdx.x * K) % np) : 0;
  for (int c = 0; c < np; c++) {
    const i64 iw = s0 + (i64)k * 512 + 64 * wv;
    const double v = (double)(iw + lane);
#pragma unroll
    for (int a = 0; a < NA; a++) __builtin_nontemporal_store(v + a, J + (i64)a * ldj + iw + lane);
    k = k + 1 == np ? 0 : k + 1;
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
int main() {
  const i64 n = 10000384;
  const size_t bytes = sizeof(double) * NA * (n + 4096);
  for (int kind = 0; kind < 2; kind++) {
    double* J = nullptr;
    hipError_t e = kind ? hipMalloc(&J, bytes) : hipExtMallocWithFlags((void**)&J, bytes, hipDeviceMallocContiguous);
    if (e != hipSuccess) { printf("allocation failed\n"); continue; }
    timeit([&] { hipLaunchKernelGGL(k_stag, dim3(512), dim3(512), 0, 0, J, n, (i64)19968, n, 0); }, 40, 1);
    printf("%s:\n", kind ? "default allocation" : "contiguous allocation");
    for (int nwg : {256, 384, 501, 512, 640, 768, 1002, 1024, 2048}) {
      i64 per = (n + nwg - 1) / nwg; per = (per + 511) / 512 * 512;
      const int grid = (int)((n + per - 1) / per);
      printf("  %4d workgroups (%6lld slots each):", grid, per);
      for (int K : {0, 1, 3, 7, 13}) printf("  K=%d %.3f", K, timeit([&] { hipLaunchKernelGGL(k_stag, dim3(grid), dim3(512), 0, 0, J, n, per, n, K); }, 2, 8));
      printf("\n");
    }
  }
  return 0;
}
