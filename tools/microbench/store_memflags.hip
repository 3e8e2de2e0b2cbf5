// The parameter-major store stream into buffers allocated with hipExtMallocWithFlags: default, fine-grained, uncached, contiguous.
// build: hipcc -O3 --offload-arch=gfx950 store_memflags.hip -o store_memflags
#include <hip/hip_runtime.h>
#include <cstdio>
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
  i64 per = (n + 511) / 512; per = (per + 511) / 512 * 512;
  const int grid = (int)((n + per - 1) / per);
  const size_t bytes = sizeof(double) * NA * n;
  double* w; hipMalloc(&w, bytes);
  timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, w, n, per, n); }, 40, 1);
  const unsigned flags[] = {hipDeviceMallocDefault, hipDeviceMallocFinegrained, hipDeviceMallocUncached, hipDeviceMallocContiguous};
  const char* names[] = {"default", "fine-grained", "uncached", "contiguous"};
  for (int rep = 0; rep < 3; rep++)
    for (int f = 0; f < 4; f++) {
      double* J = nullptr;
      if (hipExtMallocWithFlags((void**)&J, bytes, flags[f]) != hipSuccess) { printf("%s: allocation failed\n", names[f]); (void)hipGetLastError(); continue; }
      printf("%-12s: %.3f ms\n", names[f], timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, J, n, per, n); }, 3, 10));
    }
  return 0;
}
