// Is "fast memory" a property of the individual physical chunk?  N chunks of 64 MB (hipMemCreate), each mapped alone and written by
// the whole card with a plain non-temporal store stream; then the same chunks in groups of 8 written concurrently as 8 streams.
// build: hipcc -O3 --offload-arch=gfx950 chunk_speed.hip -o chunk_speed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef long long i64;
__global__ __launch_bounds__(512) void k_fill(double* __restrict__ p, i64 n) {
  for (i64 i = (i64)blockIdx.x * 512 + threadIdx.x; i < n; i += (i64)gridDim.x * 512) __builtin_nontemporal_store((double)i, p + i);
}
template <class F> static float timeit(F launch, int warm, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < warm; i++) launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1); return ms / reps;
}
int main(int argc, char** argv) {
  const int nchunk = argc > 1 ? atoi(argv[1]) : 64;
  const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<double*> va(nchunk, nullptr);
  double* warm; hipMalloc(&warm, chunk);
  timeit([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(512), 0, 0, warm, (i64)(chunk / 8)); }, 2000, 1);
  std::vector<float> t(nchunk, 0.f);
  for (int k = 0; k < nchunk; k++) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("create %d failed\n", k); return 1; }
    void* p = nullptr; hipMemAddressReserve(&p, chunk, 0, nullptr, 0);
    hipMemMap(p, chunk, 0, h, 0); hipMemSetAccess(p, chunk, &acc, 1);
    va[k] = (double*)p;
    t[k] = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(512), 0, 0, va[k], (i64)(chunk / 8)); }, 5, 40);
  }
  printf("per-chunk fill time (us), %zu MB chunks, in allocation order:\n", chunk >> 20);
  for (int k = 0; k < nchunk; k++) printf("%s%.1f", k % 16 ? " " : "\n  ", t[k] * 1e3);
  std::vector<float> s = t; std::sort(s.begin(), s.end());
  printf("\nmin %.1f  median %.1f  max %.1f us  (%.0f ... %.0f GB/s)\n", s[0] * 1e3, s[nchunk / 2] * 1e3, s.back() * 1e3,
         chunk / 1e9 / s.back() * 1e3, chunk / 1e9 / s[0] * 1e3);
  return 0;
}
