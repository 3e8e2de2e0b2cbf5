// Column stride of the parameter-major Jacobian against the store stream's rate, on a physically contiguous allocation
// (hipDeviceMallocContiguous) and on a default one: is there a stride that makes the 32 concurrent column streams fast
// regardless of the pages behind the buffer?
// build: hipcc -O3 --offload-arch=gfx950 store_stride_scan.hip -o store_stride_scan
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
template <class F> static float timeit(F launch, int warm, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < warm; i++) launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main(int argc, char** argv) {
  const i64 n = 10000384, maxpad = 1 << 20;
  const int nwg = 512;
  i64 per = (n + nwg - 1) / nwg; per = (per + 511) / 512 * 512;
  const int grid = (int)((n + per - 1) / per);
  const size_t bytes = sizeof(double) * NA * (n + maxpad);
  std::vector<i64> pads = {0, 8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 520, 768, 1024, 1032, 1536, 2048, 2056, 3072, 4096, 4104, 6144, 8192, 8200,
                           12288, 16384, 16392, 32768, 65536, 131072, 262144, 262152, 524288, 1048576};
  for (int kind = 0; kind < 2; kind++)
    for (int rep = 0; rep < 2; rep++) {
      double* J = nullptr;
      hipError_t e = kind ? hipExtMallocWithFlags((void**)&J, bytes, hipDeviceMallocContiguous) : hipMalloc(&J, bytes);
      if (e != hipSuccess) { printf("allocation failed\n"); (void)hipGetLastError(); continue; }
      timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, J, n, per, n); }, 30, 1);
      printf("%s allocation %d:", kind ? "contiguous" : "default", rep);
      for (i64 pad : pads) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, J, n + pad, per, n); }, 2, 8);
        printf(" %lld:%.3f", pad, ms);
      }
      printf("\n");
      // keep the buffer (no free) so that the next one comes from other pages
    }
  return 0;
}
