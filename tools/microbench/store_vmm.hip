// The parameter-major store stream into a buffer whose physical memory comes in separately created chunks (HIP virtual memory
// management: hipMemCreate per chunk, mapped back to back into one reserved range), chunk sizes 2 MB ... 1 GB, next to plain hipMalloc.
// build: hipcc -O3 --offload-arch=gfx950 store_vmm.hip -o store_vmm
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
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return nullptr; } } while (0)
static double* vmm_alloc(size_t bytes, size_t chunk) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (chunk < gran) chunk = gran;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void* va = nullptr; CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
  }
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  return (double*)va;
}
int main(int argc, char** argv) {
  const i64 n = 10000384;
  const int nwg = 512;
  i64 per = (n + nwg - 1) / nwg; per = (per + 511) / 512 * 512;
  const int grid = (int)((n + per - 1) / per);
  const size_t bytes = sizeof(double) * NA * n;
  double* warm; hipMalloc(&warm, bytes);
  timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, warm, n, per, n); }, 40, 1);
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  for (int rep = 0; rep < reps; rep++) {
    double* J; hipMalloc(&J, bytes);
    printf("hipMalloc %.3f |", timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, J, n, per, n); }, 3, 10));
    for (size_t mb : {2, 8, 16, 32, 48, 64, 80, 96, 128, 256, 1024}) {
      double* V = vmm_alloc(bytes, mb << 20);
      if (!V) continue;
      printf(" %zuMB %.3f", mb, timeit([&] { hipLaunchKernelGGL(k_param_major, dim3(grid), dim3(512), 0, 0, V, n, per, n); }, 3, 10));
    }
    printf("\n");
  }
  return 0;
}
