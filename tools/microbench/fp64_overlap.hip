// Do FP64 VALU FMAs and v_mfma_f64_16x16x4_f64 overlap on gfx950, or share one datapath?
// mode 0: VALU only, 1: MFMA only, 2: both in every wave (interleaved), 3: waves 0-3 VALU, waves 4-7 MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed) {
  const int wv = threadIdx.x >> 6;
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const double fa = seed * 1e-3 + (threadIdx.x & 15), fb = seed * 1e-3 + (threadIdx.x >> 4);
  const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && wv < 4);
  const bool do_mfma = MODE == 1 || MODE == 2 || (MODE == 3 && wv >= 4);
  for (int i = 0; i < iters; i++) {
    if (do_mfma) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc3, 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int u = 0; u < 8; u++) {   // 64 FMAs = 256 VALU cycles ~ 4 MFMAs
        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
template <int MODE> float run(double* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1.5);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 256 * 512 * 8);
  const int iters = 20000;
  printf("VALU only      : %.3f ms\n", run<0>(d, iters));
  printf("MFMA only      : %.3f ms\n", run<1>(d, iters));
  printf("both, one wave : %.3f ms\n", run<2>(d, iters));
  printf("split waves    : %.3f ms (4 VALU waves + 4 MFMA waves per CU, half the work of each kind)\n", run<3>(d, iters));
  return 0;
}
