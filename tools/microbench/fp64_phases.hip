// How does the FP64 pipe of a gfx950 SIMD take the fused kernel's two phases -- ~392 FP64 VALU instructions (the AD body), then
// 16 k-steps of (1 v_mfma_f64_16x16x4_f64 + 5 v_mfma_f64_4x4x4_4b_f64) -- from 1, 2, 3 or 4 resident waves?  No LDS, no memory:
// the pipe alone.  Wall clock (HIP events) per pass and SIMD, so that wave priorities cannot flatter one wave.
// mode 0: VALU phase only; 1: matrix phase only; 2: both (every wave alternates); CHAINS = independent dependency chains in the VALU phase.
// build: hipcc -O3 --offload-arch=gfx950 fp64_phases.hip -o fp64_phases
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int CHAINS>
__global__ void k(double* out, int passes, double seed) {
  double a[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x + c;
  const double m = 1.0000001, cc = 1e-9;
  d4 acc = {0, 0, 0, 0};
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  const double fa = seed * 1e-3 + (threadIdx.x & 15), fb = seed * 1e-3 + (threadIdx.x >> 4);
  for (int p = 0; p < passes; p++) {
    if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < 392 / CHAINS; u++)
#pragma unroll
        for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fma(a[c], m, cc);
    }
    if (MODE != 0) {
#pragma unroll
      for (int s = 0; s < 16; s++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fa, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(fb, fb, s2, 0, 0, 0);
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(fb, fa, s3, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s4, 0, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
  }
  double t = acc[0] + acc[1] + acc[2] + acc[3] + s0 + s1 + s2 + s3 + s4;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) t += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE, int CHAINS> void run(double* d, int waves_per_simd, const char* what) {
  const int passes = 2000, threads = 256 * waves_per_simd;     // one workgroup per CU, 4 SIMDs
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, d, passes, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, d, passes, 1.5);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // ns per pass of ONE wave's worth of work on a SIMD (the SIMD does waves_per_simd of them in that time)
  printf("%-34s chains %2d  %d w/SIMD: %8.1f ns per wave-pass and SIMD (%.0f cycles at 2.4 GHz)\n", what, CHAINS, waves_per_simd,
         ms * 1e6 / passes / waves_per_simd, ms * 1e6 / passes / waves_per_simd * 2.4);
}

int main() {
  double* d; hipMalloc(&d, 256 * 1024 * 8);
  for (int w = 1; w <= 4; w++) run<0, 8>(d, w, "VALU phase only (392 FMAs)");
  for (int w = 1; w <= 4; w++) run<0, 2>(d, w, "VALU phase only (392 FMAs)");
  for (int w = 1; w <= 4; w++) run<1, 8>(d, w, "matrix phase only (16 + 80 MFMAs)");
  for (int w = 1; w <= 4; w++) run<2, 8>(d, w, "both phases");
  for (int w = 1; w <= 4; w++) run<2, 2>(d, w, "both phases");
  return 0;
}
