// v_mfma_f64_4x4x4_4b_f64 on gfx950: what it costs next to v_mfma_f64_16x16x4_f64, and where its operands live.
//   (1) shader cycles per instruction per wave (s_memtime around an unrolled loop of independent accumulators,
//       one and two waves per SIMD), for both instructions;
//   (2) the lane -> (block, row/column, k) map of A, B and D, found by one-hot probing.
// Question behind it: a symmetric 16x16 diagonal tile of J^T J has 10 useful 4x4 blocks of 16; three 4-block
// instructions could replace one 16x16x4 if they cost a quarter of it.
// build: hipcc -O3 --offload-arch=gfx950 mfma_4x4.hip -o mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(double* out, long long* cyc, int iters, double seed) {
  const double fa = seed * 1e-3 + (threadIdx.x & 15), fb = seed * 1e-3 + (threadIdx.x >> 4);
  d4 A0 = {0, 0, 0, 0}, A1 = A0, A2 = A0, A3 = A0;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {
      A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, A0, 0, 0, 0);
      A1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, A1, 0, 0, 0);
      A2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, A2, 0, 0, 0);
      A3 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, A3, 0, 0, 0);
    } else {
      s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s1, 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s2, 0, 0, 0);
      s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s3, 0, 0, 0);
      s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s4, 0, 0, 0);
      s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s5, 0, 0, 0);
      s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s6, 0, 0, 0);
      s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb, s7, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = A0[0] + A1[1] + A2[2] + A3[3] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void k_probe(const double* a, const double* b, double* d) {
  d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}

int main() {
  double* out; long long* cyc; hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int kind = 0; kind < 2; kind++)
    for (int threads : {256, 512}) {
      long long c = 0;
      for (int rep = 0; rep < 2; rep++) {
        if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.5);
        else hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.5);
        hipDeviceSynchronize();
      }
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const int per_iter = kind == 0 ? 4 : 8, waves = threads / 256;
      // s_memtime counts at 100 MHz on this part: report the ratio between the two instructions, and wall time
      printf("%-26s %d wave(s)/SIMD : %8.3f memtime ticks per instruction per wave\n", kind == 0 ? "v_mfma_f64_16x16x4_f64" : "v_mfma_f64_4x4x4_4b_f64",
             waves, (double)c / iters / per_iter / waves);
    }
  // wall-clock version (hip events) of the same loops
  for (int kind = 0; kind < 2; kind++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.5);
    else hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-26s : %.3f ms for %d instructions per wave = %.1f ns each\n", kind == 0 ? "v_mfma_f64_16x16x4_f64" : "v_mfma_f64_4x4x4_4b_f64", ms,
           iters * (kind == 0 ? 4 : 8), ms * 1e6 / (iters * (kind == 0 ? 4 : 8)));
  }
  // layout: A one-hot at lane la, B one-hot at lane lb -> which D lanes light up
  double *a, *b, *d; hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
  std::vector<double> ha(64), hb(64), hd(64);
  printf("layout: for A one-hot at lane la (B all ones): D lanes that are non-zero\n");
  for (int la = 0; la < 64; la++) {
    for (int l = 0; l < 64; l++) { ha[l] = l == la; hb[l] = 1.0; }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
    printf("  A lane %2d ->", la);
    for (int l = 0; l < 64; l++) if (hd[l] != 0) printf(" %d", l);
    printf("\n");
  }
  printf("layout: for B one-hot at lane lb (A all ones): D lanes that are non-zero\n");
  for (int lb = 0; lb < 64; lb++) {
    for (int l = 0; l < 64; l++) { hb[l] = l == lb; ha[l] = 1.0; }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
    printf("  B lane %2d ->", lb);
    for (int l = 0; l < 64; l++) if (hd[l] != 0) printf(" %d", l);
    printf("\n");
  }
  // which (A lane, B lane) pairs meet (same k, same block): A one-hot la, B one-hot lb, any D non-zero
  printf("pairs: A lane la meets B lanes (product lands in D lane)\n");
  for (int la = 0; la < 64; la += 1) {
    printf("  A %2d:", la);
    for (int lb = 0; lb < 64; lb++) {
      for (int l = 0; l < 64; l++) { ha[l] = l == la; hb[l] = l == lb; }
      hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a, b, d);
      hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; l++) if (hd[l] != 0) printf(" B%d>D%d", lb, l);
    }
    printf("\n");
  }
  return 0;
}
