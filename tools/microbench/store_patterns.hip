// Which store pattern of a 32-column fp64 Jacobian reaches the HBM write rate of an MI355X?
// N points x 32 columns x 8 B = 2.56 GB at N = 1e7, written once, nothing read (pure write stream).
//   A  parameter-major J[a][ldj] (the layout of libgadfit_hip): a wave's pass = 32 stores of 512 B,
//      one per column, columns ldj*8 bytes apart (ldj padded by `pad` doubles)
//   B  point-major J[i][32] (the reference's JacobianT(dim, N)): a wave's pass = 16 KiB contiguous,
//      16 stores of 16 B per lane (1 KiB per wave-instruction)
//   C  parameter-major, 2 points per lane: 32 stores of 16 B per lane = 1 KiB contiguous per column
// Work split like gfh_k_sweep_gram: `nwg` workgroups of 8 waves, each on a contiguous slot range.
// build: hipcc -O3 --offload-arch=gfx950 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef long long i64;
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(512) void k_colmajor(double* __restrict__ J, i64 ldj, i64 per, i64 n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = (i64)blockIdx.x * per, e = s0 + per < n ? s0 + per : n;
  for (i64 iw = s0 + 64 * __builtin_amdgcn_readfirstlane(wv); iw < e; iw += 512) {
    const double v = (double)(iw + lane);
#pragma unroll
    for (int a = 0; a < 32; a++) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(J + (i64)a * ldj + iw, 0, 512, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v + a), rs, lane * 8, 0, AUX);
    }
  }
}

template <int AUX>
__global__ __launch_bounds__(512) void k_colmajor2(double* __restrict__ J, i64 ldj, i64 per, i64 n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = (i64)blockIdx.x * per, e = s0 + per < n ? s0 + per : n;
  for (i64 iw = s0 + 128 * __builtin_amdgcn_readfirstlane(wv); iw < e; iw += 1024) {
    const double v = (double)(iw + lane);
#pragma unroll
    for (int a = 0; a < 32; a++) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(J + (i64)a * ldj + iw, 0, 1024, 0x00020000);
      v4i pk; v2i lo = __builtin_bit_cast(v2i, v + a), hi = __builtin_bit_cast(v2i, v - a);
      pk.x = lo.x; pk.y = lo.y; pk.z = hi.x; pk.w = hi.y;
      __builtin_amdgcn_raw_buffer_store_b128(pk, rs, lane * 16, 0, AUX);
    }
  }
}

template <int AUX>
__global__ __launch_bounds__(512) void k_pointmajor(double* __restrict__ J, i64 per, i64 n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const i64 s0 = (i64)blockIdx.x * per, e = s0 + per < n ? s0 + per : n;
  for (i64 iw = s0 + 64 * __builtin_amdgcn_readfirstlane(wv); iw < e; iw += 512) {
    const double v = (double)(iw + lane);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(J + iw * 32, 0, 16384, 0x00020000);
#pragma unroll
    for (int c = 0; c < 16; c++) {
      v4i pk; v2i lo = __builtin_bit_cast(v2i, v + c), hi = __builtin_bit_cast(v2i, v - c);
      pk.x = lo.x; pk.y = lo.y; pk.z = hi.x; pk.w = hi.y;
      __builtin_amdgcn_raw_buffer_store_b128(pk, rs, c * 1024 + lane * 16, 0, AUX);
    }
  }
}

template <class F> static float timeit(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main(int argc, char** argv) {
  const i64 n = 10000384;                       // 1e7 padded to 1024 slots, as the library does
  const int reps = argc > 1 ? atoi(argv[1]) : 10;   // the first ~40 launches after an idle gap run in the power-management transient
  double* J; const i64 maxld = n + 4096;
  if (hipMalloc(&J, sizeof(double) * 32 * maxld) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(J, 0, sizeof(double) * 32 * maxld);
  const double gb = 32.0 * 8 * n / 1e9;
  for (int nwg : {512, 1024}) {
    i64 per = (n + nwg - 1) / nwg; per = (per + 1023) / 1024 * 1024;
    const int grid = (int)((n + per - 1) / per);
    for (int pad : {0, 32, 64, 96, 160, 544}) {
      const i64 ldj = n + pad;
      float ms = timeit([&] { hipLaunchKernelGGL(k_colmajor<2>, dim3(grid), dim3(512), 0, 0, J, ldj, per, n); }, reps);
      printf("A col-major 8B/lane nt   wg=%4d pad=%3d : %.3f ms  %.0f GB/s\n", grid, pad, ms, gb / ms * 1e3);
    }
    float ms = timeit([&] { hipLaunchKernelGGL(k_colmajor<0>, dim3(grid), dim3(512), 0, 0, J, n, per, n); }, reps);
    printf("A col-major 8B/lane      wg=%4d pad=  0 : %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(k_colmajor2<2>, dim3(grid), dim3(512), 0, 0, J, n, per, n); }, reps);
    printf("C col-major 16B/lane nt  wg=%4d pad=  0 : %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(k_colmajor2<2>, dim3(grid), dim3(512), 0, 0, J, n + 96, per, n); }, reps);
    printf("C col-major 16B/lane nt  wg=%4d pad= 96 : %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(k_pointmajor<2>, dim3(grid), dim3(512), 0, 0, J, per, n); }, reps);
    printf("B point-major 16B/lane nt wg=%4d        : %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(k_pointmajor<0>, dim3(grid), dim3(512), 0, 0, J, per, n); }, reps);
    printf("B point-major 16B/lane    wg=%4d        : %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
  }
  hipFree(J);
  return 0;
}
