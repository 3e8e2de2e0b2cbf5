// Issue cost of the FP64 (and helper 32-bit) VALU instructions the generated model bodies are made of, on gfx950.
// Per instruction: NI independent streams of it, unrolled, one/two/four waves per SIMD on every CU; the
// kernel reads s_memtime around the loop, so the result is shader cycles per wave-instruction per SIMD
// (independent of the clock the chip holds).  Feeds the cost model behind the generated exp() (codegen.cpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { FMA, MUL, ADD, RNDNE, LDEXP, CVTI, CMPF, CNDMASK, MAXF, LSHLADD, CMPU, RCP, SQRT, FRACT, FREXP_EXP, TRIG_PREOP, DIVSCALE, DIVFMAS, DIVFIXUP, MOV64, CND_OOP, CND_E64, CMP_CND, BFI, AND32, MINF, CMPCLASS, CVTFI, ADDU, MOV32, FMA_SGPR, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_ldexp_f64", "v_cvt_i32_f64", "v_cmp_lt_f64 (+ s_nop-free)",
  "v_cndmask_b32", "v_max_f64", "v_lshl_add_u32", "v_cmp_gt_u32", "v_rcp_f64", "v_sqrt_f64", "v_fract_f64", "v_frexp_exp_i32_f64", "v_trig_preop_f64",
  "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_mov_b64",
  "v_cndmask_b32 (dst != src)", "v_cndmask_b32 e64 (sgpr mask)", "v_cmp_lt_f64 + 2 v_cndmask (per pair)", "v_bfi_b32", "v_and_b32", "v_min_f64", "v_cmp_class_f64", "v_cvt_f64_i32", "v_add_u32", "v_mov_b32", "v_fma_f64 (sgpr, literal-free)"};

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, double seed) {
  double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const int one = 1 + (threadIdx.x & 1), j = 3 + (threadIdx.x & 3);
  const unsigned long long msk = (unsigned long long)iters * 0x9E3779B97F4A7C15ull;
  const double ms = __builtin_amdgcn_readfirstlane((int)seed) * 1.0000001;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#define A(n) a##n
#define I(n) i##n
      if (OP == FMA) {
#define X(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(n)) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (OP == MUL) {
#define X(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A(n)) : "v"(m));
        REP8(X)
#undef X
      } else if (OP == ADD) {
#define X(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(A(n)) : "v"(c));
        REP8(X)
#undef X
      } else if (OP == RNDNE) {
#define X(n) asm volatile("v_rndne_f64 %0, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == LDEXP) {
#define X(n) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(A(n)) : "v"(one));
        REP8(X)
#undef X
      } else if (OP == CVTI) {
#define X(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(I(n)) : "v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == CMPF) {
#define X(n) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(A(n)), "v"(m) : "vcc");
        REP8(X)
#undef X
      } else if (OP == CNDMASK) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(I(n)) : "v"(one) : "vcc");
        REP8(X)
#undef X
      } else if (OP == MAXF) {
#define X(n) asm volatile("v_max_f64 %0, %0, %1" : "+v"(A(n)) : "v"(m));
        REP8(X)
#undef X
      } else if (OP == LSHLADD) {
#define X(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(I(n)) : "v"(one));
        REP8(X)
#undef X
      } else if (OP == CMPU) {
#define X(n) asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(I(n)), "v"(one) : "vcc");
        REP8(X)
#undef X
      } else if (OP == RCP) {
#define X(n) asm volatile("v_rcp_f64 %0, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == SQRT) {
#define X(n) asm volatile("v_sqrt_f64 %0, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == FRACT) {
#define X(n) asm volatile("v_fract_f64 %0, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == FREXP_EXP) {
#define X(n) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(I(n)) : "v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == TRIG_PREOP) {
#define X(n) asm volatile("v_trig_preop_f64 %0, %0, %1" : "+v"(A(n)) : "v"(one));
        REP8(X)
#undef X
      } else if (OP == DIVSCALE) {
#define X(n) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(A(n)) : "v"(m) : "vcc");
        REP8(X)
#undef X
      } else if (OP == DIVFMAS) {
#define X(n) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(A(n)) : "v"(m), "v"(c) : "vcc");
        REP8(X)
#undef X
      } else if (OP == DIVFIXUP) {
#define X(n) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(A(n)) : "v"(m), "v"(c));
        REP8(X)
#undef X
      } else if (OP == CND_OOP) {
#define X(n) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(I(n)) : "v"(one), "v"(j) : "vcc");
        REP8(X)
#undef X
      } else if (OP == CND_E64) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(I(n)) : "v"(one), "s"(msk));
        REP8(X)
#undef X
      } else if (OP == CMP_CND) {
#define X(n) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(I(n)) : "v"(A(n)), "v"(m), "v"(one) : "vcc");
        REP8(X)
#undef X
      } else if (OP == BFI) {
#define X(n) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(I(n)) : "v"(one), "v"(j));
        REP8(X)
#undef X
      } else if (OP == AND32) {
#define X(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(I(n)) : "v"(j));
        REP8(X)
#undef X
      } else if (OP == MINF) {
#define X(n) asm volatile("v_min_f64 %0, %0, %1" : "+v"(A(n)) : "v"(m));
        REP8(X)
#undef X
      } else if (OP == CMPCLASS) {
#define X(n) asm volatile("v_cmp_class_f64 vcc, %0, %1" :: "v"(A(n)), "v"(one) : "vcc");
        REP8(X)
#undef X
      } else if (OP == CVTFI) {
#define X(n) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(A(n)) : "v"(I(n)));
        REP8(X)
#undef X
      } else if (OP == ADDU) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(I(n)) : "v"(one));
        REP8(X)
#undef X
      } else if (OP == MOV32) {
#define X(n) asm volatile("v_mov_b32 %0, %1" : "=v"(I(n)) : "v"(one));
        REP8(X)
#undef X
      } else if (OP == FMA_SGPR) {
#define X(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(n)) : "s"(ms), "v"(c));
        REP8(X)
#undef X
      } else if (OP == MOV64) {
#define X(n) asm volatile("v_mov_b64 %0, %1" : "=v"(A(n)) : "v"(m));
        REP8(X)
#undef X
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int OP> void run(double* d, long long* dc, int iters) {
  printf("%-28s", kNames[OP]);
  for (int wps : {1, 2, 4}) {      // waves per SIMD: blocks of 256 threads = one wave per SIMD each
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, 50, 1.5);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.5);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // cycles the SIMD spends per wave-instruction = wave's loop cycles / its instructions / waves sharing the SIMD
    printf("  %d w/SIMD: %6.2f cyc/instr/SIMD", wps, med / ((double)iters * 64) / wps);
  }
  printf("\n");
}

int main() {
  double* d; long long* dc;
  hipMalloc(&d, 1024 * 256 * 8); hipMalloc(&dc, 1024 * 4 * 8);
  const int iters = 2000;
  run<FMA>(d, dc, iters); run<MUL>(d, dc, iters); run<ADD>(d, dc, iters); run<RNDNE>(d, dc, iters); run<LDEXP>(d, dc, iters);
  run<CVTI>(d, dc, iters); run<CMPF>(d, dc, iters); run<CNDMASK>(d, dc, iters); run<MAXF>(d, dc, iters); run<LSHLADD>(d, dc, iters);
  run<CMPU>(d, dc, iters); run<RCP>(d, dc, iters); run<SQRT>(d, dc, iters); run<FRACT>(d, dc, iters); run<FREXP_EXP>(d, dc, iters);
  run<TRIG_PREOP>(d, dc, iters); run<DIVSCALE>(d, dc, iters); run<DIVFMAS>(d, dc, iters); run<DIVFIXUP>(d, dc, iters); run<MOV64>(d, dc, iters);
  run<CND_OOP>(d, dc, iters); run<CND_E64>(d, dc, iters); run<CMP_CND>(d, dc, iters); run<BFI>(d, dc, iters); run<AND32>(d, dc, iters); run<MINF>(d, dc, iters);
  run<CMPCLASS>(d, dc, iters); run<CVTFI>(d, dc, iters); run<ADDU>(d, dc, iters); run<MOV32>(d, dc, iters); run<FMA_SGPR>(d, dc, iters);
  return 0;
}
