// kernels.hip -- hand-written gfx950 kernels of the LM hot path that do not depend on the
// model: J^T J / J^T r formation on the FP64 matrix cores (STEP 2, gadfit.F90:695-699),
// the deterministic cross-workgroup reduction, the scatter into the global (dim x dim)
// system via Jacobian_indices (gadfit.F90:615-628), J^T v products (gadfit.F90:734, 849)
// and the cos(phi) sums (gadfit.F90:865-873), plus init_weights (gadfit.F90:445-470).
//
// Layout contract: J is [NA][ldj] parameter-major over padded slots; every
// gram block covers whole 256-slot tiles of ONE dataset; pad slots hold zeros in J and res.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace gfh {

typedef long long i64;
typedef double double4_t __attribute__((ext_vector_type(4)));

// The wave tree t_l += t_(l+32), (l+16), (l+8), (l+4), (l+2), (l+1) (lane 0 ends with the sum): the additions of the __shfl_down loop it
// replaces, bit for bit, with the four levels inside a row of 16 lanes as DPP moves (row_shl) instead of trips through the LDS
// crossbar (ds_bpermute).  The generated kernels carry the same function (codegen.cpp, gfh_wave_sum): k_jtv and gfh_k_omega_jt must
// add alike.
template <int N> static __device__ __forceinline__ double row_down(const double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x100 | N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x100 | N, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
static __device__ __forceinline__ double wave_sum(double t) {
  t += __shfl_down(t, 32, 64);
  t += __shfl_down(t, 16, 64);
  t += row_down<8>(t); t += row_down<4>(t); t += row_down<2>(t); t += row_down<1>(t);
  return t;
}

// --------------------------------------------------------------------------------------
// Gram kernel.  One wave owns 16x16 accumulator tiles of the (NA x NA) per-dataset Gram
// matrix; a k-step of v_mfma_f64_16x16x4_f64 consumes 4 data points.  Lane (r = l&15,
// q = l>>4) holds J[16t + r][n + 4q .. 4q+3] as one 32-byte load, i.e. a wave instruction
// reads 16 rows x 128 B (full cache lines).  The SAME fragment serves as A operand of row
// tile t and as B operand of column tile t (A[i][k] and B[k][j] have identical lane maps for
// a symmetric product), so each J element is loaded exactly once per wave.
// J^T r rides along on the VALU (2 FMAs per fragment), sum r^2 likewise.
//
// T = number of 16-row tiles (NA <= 16 T).  Partials per workgroup:
//   [pair(ti<=tj)][16][16] row-major, then JTr[16 T], then rTr.
template <int T>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ J, const i64 ldj, const int na,
                                              const double* __restrict__ res,
                                              const i64* __restrict__ gb_start, const int* __restrict__ gb_slots,
                                              double* __restrict__ partial, const int pstride) {
  constexpr int NPAIR = T * (T + 1) / 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const i64 s = gb_start[blockIdx.x];
  const i64 e = s + gb_slots[blockIdx.x];

  double4_t acc[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; p++) acc[p] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double accr[T];
#pragma unroll
  for (int t = 0; t < T; t++) accr[t] = 0.0;
  double accc = 0.0;

  const double* jrow[T];
  bool valid[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    valid[t] = (16 * t + r) < na;
    jrow[t] = J + (i64)(valid[t] ? 16 * t + r : 0) * ldj;
  }

  // Register double buffer: the 4 (T + 1) fragment loads of the NEXT 64 points are in flight while the matrix instructions of
  // this pass run (the last pass re-reads its own: no branch).  Same thread-to-point map and order of additions as the
  // single-buffered loop this replaces (0.61 -> 0.47 ms at 32 columns: the pass was load latency + 48 MFMAs, back to back).
  auto load = [&](const i64 n, double4_t (&a)[4][T], double4_t (&rr)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const i64 c = n + 16 * u + 4 * q;
#pragma unroll
      for (int t = 0; t < T; t++) a[u][t] = *reinterpret_cast<const double4_t*>(jrow[t] + c);
      rr[u] = *reinterpret_cast<const double4_t*>(res + c);
    }
  };
  double4_t a[4][T], rr[4];
  i64 n = s + 64 * wv;
  constexpr bool kDouble = T <= 2;      // (3 and 4 tiles: two buffers would not fit 256 VGPRs next to 6 / 10 accumulator tiles)
  if (kDouble && n < e) load(n, a, rr);
  for (; n < e; n += 256) {
    double4_t an[4][T], rn[4];
    if (kDouble) load(n + 256 < e ? n + 256 : n, an, rn);
    else load(n, a, rr);
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int t = 0; t < T; t++)
        if (!valid[t]) a[u][t] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int p = 0;
#pragma unroll
        for (int ti = 0; ti < T; ti++)
#pragma unroll
          for (int tj = ti; tj < T; tj++, p++)
            acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti][j], a[u][tj][j], acc[p], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < T; t++)
        accr[t] += a[u][t][0] * rr[u][0] + a[u][t][1] * rr[u][1] + a[u][t][2] * rr[u][2] + a[u][t][3] * rr[u][3];
      accc += rr[u][0] * rr[u][0] + rr[u][1] * rr[u][1] + rr[u][2] * rr[u][2] + rr[u][3] * rr[u][3];
    }
    if (kDouble) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
#pragma unroll
        for (int t = 0; t < T; t++) a[u][t] = an[u][t];
        rr[u] = rn[u];
      }
    }
  }

  // cross-wave reduction in LDS, fixed order (deterministic).
  __shared__ double sm[4][NPAIR * 256 + T * 64 + 4];
  double* mine = sm[wv];
#pragma unroll
  for (int p = 0; p < NPAIR; p++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // f64 16x16x4 C/D map: col = lane&15, row = (lane>>4) + 4*reg
      mine[p * 256 + (q + 4 * j) * 16 + r] = acc[p][j];
    }
#pragma unroll
  for (int t = 0; t < T; t++) mine[NPAIR * 256 + t * 64 + lane] = accr[t];
  if (r == 0) mine[NPAIR * 256 + T * 64 + q] = accc;
  __syncthreads();
  double* out = partial + (i64)blockIdx.x * pstride;
  for (int idx = threadIdx.x; idx < NPAIR * 256; idx += 256)
    out[idx] = ((sm[0][idx] + sm[1][idx]) + sm[2][idx]) + sm[3][idx];
  for (int idx = threadIdx.x; idx < 16 * T; idx += 256) {
    const int t = idx >> 4, rr_ = idx & 15;
    double sacc = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int qq = 0; qq < 4; qq++) sacc += sm[w][NPAIR * 256 + t * 64 + qq * 16 + rr_];
    out[NPAIR * 256 + idx] = sacc;
  }
  if (threadIdx.x == 0) {
    double sacc = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int qq = 0; qq < 4; qq++) sacc += sm[w][NPAIR * 256 + T * 64 + qq];
    out[NPAIR * 256 + 16 * T] = sacc;
  }
}

// --------------------------------------------------------------------------------------
// Up to 8 active parameters: a 16-row matrix tile would be half empty (and half of every fragment load redundant), and
// the whole per-point outer product is NA (NA + 1) / 2 + NA + 1 <= 45 multiply-adds -- nothing next to the 8 (NA + 1)
// bytes the point costs to read.  So this is a plain streaming kernel: one lane per point, the NA Jacobian entries and
// the residual as coalesced loads (two points per lane in flight), every product accumulated per lane, 8 waves per gram
// block, wave tree + waves in order at the end.  Writes the partial image of k_gram<1>.
template <int NA>
__global__ __launch_bounds__(512) void k_gram_small(const double* __restrict__ J, const i64 ldj, const double* __restrict__ res,
                                                     const i64* __restrict__ gb_start, const int* __restrict__ gb_slots,
                                                     double* __restrict__ partial, const int pstride) {
  constexpr int NP = NA * (NA + 1) / 2, NACC = NP + NA + 1;
  const i64 s = gb_start[blockIdx.x], e = s + gb_slots[blockIdx.x];
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.0;
  // two points per trip (512 apart: gb_slots is a multiple of 512), their loads issued together
  for (i64 i = s + threadIdx.x; i < e; i += 1024) {
    const bool two = i + 512 < e;
    const i64 i2 = two ? i + 512 : i;
    double j[NA], k[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) { j[a] = J[(i64)a * ldj + i]; k[a] = J[(i64)a * ldj + i2]; }
    const double r = res[i], r2 = res[i2];
    int p = 0;
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
      for (int b = a; b < NA; b++, p++) acc[p] += j[a] * j[b];
#pragma unroll
    for (int a = 0; a < NA; a++) acc[NP + a] += j[a] * r;
    acc[NP + NA] += r * r;
    if (two) {
      p = 0;
#pragma unroll
      for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = a; b < NA; b++, p++) acc[p] += k[a] * k[b];
#pragma unroll
      for (int a = 0; a < NA; a++) acc[NP + a] += k[a] * r2;
      acc[NP + NA] += r2 * r2;
    }
  }
  __shared__ double sm[8][NACC];
  __shared__ double tot[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) {
    const double t = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double t = sm[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 8; w++) t += sm[w][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  double* out = partial + (i64)blockIdx.x * pstride;
  const int idx = threadIdx.x;
  if (idx < 256) {
    int a = idx >> 4, b = idx & 15;
    if (a > b) { const int t = a; a = b; b = t; }
    out[idx] = b < NA ? tot[a * NA - a * (a - 1) / 2 + (b - a)] : 0.0;      // both triangles of the one 16 x 16 tile
  } else if (idx < 272) out[idx] = idx - 256 < NA ? tot[NP + idx - 256] : 0.0;
  else if (idx == 272) out[idx] = tot[NP + NA];
}

// --------------------------------------------------------------------------------------
// More than 64 active parameters per dataset (T > 4 tiles): the Gram image is formed in blocks of up to
// 4 x 4 tiles, one launch per block pair (gi <= gj) of the upper triangle.  A launch loads the row tiles
// R0 .. R0+TR-1 and the column tiles C0 .. C0+TC-1 of J (the same fragments when the block is on the
// diagonal) and writes its tile pairs into the SAME per-workgroup partial image k_gram<T> would write
// (pair index from the global T), so the reduction and assembly kernels do not change.  Diagonal
// launches also carry J^T r of their rows; the first one carries sum r^2.
template <int TR, int TC, bool SYM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gram_block(const double* __restrict__ J, const i64 ldj, const int na,
                                                    const double* __restrict__ res, const i64* __restrict__ gb_start,
                                                    const int* __restrict__ gb_slots, double* __restrict__ partial,
                                                    const int pstride, const int T, const int R0, const int C0) {
  constexpr int NACC = SYM ? TR * (TR + 1) / 2 : TR * TC;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const i64 s = gb_start[blockIdx.x];
  const i64 e = s + gb_slots[blockIdx.x];
  double4_t acc[NACC];
#pragma unroll
  for (int p = 0; p < NACC; p++) acc[p] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double accr[TR];
#pragma unroll
  for (int t = 0; t < TR; t++) accr[t] = 0.0;
  double accc = 0.0;
  const double* arow[TR]; bool aval[TR];
  const double* brow[TC]; bool bval[TC];
#pragma unroll
  for (int t = 0; t < TR; t++) { aval[t] = R0 + t < T && 16 * (R0 + t) + r < na; arow[t] = J + (i64)(aval[t] ? 16 * (R0 + t) + r : 0) * ldj; }
#pragma unroll
  for (int t = 0; t < TC; t++) { bval[t] = C0 + t < T && 16 * (C0 + t) + r < na; brow[t] = J + (i64)(bval[t] ? 16 * (C0 + t) + r : 0) * ldj; }
  for (i64 n = s + 64 * wv; n < e; n += 256) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const i64 c = n + 16 * u + 4 * q;
      double4_t a[TR], b[TC];
#pragma unroll
      for (int t = 0; t < TR; t++) { a[t] = *reinterpret_cast<const double4_t*>(arow[t] + c); if (!aval[t]) a[t] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
      if (!SYM) {
#pragma unroll
        for (int t = 0; t < TC; t++) { b[t] = *reinterpret_cast<const double4_t*>(brow[t] + c); if (!bval[t]) b[t] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int p = 0;
#pragma unroll
        for (int ti = 0; ti < TR; ti++)
#pragma unroll
          for (int tj = SYM ? ti : 0; tj < TC; tj++, p++)
            acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti][j], SYM ? a[tj][j] : b[tj][j], acc[p], 0, 0, 0);
      }
      if (SYM) {
        const double4_t rr = *reinterpret_cast<const double4_t*>(res + c);
#pragma unroll
        for (int t = 0; t < TR; t++) accr[t] += a[t][0] * rr[0] + a[t][1] * rr[1] + a[t][2] * rr[2] + a[t][3] * rr[3];
        if (R0 == 0) accc += rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3];
      }
    }
  }
  // Cross-wave sum, waves in order.  Up to 4 x 4 tiles every wave has an image of its own in LDS; the wide blocks (5 or 6 tiles in ONE
  // launch: 15 / 21 pair images of 2 KB) add into one image wave after wave -- the same order of additions, a quarter of the LDS.
  constexpr bool WIDE = TR > 4;
  constexpr int VEC = TR * 64 + 4;
  __shared__ double sm[WIDE ? 1 : 4][NACC * 256 + (WIDE ? 0 : VEC)];
  __shared__ double sv[WIDE ? 4 : 1][WIDE ? VEC : 1];
  double* vec = WIDE ? sv[wv] : sm[wv] + NACC * 256;
  if (!WIDE) {
    double* mine = sm[wv];
#pragma unroll
    for (int p = 0; p < NACC; p++)
#pragma unroll
      for (int j = 0; j < 4; j++) mine[p * 256 + (q + 4 * j) * 16 + r] = acc[p][j];
  } else {
    for (int w = 0; w < 4; w++) {
      if (wv == w) {
#pragma unroll
        for (int p = 0; p < NACC; p++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            double* e = &sm[0][p * 256 + (q + 4 * j) * 16 + r];
            *e = w == 0 ? acc[p][j] : *e + acc[p][j];
          }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int t = 0; t < TR; t++) vec[t * 64 + lane] = accr[t];
  if (r == 0) vec[TR * 64 + q] = accc;
  __syncthreads();
  double* out = partial + (i64)blockIdx.x * pstride;
  const int npair = T * (T + 1) / 2;
  {
    int p = 0;
    for (int ti = 0; ti < TR; ti++)
      for (int tj = SYM ? ti : 0; tj < TC; tj++, p++) {
        const int gi = R0 + ti, gj = C0 + tj;
        if (gi >= T || gj >= T) continue;
        const int gp = gi * T - gi * (gi - 1) / 2 + (gj - gi);
        for (int idx = threadIdx.x; idx < 256; idx += 256)
          out[gp * 256 + idx] = WIDE ? sm[0][p * 256 + idx]
                                     : ((sm[0][p * 256 + idx] + sm[WIDE ? 0 : 1][p * 256 + idx]) + sm[WIDE ? 0 : 2][p * 256 + idx]) + sm[WIDE ? 0 : 3][p * 256 + idx];
      }
  }
  if (SYM) {
    for (int idx = threadIdx.x; idx < 16 * TR; idx += 256) {
      const int t = idx >> 4, rr_ = idx & 15;
      if (R0 + t >= T) continue;
      double sacc = 0.0;
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) sacc += (WIDE ? sv[w] : sm[WIDE ? 0 : w] + NACC * 256)[t * 64 + qq * 16 + rr_];
      out[npair * 256 + 16 * (R0 + t) + rr_] = sacc;
    }
    if (R0 == 0 && threadIdx.x == 0) {
      double sacc = 0.0;
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) sacc += (WIDE ? sv[w] : sm[WIDE ? 0 : w] + NACC * 256)[TR * 64 + qq];
      out[npair * 256 + 16 * T] = sacc;
    }
  }
}

// --------------------------------------------------------------------------------------
// Sum workgroup partials per dataset in a fixed order (=> bitwise reproducible).
// grid = (ceil(width/32), n_datasets), block = 1024 = 32 elements x 32 slices: slice s adds
// workgroups b0+s, b0+s+32, ... and the 32 slice sums are added in slice order.
__global__ __launch_bounds__(1024) void k_reduce_partials(const double* __restrict__ partial, const int pstride,
                                                          const int width, const int* __restrict__ ds_first_gb,
                                                          double* __restrict__ out /*[nd][width]*/) {
  const int d = blockIdx.y;
  const int el = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  const int b0 = ds_first_gb[d], b1 = ds_first_gb[d + 1];
  double s = 0.0;
  if (el < width)
    for (int b = b0 + sl; b < b1; b += 32) s += partial[(i64)b * pstride + el];
  __shared__ double sm[32][33];
  sm[sl][threadIdx.x & 31] = s;
  __syncthreads();
  if (sl == 0 && el < width) {
    double t = sm[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 32; k++) t += sm[k][threadIdx.x];
    out[(i64)d * width + el] = t;
  }
}

// --------------------------------------------------------------------------------------
// Scatter the per-dataset Grams into the global normal equations through Jacobian_indices.
// packed = [JTJ (dim*dim, column-major) | JTres (dim) | chi2].  One thread per output
// element, datasets visited in order (deterministic); inv[d][col] = local active index or -1.
#define GFH_ASM_CHUNK 8
// grid = (ceil(dim/256), dim + 1): blockIdx.y = column of JTJ, or dim for the [JTres | chi2] tail; threads = rows
__global__ __launch_bounds__(256) void k_assemble(const double* __restrict__ G /*[nd][gw]*/, const int gw, const int T, const int nd,
                                                  const int dim, const int* __restrict__ inv /*[nd][dim]*/,
                                                  const int* __restrict__ owner /*[dim]*/, double* __restrict__ packed) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int col = blockIdx.y;
  const i64 nn = (i64)dim * dim;
  const int npair = T * (T + 1) / 2;
  if (col < dim) {
    if (row >= dim) return;
    // owner[c] = the only dataset whose rows touch column c (a local parameter), or -1 (global parameter):
    // an entry with a local row or column receives a contribution from that one dataset only, so the loop
    // over datasets -- and with it the order of additions -- collapses to that term
    const int orow = owner[row], ocol = owner[col];
    const int d0 = orow >= 0 ? orow : (ocol >= 0 ? ocol : 0);
    const int d1 = (orow >= 0 || ocol >= 0) ? d0 + 1 : nd;
    // GFH_ASM_CHUNK datasets per step: the index and value loads of a step are independent of each other (a
    // global x global entry walks all datasets, and one dependent load chain per dataset made this kernel
    // 21 us at 64 datasets); the additions keep the dataset order
    double s = 0.0;
    for (int d = d0; d < d1; d += GFH_ASM_CHUNK) {
      double v[GFH_ASM_CHUNK]; bool ok[GFH_ASM_CHUNK];
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) {
        ok[u] = d + u < d1;
        int a = ok[u] ? inv[(d + u) * dim + row] : -1, b = ok[u] ? inv[(d + u) * dim + col] : -1;
        ok[u] = a >= 0 && b >= 0;
        if (a > b) { int t = a; a = b; b = t; }      // upper triangle of tile pairs is stored
        const int ti = a >> 4, tj = b >> 4;
        const int p = ti * T - ti * (ti - 1) / 2 + (tj - ti);
        v[u] = ok[u] ? G[(i64)(d + u) * gw + p * 256 + (a & 15) * 16 + (b & 15)] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) if (ok[u]) s += v[u];
    }
    packed[(i64)col * dim + row] = s;
  } else if (row < dim) {
    const int orow = owner[row];
    const int d0 = orow >= 0 ? orow : 0, d1 = orow >= 0 ? orow + 1 : nd;
    double s = 0.0;
    for (int d = d0; d < d1; d += GFH_ASM_CHUNK) {
      double v[GFH_ASM_CHUNK]; bool ok[GFH_ASM_CHUNK];
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) {
        const int a = d + u < d1 ? inv[(d + u) * dim + row] : -1;
        ok[u] = a >= 0;
        v[u] = ok[u] ? G[(i64)(d + u) * gw + npair * 256 + a] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) if (ok[u]) s += v[u];
    }
    packed[nn + row] = s;
  } else if (row == dim) {
    double s = 0.0;
    for (int d = 0; d < nd; d += GFH_ASM_CHUNK) {
      double v[GFH_ASM_CHUNK];
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) v[u] = d + u < nd ? G[(i64)(d + u) * gw + npair * 256 + 16 * T] : 0.0;
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) if (d + u < nd) s += v[u];
    }
    packed[nn + dim] = s;
  }
}

// Pattern-only assembly for global fits: the normal equations of several datasets are block-arrow (the
// local parameters of different datasets do not couple), so only the entries (row <= col) that some
// dataset touches are formed and sent to the host: packed = [nnz values | JTres (dim) | chi2].  The value
// of an entry is computed exactly as in k_assemble (same terms, same order).
__global__ __launch_bounds__(256) void k_assemble_sparse(const double* __restrict__ G, const int gw, const int T, const int nd,
                                                         const int dim, const int* __restrict__ inv, const int* __restrict__ owner,
                                                         const int* __restrict__ nz_row, const int* __restrict__ nz_col, const int nnz,
                                                         double* __restrict__ packed) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int npair = T * (T + 1) / 2;
  if (idx < nnz) {
    const int row = nz_row[idx], col = nz_col[idx];
    const int orow = owner[row], ocol = owner[col];
    const int d0 = orow >= 0 ? orow : (ocol >= 0 ? ocol : 0);
    const int d1 = (orow >= 0 || ocol >= 0) ? d0 + 1 : nd;
    double s = 0.0;
    for (int d = d0; d < d1; d += GFH_ASM_CHUNK) {
      double v[GFH_ASM_CHUNK]; bool ok[GFH_ASM_CHUNK];
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) {
        ok[u] = d + u < d1;
        int a = ok[u] ? inv[(d + u) * dim + row] : -1, b = ok[u] ? inv[(d + u) * dim + col] : -1;
        ok[u] = a >= 0 && b >= 0;
        if (a > b) { int t = a; a = b; b = t; }
        const int ti = a >> 4, tj = b >> 4;
        const int p = ti * T - ti * (ti - 1) / 2 + (tj - ti);
        v[u] = ok[u] ? G[(i64)(d + u) * gw + p * 256 + (a & 15) * 16 + (b & 15)] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) if (ok[u]) s += v[u];
    }
    packed[idx] = s;
  } else if (idx < nnz + dim) {
    const int row = idx - nnz;
    const int orow = owner[row];
    const int d0 = orow >= 0 ? orow : 0, d1 = orow >= 0 ? orow + 1 : nd;
    double s = 0.0;
    for (int d = d0; d < d1; d += GFH_ASM_CHUNK) {
      double v[GFH_ASM_CHUNK]; bool ok[GFH_ASM_CHUNK];
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) {
        const int a = d + u < d1 ? inv[(d + u) * dim + row] : -1;
        ok[u] = a >= 0;
        v[u] = ok[u] ? G[(i64)(d + u) * gw + npair * 256 + a] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) if (ok[u]) s += v[u];
    }
    packed[idx] = s;
  } else if (idx == nnz + dim) {
    double s = 0.0;
    for (int d = 0; d < nd; d += GFH_ASM_CHUNK) {
      double v[GFH_ASM_CHUNK];
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) v[u] = d + u < nd ? G[(i64)(d + u) * gw + npair * 256 + 16 * T] : 0.0;
#pragma unroll
      for (int u = 0; u < GFH_ASM_CHUNK; u++) if (d + u < nd) s += v[u];
    }
    packed[idx] = s;
  }
}

// The same packed image from host-built source lists (context.cpp, prepare_active): k_assemble_sparse reaches a value through four
// dependent loads (pattern entry -> owner -> inv -> G), each ~2 us on an idle chip, which made it 24 us at config 3.  Here
// meta[idx] >= 0 is the offset of the single term in G; meta[idx] < 0 (and not INT_MIN = no term) points at [count, offsets...]
// in `list`, the terms of an entry that walks the datasets, in dataset order.  Same terms, same order of additions as k_assemble.
// host_out != nullptr (single rank): the values also go straight into the pinned result mailbox and the last workgroup to arrive
// posts the status word and the call's sequence number, as k_publish does -- one launch less per pass.
__global__ __launch_bounds__(256) void k_gather_sum(const double* __restrict__ G, const int* __restrict__ meta, const int* __restrict__ list,
                                                    const int n, double* __restrict__ out, const int* __restrict__ status, double* host_out,
                                                    unsigned* counter, unsigned long long* host_flag, const unsigned long long seq) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int m = idx < n ? meta[idx] : (int)0x80000000;
  double s = 0.0;
  if (m >= 0) s += G[m];
  else if (m != (int)0x80000000) {
    const int* __restrict__ L = list + (-(m + 1));
    const int cnt = L[0];
    for (int k = 0; k < cnt; k += 16) {
      int o[16]; double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) o[u] = k + u < cnt ? L[1 + k + u] : -1;
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = o[u] >= 0 ? G[o[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; u++) if (o[u] >= 0) s += v[u];
    }
  }
  if (idx < n) {
    out[idx] = s;
    if (host_out) __builtin_nontemporal_store(s, host_out + idx);
  }
  if (!host_out) return;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrived = atomicAdd(counter, 1u);
    if (arrived == gridDim.x - 1) {
      *counter = 0;                                          // ready for the next call (stream-ordered)
      host_out[n] = (double)*status;
      __threadfence_system();
      __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// --------------------------------------------------------------------------------------
// J^T v per gram block (v = omega or res).  partial[b][a], a < na.
// CB columns per sweep over the block's points: v[i] is loaded once per CB columns and CB + 1 independent loads
// per lane are in flight (with 2 workgroups per CU that is what keeps HBM busy: up to 32 columns = the whole
// Jacobian row of the headline model in ONE pass over v).  Every column keeps its own accumulator and its own
// order of additions, so the result does not depend on CB.
template <int CB>
__global__ __launch_bounds__(256) void k_jtv(const double* __restrict__ J, const i64 ldj, const int na,
                                             const double* __restrict__ v, const i64* __restrict__ gb_start,
                                             const int* __restrict__ gb_slots, double* __restrict__ partial,
                                             const int pstride) {
  const i64 s = gb_start[blockIdx.x], e = s + gb_slots[blockIdx.x];
  __shared__ double ws[CB][4];
  for (int a0 = 0; a0 < na; a0 += CB) {
    double acc[CB];
#pragma unroll
    for (int u = 0; u < CB; u++) acc[u] = 0.0;
    const double* __restrict__ Ja = J + (i64)a0 * ldj;
    // (gb_slots is a multiple of 512: two points per trip, their loads issued together; each accumulator still adds its
    // points in ascending order, so the sums are those of the one-point loop)
    if (a0 + CB <= na) {
      for (i64 i = s + threadIdx.x; i < e; i += 512) {
        const double v0 = v[i], v1 = v[i + 256];
        double j0[CB], j1[CB];
#pragma unroll
        for (int u = 0; u < CB; u++) { j0[u] = Ja[(i64)u * ldj + i]; j1[u] = Ja[(i64)u * ldj + i + 256]; }
#pragma unroll
        for (int u = 0; u < CB; u++) { acc[u] += j0[u] * v0; acc[u] += j1[u] * v1; }
      }
    } else {
      for (i64 i = s + threadIdx.x; i < e; i += 512) {
        const double v0 = v[i], v1 = v[i + 256];
        double j0[CB], j1[CB];
#pragma unroll
        for (int u = 0; u < CB; u++)
          if (a0 + u < na) { j0[u] = Ja[(i64)u * ldj + i]; j1[u] = Ja[(i64)u * ldj + i + 256]; }
#pragma unroll
        for (int u = 0; u < CB; u++)
          if (a0 + u < na) { acc[u] += j0[u] * v0; acc[u] += j1[u] * v1; }
      }
    }
#pragma unroll
    for (int u = 0; u < CB; u++) {
      const double t = wave_sum(acc[u]);
      if ((threadIdx.x & 63) == 0) ws[u][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < CB && a0 + (int)threadIdx.x < na)
      partial[(i64)blockIdx.x * pstride + a0 + threadIdx.x] =
          ((ws[threadIdx.x][0] + ws[threadIdx.x][1]) + ws[threadIdx.x][2]) + ws[threadIdx.x][3];
    __syncthreads();
  }
}

// out[dim] from per-dataset vectors V[d][width>=na] through inv
__global__ void k_assemble_vec(const double* __restrict__ V, const int width, const int nd, const int dim,
                               const int* __restrict__ inv, double* __restrict__ out) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= dim) return;
  double s = 0.0;
  for (int d = 0; d < nd; d++) { const int a = inv[d * dim + row]; if (a >= 0) s += V[(i64)d * width + a]; }
  out[row] = s;
}

// k_reduce_partials (width na) + k_assemble_vec + k_publish for a J^T v product as ONE single-workgroup launch
// (n_datasets * na <= 4096): the same slice sums, slice order and dataset order, so bitwise the same vector.
// host_out == nullptr: the vector stays on the device (several ranks: RCCL sums it, k_publish posts it).
__global__ __launch_bounds__(1024) void k_jtv_finish(const double* __restrict__ partial, const int pstride, const int na,
                                                     const int* __restrict__ ds_first_gb, const int nd, const int dim,
                                                     const int* __restrict__ inv, double* __restrict__ out,
                                                     const int* __restrict__ status, double* host_out,
                                                     unsigned long long* host_flag, const unsigned long long seq) {
  __shared__ double sm[32][33];
  __shared__ double V[4096];
  const int lane32 = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (int d = 0; d < nd; d++) {
    const int b0 = ds_first_gb[d], b1 = ds_first_gb[d + 1];
    for (int e0 = 0; e0 < na; e0 += 32) {
      const int el = e0 + lane32;
      double s = 0.0;
      if (el < na)
        for (int b = b0 + sl; b < b1; b += 32) s += partial[(i64)b * pstride + el];
      sm[sl][lane32] = s;
      __syncthreads();
      if (sl == 0 && el < na) {
        double t = sm[0][lane32];
#pragma unroll
        for (int k = 1; k < 32; k++) t += sm[k][lane32];
        V[d * na + el] = t;
      }
      __syncthreads();
    }
  }
  for (int row = threadIdx.x; row < dim; row += 1024) {
    double s = 0.0;
    for (int d = 0; d < nd; d++) { const int a = inv[d * dim + row]; if (a >= 0) s += V[d * na + a]; }
    out[row] = s;
    if (host_out) __builtin_nontemporal_store(s, host_out + row);
  }
  if (!host_out) return;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    host_out[dim] = (double)*status;
    __threadfence_system();
    __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// cos(phi) sums (gadfit.F90:865-873): Jdelta_i = sum_a J[a][i]*dl[ds][a];
// partial[b][0..2] = {res.Jdelta, res.res, Jdelta.Jdelta}
__global__ __launch_bounds__(256) void k_cosphi(const double* __restrict__ J, const i64 ldj, const int na,
                                                const double* __restrict__ res, const double* __restrict__ dl /*[nd][na]*/,
                                                const i64* __restrict__ gb_start, const int* __restrict__ gb_slots,
                                                const int* __restrict__ gb_ds, double* __restrict__ partial,
                                                const int pstride) {
  const i64 s = gb_start[blockIdx.x], e = s + gb_slots[blockIdx.x];
  const double* d1 = dl + (i64)gb_ds[blockIdx.x] * na;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (i64 i = s + threadIdx.x; i < e; i += 256) {
    double jd = 0.0;
    for (int a = 0; a < na; a++) jd += J[(i64)a * ldj + i] * d1[a];
    const double r = res[i];
    s0 += r * jd; s1 += r * r; s2 += jd * jd;
  }
  __shared__ double ws[3][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = s0; ws[1][threadIdx.x >> 6] = s1; ws[2][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) partial[(i64)blockIdx.x * pstride + threadIdx.x] =
      ((ws[threadIdx.x][0] + ws[threadIdx.x][1]) + ws[threadIdx.x][2]) + ws[threadIdx.x][3];
}

// Result mailbox.  The <= (dim^2+dim+1)-sized result of a pass is written by the device straight
// into pinned, host-coherent memory together with the kernels' status word; the last workgroup
// to finish then stores the call's sequence number into a host flag the calling thread spins
// on.  No copy engine, no completion-signal round trip: the host sees the result a few
// microseconds after the last kernel's stores.
__global__ __launch_bounds__(256) void k_publish(const double* __restrict__ src, const int n, const int* __restrict__ status,
                                                 double* host_out, unsigned* counter,
                                                 unsigned long long* host_flag, const unsigned long long seq) {
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < n && i < (int)(blockIdx.x + 1) * 1024; i += 256)
    __builtin_nontemporal_store(src[i], host_out + i);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrived = atomicAdd(counter, 1u);
    if (arrived == gridDim.x - 1) {
      *counter = 0;                                          // ready for the next call (stream-ordered)
      host_out[n] = (double)*status;
      __threadfence_system();
      __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The kernels' status word as one more element of a cross-rank sum: 0, 1 (quadrature workspace exhausted), 4096 (code 2),
// 2^24 (anything else) -- the sum over at most a few thousand ranks still tells which codes occurred.
__global__ void k_status_slot(const int* __restrict__ status, double* __restrict__ dst) {
  const int st = *status;
  dst[0] = st == 0 ? 0.0 : st == 1 ? 1.0 : st == 2 ? 4096.0 : 16777216.0;
}

// Keeps the part busy between an upload and the first pass of a fit (context.cpp, gfh_set_data_begin): after an idle gap the
// first ~40 launches of a series run 20-35 % slower (clock ramp, tools/probes/transient.py), and any kernel work ends that.  FP64
// arithmetic on registers for `rounds` x 256 FMAs per lane, one load per lane; the result leaves only if it is a NaN's NaN.
__global__ __launch_bounds__(256) void k_keep_warm(const double* __restrict__ x, i64 n, int rounds, double* __restrict__ sink) {
  const i64 i = ((i64)blockIdx.x * 256 + threadIdx.x) % (n > 0 ? n : 1);
  double a = x[i], b = 1.0 + 1e-9 * a;
  for (int r = 0; r < rounds; r++) {
#pragma unroll 16
    for (int k = 0; k < 256; k++) a = __builtin_fma(a, b, 1e-30);
    b = 2.0 - b;
  }
  if (a != a && b != b) sink[0] = a;
}

// Pad slots of the device layout (every dataset's range is padded to whole tiles): a real abscissa of the same dataset (so f stays
// finite), y = 0, w = 0, is_pad = 1.  One workgroup per dataset; seg[d] = {first slot, number of real points, end slot}.
__global__ __launch_bounds__(256) void k_fill_pads(const i64* __restrict__ seg, double* __restrict__ x, double* __restrict__ y,
                                                   double* __restrict__ w, unsigned char* __restrict__ is_pad) {
  const i64 s0 = seg[3 * blockIdx.x], len = seg[3 * blockIdx.x + 1], s1 = seg[3 * blockIdx.x + 2];
  const double fill = len ? x[s0 + len - 1] : 0.0;
  for (i64 sl = s0 + len + threadIdx.x; sl < s1; sl += 256) { x[sl] = fill; y[sl] = 0.0; w[sl] = 0.0; is_pad[sl] = 1; }
}

// init_weights, gadfit.F90:445-470 (w holds sigma on entry for USER)
__global__ void k_init_weights(const int type, const i64 n, const double* __restrict__ y, double* __restrict__ w,
                               const unsigned char* __restrict__ is_pad) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (is_pad[i]) { w[i] = 0.0; return; }
  const double thr = 1e2 * 2.2250738585072014e-308;
  const double yy = y[i];
  double r;
  switch (type) {
    case 0: r = 1.0; break;
    case 1: r = fabs(yy) < thr ? 0.0 : 1.0 / sqrt(yy); break;
    case 2: r = fabs(yy) < thr ? 0.0 : 1.0 / yy; break;
    case 3: r = yy; break;
    default: r = 1.0 / w[i]; break;
  }
  w[i] = r;
}

// ---------------------------------------------------------------------------- launchers
int gram_partial_stride(int T) { int n = T * (T + 1) / 2 * 256 + 16 * T + 1; return (n + 3) & ~3; }

hipError_t launch_gram(hipStream_t st, int T, const double* J, i64 ldj, int na, const double* res,
                       const i64* gb_start, const int* gb_slots, int n_gb, double* partial) {
  const int ps = gram_partial_stride(T);
  if (na <= 8) {
    switch (na) {
#define GFH_GS(N) case N: hipLaunchKernelGGL(k_gram_small<N>, dim3(n_gb), dim3(512), 0, st, J, ldj, res, gb_start, gb_slots, partial, ps); break;
      GFH_GS(1) GFH_GS(2) GFH_GS(3) GFH_GS(4) GFH_GS(5) GFH_GS(6) GFH_GS(7) GFH_GS(8)
#undef GFH_GS
    }
    return hipGetLastError();
  }
  switch (T) {
    case 1: hipLaunchKernelGGL(k_gram<1>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps); break;
    case 2: hipLaunchKernelGGL(k_gram<2>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps); break;
    case 3: hipLaunchKernelGGL(k_gram<3>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps); break;
    case 4: hipLaunchKernelGGL(k_gram<4>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps); break;
    case 5: hipLaunchKernelGGL((k_gram_block<5, 5, true>), dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps, T, 0, 0); break;
    case 6: hipLaunchKernelGGL((k_gram_block<6, 6, true>), dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps, T, 0, 0); break;
    default:
      // T > 6 (5 and 6 tiles: ONE launch that reads J once, above): blocks of up to 4 x 4 tiles over the upper triangle, diagonal blocks first.  The blocks at the edge are instantiated at
      // the number of tiles they really hold (80 parameters = 5 tiles: a 4 x 4 diagonal block, a 4 x 1 block and a 1 x 1 diagonal
      // block -- 15 tile pairs and 160 column reads per point where three 4 x 4 launches made 36 pairs and 256 reads, most of them on
      // tiles that do not exist: 2.31 -> 1.32 ms at N = 4e6; 5 and 6 tiles in one launch at two waves per SIMD: 0.62 ms, profiles/r04_p80.md)
      for (int gi = 0; gi < T; gi += 4)
        for (int gj = gi; gj < T; gj += 4) {
          const int tr = T - gi < 4 ? T - gi : 4, tc = T - gj < 4 ? T - gj : 4;
#define GFH_GB(TR_, TC_, SYM_) hipLaunchKernelGGL((k_gram_block<TR_, TC_, SYM_>), dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, gb_start, gb_slots, partial, ps, T, gi, gj)
          if (gi == gj) {
            switch (tr) { case 1: GFH_GB(1, 1, true); break; case 2: GFH_GB(2, 2, true); break; case 3: GFH_GB(3, 3, true); break; default: GFH_GB(4, 4, true); }
          } else {
            // (rows of a block off the diagonal: always four whole tiles -- only the last block row / column of the triangle is short)
            switch (tc) { case 1: GFH_GB(4, 1, false); break; case 2: GFH_GB(4, 2, false); break; case 3: GFH_GB(4, 3, false); break; default: GFH_GB(4, 4, false); }
          }
#undef GFH_GB
        }
  }
  return hipGetLastError();
}

hipError_t launch_reduce_partials(hipStream_t st, const double* partial, int pstride, int width,
                                  const int* ds_first_gb, int nd, double* out) {
  hipLaunchKernelGGL(k_reduce_partials, dim3((width + 31) / 32, nd), dim3(1024), 0, st, partial, pstride, width, ds_first_gb, out);
  return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t st, const double* G, int gw, int T, int nd, int dim, const int* inv, const int* owner, double* packed) {
  // (dim + 1) threads in x for the tail row: one more than the rows, so ceil((dim + 1) / 256) blocks
  hipLaunchKernelGGL(k_assemble, dim3((unsigned)((dim + 1 + 255) / 256), (unsigned)(dim + 1)), dim3(256), 0, st, G, gw, T, nd, dim, inv, owner, packed);
  return hipGetLastError();
}

hipError_t launch_assemble_sparse(hipStream_t st, const double* G, int gw, int T, int nd, int dim, const int* inv, const int* owner,
                                  const int* nz_row, const int* nz_col, int nnz, double* packed) {
  hipLaunchKernelGGL(k_assemble_sparse, dim3((unsigned)((nnz + dim + 1 + 255) / 256)), dim3(256), 0, st, G, gw, T, nd, dim, inv, owner, nz_row, nz_col, nnz, packed);
  return hipGetLastError();
}

hipError_t launch_gather_sum(hipStream_t st, const double* G, const int* meta, const int* list, int n, double* out, const int* status,
                             double* host_out, unsigned* counter, unsigned long long* host_flag, unsigned long long seq) {
  hipLaunchKernelGGL(k_gather_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, meta, list, n, out, status, host_out, counter, host_flag, seq);
  return hipGetLastError();
}

hipError_t launch_jtv(hipStream_t st, const double* J, i64 ldj, int na, const double* v, const i64* gb_start,
                      const int* gb_slots, int n_gb, double* partial, int pstride) {
  if (na <= 8) hipLaunchKernelGGL(k_jtv<8>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, v, gb_start, gb_slots, partial, pstride);
  else if (na <= 16) hipLaunchKernelGGL(k_jtv<16>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, v, gb_start, gb_slots, partial, pstride);
  else hipLaunchKernelGGL(k_jtv<32>, dim3(n_gb), dim3(256), 0, st, J, ldj, na, v, gb_start, gb_slots, partial, pstride);
  return hipGetLastError();
}

hipError_t launch_jtv_finish(hipStream_t st, const double* partial, int pstride, int na, const int* ds_first_gb, int nd, int dim,
                             const int* inv, double* out, const int* status, double* host_out, unsigned long long* host_flag,
                             unsigned long long seq) {
  hipLaunchKernelGGL(k_jtv_finish, dim3(1), dim3(1024), 0, st, partial, pstride, na, ds_first_gb, nd, dim, inv, out, status, host_out, host_flag, seq);
  return hipGetLastError();
}

hipError_t launch_assemble_vec(hipStream_t st, const double* V, int width, int nd, int dim, const int* inv, double* out) {
  hipLaunchKernelGGL(k_assemble_vec, dim3((dim + 255) / 256), dim3(256), 0, st, V, width, nd, dim, inv, out);
  return hipGetLastError();
}

hipError_t launch_cosphi(hipStream_t st, const double* J, i64 ldj, int na, const double* res, const double* dl,
                         const i64* gb_start, const int* gb_slots, const int* gb_ds, int n_gb, double* partial, int pstride) {
  hipLaunchKernelGGL(k_cosphi, dim3(n_gb), dim3(256), 0, st, J, ldj, na, res, dl, gb_start, gb_slots, gb_ds, partial, pstride);
  return hipGetLastError();
}

hipError_t launch_publish(hipStream_t st, const double* src, int n, const int* status, double* host_out, unsigned* counter,
                          unsigned long long* host_flag, unsigned long long seq) {
  hipLaunchKernelGGL(k_publish, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, src, n, status, host_out, counter, host_flag, seq);
  return hipGetLastError();
}

hipError_t launch_status_slot(hipStream_t st, const int* status, double* dst) {
  hipLaunchKernelGGL(k_status_slot, dim3(1), dim3(1), 0, st, status, dst);
  return hipGetLastError();
}

hipError_t launch_keep_warm(hipStream_t st, const double* x, i64 n, int rounds, double* sink) {
  hipLaunchKernelGGL(k_keep_warm, dim3(256 * 8), dim3(256), 0, st, x, n, rounds, sink);
  return hipGetLastError();
}

hipError_t launch_fill_pads(hipStream_t st, int nd, const i64* seg, double* x, double* y, double* w, unsigned char* is_pad) {
  hipLaunchKernelGGL(k_fill_pads, dim3((unsigned)nd), dim3(256), 0, st, seg, x, y, w, is_pad);
  return hipGetLastError();
}

hipError_t launch_init_weights(hipStream_t st, int type, i64 n, const double* y, double* w, const unsigned char* is_pad) {
  hipLaunchKernelGGL(k_init_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, type, n, y, w, is_pad);
  return hipGetLastError();
}

}  // namespace gfh
