// lm.cpp -- the Levenberg-Marquardt driver gadf_fit (gadfit.F90:502-1035) on the host,
// requesting sweeps from the device.  By north_star the damped normal-equation solve and the
// lambda logic stay on the host; they are restated exactly because they decide which device
// passes are requested.  Host work per iteration is O(dim^3) on a
// dim x dim matrix; everything N-sized lives in HBM behind gfh_sweep/gfh_chi2/gfh_omega.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "context.h"
#include "group.h"

using namespace gfh;

// potr_f08 (gadfit_linalg.F90:36-57) = dpotrf('U') then dpotrs, column-major, in place.  Split so a
// factor can serve two right-hand sides: gadf_fit factorises the SAME matrix twice per accelerated
// iteration (gadfit.F90:712-713 and 737-738); reusing the factor gives bitwise the same delta2.
static int potrf_upper_plain(int n, double* a) {
  auto A = [&](int i, int j) -> double& { return a[(size_t)j * n + i]; };
  for (int j = 0; j < n; j++) {
    double ajj = A(j, j);
    for (int k = 0; k < j; k++) ajj -= A(k, j) * A(k, j);
    if (!(ajj > 0.0)) { set_global_error("Cholesky factorization failed (dpotrf)."); return 1; }
    ajj = std::sqrt(ajj); A(j, j) = ajj;
    const double rinv = 1.0 / ajj;
    // row j of U: one dot product per column c.  Four columns at a time for instruction-level
    // parallelism; every individual sum keeps the k = 0..j-1 order, so results are bitwise those
    // of the plain loop (and of the oracle's).
    const double* cj = a + (size_t)j * n;
    int c = j + 1;
    for (; c + 3 < n; c += 4) {
      const double *c0 = a + (size_t)c * n, *c1 = c0 + n, *c2 = c1 + n, *c3 = c2 + n;
      double s0 = c0[j], s1 = c1[j], s2 = c2[j], s3 = c3[j];
      for (int k = 0; k < j; k++) {
        const double v = cj[k];
        s0 -= v * c0[k]; s1 -= v * c1[k]; s2 -= v * c2[k]; s3 -= v * c3[k];
      }
      A(j, c) = s0 * rinv; A(j, c + 1) = s1 * rinv; A(j, c + 2) = s2 * rinv; A(j, c + 3) = s3 * rinv;
    }
    for (; c < n; c++) {
      double s = A(j, c);
      for (int k = 0; k < j; k++) s -= A(k, j) * A(k, c);
      A(j, c) = s * rinv;
    }
  }
  return 0;
}

// ---- blocked factorisation for the normal equations of global fits (dim = n_global + n_local *
// n_datasets reaches hundreds: BASELINE config 3 has dim 259, where the plain loop above needs 7.7 ms per
// solve -- 70 times the device pass it waits for).  Left-looking by block columns of NB: the block row
// [jb, jb+nb) x [jb, n) first receives the contributions of the rows above it (dot products of
// column pairs, contiguous in memory: the 2 x 4 micro-kernel below, AVX2+FMA where the CPU has it),
// then its diagonal block is factorised by the plain algorithm and the rest of the block row follows by
// forward substitution.  Same arithmetic per entry, different (fixed) order of additions.
namespace {
typedef double v4d __attribute__((ext_vector_type(4)));

__attribute__((target("avx2,fma"))) void dots_2x4_avx2(const double* a0, const double* a1, const double* const* b, int len, double* out) {
  v4d acc[2][4];
  for (int r = 0; r < 2; r++) for (int q = 0; q < 4; q++) acc[r][q] = (v4d){0.0, 0.0, 0.0, 0.0};
  int k = 0;
  for (; k + 3 < len; k += 4) {
    v4d va0, va1, vb;
    __builtin_memcpy(&va0, a0 + k, 32); __builtin_memcpy(&va1, a1 + k, 32);
    for (int q = 0; q < 4; q++) {
      __builtin_memcpy(&vb, b[q] + k, 32);
      acc[0][q] += va0 * vb; acc[1][q] += va1 * vb;
    }
  }
  for (int q = 0; q < 4; q++) {
    double s0 = ((acc[0][q][0] + acc[0][q][1]) + acc[0][q][2]) + acc[0][q][3];
    double s1 = ((acc[1][q][0] + acc[1][q][1]) + acc[1][q][2]) + acc[1][q][3];
    for (int kk = k; kk < len; kk++) { s0 += a0[kk] * b[q][kk]; s1 += a1[kk] * b[q][kk]; }
    out[q] = s0; out[4 + q] = s1;
  }
}

void dots_2x4_plain(const double* a0, const double* a1, const double* const* b, int len, double* out) {
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < len; k++) {
    const double x0 = a0[k], x1 = a1[k];
    for (int q = 0; q < 4; q++) { s[q] += x0 * b[q][k]; s[4 + q] += x1 * b[q][k]; }
  }
  for (int q = 0; q < 8; q++) out[q] = s[q];
}

int potrf_upper_blocked(int n, double* a) {
  static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  void (*dots)(const double*, const double*, const double* const*, int, double*) = avx2 ? dots_2x4_avx2 : dots_2x4_plain;
  auto A = [&](int i, int j) -> double& { return a[(size_t)j * n + i]; };
  auto col = [&](int j) -> const double* { return a + (size_t)j * n; };
  constexpr int NB = 48;
  for (int jb = 0; jb < n; jb += NB) {
    const int nb = jb + NB < n ? NB : n - jb;
    // (1) A(i, c) -= sum_{k < jb} U(k, i) U(k, c) for i in the block row, c >= i
    if (jb > 0) {
      for (int i = jb; i < jb + nb; i += 2) {
        const int i1 = i + 1 < jb + nb ? i + 1 : i;          // odd block height: the second row repeats the first
        for (int c = i; c < n; c += 4) {
          const double* b[4]; int cc[4];
          for (int q = 0; q < 4; q++) { cc[q] = c + q < n ? c + q : c; b[q] = col(cc[q]); }
          double out[8];
          dots(col(i), col(i1), b, jb, out);
          for (int q = 0; q < 4; q++) {
            if (c + q >= n) break;
            A(i, c + q) -= out[q];
            if (i1 != i && c + q >= i1) A(i1, c + q) -= out[4 + q];
          }
        }
      }
    }
    // (2) diagonal block by the plain algorithm (rows/columns jb .. jb+nb-1, k from jb)
    for (int j = jb; j < jb + nb; j++) {
      double ajj = A(j, j);
      for (int k = jb; k < j; k++) ajj -= A(k, j) * A(k, j);
      if (!(ajj > 0.0)) { set_global_error("Cholesky factorization failed (dpotrf)."); return 1; }
      ajj = std::sqrt(ajj); A(j, j) = ajj;
      const double rinv = 1.0 / ajj;
      // (3) row j of U beyond the diagonal, inside the block and to its right: forward substitution
      for (int c = j + 1; c < n; c++) {
        double sacc = A(j, c);
        const double *cj = col(j), *ccol = col(c);
        for (int k = jb; k < j; k++) sacc -= cj[k] * ccol[k];
        A(j, c) = sacc * rinv;
      }
    }
  }
  return 0;
}
}  // namespace

static int potrf_upper(int n, double* a) {
  return n > 64 ? potrf_upper_blocked(n, a) : potrf_upper_plain(n, a);
}

static void potrs_upper(int n, const double* a, double* b) {
  auto A = [&](int i, int j) -> double { return a[(size_t)j * n + i]; };
  for (int i = 0; i < n; i++) { double t = b[i]; for (int k = 0; k < i; k++) t -= A(k, i) * b[k]; b[i] = t / A(i, i); }
  for (int k = n - 1; k >= 0; k--) if (b[k] != 0.0) { b[k] /= A(k, k); for (int i = 0; i < k; i++) b[i] -= b[k] * A(i, k); }
}

extern "C" int gfh_potr(int n, double* a, double* b) {
  if (potrf_upper(n, a)) return 1;
  potrs_upper(n, a, b);
  return 0;
}

namespace {

double ipow(double x, int n) {   // x**n with integer n, as the Fortran intrinsic
  if (n == 0) return 1.0;
  unsigned m = n < 0 ? (unsigned)(-(long)n) : (unsigned)n;
  double r = 1.0, b = x;
  while (m) { if (m & 1u) r *= b; m >>= 1; if (m) b *= b; }
  return n < 0 ? 1.0 / r : r;
}

// The dense dim x dim images of a fit.  A global fit of many curves writes and reads only the pattern of its normal equations
// (gfh_sweep's pattern-only transfer, the block-arrow solve): 4003 columns are 128 MB per image of which 1 MB is ever touched, and
// filling them with zeros was 40 ms per fit.  The memory comes zeroed from calloc (for sizes like these: fresh pages, which cost
// nothing until they are touched) and value-initialisation leaves it alone.
template <class T> struct ZeroAlloc {
  using value_type = T;
  ZeroAlloc() = default;
  template <class U> ZeroAlloc(const ZeroAlloc<U>&) {}
  T* allocate(size_t n) { void* p = calloc(n ? n : 1, sizeof(T)); if (!p) throw std::bad_alloc(); return static_cast<T*>(p); }
  void deallocate(T* p, size_t) { free(p); }
  template <class U> void construct(U*) {}
  template <class U, class A0, class... A> void construct(U* p, A0&& a0, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...); }
  template <class U> bool operator==(const ZeroAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const ZeroAlloc<U>&) const { return false; }
};
using ZVec = std::vector<double, ZeroAlloc<double>>;
inline void zero_image(ZVec& v, size_t n) { ZVec().swap(v); v.resize(n); }

struct Fit {
  gfh_ctx* c; double* pars; int na, np, nd, dim;
  const int32_t* active; std::vector<int32_t> jac;
  ZVec JTJ, lin, nextJTJ;                     // dense dim x dim images (nextJTJ: the look-ahead sweep's at the trial point)
  std::vector<double> JTres, DTD, delta1, delta2, old_delta1, JTomega, old_pars;
  std::vector<double> nextJTres;

  double dtd(const std::vector<double>& a, const std::vector<double>& b) const {
    double s = 0; for (int i = 0; i < dim; i++) s += a[i] * (DTD[i] * b[i]); return s;   // dot(a, matmul(DTD,b)), DTD diagonal
  }
  // ---- block-arrow structure of a global fit.  A column of J that only ONE dataset's rows touch (a
  // local parameter of that dataset) has no product with the local columns of any other dataset:
  // JTJ + lambda DTD = [diag(A_1 .. A_nd)  B; B^T  S] with A_d the local block of dataset d and S the
  // block of the global parameters.  The reference factorises it densely (doc/user_guide.tex:222-235
  // notes the structure is not used); at BASELINE config 3 (64 x 4 local + 3 global, dim 259) that is
  // 5.8 Mflop on the host per solve against a 0.1 ms device pass.  Here: U_d = chol(A_d),
  // W_d = U_d^-T B_d, S' = S - sum_d W_d^T W_d, U_g = chol(S') -- the Cholesky factor of the matrix with
  // the global columns ordered last, the zero blocks skipped.  Same solution up to rounding.
  std::vector<int> lcols, loff, gcols;        // local columns grouped by dataset (offsets loff[nd+1]); global columns
  std::vector<double> Ud, Wd, Ug, ytmp;       // factors: per dataset [nl*nl] and [nl*ng] (column-major), global [ng*ng]
  std::vector<int> udoff, wdoff;
  bool arrow = false;
  void find_structure() {
    std::vector<int> owner(dim, -1);           // -1 unused, d = only dataset d, -2 = several datasets
    for (int d = 0; d < nd; d++) for (int j = 0; j < na; j++) {
      int& o_ = owner[jac[(size_t)d * na + j]];
      o_ = o_ == -1 ? d : (o_ == d ? d : -2);
    }
    lcols.clear(); gcols.clear(); loff.assign(nd + 1, 0);
    for (int d = 0; d < nd; d++) { for (int col = 0; col < dim; col++) if (owner[col] == d) lcols.push_back(col); loff[d + 1] = (int)lcols.size(); }
    for (int col = 0; col < dim; col++) if (owner[col] < 0) gcols.push_back(col);
    arrow = nd > 1 && dim > 16 && (int)lcols.size() >= dim / 2;
    if (!arrow) return;
    const int ng = (int)gcols.size();
    udoff.assign(nd + 1, 0); wdoff.assign(nd + 1, 0);
    for (int d = 0; d < nd; d++) { const int nl = loff[d + 1] - loff[d]; udoff[d + 1] = udoff[d] + nl * nl; wdoff[d + 1] = wdoff[d] + nl * ng; }
    Ud.assign(udoff[nd], 0); Wd.assign(wdoff[nd], 0); Ug.assign((size_t)ng * ng, 0); ytmp.assign(dim, 0);
  }
  double Mat(int row, int col, double lambda) const { return JTJ[(size_t)col * dim + row] + (row == col ? lambda * DTD[col] : 0.0); }
  int factor_arrow(double lambda) {
    const int ng = (int)gcols.size();
    for (int a = 0; a < ng; a++) for (int b = 0; b < ng; b++) Ug[(size_t)b * ng + a] = Mat(gcols[a], gcols[b], lambda);
    for (int d = 0; d < nd; d++) {
      const int nl = loff[d + 1] - loff[d];
      if (!nl) continue;
      const int* lc = &lcols[loff[d]];
      double* U = &Ud[udoff[d]]; double* W = &Wd[wdoff[d]];
      for (int a = 0; a < nl; a++) for (int b = 0; b < nl; b++) U[(size_t)b * nl + a] = Mat(lc[a], lc[b], lambda);
      if (potrf_upper_plain(nl, U)) return 1;
      // W = U^-T B, B[a][g] = Mat(lc[a], gcols[g]); column g of W by forward substitution with U^T
      for (int g = 0; g < ng; g++) {
        double* wg = W + (size_t)g * nl;
        for (int a = 0; a < nl; a++) {
          double t = Mat(lc[a], gcols[g], lambda);
          for (int k = 0; k < a; k++) t -= U[(size_t)a * nl + k] * wg[k];
          wg[a] = t / U[(size_t)a * nl + a];
        }
      }
      // S -= W^T W (upper triangle is what potrf reads; keep it symmetric anyway)
      for (int g = 0; g < ng; g++) for (int h = 0; h <= g; h++) {
        double t = 0.0;
        for (int k = 0; k < nl; k++) t += W[(size_t)h * nl + k] * W[(size_t)g * nl + k];
        Ug[(size_t)g * ng + h] -= t; if (h != g) Ug[(size_t)h * ng + g] -= t;
      }
    }
    return ng ? potrf_upper_plain(ng, Ug.data()) : 0;
  }
  void solve_arrow(const std::vector<double>& rhs, std::vector<double>& out) {
    const int ng = (int)gcols.size();
    std::vector<double>& y = ytmp;
    std::vector<double> bg(ng);
    for (int g = 0; g < ng; g++) bg[g] = rhs[gcols[g]];
    // forward: y_d = U_d^-T b_d ; b_g -= W_d^T y_d
    for (int d = 0; d < nd; d++) {
      const int nl = loff[d + 1] - loff[d];
      const int* lc = nl ? &lcols[loff[d]] : nullptr; const double* U = nl ? &Ud[udoff[d]] : nullptr; const double* W = nl ? &Wd[wdoff[d]] : nullptr;
      double* yd = y.data() + loff[d];
      for (int a = 0; a < nl; a++) {
        double t = rhs[lc[a]];
        for (int k = 0; k < a; k++) t -= U[(size_t)a * nl + k] * yd[k];
        yd[a] = t / U[(size_t)a * nl + a];
      }
      for (int g = 0; g < ng; g++) { double t = 0.0; for (int k = 0; k < nl; k++) t += W[(size_t)g * nl + k] * yd[k]; bg[g] -= t; }
    }
    // global block: U_g^T U_g x_g = b_g
    if (ng) potrs_upper(ng, Ug.data(), bg.data());
    for (int g = 0; g < ng; g++) out[gcols[g]] = bg[g];
    // back: x_d = U_d^-1 (y_d - W_d x_g)
    for (int d = 0; d < nd; d++) {
      const int nl = loff[d + 1] - loff[d];
      if (!nl) continue;
      const int* lc = &lcols[loff[d]]; const double* U = &Ud[udoff[d]]; const double* W = &Wd[wdoff[d]];
      double* yd = y.data() + loff[d];
      for (int a = 0; a < nl; a++) { double t = yd[a]; for (int g = 0; g < ng; g++) t -= W[(size_t)g * nl + a] * bg[g]; yd[a] = t; }
      for (int a = nl - 1; a >= 0; a--) {
        double t = yd[a];
        for (int k = a + 1; k < nl; k++) t -= U[(size_t)k * nl + a] * yd[k];
        yd[a] = t / U[(size_t)a * nl + a];
        out[lc[a]] = yd[a];
      }
    }
  }

  int solve(const std::vector<double>& rhs, std::vector<double>& out, double lambda) {
    if (arrow) {
      out.assign(dim, 0.0);
      if (factor_arrow(lambda)) return fail(c, "Cholesky factorization failed (dpotrf).");
      solve_arrow(rhs, out);
      return 0;
    }
    out = rhs;                                                         // gadfit.F90:711-713
    for (int col = 0; col < dim; col++)
      for (int row = 0; row < dim; row++)
        lin[(size_t)col * dim + row] = JTJ[(size_t)col * dim + row] + (row == col ? lambda * DTD[col] : 0.0);
    if (potrf_upper(dim, lin.data())) return fail(c, "Cholesky factorization failed (dpotrf).");
    potrs_upper(dim, lin.data(), out.data());
    return 0;
  }
  // second right-hand side against the factor left in `lin` by solve() (same JTJ, lambda, DTD)
  void solve_again(const std::vector<double>& rhs, std::vector<double>& out) {
    if (arrow) { out.assign(dim, 0.0); solve_arrow(rhs, out); return; }
    out = rhs;
    potrs_upper(dim, lin.data(), out.data());
  }
  void restore() { for (int d = 0; d < nd; d++) for (int j = 0; j < na; j++) pars[d * np + active[j]] = old_pars[d * na + j]; }
  void save() { for (int d = 0; d < nd; d++) for (int j = 0; j < na; j++) old_pars[d * na + j] = pars[d * np + active[j]]; }
};

}  // namespace

// The damped solve exactly as gfh_fit performs it (host only; tests and callers with their own LM loop).
extern "C" int gfh_solve_damped(int n_datasets, int n_act, const int32_t* jac_idx, int dim, const double* JTJ,
                                const double* DTD, double lambda, const double* rhs, double* out, int use_structure) {
  if (n_datasets < 1 || n_act < 1 || dim < 1 || !jac_idx || !JTJ || !DTD || !rhs || !out) { set_global_error("gfh_solve_damped: bad arguments"); return 1; }
  Fit f; f.c = nullptr; f.pars = nullptr; f.na = n_act; f.np = 0; f.nd = n_datasets; f.dim = dim; f.active = nullptr;
  f.jac.assign(jac_idx, jac_idx + (size_t)n_datasets * n_act);
  for (int v : f.jac) if (v < 0 || v >= dim) { set_global_error("gfh_solve_damped: Jacobian index out of range"); return 1; }
  f.JTJ.assign(JTJ, JTJ + (size_t)dim * dim); f.DTD.assign(DTD, DTD + dim); zero_image(f.lin, (size_t)dim * dim);
  f.find_structure();
  if (!use_structure) f.arrow = false;
  std::vector<double> b(rhs, rhs + dim), x;
  if (f.arrow ? f.factor_arrow(lambda) : 0) return 1;
  if (f.arrow) { x.assign(dim, 0.0); f.solve_arrow(b, x); }
  else {
    x = b;
    for (int col = 0; col < dim; col++) for (int row = 0; row < dim; row++) f.lin[(size_t)col * dim + row] = f.Mat(row, col, lambda);
    if (potrf_upper(dim, f.lin.data())) return 1;
    potrs_upper(dim, f.lin.data(), x.data());
  }
  memcpy(out, x.data(), sizeof(double) * dim);
  return 0;
}

extern "C" int gfh_fit(gfh_ctx* c, double* pars, int na, const int32_t* active, const int32_t* is_global,
                       gfh_fit_options* o, gfh_fit_result* r) {
  if (!c) return 1;
  if (c->grp) {
    // device group: the whole LM loop runs on every member's thread, as on every coarray image of the reference
    // (the sums are identical bits on all members, so the replicated host logic takes the same path); member 0 answers
    const int n = gfh::group_size(c);
    gfh_ctx* k0 = gfh::group_member(c, 0);
    if (!k0->has_model || !k0->nd) return fail(c, "gfh_fit: model and data must be set first");
    const size_t np = (size_t)k0->nd * k0->model.n_pars;
    std::vector<std::vector<double>> P((size_t)n, std::vector<double>(pars, pars + np));
    gfh_fit_options dflt; memset(&dflt, 0, sizeof dflt); dflt.umnigh_a = 0.5;
    std::vector<gfh_fit_options> O((size_t)n, o ? *o : dflt);
    std::vector<gfh_fit_result> R((size_t)n);
    for (int q = 1; q < n; q++) O[(size_t)q].verbosity = 0;                  // one image prints (gadfit.F90:886)
    const int rc = gfh::group_run(c, [&](gfh_ctx* k, int q) -> int {
      return gfh_fit(k, P[(size_t)q].data(), na, active, is_global, &O[(size_t)q], &R[(size_t)q]); });
    memcpy(pars, P[0].data(), sizeof(double) * np);
    if (o) o->umnigh_a = O[0].umnigh_a;
    if (r) *r = R[0];
    return rc;
  }
  if (c->device < 0) return fail(c, "no GPU bound to this context (libgadfit_hip has no CPU fallback)");
  if (!c->has_model || !c->nd) return fail(c, "gfh_fit: model and data must be set first");
  if (na < 1) return fail(c, "There are no active parameters.");                                     // gadfit.F90:602-603
  if (!pars || !active || !is_global) return fail(c, "gfh_fit: null argument");
  if (na > c->model.n_pars) return fail(c, "gfh_fit: more active parameters than the model has");
  for (int j = 0; j < na; j++)          // active[] indexes is_global[] and the parameter block: checked before anything reads through it
    if (active[j] < 0 || active[j] >= c->model.n_pars) return fail(c, "gfh_fit: active parameter index out of range");
  gfh_fit_options defaults; memset(&defaults, 0, sizeof defaults); defaults.umnigh_a = 0.5;
  if (!o) o = &defaults;
  gfh::Range range("gadfit gfh_fit");
  double lambda = o->has_lambda ? o->lambda : 1.0;                                                    // gadfit.F90:568-584
  const double lam_up = o->has_lam_up ? o->lam_up : 10.0, lam_down = o->has_lam_down ? o->lam_down : 10.0;
  int lam_incs = 2;
  if (o->has_lam_incs) { if (o->lam_incs < 1) return fail(c, "Input parameter lam_incs must be at least 1."); lam_incs = o->lam_incs; }
  const int uphill = o->has_uphill ? o->uphill : 0;
  const bool nielsen = o->has_nielsen && o->nielsen, umnigh = o->has_umnigh && o->umnigh;
  const double umnigh_m = std::exp(-0.2);

  Fit f; f.c = c; f.pars = pars; f.na = na; f.np = c->model.n_pars; f.nd = c->nd; f.active = active;
  f.jac.resize((size_t)f.nd * na);
  const int dim = f.dim = gfh_jacobian_indices(f.nd, na, active, is_global, f.jac.data());          // gadfit.F90:615-631
  f.find_structure();
  zero_image(f.JTJ, (size_t)dim * dim); f.JTres.assign(dim, 0); f.DTD.assign(dim, 0); f.delta1.assign(dim, 0);
  f.delta2.assign(dim, 0); f.old_delta1.assign(dim, 0); f.JTomega.assign(dim, 0);
  if (!f.arrow) zero_image(f.lin, (size_t)dim * dim);       // (the dense factor; a global fit of many curves is solved block by block)
  f.old_pars.assign((size_t)na * f.nd, 0);
  if (o->DTD_min) for (int i = 0; i < dim; i++) f.DTD[i] = o->DTD_min[i];                           // gadfit.F90:641-646
  long long dof_ll = (long long)c->n_total - dim;                                                    // gadfit.F90:648-657
  if (dof_ll < 0) return fail(c, "More independent fitting parameters than data points.");
  const double dof = dof_ll == 0 ? 1.0 : (double)dof_ll;
  f.save();
  gfh_fit_result local; if (!r) r = &local;
  memset(r, 0, sizeof *r); r->dim = dim; r->exit_reason = -1;
  r->dof = dof_ll == 0 ? 1 : (dof_ll > 2147483647LL ? 2147483647 : (int)dof_ll);      // (the fit itself uses the 64-bit count)
  int iterations = 0;
  double old_chi2 = 0, new_chi2 = 0, old_old_chi2 = 0, acc_ratio = 0, beta = 0, sweep_chi2 = 0;
  auto t0 = std::chrono::steady_clock::now();
  // f.JTJ / f.nextJTJ start as zeros and are only ever written by gfh_sweep with this fit's pattern: the
  // pattern-only transfer of global fits need not clear them again
  c->jtj_prezeroed = true;
  auto finish = [&](int rc) {
    c->jtj_prezeroed = false;
    r->iterations = iterations; r->lambda = lambda; r->chi2 = old_chi2;
    r->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
  };
  // keep_jacobian = 2: J goes to HBM only if this fit reads it back (the grad_chi2 / cos_phi tests, and
  // STEP 3's J^T omega where gfh_k_omega_jt is not available: models with integrate(), robust losses); the fused kernel forms J^T J / J^T r from registers either way
  if (c->keep_jacobian == 2) {
    set_store_j(c, (o->has_accth && o->accth > 1.17549435e-38 && omega_needs_jacobian(c, na)) || o->has_grad_chi2 || o->has_cos_phi);
    set_store_res(c, o->has_grad_chi2 || o->has_cos_phi);
  }
  if (gfh_set_active(c, active, na, f.jac.data(), dim)) return finish(1);
  // Look-ahead (gadfit_hip.h, gfh_set_lookahead): the fused sweep already returns sum r^2, so the
  // FIRST trial chi2() of an iteration (gadfit.F90:753) is taken from a sweep at the trial point;
  // when the step is accepted that sweep IS the next iteration's STEP 1+2 (same parameters, same
  // kernel, same numbers), so an accepted iteration costs one N-sized pass instead of two.  Armed
  // while the previous first trial was accepted.  Off when the convergence tests read the device
  // J/res pair the reference has at that point (old J, new res: gadfit.F90:849-850, 865-873).
  // adaptive parallelism (load_balancing, gadfit.F90:672-673): the ranges may be re-cut before an iteration, so no sweep is handed over
  const bool balancing = c->load_balancing && c->nranks > 1;
  // ... and only where the sweep's sum r^2 is bitwise what chi2() returns at the same parameters: the fused kernel (same
  // partition and order of additions as gfh_k_chi2) with shared reciprocals (GADFIT_HIP_FAST_DIV=0 keeps the reference's
  // two division forms, whose values differ by rounding between the active and the passive evaluation)
  const bool la_ok = c->lookahead != 0 && !o->has_grad_chi2 && !o->has_cos_phi && c->gen.loss == 0 && !balancing &&
                     sweep_chi2_is_bitwise(c) && c->gen.fast_div;
  // WHEN to speculate is a prediction -- whichever kernel runs, the trial chi2 is the same bits -- and a wrong one costs a sweep
  // thrown away (sweep - chi2(): 0.36 ms at the headline size, what four right ones save).  Round 6:
  //   * the first iteration of a fit has no history: it speculates where the first step is substantially damped (lambda >= 0.1; the
  //     reference's default is 1) and runs the reference's schedule where the caller asks for a near Gauss-Newton first step;
  //   * later iterations speculate iff the previous iteration accepted its first trial (after a rejected first trial the next
  //     iteration runs the reference's schedule, chi2 kernel first) ...
  //   * ... and lambda is not at or below a level rejected within the last 4 iterations: the usual end state of the plain
  //     lambda/10 - lambda*10 rule alternates "accepted at L, rejected at L/10, accepted at L", and that rejection is known ahead.
  // bench.py's rejecting_fit leg (40 %-off start, lambda0 = 1e-6): no sweep thrown away, the reference's schedule pass for pass;
  // the headline fits (all first trials accepted): one N-sized pass per iteration as before.
  bool have_next = false, prev_first_accepted = true;
  double la_lam_rej = -1.0; int la_rej_age = 1 << 20;
  if (la_ok) { zero_image(f.nextJTJ, (size_t)dim * dim); f.nextJTres.assign(dim, 0); }
  // old_chi2 = chi2() before the loop (gadfit.F90:670).  With look-ahead the first STEP 1+2 pass -- same
  // parameters -- returns that sum r^2 itself and is handed to the first iteration: one N-sized pass less per fit.
  if (la_ok) {
    if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.nextJTJ.data(), f.nextJTres.data(), &old_chi2)) return finish(1);
    have_next = true; r->n_lookahead++;
  } else if (gfh_chi2(c, pars, &old_chi2)) return finish(1);
  r->n_chi2++;
  for (;;) {
    if (balancing && iterations > 0 && gfh_rebalance(c, nullptr)) return finish(1);                 // re_initialize, gadfit.F90:672-673
    // STEP 1 + 2 (gadfit.F90:675-701)
    if (have_next) { f.JTJ.swap(f.nextJTJ); f.JTres.swap(f.nextJTres); have_next = false; }
    else if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.JTJ.data(), f.JTres.data(), &sweep_chi2)) return finish(1);
    r->n_sweeps++;
    for (int i = 0; i < dim; i++) {                                                                 // gadfit.F90:702-710
      const double d = f.JTJ[(size_t)i * dim + i];
      if (o->has_damp_max && !o->damp_max) f.DTD[i] = d; else f.DTD[i] = f.DTD[i] > d ? f.DTD[i] : d;
    }
    // GADFIT_HIP_DUMP_FIRST_PASS=<file>: J^T J, J^T r and chi2 of the FIRST pass of this fit (the start parameters: no solve and no
    // accept / reject has touched them, so they compare with the oracle free of the fit's conditioning -- tests/test_gpu_fortran_fuzz.py),
    // written by rank 0 with 17 digits; every later fit of the process appends.
    if (iterations == 0 && c->rank == 0) if (const char* dump = getenv("GADFIT_HIP_DUMP_FIRST_PASS")) if (FILE* fh = fopen(dump, "a")) {
      fprintf(fh, "first_pass dim %d chi2 %.17g\nJTres", dim, old_chi2);
      for (int j = 0; j < dim; j++) fprintf(fh, " %.17g", f.JTres[(size_t)j]);
      fprintf(fh, "\nJTJ");
      for (size_t j = 0; j < (size_t)dim * dim; j++) fprintf(fh, " %.17g", f.JTJ[j]);
      fprintf(fh, "\n");
      fclose(fh);
    }
    if (getenv("GADFIT_HIP_TRACE_SUMS")) {          // (debugging: what the iteration's pass returned)
      fprintf(stderr, "iteration %d: chi2 %.17g  JTres", iterations + 1, old_chi2);
      for (int j = 0; j < dim && j < 8; j++) fprintf(stderr, " %.10g", f.JTres[(size_t)j]);
      fprintf(stderr, "  JTJ");
      for (int j = 0; j < dim && j < 4; j++) for (int k = 0; k < dim && k < 4; k++) fprintf(stderr, " %.10g", f.JTJ[(size_t)j * dim + k]);
      fprintf(stderr, "  lambda %.6g\n", lambda);
    }
    if (f.solve(f.JTres, f.delta1, lambda)) return finish(1);
    if (o->has_accth && o->accth > 1.17549435e-38) {                                                // STEP 3, gadfit.F90:715-743
      if (gfh_omega(c, pars, f.delta1.data(), f.JTomega.data())) return finish(1);
      r->n_omega++;
      f.solve_again(f.JTomega, f.delta2);                                                             // gadfit.F90:736-738
      acc_ratio = std::sqrt(f.dtd(f.delta2, f.delta2) / f.dtd(f.delta1, f.delta1));
      if (acc_ratio > o->accth) std::fill(f.delta2.begin(), f.delta2.end(), 0.0);
    }
    for (int d = 0; d < f.nd; d++) for (int j = 0; j < na; j++) {                                   // gadfit.F90:745-750
      double& p = pars[d * f.np + active[j]];
      p = p + f.delta1[f.jac[d * na + j]] + 0.5 * f.delta2[f.jac[d * na + j]];
    }
    bool quit = false;
    bool first_accepted = false;
    double iter_lam_rej = -1.0;                   // (largest lambda whose trial this iteration rejected)
    for (int i = 1; i <= lam_incs + 1; i++) {                                                       // STEP 4, gadfit.F90:752-819
      // no look-ahead in the iteration that max_iter ends anyway (its Jacobian would not be used)
      const bool la_predicts_accept = iterations == 0 ? lambda >= 0.1
                                                      : prev_first_accepted && !(la_rej_age <= 4 && lambda <= la_lam_rej * (1.0 + 1e-9));
      const bool spec = la_ok && i == 1 && la_predicts_accept && !(o->has_max_iter && iterations + 1 >= o->max_iter);
      const double lambda_of_trial = lambda;
      if (spec) {
        if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.nextJTJ.data(), f.nextJTres.data(), &new_chi2)) return finish(1);
        r->n_lookahead++;
      } else if (gfh_chi2(c, pars, &new_chi2)) return finish(1);
      r->n_chi2++;
      if (getenv("GADFIT_HIP_TRACE_SUMS")) {
        fprintf(stderr, "  trial %d (%s): lambda %.6g  chi2 %.17g  active parameters of dataset 1:", i, spec ? "sweep" : "chi2", lambda, new_chi2);
        for (int j = 0; j < na && j < 8; j++) fprintf(stderr, " %.17g", pars[active[j]]);
        fprintf(stderr, "\n");
      }
      if (iterations == 0) beta = 0.0;
      else beta = f.dtd(f.delta1, f.old_delta1) / std::sqrt(f.dtd(f.delta1, f.delta1)) / std::sqrt(f.dtd(f.old_delta1, f.old_delta1));
      if (ipow(1.0 - beta, uphill) * new_chi2 < old_chi2) {                                         // gadfit.F90:761
        if (nielsen) {                                                                              // gadfit.F90:762-767
          double q = 0;
          for (int col = 0; col < dim; col++) {
            double s = 0;
            for (int k = 0; k < dim; k++) s += (f.JTJ[(size_t)k * dim + col] + (k == col ? lambda * f.DTD[col] : 0.0)) * f.delta1[k];
            q += f.delta1[col] * s;
          }
          const double rho = (old_chi2 - new_chi2) / 2 / q;
          const double t = 1 - ipow(2 * rho - 1, 3), lo = 1 / lam_down;
          lambda = lambda * (lo > t ? lo : t);
        }
        if (umnigh) {                                                                               // gadfit.F90:768-779
          if (new_chi2 < old_chi2 && beta >= 0.0) {
            o->umnigh_a = o->umnigh_a * umnigh_m + 1.0 - umnigh_m;
            double t = ipow(1.0 - std::fabs(2.0 * o->umnigh_a - 1.0), 2);
            t = t > 1e-2 ? t : 1e-2; t = t < 1.0 ? t : 1.0;
            lambda = lambda * t;
          } else {
            o->umnigh_a = o->umnigh_a * umnigh_m + (1.0 - umnigh_m) / 2.0;
            if (new_chi2 >= old_chi2) {
              double t = 1.0 - std::fabs(2.0 * o->umnigh_a - 1.0);
              t = t > 1.0 ? t : 1.0; t = t < 10.0 ? t : 10.0;
              lambda = lambda / t;
            }
          }
        }
        if (!(nielsen || umnigh)) lambda = lambda / lam_down;                                       // gadfit.F90:780-782
        have_next = spec; first_accepted = i == 1;
        break;
      } else if (i <= lam_incs) {                                                                   // gadfit.F90:785-808
        iter_lam_rej = iter_lam_rej > lambda_of_trial ? iter_lam_rej : lambda_of_trial;
        if (umnigh) {
          o->umnigh_a = o->umnigh_a * umnigh_m;
          double t = 1.0 - std::fabs(2.0 * o->umnigh_a - 1.0);
          if (beta < 0.0) { t = t * t; t = t > 1e-2 ? t : 1e-2; } else { t = t > 0.1 ? t : 0.1; }
          t = t < 1.0 ? t : 1.0;
          lambda = lambda * t;
        } else lambda = lam_up * lambda;
        f.restore();
        if (f.solve(f.JTres, f.delta1, lambda)) return finish(1);
        for (int d = 0; d < f.nd; d++) for (int j = 0; j < na; j++) pars[d * f.np + active[j]] += f.delta1[f.jac[d * na + j]];
      } else {                                                                                      // gadfit.F90:809-816
        f.restore();
        if (o->verbosity) printf(" Lambda increased %d times in a row.\n", lam_incs + 1);
        r->exit_reason = 7; quit = true; break;
      }
    }
    if (quit) break;
    // what the next iteration's prediction reads (see above)
    prev_first_accepted = first_accepted;
    if (iter_lam_rej >= 0.0) { la_lam_rej = iter_lam_rej; la_rej_age = 0; } else if (la_rej_age < (1 << 20)) la_rej_age++;
    f.save();                                                                                       // gadfit.F90:821-827
    f.old_delta1 = f.delta1;
    old_old_chi2 = old_chi2;
    old_chi2 = old_chi2 < new_chi2 ? old_chi2 : new_chi2;
    iterations++;
    if (o->verbosity) printf(" iteration %d  lambda %.6g  chi2/DOF %.15g\n", iterations, lambda, new_chi2 / dof);
    // STEP 5 (gadfit.F90:835-915)
    if (o->has_chi2_abs && old_chi2 / dof < o->chi2_abs) { r->exit_reason = 1; break; }
    if (o->has_chi2_rel && (old_old_chi2 - old_chi2) / old_chi2 < o->chi2_rel) { r->exit_reason = 2; break; }
    if (o->has_grad_chi2) {                                                                         // gadfit.F90:848-860
      std::vector<double> g(dim);
      if (gfh_aux(c, 0, nullptr, g.data())) return finish(1);
      f.JTres = g;
      double s = 0; for (double v : g) s += v * v;
      if (2 * std::sqrt(s) < o->grad_chi2) { r->exit_reason = 3; break; }
    }
    if (o->has_cos_phi) {                                                                           // gadfit.F90:861-884
      double s3[3];
      if (gfh_aux(c, 1, f.delta1.data(), s3)) return finish(1);
      if (std::fabs(s3[0]) / std::sqrt(s3[1]) / std::sqrt(s3[2]) < o->cos_phi) { r->exit_reason = 4; break; }
    }
    if (o->has_rel_error) {                                                                         // gadfit.F90:885-898
      bool all = true;
      for (int d = 0; d < f.nd && all; d++) for (int j = 0; j < na; j++)
        if (std::fabs(f.delta1[f.jac[d * na + j]] / pars[d * f.np + active[j]]) > o->rel_error) { all = false; break; }
      if (all) { r->exit_reason = 5; break; }
    }
    if (o->has_rel_error_global) {                                                                  // gadfit.F90:899-910
      bool any = false;
      for (int j = 0; j < na; j++)
        if (is_global[active[j]] && std::fabs(f.delta1[f.jac[j]] / pars[active[j]]) > o->rel_error_global) any = true;
      if (!any) { r->exit_reason = 6; break; }
    }
    if (o->has_max_iter && iterations >= o->max_iter) { r->exit_reason = 0; break; }                // gadfit.F90:911-915
  }
  return finish(0);
}


// n_iter iterations of the basic scheme with no convergence exits (see gadfit_hip.h).
extern "C" int gfh_lm_iterate(gfh_ctx* c, double* pars, int na, const int32_t* active, const int32_t* is_global,
                              int n_iter, double* state3, double* DTD) {
  if (!c) return 1;
  if (c->grp) {
    const int n = gfh::group_size(c);
    gfh_ctx* k0 = gfh::group_member(c, 0);
    if (!k0->has_model || !k0->nd) return fail(c, "gfh_lm_iterate: model and data must be set first");
    const size_t np = (size_t)k0->nd * k0->model.n_pars;
    std::vector<int32_t> jtmp((size_t)k0->nd * na);
    const int gdim = gfh_jacobian_indices(k0->nd, na, active, is_global, jtmp.data());
    std::vector<std::vector<double>> P((size_t)n, std::vector<double>(pars, pars + np));
    std::vector<std::vector<double>> S((size_t)n, std::vector<double>(state3, state3 + 3));
    std::vector<std::vector<double>> D((size_t)n, std::vector<double>(DTD, DTD + gdim));
    const int rc = gfh::group_run(c, [&](gfh_ctx* k, int q) -> int {
      return gfh_lm_iterate(k, P[(size_t)q].data(), na, active, is_global, n_iter, S[(size_t)q].data(), D[(size_t)q].data()); });
    memcpy(pars, P[0].data(), sizeof(double) * np);
    memcpy(state3, S[0].data(), sizeof(double) * 3);
    memcpy(DTD, D[0].data(), sizeof(double) * gdim);
    return rc;
  }
  if (c->device < 0) return fail(c, "no GPU bound to this context (libgadfit_hip has no CPU fallback)");
  if (!c->has_model || !c->nd) return fail(c, "gfh_lm_iterate: model and data must be set first");
  if (na < 1 || na > c->model.n_pars || !pars || !active || !is_global || !state3 || !DTD) return fail(c, "gfh_lm_iterate: bad arguments");
  for (int j = 0; j < na; j++)
    if (active[j] < 0 || active[j] >= c->model.n_pars) return fail(c, "gfh_lm_iterate: active parameter index out of range");
  Fit f; f.c = c; f.pars = pars; f.na = na; f.np = c->model.n_pars; f.nd = c->nd; f.active = active;
  f.jac.resize((size_t)f.nd * na);
  const int dim = f.dim = gfh_jacobian_indices(f.nd, na, active, is_global, f.jac.data());
  f.find_structure();
  zero_image(f.JTJ, (size_t)dim * dim); f.JTres.assign(dim, 0); f.DTD.assign(DTD, DTD + dim); f.delta1.assign(dim, 0);
  if (!f.arrow) zero_image(f.lin, (size_t)dim * dim);
  f.old_pars.assign((size_t)na * f.nd, 0);
  zero_image(f.nextJTJ, (size_t)dim * dim); f.nextJTres.assign(dim, 0);
  double lambda = state3[0], old_chi2 = state3[1], sweep_chi2 = 0, new_chi2 = 0;
  if (gfh_set_active(c, active, na, f.jac.data(), dim)) return 1;
  if (old_chi2 < 0 && gfh_chi2(c, pars, &old_chi2)) return 1;
  // look-ahead as in gfh_fit: the trial chi2 is the sum r^2 of a sweep at the trial point, which
  // an accepted step hands to the next iteration.  The hand-over does not cross calls: the first
  // iteration of every call sweeps, and a look-ahead of the last iteration is not started.
  const bool la_ok = c->lookahead != 0 && c->gen.loss == 0 && sweep_chi2_is_bitwise(c) && c->gen.fast_div;
  bool la_armed = la_ok, have_next = false;
  for (int it = 0; it < n_iter; it++) {
    f.save();
    if (have_next) { f.JTJ.swap(f.nextJTJ); f.JTres.swap(f.nextJTres); have_next = false; }
    else if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.JTJ.data(), f.JTres.data(), &sweep_chi2)) return 1;
    for (int i = 0; i < dim; i++) { const double d = f.JTJ[(size_t)i * dim + i]; f.DTD[i] = f.DTD[i] > d ? f.DTD[i] : d; }
    if (f.solve(f.JTres, f.delta1, lambda)) return 1;
    for (int d = 0; d < f.nd; d++) for (int j = 0; j < na; j++) pars[d * f.np + active[j]] += f.delta1[f.jac[d * na + j]];
    const bool spec = la_armed && it + 1 < n_iter;
    if (spec) { if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.nextJTJ.data(), f.nextJTres.data(), &new_chi2)) return 1; }
    else if (gfh_chi2(c, pars, &new_chi2)) return 1;
    if (new_chi2 < old_chi2) { old_chi2 = new_chi2; lambda /= 10.0; state3[2] += 1.0; have_next = spec; la_armed = la_ok; }
    else { f.restore(); lambda *= 10.0; la_armed = false; }
  }
  state3[0] = lambda; state3[1] = old_chi2;
  for (int i = 0; i < dim; i++) DTD[i] = f.DTD[i];
  return 0;
}
