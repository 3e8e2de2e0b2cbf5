// ref_cxx_driver.cpp -- TEST INFRASTRUCTURE (oracle/): a caller of the REFERENCE'S OWN C++ automatic differentiation,
// compiled together with the reference's sources where they lie under /root/reference/c++/gadfit (automatic_differentiation.cpp,
// fit_function.cpp, exceptions.cpp, lapack_fallback.cpp -- each compiles as it is) into oracle/_ref/libgadfit_refcxx.so by
// oracle/Makefile.  Nothing of the reference is copied: this file is user code against its public headers, as c++/tests are.
//
// What it is for: (1) pinning oracle/gadfit_oracle.c on the hot path against outputs of the reference itself at sizes no
// known-answer test of the reference holds (residuals, Jacobian rows, J^T J, J^T r of the bench models at seeded inputs:
// tests/test_oracle_vs_reference_cxx.py); (2) bench.py's `cpu_baseline` of kind "reference".
//
// The loop below is the per-point loop of LMsolver::computeLeftHandSide / computeRightHandSide / chi2
// (c++/gadfit/lm_solver.cpp:286-346, 513-529) written out by a caller: activate the parameters for the reverse mode, seed the
// tape once per thread, per point evaluate the fit function on AdVar, run gadfit::returnSweep, store the Jacobian row
// [point][parameter], then the reference's own dsyrk / dgemv (its vendored fallback, lapack_fallback.cpp -- the container has no
// LAPACK to link lapack.cpp against).  lm_solver.cpp itself is NOT built: it includes <spdlog/spdlog.h>, which the image lacks,
// and the rules of this build forbid a stand-in header -- so the class LMsolver (lambda loop, Cholesky) is not timed; what is
// timed is everything N-sized that LMsolver::fit does per iteration, through the reference's own AD and linear algebra.
#include <chrono>
#include <cmath>
#include <vector>

#include <omp.h>

#include "automatic_differentiation.h"
#include "fit_function.h"
#include "lapack.h"
#include "numerical_integration.h"

#include "gadfit_tape.h"      // this repository's tape format (include/): the interpreter at the end evaluates one through the reference's AdVar

namespace {

using gadfit::AdVar;

// the bench models of tests/models.py, as a user of the reference's C++ API writes them
AdVar model_gauss8(const std::vector<AdVar>& p, const double x) {
  AdVar y { 0.0, 0.0, 0.0, gadfit::passive_idx };
  for (int k = 0; k < 8; k++) {
    const AdVar d = x - p[4 * k + 1];
    const AdVar t = p[4 * k] * exp(-pow(d / p[4 * k + 2], 2)) * (1 + p[4 * k + 3] * d);   // tests/models.py: model_gauss8
    if (k == 0) y = t; else y = y + t;
  }
  return y;
}

AdVar model_exp4(const std::vector<AdVar>& p, const double x) {
  AdVar y = p[0] * exp(-(x / p[1]));
  for (int k = 1; k < 4; k++) y = y + p[2 * k] * exp(-(x / p[2 * k + 1]));
  return y;
}

// BASELINE config 4 / reference test 2 (fortran/tests/2_integral_single.F90:27-46; tests/golden/goldens.py: model_integral_single):
// pi * int_0^x t^a exp(-b t^2) dt through the reference's own adaptive Gauss-Kronrod rule and its AD (numerical_integration.cpp:242-310)
double g_rel_error = 1e-10;
AdVar integrand_single(const std::vector<AdVar>& q, const AdVar& t) { return pow(t, q[0]) * exp(-(q[1] * pow(t, 2))); }
AdVar model_integral_single(const std::vector<AdVar>& p, const double x) {
  return 3.14159265358979323846 * gadfit::integrate(integrand_single, p, 0.0, x, g_rel_error);
}

// BASELINE config 3's per-dataset function (tests/models.py: model_global7): 4 local + 3 shared parameters
AdVar model_global7(const std::vector<AdVar>& p, const double x) {
  return p[0] * exp(-(x / p[4])) + p[1] * exp(-(x / p[5])) + p[2] * x * exp(-(x / p[6])) + p[3];
}

gadfit::fitSignature pick(int model) {
  return model == 0 ? gadfit::fitSignature(model_gauss8) : model == 1 ? gadfit::fitSignature(model_exp4)
       : model == 2 ? gadfit::fitSignature(model_integral_single) : gadfit::fitSignature(model_global7);
}

}  // namespace

extern "C" {

int refcxx_n_pars(int model) { return model == 0 ? 32 : model == 1 ? 8 : model == 2 ? 2 : 7; }
void refcxx_set_rel_error(double e) { g_rel_error = e; }

// One STEP 1 + STEP 2 pass with all parameters active.  jac: [n][n_par] row-major (lm_solver.cpp:315-317), res: [n], JTJ: [n_par^2],
// JTres: [n_par]; seconds[0] = the Jacobian loop, seconds[1] = dsyrk + dgemv.  Returns 0.
int refcxx_sweep(int model, long n, const double* x, const double* y, const double* sigma, const double* pars, int n_threads,
                 double* jac_out, double* res_out, double* JTJ, double* JTres, double* seconds) {
  const int np = refcxx_n_pars(model);
  gadfit::FitFunction f(pick(model));
  for (int j = 0; j < np; j++) f.par(j) = AdVar(pars[j], 0.0, 0.0, gadfit::passive_idx);
  std::vector<double> jac((size_t)n * np), res((size_t)n), jtj((size_t)np * np), jtr((size_t)np);
  std::vector<double> adjoints;
  if (model == 2) gadfit::initIntegration();       // (the rule's tables and the per-thread workspaces: what LMsolver's users call once)
  const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel num_threads(n_threads) firstprivate(f) private(adjoints)
  {
    for (int j = 0; j < np; j++) { f.activateParReverse(j, j); gadfit::addADSeed(f.par(j)); }
#pragma omp for nowait
    for (long i = 0; i < n; i++) {
      const double r = (y[i] - f(x[i]).val) / sigma[i];
      res[(size_t)i] = r;
      gadfit::returnSweep(np - 1, adjoints);
      for (int j = 0; j < np; j++) jac[(size_t)i * np + j] = adjoints[j] / sigma[i];
    }
  }
  const auto t1 = std::chrono::steady_clock::now();
  omp_set_num_threads(n_threads);
  gadfit::dsyrk('t', np, (int)n, jac, jtj);
  gadfit::dgemv('t', (int)n, np, jac, res, jtr);
  const auto t2 = std::chrono::steady_clock::now();
  if (jac_out) for (size_t k = 0; k < jac.size(); k++) jac_out[k] = jac[k];
  if (res_out) for (size_t k = 0; k < res.size(); k++) res_out[k] = res[k];
  if (JTJ) for (size_t k = 0; k < jtj.size(); k++) JTJ[k] = jtj[k];
  if (JTres) for (size_t k = 0; k < jtr.size(); k++) JTres[k] = jtr[k];
  if (seconds) { seconds[0] = std::chrono::duration<double>(t1 - t0).count(); seconds[1] = std::chrono::duration<double>(t2 - t1).count(); }
  return 0;
}

// STEP 3's second directional derivatives: the loop of LMsolver::computeDeltas (lm_solver.cpp:360-380) -- every parameter active in
// forward mode with seed delta1[j], omega_i = f(x_i).dd / sigma_i (the C++ side's sign: the Fortran side and the oracle carry -dd * w)
int refcxx_omega(int model, long n, const double* x, const double* sigma, const double* pars, const double* delta1, int n_threads, double* omega) {
  const int np = refcxx_n_pars(model);
  gadfit::FitFunction f(pick(model));
  for (int j = 0; j < np; j++) f.par(j) = AdVar(pars[j], 0.0, 0.0, gadfit::passive_idx);
  if (model == 2) gadfit::initIntegration();
#pragma omp parallel num_threads(n_threads) firstprivate(f)
  {
    for (int j = 0; j < np; j++) f.activateParForward(j, delta1[j]);
#pragma omp for nowait
    for (long i = 0; i < n; i++) omega[i] = f(x[i]).dd / sigma[i];
  }
  return 0;
}

// chi2 with all parameters passive (lm_solver.cpp:513-529)
double refcxx_chi2(int model, long n, const double* x, const double* y, const double* sigma, const double* pars, int n_threads, double* seconds) {
  const int np = refcxx_n_pars(model);
  gadfit::FitFunction f(pick(model));
  for (int j = 0; j < np; j++) f.par(j) = AdVar(pars[j], 0.0, 0.0, gadfit::passive_idx);
  double sum = 0.0;
  if (model == 2) gadfit::initIntegration();
  const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel num_threads(n_threads) firstprivate(f)
  {
#pragma omp for reduction(+ : sum) nowait
    for (long i = 0; i < n; i++) {
      const double r = (y[i] - f(x[i]).val) / sigma[i];
      sum += r * r;
    }
  }
  if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return sum;
}


// ---- a model TAPE (include/gadfit_tape.h: what the Python tracer and the Fortran recorder hand to the library and to the oracle)
// evaluated through the REFERENCE'S OWN AdVar operators: every elemental of the operator set, in whichever (advar, advar) / (advar,
// real) / (real, advar) variant the operands' static types select, so that the oracle's restated elementals can be held against the
// reference's on random expressions (tests/test_oracle_random_models_vs_reference_cxx.py).  eval() tapes without integrate().
namespace {
struct TVal { bool real; double r; AdVar a; };
bool g_integration_ready = false;

// sub-tape `sub` of t: 0 = eval() (X = x, PARAM = pars), an integrand otherwise (IVAR = ivar, IPARAM = pars)
AdVar tape_eval_sub(const gfh_tape* t, const int sub, const double x, const AdVar& ivar, const std::vector<AdVar>& p, bool* ok) {
  const gfh_subtape& st = t->sub[sub];
  std::vector<TVal> v((size_t)st.n_nodes);
  const AdVar none { 0.0, 0.0, 0.0, gadfit::passive_idx };
  auto as_advar = [&](const TVal& q) { return q.real ? AdVar(q.r, 0.0, 0.0, gadfit::passive_idx) : q.a; };
  for (int k = 0; k < st.n_nodes; k++) {
    const gfh_node& n = st.nodes[k];
    TVal& o = v[(size_t)k];
    o.real = (n.flags & GFH_F_REAL) != 0; o.r = 0.0; o.a = none;
    const TVal* A = (n.a >= 0 && n.a < k) ? &v[(size_t)n.a] : nullptr;
    const TVal* B = (n.b >= 0 && n.b < k) ? &v[(size_t)n.b] : nullptr;
    switch (n.op) {
      case GFH_CONST: o.r = n.c; break;
      case GFH_X: o.r = x; break;
      case GFH_PARAM: case GFH_IPARAM: o.a = p[(size_t)n.a]; break;
      case GFH_IVAR: o.a = ivar; break;
      case GFH_LIFT: o.a = AdVar(A->r, 0.0, 0.0, gadfit::passive_idx); break;
      case GFH_NEG: o.r = -A->r; break;
      case GFH_VAL: o.r = A->a.val; break;
      case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW:
        if (A->real && B->real) {
          o.r = n.op == GFH_ADD ? A->r + B->r : n.op == GFH_SUB ? A->r - B->r : n.op == GFH_MUL ? A->r * B->r : n.op == GFH_DIV ? A->r / B->r : std::pow(A->r, B->r);
        } else if (!A->real && !B->real) {
          o.a = n.op == GFH_ADD ? A->a + B->a : n.op == GFH_SUB ? A->a - B->a : n.op == GFH_MUL ? A->a * B->a : n.op == GFH_DIV ? A->a / B->a : pow(A->a, B->a);
        } else if (!A->real) {
          o.a = n.op == GFH_ADD ? A->a + B->r : n.op == GFH_SUB ? A->a - B->r : n.op == GFH_MUL ? A->a * B->r : n.op == GFH_DIV ? A->a / B->r : pow(A->a, B->r);
        } else {
          o.a = n.op == GFH_ADD ? A->r + B->a : n.op == GFH_SUB ? A->r - B->a : n.op == GFH_MUL ? A->r * B->a : n.op == GFH_DIV ? A->r / B->a : pow(A->r, B->a);
        }
        break;
      case GFH_POWI:      // a ** n, integer n in b (the C++ side has no integer form of its own: pow(AdVar, Number))
        if (A->real) o.r = std::pow(A->r, (double)n.b); else o.a = pow(A->a, n.b);
        break;
#define GFH_UN(OP_, FN_) case OP_: if (A->real) o.r = std::FN_(A->r); else o.a = FN_(A->a); break;
      GFH_UN(GFH_EXP, exp) GFH_UN(GFH_SQRT, sqrt) GFH_UN(GFH_LOG, log) GFH_UN(GFH_SIN, sin) GFH_UN(GFH_COS, cos) GFH_UN(GFH_TAN, tan)
      GFH_UN(GFH_ASIN, asin) GFH_UN(GFH_ACOS, acos) GFH_UN(GFH_ATAN, atan) GFH_UN(GFH_SINH, sinh) GFH_UN(GFH_COSH, cosh) GFH_UN(GFH_TANH, tanh)
      GFH_UN(GFH_ASINH, asinh) GFH_UN(GFH_ACOSH, acosh) GFH_UN(GFH_ATANH, atanh) GFH_UN(GFH_ERF, erf)
#undef GFH_UN
      case GFH_ABS: if (A->real) o.r = std::fabs(A->r); else o.a = abs(A->a); break;
      case GFH_GUARD_GT: case GFH_GUARD_LT: break;       // (a comparison on the recorded path: no value)
      case GFH_INTEGRATE: {
        // integrate(f, pars, lower, upper [, rel_error, abs_error]) through the reference's own gadfit::integrate (numerical_integration.cpp:
        // 242-310 and its AdVar-bound forms): finite bounds only (the C++ side has no infinite ones), the 15-point rule (its only one)
        const gfh_integral& in = t->integrals[n.a];
        if (in.lower_inf || in.upper_inf || (t->gk_points && t->gk_points != 15)) { *ok = false; return none; }
        if (!g_integration_ready) { gadfit::initIntegration(gadfit::default_workspace_size, 2); g_integration_ready = true; }
        std::vector<AdVar> q((size_t)in.n_ipars);
        for (int j = 0; j < in.n_ipars; j++) q[(size_t)j] = as_advar(v[(size_t)t->ipar_nodes[in.ipar_off + j]]);
        const int isub = in.integrand;
        gadfit::integrandSignature f = [t, isub, x, ok](const std::vector<AdVar>& qq, const AdVar& tt) { return tape_eval_sub(t, isub, x, tt, qq, ok); };
        const double rel = in.rel_error >= 0 ? in.rel_error : (in.depth <= 1 ? t->rel_error_outer : t->rel_error_inner);
        const double abse = in.abs_error >= 0 ? in.abs_error : 0.0;
        const TVal& lo = v[(size_t)in.lower]; const TVal& hi = v[(size_t)in.upper];
        if (lo.real && hi.real) o.a = gadfit::integrate(f, q, lo.r, hi.r, rel, abse);
        else if (lo.real) o.a = gadfit::integrate(f, q, lo.r, hi.a, rel, abse);
        else if (hi.real) o.a = gadfit::integrate(f, q, lo.a, hi.r, rel, abse);
        else o.a = gadfit::integrate(f, q, lo.a, hi.a, rel, abse);
        break;
      }
      default: *ok = false; return none;                 // (auxiliary columns: not this interpreter's)
    }
  }
  const TVal& y = v[(size_t)st.result];
  return y.real ? AdVar(y.r, 0.0, 0.0, gadfit::passive_idx) : y.a;
}

AdVar tape_eval(const gfh_tape* t, const double x, const std::vector<AdVar>& p, bool* ok) {
  *ok = true;
  const AdVar none { 0.0, 0.0, 0.0, gadfit::passive_idx };
  return tape_eval_sub(t, 0, x, none, p, ok);
}
}  // namespace

// value and reverse-mode gradient (one entry per active parameter, in order) of the tape's eval() at x; returns 0, or 1 for a tape
// this interpreter does not take
int refcxx_tape_reverse(const gfh_tape* t, double x, const double* pars, const int* active_mask, double* val, double* grad) {
  std::vector<AdVar> p((size_t)t->n_pars);
  int na = 0;
  for (int j = 0; j < t->n_pars; j++) {
    p[(size_t)j] = AdVar(pars[j], 0.0, 0.0, active_mask[j] ? na : gadfit::passive_idx);
    if (active_mask[j]) { gadfit::addADSeed(p[(size_t)j]); na++; }
  }
  bool ok = true;
  const AdVar y = tape_eval(t, x, p, &ok);
  *val = y.val;
  for (int j = 0; j < na; j++) grad[j] = 0.0;
  if (na) {
    // (a passive result -- it does not depend on an active parameter, whatever else was recorded on the way: returnSweep seeds the LAST
    // forward value, which is then not the result; the sweep still runs, to rewind the tape)
    std::vector<double> adjoints;
    gadfit::returnSweep(na - 1, adjoints);
    if (y.idx > gadfit::passive_idx) for (int j = 0; j < na; j++) grad[j] = adjoints[(size_t)j];
  }
  return ok ? 0 : 1;
}

// forward mode: out3 = {val, d, dd} with the active parameters seeded by dseed[j] (dd = 0)
int refcxx_tape_forward(const gfh_tape* t, double x, const double* pars, const int* active_mask, const double* dseed, double* out3) {
  std::vector<AdVar> p((size_t)t->n_pars);
  for (int j = 0; j < t->n_pars; j++)
    p[(size_t)j] = active_mask[j] ? AdVar(pars[j], dseed[j], 0.0, gadfit::forward_active_idx) : AdVar(pars[j], 0.0, 0.0, gadfit::passive_idx);
  bool ok = true;
  const AdVar y = tape_eval(t, x, p, &ok);
  out3[0] = y.val; out3[1] = y.idx == gadfit::forward_active_idx ? y.d : 0.0; out3[2] = y.idx == gadfit::forward_active_idx ? y.dd : 0.0;
  return ok ? 0 : 1;
}


// the damped normal-equation solve through the reference's own vendored Cholesky (lapack_fallback.cpp: dpptrf + dpptrs, as
// LMsolver::computeDeltas calls them, lm_solver.cpp:351-354): a = n x n symmetric positive definite (row-major = column-major), b = right-hand side, overwritten
int refcxx_potr(int n, const double* a, double* b) {
  std::vector<double> A(a, a + (size_t)n * n), B(b, b + n);
  gadfit::dpptrf(n, A);
  gadfit::dpptrs(n, A, B);
  for (int i = 0; i < n; i++) b[i] = B[(size_t)i];
  return 0;
}

}  // extern "C"
