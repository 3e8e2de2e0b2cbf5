// lm.cpp -- the Levenberg-Marquardt driver gadf_fit (gadfit.F90:502-1035) on the host,
// requesting sweeps from the device.  By north_star the damped normal-equation solve and the
// lambda logic stay on the host; they are restated exactly because they decide which device
// passes are requested.  Host work per iteration is O(dim^3) on a
// dim x dim matrix; everything N-sized lives in HBM behind gfh_sweep/gfh_chi2/gfh_omega.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "context.h"

using namespace gfh;

// potr_f08 (gadfit_linalg.F90:36-57) = dpotrf('U') then dpotrs, column-major, in place.  Split so a
// factor can serve two right-hand sides: gadf_fit factorises the SAME matrix twice per accelerated
// iteration (gadfit.F90:712-713 and 737-738); reusing the factor gives bitwise the same delta2.
static int potrf_upper(int n, double* a) {
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

struct Fit {
  gfh_ctx* c; double* pars; int na, np, nd, dim;
  const int32_t* active; std::vector<int32_t> jac;
  std::vector<double> JTJ, JTres, DTD, delta1, delta2, old_delta1, lin, JTomega, old_pars;
  std::vector<double> nextJTJ, nextJTres;     // look-ahead sweep results at the trial point

  double dtd(const std::vector<double>& a, const std::vector<double>& b) const {
    double s = 0; for (int i = 0; i < dim; i++) s += a[i] * (DTD[i] * b[i]); return s;   // dot(a, matmul(DTD,b)), DTD diagonal
  }
  int solve(const std::vector<double>& rhs, std::vector<double>& out, double lambda) {
    out = rhs;                                                         // gadfit.F90:711-713
    for (int col = 0; col < dim; col++)
      for (int row = 0; row < dim; row++)
        lin[(size_t)col * dim + row] = JTJ[(size_t)col * dim + row] + (row == col ? lambda * DTD[col] : 0.0);
    if (potrf_upper(dim, lin.data())) return fail(c, gfh_last_error(nullptr));
    potrs_upper(dim, lin.data(), out.data());
    return 0;
  }
  // second right-hand side against the factor left in `lin` by solve() (same JTJ, lambda, DTD)
  void solve_again(const std::vector<double>& rhs, std::vector<double>& out) {
    out = rhs;
    potrs_upper(dim, lin.data(), out.data());
  }
  void restore() { for (int d = 0; d < nd; d++) for (int j = 0; j < na; j++) pars[d * np + active[j]] = old_pars[d * na + j]; }
  void save() { for (int d = 0; d < nd; d++) for (int j = 0; j < na; j++) old_pars[d * na + j] = pars[d * np + active[j]]; }
};

}  // namespace

extern "C" int gfh_fit(gfh_ctx* c, double* pars, int na, const int32_t* active, const int32_t* is_global,
                       gfh_fit_options* o, gfh_fit_result* r) {
  if (!c) return 1;
  if (c->device < 0) return fail(c, "no GPU bound to this context (libgadfit_hip has no CPU fallback)");
  if (!c->has_model || !c->nd) return fail(c, "gfh_fit: model and data must be set first");
  if (na < 1) return fail(c, "There are no active parameters.");                                     // gadfit.F90:602-603
  gfh_fit_options defaults; memset(&defaults, 0, sizeof defaults); defaults.umnigh_a = 0.5;
  if (!o) o = &defaults;
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
  f.JTJ.assign((size_t)dim * dim, 0); f.JTres.assign(dim, 0); f.DTD.assign(dim, 0); f.delta1.assign(dim, 0);
  f.delta2.assign(dim, 0); f.old_delta1.assign(dim, 0); f.lin.assign((size_t)dim * dim, 0); f.JTomega.assign(dim, 0);
  f.old_pars.assign((size_t)na * f.nd, 0);
  if (o->DTD_min) for (int i = 0; i < dim; i++) f.DTD[i] = o->DTD_min[i];                           // gadfit.F90:641-646
  long long dof_ll = (long long)c->n_total - dim;                                                    // gadfit.F90:648-657
  if (dof_ll < 0) return fail(c, "More independent fitting parameters than data points.");
  const double dof = dof_ll == 0 ? 1.0 : (double)dof_ll;
  f.save();
  gfh_fit_result local; if (!r) r = &local;
  memset(r, 0, sizeof *r); r->dim = dim; r->dof = (int)(dof_ll == 0 ? 1 : dof_ll); r->exit_reason = -1;
  int iterations = 0;
  double old_chi2 = 0, new_chi2 = 0, old_old_chi2 = 0, acc_ratio = 0, beta = 0, sweep_chi2 = 0;
  auto t0 = std::chrono::steady_clock::now();
  auto finish = [&](int rc) {
    r->iterations = iterations; r->lambda = lambda; r->chi2 = old_chi2;
    r->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
  };
  // keep_jacobian = 2: J goes to HBM only if this fit reads it back (the grad_chi2 / cos_phi tests, and
  // STEP 3's J^T omega where gfh_k_omega_jt is not available: models with integrate(), robust losses); the fused kernel forms J^T J / J^T r from registers either way
  if (c->keep_jacobian == 2)
    set_store_j(c, (o->has_accth && o->accth > 1.17549435e-38 && omega_needs_jacobian(c)) || o->has_grad_chi2 || o->has_cos_phi);
  if (gfh_set_active(c, active, na, f.jac.data(), dim)) return finish(1);
  if (gfh_chi2(c, pars, &old_chi2)) return finish(1);                                               // gadfit.F90:670
  r->n_chi2++;
  // Look-ahead (gadfit_hip.h, gfh_set_lookahead): the fused sweep already returns sum r^2, so the
  // FIRST trial chi2() of an iteration (gadfit.F90:753) is taken from a sweep at the trial point;
  // when the step is accepted that sweep IS the next iteration's STEP 1+2 (same parameters, same
  // kernel, same numbers), so an accepted iteration costs one N-sized pass instead of two.  Armed
  // while the previous first trial was accepted.  Off when the convergence tests read the device
  // J/res pair the reference has at that point (old J, new res: gadfit.F90:849-850, 865-873).
  const bool la_ok = c->lookahead != 0 && !o->has_grad_chi2 && !o->has_cos_phi && c->gen.loss == 0;
  bool la_armed = la_ok, have_next = false;
  if (la_ok) { f.nextJTJ.assign((size_t)dim * dim, 0); f.nextJTres.assign(dim, 0); }
  for (;;) {
    // STEP 1 + 2 (gadfit.F90:675-701)
    if (have_next) { f.JTJ.swap(f.nextJTJ); f.JTres.swap(f.nextJTres); have_next = false; }
    else if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.JTJ.data(), f.JTres.data(), &sweep_chi2)) return finish(1);
    r->n_sweeps++;
    for (int i = 0; i < dim; i++) {                                                                 // gadfit.F90:702-710
      const double d = f.JTJ[(size_t)i * dim + i];
      if (o->has_damp_max && !o->damp_max) f.DTD[i] = d; else f.DTD[i] = f.DTD[i] > d ? f.DTD[i] : d;
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
    for (int i = 1; i <= lam_incs + 1; i++) {                                                       // STEP 4, gadfit.F90:752-819
      // no look-ahead in the iteration that max_iter ends anyway (its Jacobian would not be used)
      const bool spec = la_armed && i == 1 && !(o->has_max_iter && iterations + 1 >= o->max_iter);
      if (spec) {
        if (gfh_sweep(c, pars, active, na, f.jac.data(), dim, f.nextJTJ.data(), f.nextJTres.data(), &new_chi2)) return finish(1);
        r->n_lookahead++;
      } else if (gfh_chi2(c, pars, &new_chi2)) return finish(1);
      r->n_chi2++;
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
    la_armed = la_ok && first_accepted;
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
  if (c->device < 0) return fail(c, "no GPU bound to this context (libgadfit_hip has no CPU fallback)");
  if (!c->has_model || !c->nd) return fail(c, "gfh_lm_iterate: model and data must be set first");
  Fit f; f.c = c; f.pars = pars; f.na = na; f.np = c->model.n_pars; f.nd = c->nd; f.active = active;
  f.jac.resize((size_t)f.nd * na);
  const int dim = f.dim = gfh_jacobian_indices(f.nd, na, active, is_global, f.jac.data());
  f.JTJ.assign((size_t)dim * dim, 0); f.JTres.assign(dim, 0); f.DTD.assign(DTD, DTD + dim); f.delta1.assign(dim, 0);
  f.lin.assign((size_t)dim * dim, 0); f.old_pars.assign((size_t)na * f.nd, 0);
  f.nextJTJ.assign((size_t)dim * dim, 0); f.nextJTres.assign(dim, 0);
  double lambda = state3[0], old_chi2 = state3[1], sweep_chi2 = 0, new_chi2 = 0;
  if (gfh_set_active(c, active, na, f.jac.data(), dim)) return 1;
  if (old_chi2 < 0 && gfh_chi2(c, pars, &old_chi2)) return 1;
  // look-ahead as in gfh_fit: the trial chi2 is the sum r^2 of a sweep at the trial point, which
  // an accepted step hands to the next iteration.  The hand-over does not cross calls: the first
  // iteration of every call sweeps, and a look-ahead of the last iteration is not started.
  const bool la_ok = c->lookahead != 0 && c->gen.loss == 0;
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
