/* gadfit_oracle.h -- CPU ORACLE for the gadfit LM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm for the hot path
 * (raullaasner/gadfit v2.0.1, Fortran side): operator-overloading AD with a per-point
 * reverse tape (fortran/gadfit/automatic_differentiation.F90), forward-mode (val,d,dd)
 * arithmetic, adaptive Gauss-Kronrod quadrature through AD
 * (fortran/gadfit/numerical_integration.F90) and the gadf_fit Levenberg-Marquardt driver
 * (fortran/gadfit/gadfit.F90:502-1035).  Every function cites the lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library -- as the checker / reported baseline, never as the product path.
 *
 * Parity status: PINNED.  tests/test_oracle_goldens.py checks this code against the
 * reference's own known-answer tests (fortran/tests/ad_forward_mode.F90,
 * ad_reverse_mode.F90, 1_gaussian.F90, 2_integral_single.F90, 3_integral_double.F90,
 * 4_multiple_curves.F90).  The reference itself is not buildable here under the rules
 * (Fortran side needs coarrays, which flang 22 cannot lower; C++ side needs spdlog), so
 * there is no oracle/_ref.
 */
#ifndef GADFIT_ORACLE_H
#define GADFIT_ORACLE_H
#include <stdint.h>
#include "../include/gadfit_tape.h"

#ifdef __cplusplus
extern "C" {
#endif

/* data_error_type, gadfit.F90:45-48 */
enum { ORC_NONE = 0, ORC_SQRT_Y = 1, ORC_PROPTO_Y = 2, ORC_INVERSE_Y = 3, ORC_USER = 4 };

/* gadf_fit optional arguments (gadfit.F90:502-511).  has_* mirrors present(). */
typedef struct orc_fit_options {
  double lambda, lam_up, lam_down, accth, grad_chi2, cos_phi, rel_error, rel_error_global,
         chi2_rel, chi2_abs;
  int has_lambda, has_lam_up, has_lam_down, has_accth, has_grad_chi2, has_cos_phi,
      has_rel_error, has_rel_error_global, has_chi2_rel, has_chi2_abs;
  const double* DTD_min; /* NULL = absent; length dim */
  int lam_incs, has_lam_incs;
  int uphill, has_uphill;
  int max_iter, has_max_iter;
  int damp_max, has_damp_max;
  int nielsen, has_nielsen;
  int umnigh, has_umnigh;
  int n_images;          /* emulate num_images(): partition + co_sum order; >=1 */
  double umnigh_a;       /* in/out: the implicitly SAVEd local (gadfit.F90:515) */
} orc_fit_options;

typedef struct orc_problem {
  const gfh_tape* tape;
  int n_datasets;
  const int64_t* data_positions; /* n_datasets+1 offsets, 0-based, [0]=0 */
  const double* x; const double* y; const double* w; /* weights as used: (y-f)*w */
  int n_pars;
  double* pars;            /* [n_datasets][n_pars], in/out */
  int n_active;
  const int32_t* active_pars; /* 0-based parameter indices, compacted, ascending as given */
  const int32_t* is_global;   /* [n_pars] */
  int loss;                   /* robust cost of the C++ solver (lm_solver.h:76-83, lm_solver.cpp:255-284): 0 linear, 1 cauchy, 2 huber */
  const double* aux;          /* auxiliary per-point columns [tape->n_aux][N] (GFH_AUX nodes) or NULL */
  int finite_diff;            /* use_ad = .false. (gadfit.F90:583-584, 600): parameters passive, grad_finite / dir_deriv_2nd_finite */
  /* A branching eval() (the reference runs the user's code at every point, gadfit.F90:679-690, and its `if`s on advar compare
   * val, AD:315-395).  Each recorded path through eval() is one tape whose comparisons are guard nodes (gadfit_tape.h); THE path
   * of a point is the tape all of whose guards come out as recorded when it is evaluated at that point with the current
   * parameters.  n_variants = 0: `tape` alone.  hint (or NULL): per point, the tape to prefer among those whose guards hold --
   * for tapes that differ without a guard (control flow on plain reals, which only the recording host can see). */
  int n_variants;
  const gfh_tape* const* variants;
  const double* hint;
} orc_problem;

/* Per-iteration record for fixtures/tests (all optional, may be NULL). */
typedef struct orc_fit_result {
  int iterations;
  int dim;
  double lambda;      /* final */
  double chi2;        /* old_chi2 at exit */
  int dof;
  int exit_reason;    /* 0 max_iter,1 chi2_abs,2 chi2_rel,3 grad,4 cos_phi,5 rel_error,6 rel_error_global,7 lambda increased */
  int n_sweeps, n_chi2, n_omega; /* requests made */
  /* first-iteration snapshots, caller-allocated (dim*dim, dim, dim, dim) or NULL */
  double* JTJ0; double* JTres0; double* delta1_0; double* delta2_0;
  double chi2_0;
} orc_fit_result;

const char* orc_last_error(void);

/* init_weights, gadfit.F90:445-470.  sigma only read for ORC_USER. */
void orc_init_weights(int error_type, int64_t n, const double* y, const double* sigma, double* w);

/* Jacobian_indices and dim, gadfit.F90:615-631.  jac_idx [n_datasets][n_active], 0-based columns. */
int orc_jacobian_indices(int n_datasets, int n_active, const int32_t* active_pars,
                         const int32_t* is_global, int32_t* jac_idx);

/* img_bounds for image `image` (0-based) of n_images, even weights (gadfit.F90:977-1002).
 * bounds[n_datasets+1], 0-based half-open sub-ranges into the concatenated arrays. */
void orc_img_bounds(int n_images, int image, int n_datasets, const int64_t* data_positions,
                    int64_t* bounds);

/* STEP 1+2 for all points (gadfit.F90:675-701): JTJ dim*dim column-major, JTres dim.
 * Optionally returns res (N) and JacobianT (dim x N, parameter fastest). */
int orc_sweep(const orc_problem* p, int n_images, double* JTJ, double* JTres,
              double* res_out, double* JT_out);
/* chi2(), gadfit.F90:1015-1034 */
int orc_chi2(const orc_problem* p, int n_images, double* chi2, double* res_out);
/* STEP 3 omega vector and J^T omega (gadfit.F90:715-735); delta1 length dim; JT (dim x N) from orc_sweep */
int orc_omega(const orc_problem* p, const double* delta1, const double* JT, double* omega_out,
              double* JTomega);

/* The full gadf_fit (gadfit.F90:502-1035). */
int orc_fit(orc_problem* p, orc_fit_options* o, orc_fit_result* r);

/* Single-point AD probes, for the ad_forward_mode / ad_reverse_mode goldens.
 * active[n_pars]: nonzero = active.  grad: adjoints of active params in parameter order. */
int orc_eval_reverse(const gfh_tape* t, double x, const double* pars, const int32_t* active,
                     double* val, double* grad);
/* d_seed/dd_seed per parameter (only read for active ones). out3 = val, d, dd */
int orc_eval_forward(const gfh_tape* t, double x, const double* pars, const int32_t* active,
                     const double* d_seed, const double* dd_seed, double* out3);

/* potr_f08 = dpotrf('U') + dpotrs (gadfit_linalg.F90:36-57); a is n*n column-major, destroyed */
int orc_potr(int n, double* a, double* b);

/* statistics of the last orc_eval_* / sweep: max tape sizes (ad_memory_report analogue) */
void orc_tape_stats(int* max_trace, int* max_index, int* max_const);
/* quadrature work since the last call: integrand evaluations of the bisections, of the final passes, calls, final intervals */
void orc_quad_counters(long long* out4);

#ifdef __cplusplus
}
#endif
#endif
