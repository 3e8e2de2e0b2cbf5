/* gadfit_oracle.c -- CPU ORACLE (test infrastructure, see gadfit_oracle.h).
 *
 * Restates, in plain C, the reference's per-data-point AD sweep and LM driver:
 *   AD   = /root/reference/fortran/gadfit/automatic_differentiation.F90
 *   NI   = /root/reference/fortran/gadfit/numerical_integration.F90
 *   GF   = /root/reference/fortran/gadfit/gadfit.F90
 *   LA   = /root/reference/fortran/gadfit/gadfit_linalg.F90 (+ LAPACK dpotf2/dpotrs semantics)
 * The user's eval() is replaced by walking the model tape (include/gadfit_tape.h) and
 * calling the restated elementals in program order -- exactly the call sequence operator
 * overloading produces in the reference.
 */
#include "gadfit_oracle.h"
#include "gk_tables.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ errors */
static char g_err[512];
const char* orc_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ------------------------------------------------------------------ AD state */
/* AD:36-50 op codes */
enum { ADD_A_A = -1, ADD_SUBTRACT_A_R = -2, SUBTRACT_A_A = -3, SUBTRACT_R_A = -4,
       MULTIPLY_A_A = -5, MULTIPLY_DIVIDE_A_R = -6, DIVIDE_A_A = -7, DIVIDE_R_A = -8,
       POWER_A_A = -9, POWER_A_R = -10, POWER_R_A = -11, POWER_INTEGER = -12, ABS_A = -13,
       EXP_A = -14, SQRT_A = -15, LOG_A = -16, SIN_A = -17, COS_A = -18, TAN_A = -19,
       ASIN_A = -20, ACOS_A = -21, ATAN_A = -22, SINH_A = -23, COSH_A = -24, TANH_A = -25,
       ASINH_A = -26, ACOSH_A = -27, ATANH_A = -28, ERF_A = -29,
       INT_LOWER_BOUND = -31, INT_UPPER_BOUND = -32, INT_BOTH_BOUNDS = -33 };

/* AD:65-68 */
typedef struct advar { double val, d, dd; int index; } advar;

/* AD:233-241 (module state).  1-based like the reference: slot 0 unused. */
static double *forward_values, *adjoints, *ad_constants;
static int *trace;
static int cap_sweep, cap_trace, cap_const;
static int trace_count, index_count, const_count;
static int reverse_mode = 0;
static int max_trace_count, max_index_count, max_const_count;
static int ad_overflow;

/* sqrtpi, gadf_constants.F90:29-33 */
static const double SQRTPI = 1.772453850905516027298167483341145;

static void ad_reserve(int sweep) {
  /* ad_init_reverse, AD:272-313: sweep, 4*sweep trace, sweep/2 constants.  The oracle
   * simply grows on demand (the reference errors out after the fact, AD:1481-1484). */
  if (sweep <= cap_sweep) return;
  forward_values = (double*)realloc(forward_values, sizeof(double) * (sweep + 1));
  adjoints = (double*)realloc(adjoints, sizeof(double) * (sweep + 1));
  ad_constants = (double*)realloc(ad_constants, sizeof(double) * (sweep + 1));
  trace = (int*)realloc(trace, sizeof(int) * (4 * sweep + 4));
  cap_sweep = sweep; cap_trace = 4 * sweep; cap_const = sweep;
}
static inline void ad_room(void) {
  if (index_count + 2 > cap_sweep || trace_count + 8 > cap_trace || const_count + 4 > cap_const)
    ad_reserve(cap_sweep ? 2 * cap_sweep : 10000);
}
void orc_tape_stats(int* a, int* b, int* c) { *a = max_trace_count; *b = max_index_count; *c = max_const_count; }

static inline advar passive(double v) { advar y = { v, 0.0, 0.0, 0 }; return y; }

/* Fortran x**n with integer n: repeated multiplication (compiler intrinsic). */
static double powi(double x, int n) {
  if (n == 0) return 1.0;
  unsigned m = n < 0 ? (unsigned)(-(long)n) : (unsigned)n;
  double r = 1.0, b = x;
  while (m) { if (m & 1u) r *= b; m >>= 1; if (m) b *= b; }
  return n < 0 ? 1.0 / r : r;
}

/* record helpers */
static inline void rec_bin(advar* y, const advar* x1, const advar* x2, int op) {
  ad_room();
  index_count++; y->index = index_count; forward_values[index_count] = y->val;
  trace[trace_count + 1] = x1->index; trace[trace_count + 2] = x2->index;
  trace[trace_count + 3] = y->index;  trace[trace_count + 4] = op; trace_count += 4;
}
static inline void rec_un(advar* y, const advar* x, int op) {
  ad_room();
  index_count++; y->index = index_count; forward_values[index_count] = y->val;
  trace[trace_count + 1] = x->index; trace[trace_count + 2] = y->index;
  trace[trace_count + 3] = op; trace_count += 3;
}
static inline void rec_un_c(advar* y, const advar* x, int op, double c) {
  ad_room();
  index_count++; const_count++; y->index = index_count; forward_values[index_count] = y->val;
  ad_constants[const_count] = c;
  trace[trace_count + 1] = x->index; trace[trace_count + 2] = y->index;
  trace[trace_count + 3] = op; trace_count += 3;
}

/* ------------------------------------------------------------------ elementals */
/* AD:481-500 */
static advar add_advar_real(advar x1, double x2) {
  advar y = passive(x1.val + x2);
  if (x1.index != 0) {
    if (reverse_mode) rec_un(&y, &x1, ADD_SUBTRACT_A_R);
    else { y.d = x1.d; y.dd = x1.dd; y.index = 1; }
  }
  return y;
}
/* AD:526-545 */
static advar add_real_advar(double x1, advar x2) {
  advar y = passive(x1 + x2.val);
  if (x2.index != 0) {
    if (reverse_mode) rec_un(&y, &x2, ADD_SUBTRACT_A_R);
    else { y.d = x2.d; y.dd = x2.dd; y.index = 1; }
  }
  return y;
}
/* AD:454-479 */
static advar add_advar_advar(advar x1, advar x2) {
  if (x1.index != 0 && x2.index != 0) {
    advar y = passive(x1.val + x2.val);
    if (reverse_mode) rec_bin(&y, &x1, &x2, ADD_A_A);
    else { y.d = x1.d + x2.d; y.dd = x1.dd + x2.dd; y.index = 1; }
    return y;
  } else if (x1.index != 0) return add_advar_real(x1, x2.val);
  else if (x2.index != 0) return add_real_advar(x1.val, x2);
  return passive(x1.val + x2.val);
}
/* AD:603-625 */
static advar subtract_advar_real(advar x1, double x2) {
  advar y = passive(x1.val - x2);
  if (x1.index != 0) {
    if (reverse_mode) rec_un(&y, &x1, ADD_SUBTRACT_A_R);
    else { y.d = x1.d; y.dd = x1.dd; y.index = 1; }
  }
  return y;
}
/* AD:648-667 */
static advar subtract_real_advar(double x1, advar x2) {
  advar y = passive(x1 - x2.val);
  if (x2.index != 0) {
    if (reverse_mode) rec_un(&y, &x2, SUBTRACT_R_A);
    else { y.d = -x2.d; y.dd = -x2.dd; y.index = 1; }
  }
  return y;
}
/* AD:571-596 */
static advar subtract_advar_advar(advar x1, advar x2) {
  if (x1.index != 0 && x2.index != 0) {
    advar y = passive(x1.val - x2.val);
    if (reverse_mode) rec_bin(&y, &x1, &x2, SUBTRACT_A_A);
    else { y.d = x1.d - x2.d; y.dd = x1.dd - x2.dd; y.index = 1; }
    return y;
  } else if (x1.index != 0) return subtract_advar_real(x1, x2.val);
  else if (x2.index != 0) return subtract_real_advar(x1.val, x2);
  return passive(x1.val - x2.val);
}
/* AD:720-741 */
static advar multiply_advar_real(advar x1, double x2) {
  advar y = passive(x1.val * x2);
  if (x1.index != 0) {
    if (reverse_mode) rec_un_c(&y, &x1, MULTIPLY_DIVIDE_A_R, x2);
    else { y.d = x1.d * x2; y.dd = x1.dd * x2; y.index = 1; }
  }
  return y;
}
/* AD:767-788 */
static advar multiply_real_advar(double x1, advar x2) {
  advar y = passive(x1 * x2.val);
  if (x2.index != 0) {
    if (reverse_mode) rec_un_c(&y, &x2, MULTIPLY_DIVIDE_A_R, x1);
    else { y.d = x1 * x2.d; y.dd = x1 * x2.dd; y.index = 1; }
  }
  return y;
}
/* AD:693-718 */
static advar multiply_advar_advar(advar x1, advar x2) {
  if (x1.index != 0 && x2.index != 0) {
    advar y = passive(x1.val * x2.val);
    if (reverse_mode) rec_bin(&y, &x1, &x2, MULTIPLY_A_A);
    else {
      y.d = x1.val * x2.d + x1.d * x2.val;
      y.dd = x1.val * x2.dd + 2 * x1.d * x2.d + x1.dd * x2.val;
      y.index = 1;
    }
    return y;
  } else if (x1.index != 0) return multiply_advar_real(x1, x2.val);
  else if (x2.index != 0) return multiply_real_advar(x1.val, x2);
  return passive(x1.val * x2.val);
}
/* AD:843-866: multiplies by the reciprocal, also when passive */
static advar divide_advar_real(advar x1, double x2) {
  double inv_x2 = 1 / x2;
  advar y = passive(x1.val * inv_x2);
  if (x1.index != 0) {
    if (reverse_mode) rec_un_c(&y, &x1, MULTIPLY_DIVIDE_A_R, inv_x2);
    else { y.d = x1.d * inv_x2; y.dd = x1.dd * inv_x2; y.index = 1; }
  }
  return y;
}
/* AD:892-913 */
static advar divide_real_advar(double x1, advar x2) {
  advar y = passive(x1 / x2.val);
  if (x2.index != 0) {
    if (reverse_mode) rec_un_c(&y, &x2, DIVIDE_R_A, x1);
    else {
      y.d = -y.val * x2.d / x2.val;
      y.dd = (-y.val * x2.dd - 2 * y.d * x2.d) / x2.val;
      y.index = 1;
    }
  }
  return y;
}
/* AD:814-841 */
static advar divide_advar_advar(advar x1, advar x2) {
  if (x1.index != 0 && x2.index != 0) {
    double inv_x2 = 1 / x2.val;
    advar y = passive(x1.val * inv_x2);
    if (reverse_mode) rec_bin(&y, &x1, &x2, DIVIDE_A_A);
    else {
      y.d = (x1.d - y.val * x2.d) * inv_x2;
      y.dd = (x1.dd - y.val * x2.dd - 2 * y.d * x2.d) * inv_x2;
      y.index = 1;
    }
    return y;
  } else if (x1.index != 0) return divide_advar_real(x1, x2.val);
  else if (x2.index != 0) return divide_real_advar(x1.val, x2);
  return passive(x1.val / x2.val);
}
/* AD:939-957 */
static advar abs_advar(advar x) {
  advar y = passive(fabs(x.val));
  if (x.index != 0) {
    if (reverse_mode) rec_un(&y, &x, ABS_A);
    else { double s = copysign(1.0, x.val); y.d = x.d * s; y.dd = x.dd * s; y.index = 1; }
  }
  return y;
}
/* AD:990-1012 */
static advar power_advar_real(advar x1, double x2) {
  advar y = passive(pow(x1.val, x2));
  if (x1.index != 0) {
    if (reverse_mode) rec_un_c(&y, &x1, POWER_A_R, x2);
    else {
      y.d = x1.d * x2 * pow(x1.val, x2 - 1);
      double inv_x1 = 1 / x1.val;
      y.dd = y.d * y.d / y.val + y.val * x2 * (x1.dd - x1.d * x1.d * inv_x1) * inv_x1;
      y.index = 1;
    }
  }
  return y;
}
/* AD:1061-1083 */
static advar power_real_advar(double x1, advar x2) {
  advar y = passive(pow(x1, x2.val));
  if (x2.index != 0) {
    if (reverse_mode) rec_un_c(&y, &x2, POWER_R_A, x1);
    else {
      double log_value = log(x1);
      y.d = y.val * x2.d * log_value;
      y.dd = y.d * y.d / y.val + y.val * x2.dd * log_value;
      y.index = 1;
    }
  }
  return y;
}
/* AD:959-988 */
static advar power_advar_advar(advar x1, advar x2) {
  if (x1.index != 0 && x2.index != 0) {
    advar y = passive(pow(x1.val, x2.val));
    if (reverse_mode) rec_bin(&y, &x1, &x2, POWER_A_A);
    else {
      double log_value = log(x1.val);
      y.d = y.val * x2.d * log_value + x1.d * x2.val * pow(x1.val, x2.val - 1);
      double inv_x1 = 1 / x1.val;
      y.dd = y.d * y.d / y.val + y.val * (x2.dd * log_value +
             (2 * x2.d * x1.d + x2.val * (x1.dd - x1.d * x1.d * inv_x1)) * inv_x1);
      y.index = 1;
    }
    return y;
  } else if (x1.index != 0) return power_advar_real(x1, x2.val);
  else if (x2.index != 0) return power_real_advar(x1.val, x2);
  return passive(pow(x1.val, x2.val));
}
/* AD:1033-1059: the integer exponent lives in the trace */
static advar power_advar_integer(advar x1, int x2) {
  advar y = passive(powi(x1.val, x2));
  if (x1.index != 0) {
    if (reverse_mode) {
      ad_room();
      index_count++; y.index = index_count; forward_values[index_count] = y.val;
      trace[trace_count + 1] = x1.index; trace[trace_count + 2] = x2;
      trace[trace_count + 3] = y.index;  trace[trace_count + 4] = POWER_INTEGER; trace_count += 4;
    } else {
      double inv_x1 = 1 / x1.val;
      y.d = y.val * x2 * x1.d * inv_x1;
      y.dd = y.d * y.d / y.val + y.val * x2 * (x1.dd - x1.d * x1.d * inv_x1) * inv_x1;
      y.index = 1;
    }
  }
  return y;
}

/* unary elementals AD:1110-1459.  fwd: computes d, dd from x and y (val already set) */
#define UNARY(name, OP, valexpr, fwdcode)                               \
  static advar name(advar x) {                                           \
    advar y = passive(valexpr);                                          \
    if (x.index != 0) {                                                  \
      if (reverse_mode) rec_un(&y, &x, OP);                              \
      else { fwdcode; y.index = 1; }                                     \
    }                                                                    \
    return y;                                                            \
  }
UNARY(exp_advar, EXP_A, exp(x.val), { y.d = x.d * y.val; y.dd = x.dd * y.val + x.d * y.d; })          /* AD:1110-1128 */
UNARY(sqrt_advar, SQRT_A, sqrt(x.val), { double inv_y = 1 / y.val; y.d = x.d / 2 * inv_y;
                                          y.dd = (x.dd * inv_y - y.d * x.d / x.val) / 2; })            /* AD:1130-1150 */
UNARY(log_advar, LOG_A, log(x.val), { double inv_x = 1 / x.val; y.d = x.d * inv_x;
                                       y.dd = (x.dd - x.d * y.d) * inv_x; })                           /* AD:1152-1172 */
UNARY(sin_advar, SIN_A, sin(x.val), { double c = cos(x.val); y.d = x.d * c;
                                       y.dd = x.dd * c - x.d * x.d * y.val; })                         /* AD:1174-1194 */
UNARY(cos_advar, COS_A, cos(x.val), { double s = -sin(x.val); y.d = x.d * s;
                                       y.dd = x.dd * s - x.d * x.d * y.val; })                         /* AD:1196-1216 */
UNARY(tan_advar, TAN_A, tan(x.val), { double t = 1 / cos(x.val); t = t * t; y.d = x.d * t;
                                       y.dd = x.dd * t + 2 * x.d * y.val * y.d; })                     /* AD:1218-1239 */
UNARY(asin_advar, ASIN_A, asin(x.val), { double t = 1 / sqrt(1 - x.val * x.val); y.d = x.d * t;
                                          y.dd = t * (x.dd + x.val * y.d * y.d); })                    /* AD:1241-1261 */
UNARY(acos_advar, ACOS_A, acos(x.val), { double t = -1 / sqrt(1 - x.val * x.val); y.d = x.d * t;
                                          y.dd = t * (x.dd + x.val * y.d * y.d); })                    /* AD:1263-1283 */
UNARY(atan_advar, ATAN_A, atan(x.val), { double t = 1 / (1 + x.val * x.val); y.d = x.d * t;
                                          y.dd = x.dd * t - 2 * x.val * y.d * y.d; })                  /* AD:1285-1305 */
UNARY(sinh_advar, SINH_A, sinh(x.val), { double c = cosh(x.val); y.d = x.d * c;
                                          y.dd = x.dd * c + y.val * x.d * x.d; })                      /* AD:1307-1327 */
UNARY(cosh_advar, COSH_A, cosh(x.val), { double s = sinh(x.val); y.d = x.d * s;
                                          y.dd = x.dd * s + y.val * x.d * x.d; })                      /* AD:1329-1349 */
UNARY(tanh_advar, TANH_A, tanh(x.val), { double ch = cosh(x.val); double t = 1 / (ch * ch); y.d = x.d * t;
                                          y.dd = x.dd * t - 2 * y.val * x.d * y.d; })                  /* AD:1351-1371 */
UNARY(asinh_advar, ASINH_A, asinh(x.val), { double t = 1 / sqrt(1 + x.val * x.val); y.d = x.d * t;
                                             y.dd = t * (x.dd - x.val * y.d * y.d); })                 /* AD:1373-1393 */
UNARY(acosh_advar, ACOSH_A, acosh(x.val), { double t = 1 / sqrt(x.val * x.val - 1); y.d = x.d * t;
                                             y.dd = t * (x.dd - x.val * y.d * y.d); })                 /* AD:1395-1415 */
UNARY(atanh_advar, ATANH_A, atanh(x.val), { double t = 1 / (1 - x.val * x.val); y.d = x.d * t;
                                             y.dd = (x.dd + 2 * x.val * x.d * y.d) * t; })             /* AD:1417-1437 */
UNARY(erf_advar, ERF_A, erf(x.val), { double t = 2 / SQRTPI * exp(-(x.val * x.val)); y.d = x.d * t;
                                       y.dd = (x.dd - 2 * x.d * x.d * x.val) * t; })                   /* AD:1439-1459 */

/* ------------------------------------------------------------------ ad_grad, AD:1476-1659 */
static void ad_grad(int num_parameters) {
  if (trace_count > max_trace_count) max_trace_count = trace_count;
  if (index_count > max_index_count) max_index_count = index_count;
  if (const_count > max_const_count) max_const_count = const_count;
  for (int k = 1; k < index_count; k++) adjoints[k] = 0.0;
  adjoints[index_count] = 1.0;
#define T(k) trace[i - (k)]
#define ADJ(k) adjoints[T(k)]
#define FV(k) forward_values[T(k)]
  int i = trace_count;
  while (i > 0) {
    switch (trace[i]) {
    case ADD_A_A: ADJ(3) = ADJ(3) + ADJ(1); ADJ(2) = ADJ(2) + ADJ(1); i -= 4; break;
    case ADD_SUBTRACT_A_R: ADJ(2) = ADJ(2) + ADJ(1); i -= 3; break;
    case SUBTRACT_A_A: ADJ(3) = ADJ(3) + ADJ(1); ADJ(2) = ADJ(2) - ADJ(1); i -= 4; break;
    case SUBTRACT_R_A: ADJ(2) = ADJ(2) - ADJ(1); i -= 3; break;
    case MULTIPLY_A_A:
      ADJ(3) = ADJ(3) + ADJ(1) * FV(2); ADJ(2) = ADJ(2) + ADJ(1) * FV(3); i -= 4; break;
    case MULTIPLY_DIVIDE_A_R:
      ADJ(2) = ADJ(2) + ADJ(1) * ad_constants[const_count]; const_count--; i -= 3; break;
    case DIVIDE_A_A:
      ADJ(3) = ADJ(3) + ADJ(1) / FV(2);
      ADJ(2) = ADJ(2) - ADJ(1) * FV(1) / FV(2); i -= 4; break;
    case DIVIDE_R_A:
      ADJ(2) = ADJ(2) - ADJ(1) * ad_constants[const_count] / FV(2) / FV(2);
      const_count--; i -= 3; break;
    case POWER_A_A:
      ADJ(3) = ADJ(3) + ADJ(1) * FV(2) * pow(FV(3), FV(2) - 1);
      ADJ(2) = ADJ(2) + ADJ(1) * log(FV(3)) * pow(FV(3), FV(2)); i -= 4; break;
    case POWER_A_R:
      ADJ(2) = ADJ(2) + ADJ(1) * ad_constants[const_count] * pow(FV(2), ad_constants[const_count] - 1);
      const_count--; i -= 3; break;
    case POWER_R_A:
      ADJ(2) = ADJ(2) + ADJ(1) * log(ad_constants[const_count]) * pow(ad_constants[const_count], FV(2));
      const_count--; i -= 3; break;
    case POWER_INTEGER:
      ADJ(3) = ADJ(3) + ADJ(1) * T(2) * powi(FV(3), T(2) - 1); i -= 4; break;
    case ABS_A:
      if (FV(2) < 0) ADJ(2) = ADJ(2) - ADJ(1); else ADJ(2) = ADJ(2) + ADJ(1);
      i -= 3; break;
    case EXP_A: ADJ(2) = ADJ(2) + ADJ(1) * FV(1); i -= 3; break;
    case SQRT_A: ADJ(2) = ADJ(2) + ADJ(1) / 2.0 / FV(1); i -= 3; break;
    case LOG_A: ADJ(2) = ADJ(2) + ADJ(1) / FV(2); i -= 3; break;
    case SIN_A: ADJ(2) = ADJ(2) + ADJ(1) * cos(FV(2)); i -= 3; break;
    case COS_A: ADJ(2) = ADJ(2) - ADJ(1) * sin(FV(2)); i -= 3; break;
    case TAN_A: { double c = cos(FV(2)); ADJ(2) = ADJ(2) + ADJ(1) / (c * c); i -= 3; break; }
    case ASIN_A: ADJ(2) = ADJ(2) + ADJ(1) / sqrt(1 - FV(2) * FV(2)); i -= 3; break;
    case ACOS_A: ADJ(2) = ADJ(2) - ADJ(1) / sqrt(1 - FV(2) * FV(2)); i -= 3; break;
    case ATAN_A: ADJ(2) = ADJ(2) + ADJ(1) / (1 + FV(2) * FV(2)); i -= 3; break;
    case SINH_A: ADJ(2) = ADJ(2) + ADJ(1) * cosh(FV(2)); i -= 3; break;
    case COSH_A: ADJ(2) = ADJ(2) + ADJ(1) * sinh(FV(2)); i -= 3; break;
    case TANH_A: { double c = cosh(FV(2)); ADJ(2) = ADJ(2) + ADJ(1) / (c * c); i -= 3; break; }
    case ASINH_A: ADJ(2) = ADJ(2) + ADJ(1) / sqrt(FV(2) * FV(2) + 1); i -= 3; break;
    case ACOSH_A: ADJ(2) = ADJ(2) + ADJ(1) / sqrt(FV(2) * FV(2) - 1); i -= 3; break;
    case ATANH_A: ADJ(2) = ADJ(2) + ADJ(1) / (1 - FV(2) * FV(2)); i -= 3; break;
    case ERF_A: ADJ(2) = ADJ(2) + ADJ(1) * 2.0 / SQRTPI * exp(-(FV(2) * FV(2))); i -= 3; break;
    case INT_BOTH_BOUNDS:
      ADJ(2) = ADJ(2) + ADJ(1) * ad_constants[const_count]; const_count--;
      ADJ(3) = ADJ(3) - ADJ(1) * ad_constants[const_count]; const_count--; i -= 4; break;
    case INT_LOWER_BOUND:
      ADJ(2) = ADJ(2) - ADJ(1) * ad_constants[const_count]; const_count--; i -= 3; break;
    case INT_UPPER_BOUND:
      ADJ(2) = ADJ(2) + ADJ(1) * ad_constants[const_count]; const_count--; i -= 3; break;
    default:
      ad_overflow = 1; i = 0; break;
    }
  }
#undef T
#undef ADJ
#undef FV
  trace_count = 0;
  index_count = num_parameters;
}

/* ------------------------------------------------------------------ tape interpreter */
typedef struct ival { advar a; double r; } ival; /* r used when node is real-typed */

typedef struct frame {
  const gfh_tape* t;
  double x;            /* data abscissa */
  advar* pars;         /* this%pars(:) */
  const double* aux;   /* this point's auxiliary real inputs: column k at aux[k*aux_ld] (GFH_AUX), or NULL */
  int64_t aux_ld;
} frame;

/* integrand call context */
typedef struct icall {
  const frame* fr; int sub; advar* ipars; int n_ipars;
  int transform; double tb; /* 0 none; 1: f(tb-1+1/x)/x**2 ; 2: f(tb+1-1/x)/x**2 (NI:314-318, 347-351) */
} icall;

static int guard_mismatch = 0;   /* set by eval_sub when a guard of eval()'s tape comes out differently from its recorded outcome */
static int sub_mismatch = 0;     /* the same for a guard inside an integrand's sub-tape */
/* Integrands that compare AD variables (AD:315-395 inside the function handed to integrate()): the reference takes the branch anew
 * at every abscissa of the quadrature.  A recording follows ONE path through the integrand; the recordings of a problem that share
 * eval()'s own path (fam_tapes, set by eval_point) differ in the paths their integrands took, and every evaluation of an integrand
 * uses the recording whose comparisons hold at that abscissa (eval_integrand). */
static const gfh_tape* const* fam_tapes = NULL;
static int fam_n = 0;
static int integrand_uncovered = 0;
static advar eval_sub(const frame* fr, int sub, advar ivar, advar* ipars);
static advar do_integrate(const frame* fr, const gfh_integral* in, advar lower, advar upper,
                          int lower_is_advar, int upper_is_advar, advar* ipars);

static advar eval_sub(const frame* fr, int sub, advar ivar, advar* ipars) {
  const gfh_subtape* st = &fr->t->sub[sub];
  int n = st->n_nodes;
  ival stackbuf[256];
  ival* v = n <= 256 ? stackbuf : (ival*)malloc(sizeof(ival) * n);
  for (int k = 0; k < n; k++) {
    const gfh_node* nd = &st->nodes[k];
    int isreal = nd->flags & GFH_F_REAL;
    switch (nd->op) {
    case GFH_CONST: v[k].r = nd->c; break;
    case GFH_X: v[k].r = fr->x; break;
    case GFH_AUX: v[k].r = fr->aux ? fr->aux[(int64_t)nd->a * fr->aux_ld] : 0.0; break;   /* tabulated real function of x */
    case GFH_PARAM: v[k].a = fr->pars[nd->a]; break;
    case GFH_IVAR: v[k].a = ivar; break;
    case GFH_IPARAM: v[k].a = ipars[nd->a]; break;
    case GFH_LIFT: v[k].a = passive(v[nd->a].r); break;              /* AD:401-447 */
    case GFH_VAL: v[k].r = v[nd->a].a.val; break;                     /* p%val in plain real arithmetic: a number, no derivative */
    case GFH_NEG: v[k].r = -v[nd->a].r; break;
    case GFH_ADD: case GFH_SUB: case GFH_MUL: case GFH_DIV: case GFH_POW: {
      int ra = st->nodes[nd->a].flags & GFH_F_REAL, rb = st->nodes[nd->b].flags & GFH_F_REAL;
      if (isreal) {
        double p = v[nd->a].r, q = v[nd->b].r, r;
        switch (nd->op) { case GFH_ADD: r = p + q; break; case GFH_SUB: r = p - q; break;
          case GFH_MUL: r = p * q; break; case GFH_DIV: r = p / q; break; default: r = pow(p, q); }
        v[k].r = r;
      } else if (!ra && !rb) {
        advar p = v[nd->a].a, q = v[nd->b].a;
        switch (nd->op) { case GFH_ADD: v[k].a = add_advar_advar(p, q); break;
          case GFH_SUB: v[k].a = subtract_advar_advar(p, q); break;
          case GFH_MUL: v[k].a = multiply_advar_advar(p, q); break;
          case GFH_DIV: v[k].a = divide_advar_advar(p, q); break;
          default: v[k].a = power_advar_advar(p, q); }
      } else if (!ra) {
        advar p = v[nd->a].a; double q = v[nd->b].r;
        switch (nd->op) { case GFH_ADD: v[k].a = add_advar_real(p, q); break;
          case GFH_SUB: v[k].a = subtract_advar_real(p, q); break;
          case GFH_MUL: v[k].a = multiply_advar_real(p, q); break;
          case GFH_DIV: v[k].a = divide_advar_real(p, q); break;
          default: v[k].a = power_advar_real(p, q); }
      } else {
        double p = v[nd->a].r; advar q = v[nd->b].a;
        switch (nd->op) { case GFH_ADD: v[k].a = add_real_advar(p, q); break;
          case GFH_SUB: v[k].a = subtract_real_advar(p, q); break;
          case GFH_MUL: v[k].a = multiply_real_advar(p, q); break;
          case GFH_DIV: v[k].a = divide_real_advar(p, q); break;
          default: v[k].a = power_real_advar(p, q); }
      }
      break;
    }
    case GFH_POWI:
      if (isreal) v[k].r = powi(v[nd->a].r, nd->b);
      else v[k].a = power_advar_integer(v[nd->a].a, nd->b);
      break;
#define UN(OPC, fn, rfn) case OPC: if (isreal) v[k].r = rfn(v[nd->a].r); else v[k].a = fn(v[nd->a].a); break;
    UN(GFH_ABS, abs_advar, fabs) UN(GFH_EXP, exp_advar, exp) UN(GFH_SQRT, sqrt_advar, sqrt)
    UN(GFH_LOG, log_advar, log) UN(GFH_SIN, sin_advar, sin) UN(GFH_COS, cos_advar, cos)
    UN(GFH_TAN, tan_advar, tan) UN(GFH_ASIN, asin_advar, asin) UN(GFH_ACOS, acos_advar, acos)
    UN(GFH_ATAN, atan_advar, atan) UN(GFH_SINH, sinh_advar, sinh) UN(GFH_COSH, cosh_advar, cosh)
    UN(GFH_TANH, tanh_advar, tanh) UN(GFH_ASINH, asinh_advar, asinh) UN(GFH_ACOSH, acosh_advar, acosh)
    UN(GFH_ATANH, atanh_advar, atanh) UN(GFH_ERF, erf_advar, erf)
#undef UN
    case GFH_INTEGRATE: {
      const gfh_integral* in = &fr->t->integrals[nd->a];
      advar lp[64];
      for (int q = 0; q < in->n_ipars; q++) {
        int nn = fr->t->ipar_nodes[in->ipar_off + q];
        lp[q] = (st->nodes[nn].flags & GFH_F_REAL) ? passive(v[nn].r) : v[nn].a;
      }
      advar lo = passive(0), up = passive(0); int la = 0, ua = 0;
      if (!in->lower_inf) { if (st->nodes[in->lower].flags & GFH_F_REAL) lo = passive(v[in->lower].r);
                            else { lo = v[in->lower].a; la = 1; } }
      if (!in->upper_inf) { if (st->nodes[in->upper].flags & GFH_F_REAL) up = passive(v[in->upper].r);
                            else { up = v[in->upper].a; ua = 1; } }
      v[k].a = do_integrate(fr, in, lo, up, la, ua, lp);
      break;
    }
    case GFH_GUARD_GT: case GFH_GUARD_LT: {
      /* advar_gt_advar ... qp_lt_advar, AD:315-395: the comparison looks at val only.  The tape records one path through
       * eval(); it is this point's path only if the comparison comes out as it did when the path was recorded. */
      const double p = (st->nodes[nd->a].flags & GFH_F_REAL) ? v[nd->a].r : v[nd->a].a.val;
      const double q = (st->nodes[nd->b].flags & GFH_F_REAL) ? v[nd->b].r : v[nd->b].a.val;
      const int out = nd->op == GFH_GUARD_GT ? p > q : p < q;
      if (out != ((nd->flags & GFH_F_TAKEN) ? 1 : 0)) { if (sub == 0) guard_mismatch = 1; else sub_mismatch = 1; }
      v[k].a = passive(NAN); v[k].r = NAN;
      break;
    }
    default: v[k].a = passive(NAN); v[k].r = NAN; break;
    }
  }
  const gfh_node* rn = &st->nodes[st->result];
  advar y = (rn->flags & GFH_F_REAL) ? passive(v[st->result].r) : v[st->result].a;
  if (v != stackbuf) free(v);
  return y;
}

/* ------------------------------------------------------------------ quadrature, NI */
#define WS_SIZE 4096 /* capacity; the size in force is the tape's ws_size / ws_size_inner, default DEFAULT_WORKSPACE_SIZE = 1000 (NI:40, 84-98, 114-135) */
typedef struct workspace { advar sums[WS_SIZE]; double lower[WS_SIZE], upper[WS_SIZE], abs_error[WS_SIZE]; } workspace;
static workspace* ws[2];
static int int_order = 0;                 /* NI:209 */
static int max_ws[2];
/* work counters of the quadrature (bench.py's algorithmic floor of BASELINE config 4): integrand evaluations of the bisection
 * (NI:241-263: values only), of the final pass (NI:268-275: through AD where the parameters are active), calls of
 * integrate_real_real, intervals at their ends.  orc_quad_counters reads and clears them. */
static long long quad_evals_bisect, quad_evals_final, quad_calls, quad_intervals;
static int quad_in_final;
void orc_quad_counters(long long* out4) {
  out4[0] = quad_evals_bisect; out4[1] = quad_evals_final; out4[2] = quad_calls; out4[3] = quad_intervals;
  quad_evals_bisect = quad_evals_final = quad_calls = quad_intervals = 0;
}

static int sub_has_guards(const gfh_tape* t, int sub) {
  if (sub < 0 || sub >= t->n_subtapes) return 0;
  for (int k = 0; k < t->sub[sub].n_nodes; k++) if (t->sub[sub].nodes[k].op == GFH_GUARD_GT || t->sub[sub].nodes[k].op == GFH_GUARD_LT) return 1;
  return 0;
}
/* the integrand of sub-tape `sub` at abscissa x: through the recording (of fr->t or of a tape of its family) whose comparisons hold
 * there -- found with values only (passive copies: nothing is written to the AD tape), then evaluated for real */
static advar eval_integrand(const frame* fr, int sub, advar x, advar* ipars, int n_ipars) {
  int any = sub_has_guards(fr->t, sub);
  for (int c = 0; c < fam_n && !any; c++) any = sub_has_guards(fam_tapes[c], sub);
  if (!any) return eval_sub(fr, sub, x, ipars);
  advar pq[64];
  const int nq = n_ipars < 64 ? n_ipars : 64;
  for (int k = 0; k < nq; k++) pq[k] = passive(ipars[k].val);
  const int saved = sub_mismatch;
  const gfh_tape* chosen = NULL;
  for (int c = -1; c < fam_n && !chosen; c++) {
    const gfh_tape* t = c < 0 ? fr->t : fam_tapes[c];
    if (c >= 0 && t == fr->t) continue;
    if (sub >= t->n_subtapes) continue;
    frame f2 = *fr; f2.t = t;
    sub_mismatch = 0;
    (void)eval_sub(&f2, sub, passive(x.val), pq);
    if (!sub_mismatch) chosen = t;
  }
  sub_mismatch = saved;
  if (!chosen) { integrand_uncovered = 1; return passive(NAN); }
  frame f2 = *fr; f2.t = chosen;
  const advar y = eval_sub(&f2, sub, x, ipars);
  sub_mismatch = saved;
  return y;
}

static advar icall_f(const icall* c, advar x) {
  if (!c->transform) return eval_integrand(c->fr, c->sub, x, c->ipars, c->n_ipars);
  /* NI:317 / 350: y = f(b -+ 1 +- 1/x, pars)/x**2 with x an advar */
  advar inv = divide_real_advar(1.0, x);
  advar arg = c->transform == 1 ? add_real_advar(c->tb - 1, inv)
                                : subtract_real_advar(c->tb + 1, inv);
  advar fv = eval_integrand(c->fr, c->sub, arg, c->ipars, c->n_ipars);
  return divide_advar_advar(fv, power_advar_integer(x, 2));
}

/* NI:636-664 */
static advar gauss_kronrod(const icall* c, const gk_rule* rule, double lower, double upper, double* abs_error) {
  double scale = (upper - lower) / 2, shift = (lower + upper) / 2, sum_gauss = 0.0;
  advar y = passive(0.0);
  if (quad_in_final) quad_evals_final += rule->n; else quad_evals_bisect += rule->n;
  for (int i = 1; i <= rule->n; i++) {
    advar arg = passive(scale * rule->roots[i - 1] + shift);
    advar f_value = icall_f(c, arg);
    if (i % 2 == 0) sum_gauss = sum_gauss + rule->wg[i / 2 - 1] * f_value.val;
    y = add_advar_advar(y, multiply_real_advar(rule->wk[i - 1], f_value));
  }
  y = multiply_real_advar(scale, y);
  *abs_error = fabs(y.val - scale * sum_gauss);
  return y;
}

static int quad_failed;
/* NI:193-284 */
static advar integrate_real_real(const icall* c, const gfh_integral* in, double lower, double upper) {
  const gfh_tape* t = c->fr->t;
  const gk_rule* rule = gk_rule_by_points(t->gk_points ? t->gk_points : 15);
  int_order++;
  workspace* w = ws[int_order - 1];
  if (!w) w = ws[int_order - 1] = (workspace*)malloc(sizeof(workspace));
  /* default tolerances NI:61-62, 117-119, 132: resolved by the tape producer */
  double rel_error_loc = in->rel_error >= 0 ? in->rel_error
                        : (int_order == 1 ? t->rel_error_outer : t->rel_error_inner);
  double abs_error_loc = in->abs_error >= 0 ? in->abs_error : 0.0;
  int saved[64];
  for (int q = 0; q < c->n_ipars; q++) { saved[q] = c->ipars[q].index; c->ipars[q].index = 0; } /* NI:238-239 */
  w->lower[0] = lower; w->upper[0] = upper;
  w->sums[0] = gauss_kronrod(c, rule, lower, upper, &w->abs_error[0]);
  /* size(ws(int_order)%sums): workspace_init(workspace_size), user-sized through init_integration / init_integration_dbl (NI:84-98, 120, 134) */
  int ws_size = int_order == 1 ? t->ws_size : t->ws_size_inner;
  if (ws_size <= 0) ws_size = 1000;
  if (ws_size > WS_SIZE) ws_size = WS_SIZE;
  for (int current_size = 1; current_size <= ws_size - 1; current_size++) {      /* NI:251 */
    int m = 0; /* maxloc: first maximum */
    for (int q = 1; q < current_size; q++) if (w->abs_error[q] > w->abs_error[m]) m = q;
    double aa = w->lower[m], bb = w->upper[m], middle = (aa + bb) / 2;
    w->sums[m] = gauss_kronrod(c, rule, aa, middle, &w->abs_error[m]);
    w->sums[current_size] = gauss_kronrod(c, rule, middle, bb, &w->abs_error[current_size]);
    w->upper[m] = middle; w->lower[current_size] = middle; w->upper[current_size] = bb;
    double errors_sum = 0, sums_sum = 0;
    for (int q = 0; q <= current_size; q++) { errors_sum += w->abs_error[q]; sums_sum += w->sums[q].val; }
    if (errors_sum < abs_error_loc || errors_sum / sums_sum < rel_error_loc) {
      for (int q = 0; q < c->n_ipars; q++) c->ipars[q].index = saved[q];        /* NI:269 */
      advar y = passive(0.0);
      double dummy;
      const int outer_final = quad_in_final;
      quad_in_final = 1; quad_calls++; quad_intervals += current_size + 1;
      for (int q = 0; q <= current_size; q++)
        y = add_advar_advar(y, gauss_kronrod(c, rule, w->lower[q], w->upper[q], &dummy));
      quad_in_final = outer_final;
      if (current_size > max_ws[int_order - 1]) max_ws[int_order - 1] = current_size;
      int_order--;
      return y;
    }
  }
  quad_failed = 1; /* NI:282-283 */
  for (int q = 0; q < c->n_ipars; q++) c->ipars[q].index = saved[q];
  int_order--;
  return passive(NAN);
}

/* NI:291-369: infinite bounds.  *_inf: +1 INFINITY, -1 -INFINITY */
static advar integrate_inf_real(icall* c, const gfh_integral* in, int lower_inf, double upper);
static advar integrate_real_inf(icall* c, const gfh_integral* in, double lower, int upper_inf) {
  if (upper_inf > 0) {
    icall c2 = *c; c2.transform = 1; c2.tb = lower;
    return integrate_real_real(&c2, in, 0.0, 1.0);
  }
  advar y = integrate_inf_real(c, in, -1, lower);
  return subtract_real_advar(0.0, y);   /* unary minus = 0 - y, AD:598-601 */
}
static advar integrate_inf_real(icall* c, const gfh_integral* in, int lower_inf, double upper) {
  if (lower_inf < 0) {
    icall c2 = *c; c2.transform = 2; c2.tb = upper;
    return integrate_real_real(&c2, in, 0.0, 1.0);
  }
  advar y = integrate_real_inf(c, in, upper, +1);
  return subtract_real_advar(0.0, y);
}

/* Leibniz terms for an active bound.  sign=+1 upper, -1 lower.
 * Reverse: NI:464-479, 511-526, 558-573, 605-620; forward: NI:480-487, 527-534. */
static void bound_term(icall* c, advar* y, advar bound, int sign, int index_first) {
  if (reverse_mode) {
    int saved[64];
    if (index_first && y->index == 0) { ad_room(); index_count++; forward_values[index_count] = y->val; y->index = index_count; }
    int sw = (y->index != 0);
    if (sw) for (int q = 0; q < c->n_ipars; q++) { saved[q] = c->ipars[q].index; c->ipars[q].index = 0; }
    ad_room();
    const_count++;
    ad_constants[const_count] = icall_f(c, passive(bound.val)).val;
    if (sw) for (int q = 0; q < c->n_ipars; q++) c->ipars[q].index = saved[q];
    if (!index_first && y->index == 0) { ad_room(); index_count++; forward_values[index_count] = y->val; y->index = index_count; }
    trace[trace_count + 1] = bound.index; trace[trace_count + 2] = y->index;
    trace[trace_count + 3] = sign > 0 ? INT_UPPER_BOUND : INT_LOWER_BOUND; trace_count += 3;
  } else {
    advar arg = passive(bound.val);
    advar f1 = icall_f(c, arg);
    y->d = y->d + sign * bound.d * f1.val;
    advar dummy = icall_f(c, arg);
    advar dir_deriv = icall_f(c, bound);
    y->dd = y->dd + sign * (bound.dd * dummy.val + bound.d * (dir_deriv.d + dummy.d));
    if (y->index == 0) y->index = 1;
  }
}

static advar do_integrate(const frame* fr, const gfh_integral* in, advar lower, advar upper,
                          int lower_is_advar, int upper_is_advar, advar* ipars) {
  icall c = { fr, in->integrand, ipars, in->n_ipars, 0, 0.0 };
  advar y;
  (void)lower_is_advar; (void)upper_is_advar;
  if (in->lower_inf && in->upper_inf) {             /* NI:354-369 */
    advar y1 = integrate_inf_real(&c, in, in->lower_inf, 0.0);
    advar y2 = integrate_real_inf(&c, in, 0.0, in->upper_inf);
    return add_advar_advar(y1, y2);
  }
  if (in->upper_inf) {                               /* NI:291-319, 586-630 */
    y = integrate_real_inf(&c, in, lower.val, in->upper_inf);
    if (lower.index != 0) bound_term(&c, &y, lower, -1, 0);
    return y;
  }
  if (in->lower_inf) {                               /* NI:324-352, 539-583 */
    y = integrate_inf_real(&c, in, in->lower_inf, upper.val);
    if (upper.index != 0) bound_term(&c, &y, upper, +1, 0);
    return y;
  }
  y = integrate_real_real(&c, in, lower.val, upper.val);
  if (lower.index != 0 && upper.index != 0) {        /* NI:403-439 */
    if (reverse_mode) {
      int saved[64];
      if (y.index == 0) { ad_room(); index_count++; forward_values[index_count] = y.val; y.index = index_count; }
      for (int q = 0; q < c.n_ipars; q++) { saved[q] = c.ipars[q].index; c.ipars[q].index = 0; }
      ad_room();
      const_count++; ad_constants[const_count] = icall_f(&c, passive(lower.val)).val;
      const_count++; ad_constants[const_count] = icall_f(&c, passive(upper.val)).val;
      for (int q = 0; q < c.n_ipars; q++) c.ipars[q].index = saved[q];
      trace[trace_count + 1] = lower.index; trace[trace_count + 2] = upper.index;
      trace[trace_count + 3] = y.index; trace[trace_count + 4] = INT_BOTH_BOUNDS; trace_count += 4;
    } else {
      advar arg = passive(lower.val);
      y.d = y.d - lower.d * icall_f(&c, arg).val;
      arg.val = upper.val;
      y.d = y.d + upper.d * icall_f(&c, arg).val;
      arg.val = lower.val;
      advar dummy = icall_f(&c, arg), dir_deriv = icall_f(&c, lower);
      y.dd = y.dd - lower.dd * dummy.val - lower.d * (dir_deriv.d + dummy.d);
      arg.val = upper.val;
      dummy = icall_f(&c, arg); dir_deriv = icall_f(&c, upper);
      y.dd = y.dd + upper.dd * dummy.val + upper.d * (dir_deriv.d + dummy.d);
      if (y.index == 0) y.index = 1;
    }
  } else if (lower.index != 0) bound_term(&c, &y, lower, -1, 1);   /* NI:445-489 */
  else if (upper.index != 0) bound_term(&c, &y, upper, +1, 1);     /* NI:492-536 */
  return y;
}

/* ------------------------------------------------------------------ linear algebra */
/* potr_f08, LA:36-57: dpotrf('U') (unblocked dpotf2 semantics) + dpotrs. Column-major. */
int orc_potr(int n, double* a, double* b) {
#define A(i, j) a[(size_t)(j) * n + (i)]
  for (int j = 0; j < n; j++) {
    double ajj = A(j, j);
    for (int k = 0; k < j; k++) ajj -= A(k, j) * A(k, j);
    if (ajj <= 0.0 || ajj != ajj) FAIL("Cholesky factorization failed (dpotrf).");
    ajj = sqrt(ajj); A(j, j) = ajj;
    double rinv = 1.0 / ajj;
    for (int c = j + 1; c < n; c++) {
      double s = A(j, c);
      for (int k = 0; k < j; k++) s -= A(k, j) * A(k, c);
      A(j, c) = s * rinv;
    }
  }
  /* U^T y = b */
  for (int i = 0; i < n; i++) { double t = b[i]; for (int k = 0; k < i; k++) t -= A(k, i) * b[k]; b[i] = t / A(i, i); }
  /* U x = y */
  for (int k = n - 1; k >= 0; k--) if (b[k] != 0.0) { b[k] = b[k] / A(k, k); for (int i = 0; i < k; i++) b[i] -= b[k] * A(i, k); }
#undef A
  return 0;
}

/* ------------------------------------------------------------------ setup helpers */
void orc_init_weights(int e, int64_t n, const double* y, const double* sigma, double* w) {
  const double thr = 1e2 * 2.2250738585072014e-308; /* 1d2*tiny(1.0_kp), GF:450 */
  for (int64_t i = 0; i < n; i++) {
    switch (e) {
    case ORC_NONE: w[i] = 1.0; break;
    case ORC_SQRT_Y: w[i] = fabs(y[i]) < thr ? 0.0 : 1.0 / sqrt(y[i]); break;
    case ORC_PROPTO_Y: w[i] = fabs(y[i]) < thr ? 0.0 : 1.0 / y[i]; break;
    case ORC_INVERSE_Y: w[i] = y[i]; break;
    default: w[i] = 1.0 / sigma[i]; break;
    }
  }
}

int orc_jacobian_indices(int nd, int na, const int32_t* active_pars, const int32_t* is_global, int32_t* jac) {
  int shift = 0; /* GF:618-628 */
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < na; j++) {
      if (is_global[active_pars[j]]) { jac[i * na + j] = j; if (i > 0) shift++; }
      else jac[i * na + j] = j + i * na - shift;
    }
  return nd * na - shift;
}

void orc_img_bounds(int n_images, int image, int nd, const int64_t* dp, int64_t* b) {
  /* GF:974-1002 with img_weights = 1/num_images */
  int64_t N = dp[nd];
  int64_t* sizes = (int64_t*)malloc(sizeof(int64_t) * n_images);
  int64_t tmp = 0;
  for (int i = 0; i < n_images; i++) { sizes[i] = (int64_t)((1.0 / n_images) * (double)N); tmp += sizes[i]; }
  for (int i = 0; i < n_images; i++) if (i + 1 <= N - tmp) sizes[i]++;
  int64_t prev_size = 0; for (int i = 0; i < image; i++) prev_size += sizes[i];
  int64_t my_size = sizes[image];
  for (int i = 0; i <= nd; i++) b[i] = prev_size;
  for (int i = 1; i <= nd; i++) {
    int64_t cur_length = dp[i] - dp[i - 1];
    if (prev_size >= cur_length) { prev_size -= cur_length; b[i] = b[i - 1]; /* unchanged */
      /* reference leaves img_bounds(i) at its initial prev_size+1; sub-range is empty as
         img_bounds(i) == img_bounds(i-1) only holds when both untouched */ }
    else if (my_size > 0) {
      if (my_size + prev_size >= cur_length) { b[i] = b[i - 1] + cur_length - prev_size; my_size = my_size + prev_size - cur_length; prev_size = 0; }
      else { b[i] = b[i - 1] + my_size; my_size -= cur_length; }
    } else b[i] = b[i - 1];
  }
  free(sizes);
}

/* ------------------------------------------------------------------ hot path */
static void load_pars(const orc_problem* p, int ds, advar* pa, int with_index) {
  for (int k = 0; k < p->n_pars; k++) pa[k] = passive(p->pars[ds * p->n_pars + k]);
  if (with_index) for (int j = 0; j < p->n_active; j++) pa[p->active_pars[j]].index = j + 1; /* GF:608-610 */
}

static int images_of(int n_images) { return n_images < 1 ? 1 : n_images; }

static int tape_has_guards(const gfh_tape* t) {
  for (int k = 0; k < t->sub[0].n_nodes; k++) if (t->sub[0].nodes[k].op == GFH_GUARD_GT || t->sub[0].nodes[k].op == GFH_GUARD_LT) return 1;
  return 0;
}
static int no_variant_covers = 0;   /* a point none of the recorded paths is valid for (reported by the callers) */

/* eval() at data point i: the reference calls the user's function, which branches on values as it goes (GF:681; AD:315-395).
 * Here: find the recorded path whose comparisons all come out as recorded at this point and these parameter VALUES (a trial
 * evaluation with every parameter passive, so nothing is written to the AD tape), then evaluate that path for real. */
static advar eval_point(const orc_problem* p, frame* fr, int64_t i) {
  const int nv = p->n_variants > 0 ? p->n_variants : 1;
  const gfh_tape* const* vs = p->n_variants > 0 ? p->variants : &p->tape;
  fam_tapes = NULL; fam_n = 0;
  if (nv == 1 && !tape_has_guards(vs[0])) { fr->t = vs[0]; return eval_sub(fr, 0, passive(0), NULL); }
  advar pp[256];
  const int np = p->n_pars < 256 ? p->n_pars : 256;
  for (int k = 0; k < np; k++) pp[k] = passive(fr->pars[k].val);
  frame probe = *fr; probe.pars = pp;
  const int want = p->hint ? (int)p->hint[i] : -1;
  int chosen = -1;
  /* (while eval()'s own comparisons are tried, integrands may look for their path in any recording) */
  fam_tapes = vs; fam_n = nv;
  static const gfh_tape* fam_buf[4096];
  int nf = 0;
  for (int v = 0; v < nv; v++) {
    probe.t = vs[v]; guard_mismatch = 0;
    (void)eval_sub(&probe, 0, passive(0), NULL);
    if (guard_mismatch) continue;
    if (nf < 4096) fam_buf[nf++] = vs[v];
    if (chosen < 0) chosen = v;
    if (v == want) { chosen = v; break; }
  }
  guard_mismatch = 0;
  if (chosen < 0) { no_variant_covers = 1; fr->t = vs[0]; fam_tapes = NULL; fam_n = 0; return passive(NAN); }
  fr->t = vs[chosen];
  /* the recordings that take eval()'s path of this point: between them they hold the paths of its integrands */
  fam_tapes = fam_buf; fam_n = nf;
  const advar y = eval_sub(fr, 0, passive(0), NULL);
  fam_tapes = NULL; fam_n = 0;
  return y;
}

/* Square root of the derivative of the loss function, z = res^2 (lm_solver.cpp:255-284):
 * cauchy rho = ln(1+z); huber rho = z (z<=1) or 2 sqrt(z) - 1; linear rho = z. */
static double loss_scale(int loss, double res) {
  switch (loss) {
    case 1: return sqrt(1.0 / (1.0 + res * res));
    case 2: return res * res > 1.0 ? sqrt(1.0 / fabs(res)) : 1.0;
    default: return 1.0;
  }
}

int orc_sweep(const orc_problem* p, int n_images, double* JTJ, double* JTres, double* res_out, double* JT_out) {
  int nd = p->n_datasets, na = p->n_active;
  int32_t* jac = (int32_t*)malloc(sizeof(int32_t) * nd * na);
  int dim = orc_jacobian_indices(nd, na, p->active_pars, p->is_global, jac);
  int64_t N = p->data_positions[nd];
  int P = images_of(n_images);
  double* JTJ_img = (double*)calloc((size_t)dim * dim, sizeof(double));
  double* JTr_img = (double*)calloc(dim, sizeof(double));
  double* row = (double*)malloc(sizeof(double) * dim);
  int64_t* b = (int64_t*)malloc(sizeof(int64_t) * (nd + 1));
  advar* pa = (advar*)malloc(sizeof(advar) * p->n_pars);
  memset(JTJ, 0, sizeof(double) * dim * dim); memset(JTres, 0, sizeof(double) * dim);
  ad_reserve(10000); reverse_mode = 1; quad_failed = 0; ad_overflow = 0; no_variant_covers = 0; integrand_uncovered = 0;
  frame fr = { p->tape, 0.0, pa, NULL, 0 };
  for (int img = 0; img < P; img++) {
    orc_img_bounds(P, img, nd, p->data_positions, b);
    memset(JTJ_img, 0, sizeof(double) * dim * dim); memset(JTr_img, 0, sizeof(double) * dim);
    for (int j = 0; j < nd; j++) {
      load_pars(p, j, pa, p->finite_diff ? 0 : 1);                                          /* GF:600-611: indices only with AD */
      for (int k = 1; k <= na; k++) forward_values[k] = pa[p->active_pars[k - 1]].val;     /* GF:679 */
      for (int64_t i = b[j]; i < b[j + 1]; i++) {
        index_count = na; trace_count = 0; const_count = 0;
        fr.x = p->x[i]; fr.aux = p->aux ? p->aux + i : NULL; fr.aux_ld = p->data_positions[p->n_datasets];
        advar f = eval_point(p, &fr, i);                                       /* GF:681 */
        double res = (p->y[i] - f.val) * p->w[i];                                           /* GF:682-683 */
        /* robust cost: residual and Jacobian row scaled by sqrt(rho'), chi2() stays plain (lm_solver.cpp:303-317, 513-529) */
        const double ls = p->loss ? loss_scale(p->loss, res) : 1.0;
        if (p->loss) res = ls * res;
        memset(row, 0, sizeof(double) * dim);
        if (p->finite_diff) {
          /* GF:686-687 -> grad_finite, fitfunction.F90:155-174 */
          for (int k = 0; k < na; k++) {
            advar* q = &pa[p->active_pars[k]];
            const double saved_value = q->val;
            double step = sqrt(DBL_EPSILON) * saved_value;                                  /* FF:164 */
            if (!(fabs(step) > DBL_MIN)) {                                                  /* FF:165-167 */
              snprintf(g_err, sizeof g_err, "Absolute value of parameter %d is too small.", p->active_pars[k] + 1);
              free(JTJ_img); free(JTr_img); free(row); free(b); free(pa); free(jac);
              return 1;
            }
            q->val = q->val + step;                                                         /* FF:168 */
            step = q->val - saved_value;                                                    /* FF:169 */
            double g = eval_point(p, &fr, i).val;                              /* FF:170 */
            *q = passive(saved_value);                                                      /* FF:171 */
            g = (g - eval_point(p, &fr, i).val) / step;                        /* FF:172 */
            row[jac[j * na + k]] = p->loss ? (ls * g) * p->w[i] : g * p->w[i];              /* GF:689-690 */
          }
        } else if (f.index != 0) {
          /* the function result must be the last value written (AD:1489-1490) */
          if (f.index != index_count) { free(jac); FAIL("tape result is not the last AD variable"); }
          ad_grad(na);                                                                      /* GF:685 */
          if (!p->loss) for (int k = 0; k < na; k++) row[jac[j * na + k]] = adjoints[k + 1] * p->w[i];    /* GF:689-690 */
          else for (int k = 0; k < na; k++) row[jac[j * na + k]] = (ls * adjoints[k + 1]) * p->w[i];
        }
        /* STEP 2, GF:696-698: JTJ = matmul(JacobianT, Jacobian), JTres = matmul(JacobianT, res) */
        for (int c = 0; c < dim; c++) {
          if (row[c] == 0.0) continue;
          for (int r = 0; r < dim; r++) JTJ_img[(size_t)c * dim + r] += row[r] * row[c];
          JTr_img[c] += row[c] * res;
        }
        if (res_out) res_out[i] = res;
        if (JT_out) memcpy(JT_out + (size_t)i * dim, row, sizeof(double) * dim);
      }
    }
    /* co_sum in image order, misc.F90:133-170 */
    for (int q = 0; q < dim * dim; q++) JTJ[q] = img == 0 ? JTJ_img[q] : JTJ[q] + JTJ_img[q];
    for (int q = 0; q < dim; q++) JTres[q] = img == 0 ? JTr_img[q] : JTres[q] + JTr_img[q];
  }
  (void)N;
  free(JTJ_img); free(JTr_img); free(row); free(b); free(pa); free(jac);
  if (no_variant_covers) FAIL("eval() takes a branch at some data point that none of the recorded variants covers");
  if (integrand_uncovered) FAIL("an integrand takes a branch at some abscissa that none of the recordings covers");
  if (quad_failed) FAIL("Number of iterations was insufficient. Increase either workspace size or the error bound(s).");
  if (ad_overflow) FAIL("corrupt trace");
  return 0;
}

int orc_chi2(const orc_problem* p, int n_images, double* chi2, double* res_out) {
  int nd = p->n_datasets; int P = images_of(n_images);
  int64_t* b = (int64_t*)malloc(sizeof(int64_t) * (nd + 1));
  advar* pa = (advar*)malloc(sizeof(advar) * p->n_pars);
  frame fr = { p->tape, 0.0, pa, NULL, 0 };
  double total = 0; quad_failed = 0; no_variant_covers = 0; integrand_uncovered = 0;
  int rm = reverse_mode; reverse_mode = 1;
  for (int img = 0; img < P; img++) {
    orc_img_bounds(P, img, nd, p->data_positions, b);
    double sum = 0;
    for (int j = 0; j < nd; j++) {
      load_pars(p, j, pa, 0);                              /* GF:1022: all passive */
      for (int64_t i = b[j]; i < b[j + 1]; i++) {
        fr.x = p->x[i]; fr.aux = p->aux ? p->aux + i : NULL; fr.aux_ld = p->data_positions[p->n_datasets];
        advar f = eval_point(p, &fr, i);
        double res = (p->y[i] - f.val) * p->w[i];
        if (res_out) res_out[i] = res;
        sum += res * res;                                  /* GF:1030 dot_product */
      }
    }
    total = img == 0 ? sum : total + sum;                  /* co_sum */
  }
  reverse_mode = rm;
  free(b); free(pa);
  if (no_variant_covers) FAIL("eval() takes a branch at some data point that none of the recorded variants covers");
  if (integrand_uncovered) FAIL("an integrand takes a branch at some abscissa that none of the recordings covers");
  if (quad_failed) FAIL("quadrature workspace exhausted");
  *chi2 = total;
  return 0;
}

int orc_omega(const orc_problem* p, const double* delta1, const double* JT, double* omega_out, double* JTomega) {
  int nd = p->n_datasets, na = p->n_active;
  int32_t* jac = (int32_t*)malloc(sizeof(int32_t) * nd * na);
  int dim = orc_jacobian_indices(nd, na, p->active_pars, p->is_global, jac);
  advar* pa = (advar*)malloc(sizeof(advar) * p->n_pars);
  frame fr = { p->tape, 0.0, pa, NULL, 0 };
  memset(JTomega, 0, sizeof(double) * dim);
  reverse_mode = 0; quad_failed = 0; no_variant_covers = 0; integrand_uncovered = 0;  /* GF:716 */
  double* saved = (double*)malloc(sizeof(double) * (na > 0 ? na : 1));
  for (int j = 0; j < nd; j++) {
    load_pars(p, j, pa, p->finite_diff ? 0 : 1);
    for (int k = 0; k < na; k++) pa[p->active_pars[k]].d = delta1[jac[j * na + k]];  /* GF:719 */
    for (int64_t i = p->data_positions[j]; i < p->data_positions[j + 1]; i++) {
      fr.x = p->x[i]; fr.aux = p->aux ? p->aux + i : NULL; fr.aux_ld = p->data_positions[p->n_datasets];
      double om;
      if (p->finite_diff) {
        /* GF:725-728 -> dir_deriv_2nd_finite, fitfunction.F90:188-203; dir = delta1(Jacobian_indices(:,j)) */
        const double h = sqrt(sqrt(DBL_EPSILON));
        for (int k = 0; k < na; k++) saved[k] = pa[p->active_pars[k]].val;
        for (int k = 0; k < na; k++) pa[p->active_pars[k]].val = pa[p->active_pars[k]].val + h * delta1[jac[j * na + k]];
        double y = eval_point(p, &fr, i).val;
        for (int k = 0; k < na; k++) pa[p->active_pars[k]].val = saved[k] - h * delta1[jac[j * na + k]];
        y = y + eval_point(p, &fr, i).val;
        for (int k = 0; k < na; k++) pa[p->active_pars[k]].val = saved[k];
        y = y - 2 * eval_point(p, &fr, i).val;
        y = y / sqrt(DBL_EPSILON);
        om = -y * p->w[i];
        if (omega_out) omega_out[i] = om;
        for (int c = 0; c < dim; c++) JTomega[c] += JT[(size_t)i * dim + c] * om;
        continue;
      }
      advar f = eval_point(p, &fr, i);
      om = -f.dd * p->w[i];                                 /* GF:723 */
      if (omega_out) omega_out[i] = om;
      for (int c = 0; c < dim; c++) JTomega[c] += JT[(size_t)i * dim + c] * om;      /* GF:734 */
    }
  }
  reverse_mode = 1;                                         /* GF:733 */
  free(pa); free(jac); free(saved);
  if (no_variant_covers) FAIL("eval() takes a branch at some data point that none of the recorded variants covers");
  if (integrand_uncovered) FAIL("an integrand takes a branch at some abscissa that none of the recordings covers");
  if (quad_failed) FAIL("quadrature workspace exhausted");
  return 0;
}

/* ------------------------------------------------------------------ gadf_fit, GF:502-1035 */
static double dtd_dot(int dim, const double* DTD, const double* a, const double* b) {
  /* dot_product(a, matmul(DTD, b)) with diagonal DTD */
  double s = 0; for (int i = 0; i < dim; i++) s += a[i] * (DTD[i] * b[i]); return s;
}

int orc_fit(orc_problem* p, orc_fit_options* o, orc_fit_result* r) {
  int nd = p->n_datasets, na = p->n_active, np = p->n_pars;
  if (na == 0) FAIL("There are no active parameters.");
  double lambda = o->has_lambda ? o->lambda : 1.0;          /* GF:568-584 */
  double lam_up = o->has_lam_up ? o->lam_up : 10.0;
  double lam_down = o->has_lam_down ? o->lam_down : 10.0;
  int lam_incs = 2;
  if (o->has_lam_incs) { if (o->lam_incs < 1) FAIL("Input parameter lam_incs must be at least 1."); lam_incs = o->lam_incs; }
  int uphill = o->has_uphill ? o->uphill : 0;
  int P = images_of(o->n_images);
  int32_t* jac = (int32_t*)malloc(sizeof(int32_t) * nd * na);
  int dim = orc_jacobian_indices(nd, na, p->active_pars, p->is_global, jac);
  int64_t N = p->data_positions[nd];
  double *JTJ = (double*)calloc((size_t)dim * dim, 8), *JTres = (double*)calloc(dim, 8), *DTD = (double*)calloc(dim, 8),
         *delta1 = (double*)calloc(dim, 8), *delta2 = (double*)calloc(dim, 8), *old_delta1 = (double*)calloc(dim, 8),
         *lin = (double*)malloc((size_t)dim * dim * 8), *JTomega = (double*)calloc(dim, 8),
         *old_pars = (double*)malloc(sizeof(double) * na * nd), *res = (double*)malloc(sizeof(double) * N),
         *JT = (double*)malloc(sizeof(double) * (size_t)N * dim);
  int rc = -1, iterations = 0;
  if (o->DTD_min) for (int i = 0; i < dim; i++) DTD[i] = o->DTD_min[i];   /* GF:641-646 */
  int dof = (int)(N - dim);                                               /* GF:648-657 */
  if (dof < 0) { snprintf(g_err, sizeof g_err, "More independent fitting parameters than data points."); goto done; }
  if (dof == 0) dof = 1;
  for (int i = 0; i < nd; i++) for (int j = 0; j < na; j++) old_pars[i * na + j] = p->pars[i * np + p->active_pars[j]];
  double old_chi2, new_chi2 = 0, old_old_chi2 = 0, acc_ratio = 0, beta = 0;
  if (r) { r->n_sweeps = r->n_chi2 = r->n_omega = 0; r->dim = dim; r->dof = dof; r->exit_reason = -1; }
  if (orc_chi2(p, P, &old_chi2, res)) goto done;                          /* GF:670 */
  if (r) r->n_chi2++;
#define SOLVE(rhs_src, out) do { for (int q = 0; q < dim; q++) out[q] = rhs_src[q]; \
    for (int c = 0; c < dim; c++) for (int rr = 0; rr < dim; rr++) lin[(size_t)c * dim + rr] = JTJ[(size_t)c * dim + rr] + (rr == c ? lambda * DTD[c] : lambda * 0.0); \
    if (orc_potr(dim, lin, out)) goto done; } while (0)
  for (;;) {
    if (orc_sweep(p, P, JTJ, JTres, res, JT)) goto done;                  /* STEP 1+2, GF:675-701 */
    if (r) r->n_sweeps++;
    for (int i = 0; i < dim; i++) {                                       /* GF:702-710 */
      double d = JTJ[(size_t)i * dim + i];
      if (o->has_damp_max && !o->damp_max) DTD[i] = d; else DTD[i] = DTD[i] > d ? DTD[i] : d;
    }
    SOLVE(JTres, delta1);                                                 /* GF:711-713 */
    if (o->has_accth && o->accth > 1.17549435e-38f) {                     /* STEP 3, GF:715-743 */
      if (orc_omega(p, delta1, JT, NULL, JTomega)) goto done;
      if (r) r->n_omega++;
      SOLVE(JTomega, delta2);
      acc_ratio = sqrt(dtd_dot(dim, DTD, delta2, delta2) / dtd_dot(dim, DTD, delta1, delta1));
      if (acc_ratio > o->accth) for (int q = 0; q < dim; q++) delta2[q] = 0.0;
    }
    if (r && iterations == 0) {
      if (r->JTJ0) memcpy(r->JTJ0, JTJ, sizeof(double) * dim * dim);
      if (r->JTres0) memcpy(r->JTres0, JTres, sizeof(double) * dim);
      if (r->delta1_0) memcpy(r->delta1_0, delta1, sizeof(double) * dim);
      if (r->delta2_0) memcpy(r->delta2_0, delta2, sizeof(double) * dim);
      r->chi2_0 = old_chi2;
    }
    for (int i = 0; i < nd; i++) for (int j = 0; j < na; j++)             /* GF:745-750 */
      p->pars[i * np + p->active_pars[j]] = p->pars[i * np + p->active_pars[j]] + delta1[jac[i * na + j]] + 0.5 * delta2[jac[i * na + j]];
    int accepted = 0;
    for (int i = 1; i <= lam_incs + 1; i++) {                             /* STEP 4, GF:752-819 */
      if (orc_chi2(p, P, &new_chi2, res)) goto done;
      if (r) r->n_chi2++;
      if (iterations == 0) beta = 0.0;
      else beta = dtd_dot(dim, DTD, delta1, old_delta1) / sqrt(dtd_dot(dim, DTD, delta1, delta1)) / sqrt(dtd_dot(dim, DTD, old_delta1, old_delta1));
      if (powi(1.0 - beta, uphill) * new_chi2 < old_chi2) {               /* GF:761 */
        if (o->has_nielsen && o->nielsen) {                               /* GF:762-767 */
          double q = 0;
          for (int c = 0; c < dim; c++) { double s = 0; for (int k = 0; k < dim; k++) s += (JTJ[(size_t)k * dim + c] + (k == c ? lambda * DTD[c] : 0.0)) * delta1[k]; q += delta1[c] * s; }
          double rho = (old_chi2 - new_chi2) / 2 / q;
          double t = 1 - powi(2 * rho - 1, 3), lo = 1 / lam_down;
          lambda = lambda * (lo > t ? lo : t);
        }
        if (o->has_umnigh && o->umnigh) {                                 /* GF:768-779 */
          const double m = exp(-0.2);
          if (new_chi2 < old_chi2 && beta >= 0.0) {
            o->umnigh_a = o->umnigh_a * m + 1.0 - m;
            double t = powi(1.0 - fabs(2.0 * o->umnigh_a - 1.0), 2); t = t > 1e-2 ? t : 1e-2; t = t < 1.0 ? t : 1.0;
            lambda = lambda * t;
          } else {
            o->umnigh_a = o->umnigh_a * m + (1.0 - m) / 2.0;
            if (new_chi2 >= old_chi2) { double t = 1.0 - fabs(2.0 * o->umnigh_a - 1.0); t = t > 1.0 ? t : 1.0; t = t < 10.0 ? t : 10.0; lambda = lambda / t; }
          }
        }
        if (!((o->has_nielsen && o->nielsen) || (o->has_umnigh && o->umnigh))) lambda = lambda / lam_down;  /* GF:780-782 */
        accepted = 1; break;
      } else if (i <= lam_incs) {                                         /* GF:785-808 */
        if (o->has_umnigh && o->umnigh) {
          const double m = exp(-0.2);
          o->umnigh_a = o->umnigh_a * m;
          double t = 1.0 - fabs(2.0 * o->umnigh_a - 1.0);
          if (beta < 0.0) { t = t * t; t = t > 1e-2 ? t : 1e-2; } else { t = t > 0.1 ? t : 0.1; }
          t = t < 1.0 ? t : 1.0;
          lambda = lambda * t;
        } else lambda = lam_up * lambda;
        for (int a = 0; a < nd; a++) for (int j = 0; j < na; j++) p->pars[a * np + p->active_pars[j]] = old_pars[a * na + j];
        SOLVE(JTres, delta1);
        for (int a = 0; a < nd; a++) for (int j = 0; j < na; j++) p->pars[a * np + p->active_pars[j]] += delta1[jac[a * na + j]];
      } else {                                                            /* GF:809-816 */
        for (int a = 0; a < nd; a++) for (int j = 0; j < na; j++) p->pars[a * np + p->active_pars[j]] = old_pars[a * na + j];
        if (r) r->exit_reason = 7;
        rc = 0; goto done;
      }
    }
    (void)accepted;
    for (int a = 0; a < nd; a++) for (int j = 0; j < na; j++) old_pars[a * na + j] = p->pars[a * np + p->active_pars[j]]; /* GF:821-827 */
    memcpy(old_delta1, delta1, sizeof(double) * dim);
    old_old_chi2 = old_chi2;
    old_chi2 = old_chi2 < new_chi2 ? old_chi2 : new_chi2;
    iterations++;
    /* STEP 5, GF:835-915 */
    if (o->has_chi2_abs && old_chi2 / dof < o->chi2_abs) { if (r) r->exit_reason = 1; break; }
    if (o->has_chi2_rel && (old_old_chi2 - old_chi2) / old_chi2 < o->chi2_rel) { if (r) r->exit_reason = 2; break; }
    if (o->has_grad_chi2) {                                               /* GF:848-860: res from last chi2(), old JT */
      double s = 0;
      for (int c = 0; c < dim; c++) { double g = 0; for (int64_t i = 0; i < N; i++) g += JT[(size_t)i * dim + c] * res[i]; JTres[c] = g; s += g * g; }
      if (2 * sqrt(s) < o->grad_chi2) { if (r) r->exit_reason = 3; break; }
    }
    if (o->has_cos_phi) {                                                 /* GF:861-884 */
      double rj = 0, rr = 0, jj = 0;
      for (int64_t i = 0; i < N; i++) { double jd = 0; for (int c = 0; c < dim; c++) jd += JT[(size_t)i * dim + c] * delta1[c]; rj += res[i] * jd; rr += res[i] * res[i]; jj += jd * jd; }
      if (fabs(rj) / sqrt(rr) / sqrt(jj) < o->cos_phi) { if (r) r->exit_reason = 4; break; }
    }
    if (o->has_rel_error) {                                               /* GF:885-898 */
      int all = 1;
      for (int a = 0; a < nd && all; a++) for (int j = 0; j < na; j++)
        if (fabs(delta1[jac[a * na + j]] / p->pars[a * np + p->active_pars[j]]) > o->rel_error) { all = 0; break; }
      if (all) { if (r) r->exit_reason = 5; break; }
    }
    if (o->has_rel_error_global) {                                        /* GF:899-910 */
      int any = 0;
      for (int j = 0; j < na; j++) if (p->is_global[p->active_pars[j]] && fabs(delta1[jac[j]] / p->pars[p->active_pars[j]]) > o->rel_error_global) any = 1;
      if (!any) { if (r) r->exit_reason = 6; break; }
    }
    if (o->has_max_iter && iterations >= o->max_iter) { if (r) r->exit_reason = 0; break; }  /* GF:911-915 */
  }
  rc = 0;
done:
  if (r) { r->iterations = iterations; r->lambda = lambda; r->chi2 = old_chi2; }
  free(jac); free(JTJ); free(JTres); free(DTD); free(delta1); free(delta2); free(old_delta1); free(lin);
  free(JTomega); free(old_pars); free(res); free(JT);
  return rc;
}

/* ------------------------------------------------------------------ single-point probes */
int orc_eval_reverse(const gfh_tape* t, double x, const double* pars, const int32_t* active, double* val, double* grad) {
  advar pa[256]; int na = 0;
  if (t->n_pars > 256) FAIL("too many parameters");
  ad_reserve(10000); reverse_mode = 1; quad_failed = 0;
  for (int k = 0; k < t->n_pars; k++) { pa[k] = passive(pars[k]); if (active[k]) { pa[k].index = ++na; forward_values[na] = pars[k]; } }
  index_count = na; trace_count = 0; const_count = 0;
  frame fr = { t, x, pa, NULL, 0 };
  advar f = eval_sub(&fr, 0, passive(0), NULL);
  *val = f.val;
  if (na > 0) {
    if (f.index != 0) { ad_grad(na); for (int k = 0; k < na; k++) grad[k] = adjoints[k + 1]; }
    else for (int k = 0; k < na; k++) grad[k] = 0.0;
  }
  if (quad_failed) FAIL("quadrature workspace exhausted");
  return 0;
}

int orc_eval_forward(const gfh_tape* t, double x, const double* pars, const int32_t* active,
                     const double* d_seed, const double* dd_seed, double* out3) {
  advar pa[256];
  if (t->n_pars > 256) FAIL("too many parameters");
  reverse_mode = 0; quad_failed = 0;
  for (int k = 0; k < t->n_pars; k++) { pa[k] = passive(pars[k]); if (active[k]) { pa[k].index = -1; pa[k].d = d_seed[k]; pa[k].dd = dd_seed ? dd_seed[k] : 0.0; } }
  frame fr = { t, x, pa, NULL, 0 };
  advar f = eval_sub(&fr, 0, passive(0), NULL);
  reverse_mode = 1;
  out3[0] = f.val; out3[1] = f.d; out3[2] = f.dd;
  if (quad_failed) FAIL("quadrature workspace exhausted");
  return 0;
}
