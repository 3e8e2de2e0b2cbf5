"""Branching fitting functions for the tests of guards / variants (include/gadfit_tape.h, GFH_GUARD_*): eval() bodies whose
control flow depends on comparisons of AD variables -- which the reference allows (automatic_differentiation.F90:315-395
exports `>` and `<` on advar; gadfit.F90:679-690 evaluates eval() afresh at every point) -- with numpy twins."""
import numpy as np

from gadfit_amd.ad import exp
from tests import models as M


# ---- two segments, breakpoint = parameter 1 (active or passive): a line up to the break, a decay after it ------------------
def model_piecewise2(p, x):
    if x < p[1]:                                  # real < advar (dp_lt_advar, AD:380-384)
        return p[0] + p[2] * (x - p[1])
    return p[0] * exp(-((x - p[1]) / p[3]))


def piecewise2_numpy(p, x):
    return np.where(x < p[1], p[0] + p[2] * (x - p[1]), p[0] * np.exp(-(x - p[1]) / p[3]))


def piecewise2_grad_numpy(p, x):
    """analytic gradient (columns = parameters 0..3) of piecewise2 at fixed branch"""
    lo = x < p[1]
    e = np.exp(-(x - p[1]) / p[3])
    g = np.zeros((x.size, 4))
    g[:, 0] = np.where(lo, 1.0, e)
    g[:, 1] = np.where(lo, -p[2], p[0] * e / p[3])
    g[:, 2] = np.where(lo, x - p[1], 0.0)
    g[:, 3] = np.where(lo, 0.0, p[0] * e * (x - p[1]) / p[3] ** 2)
    return g


PIECEWISE2_TRUTH = np.array([4.0, 37.3, 0.08, 11.0])


# ---- three segments, two breakpoints (parameters 1 and 2) ----------------------------------------------------------------
def model_piecewise3(p, x):
    if x < p[1]:
        return p[0] * exp(-((p[1] - x) / p[3]))          # rising flank
    if p[2] > x:                                         # advar > real (advar_gt_dp, AD:335-339)
        return p[0] + p[4] * (x - p[1])                  # plateau with a tilt
    top = p[0] + p[4] * (p[2] - p[1])
    return top * exp(-((x - p[2]) / p[5]))               # falling flank


def piecewise3_numpy(p, x):
    a = p[0] * np.exp(-(p[1] - x) / p[3])
    b = p[0] + p[4] * (x - p[1])
    c = (p[0] + p[4] * (p[2] - p[1])) * np.exp(-(x - p[2]) / p[5])
    return np.where(x < p[1], a, np.where(p[2] > x, b, c))


PIECEWISE3_TRUTH = np.array([3.0, 21.7, 58.4, 6.0, 0.03, 9.0])


# ---- max(p1, p2 x): Python's max() compares with `>` (advar_gt_advar, AD:315-318) -------------------------------------------
def model_max(p, x):
    return max(p[0], p[1] * x) + p[2] * exp(-(x / p[3]))


def max_numpy(p, x):
    return np.where(p[1] * x > p[0], p[1] * x, p[0]) + p[2] * np.exp(-x / p[3])


MAX_TRUTH = np.array([2.5, 0.06, 3.0, 8.0])


# ---- a comparison of two parameters: the same outcome for every point, may flip during a fit --------------------------------
def model_par_order(p, x):
    if p[0] > p[1]:
        return p[0] * exp(-(x / p[2])) + p[1]
    return p[1] * exp(-(x / p[2])) + p[0]


def par_order_numpy(p, x):
    hi, lo = max(p[0], p[1]), min(p[0], p[1])
    return hi * np.exp(-x / p[2]) + lo


# ---- clipped term: nested comparisons, 4 leaf paths of which data may visit only some ---------------------------------------
def model_clip(p, x):
    t = p[0] * (x - p[1])
    if t < 0.0:                                   # advar < real
        t = 0.0 * t
    if t > p[2]:
        t = p[2] + 0.0 * t
    return t + p[3]


def clip_numpy(p, x):
    return np.clip(p[0] * (x - p[1]), 0.0, p[2]) + p[3]


CLIP_TRUTH = np.array([0.2, 20.0, 6.0, 1.0])


def make_data(fn_numpy, truth, n, x_lo=0.0, x_hi=100.0, seed=M.SEED):
    return M.make_single(fn_numpy, truth, n, x_lo, x_hi, seed)


# ---- two segments with a plain-real factor on the second one (in a Fortran eval() the factor is real(kp) arithmetic on x, which
# reaches the device as an auxiliary per-point column; here the same arithmetic on the symbolic x is simply recorded) -------------
def model_piecewise_aux(p, x):
    if x < p[1]:
        return p[0] + p[2] * (x - p[1])
    g = 1.0 / (1.0 + 1.0e-4 * x ** 2)                 # real(kp) function of x alone
    return p[0] * exp(-((x - p[1]) / p[3])) * (g / (1.0 / (1.0 + 1.0e-4 * p[1] ** 2)))


def piecewise_aux_numpy(p, x):
    g = 1.0 / (1.0 + 1.0e-4 * x ** 2)
    gb = 1.0 / (1.0 + 1.0e-4 * p[1] ** 2)
    return np.where(x < p[1], p[0] + p[2] * (x - p[1]), p[0] * np.exp(-(x - p[1]) / p[3]) * g / gb)


# ---- 8 skewed Gaussians whose sum saturates at a level (parameter 32): 32 or 33 active parameters, i.e. the fused kernel's matrix
# stage behind per-lane variant bodies ----------------------------------------------------------------------------------------
def model_gauss8_saturating(p, x):
    y = M.model_gauss8(p, x)
    if y > p[32]:                                 # advar > advar
        y = p[32] + 0.0 * y
    return y


def gauss8_saturating_numpy(p, x):
    return np.minimum(M.gauss8_numpy(p, x), p[32])


def gauss8_saturating_truth():
    return np.concatenate([M.gauss8_truth(), [3.2]])


# ---- a quadrature on either side of a breakpoint: int_0^x up to the break, int_0^break plus a line after it (the second site has an
# ACTIVE upper bound; both sites share one integrand sub-tape) ------------------------------------------------------------------
def model_integral_then_line(p, x):
    from gadfit_amd.ad import integrate

    def integrand(t, q):
        return t ** q[0] * exp(-(q[1] * t ** 2))
    if x < p[2]:
        return integrate(integrand, [p[0], p[1]], 0.0, x)
    return integrate(integrand, [p[0], p[1]], 0.0, p[2]) + p[3] * (x - p[2])


INTEGRAL_THEN_LINE_TRUTH = np.array([2.0, 0.9, 1.6, -0.05])


def integral_then_line_numpy(p, x):
    from scipy import integrate as si
    f = lambda t: t ** p[0] * np.exp(-p[1] * t * t)
    top = si.quad(f, 0.0, p[2], epsabs=0, epsrel=1e-13)[0]
    return np.array([si.quad(f, 0.0, xi, epsabs=0, epsrel=1e-13)[0] if xi < p[2] else top + p[3] * (xi - p[2]) for xi in x])


# ---- a window holding two of 400001 points (tests/fortran/fit_rare_branch.F90: the sampled recordings of gadf_fit miss it) -------
def model_rare(p, x):
    y = p[0] * exp(-(x / p[1])) + p[2]
    if x > p[3]:
        if x < p[4]:
            y = y + p[5]
    return y


def rare_data(n=400001, inside=2):
    """`inside` consecutive points from index 200001 on lie in the window (w0, w1)"""
    i = np.arange(n, dtype=np.float64)
    x = 100.0 * i / (n - 1)
    w0 = 0.5 * (x[200000] + x[200001]); w1 = 0.5 * (x[200000 + inside] + x[200001 + inside])
    y = 5.0 * np.exp(-(x / 20.0)) + 1.0 + 1.0e-3 * np.sin(np.mod(37 * np.arange(n), 1000).astype(np.float64))
    y[(x > w0) & (x < w1)] += 0.5
    return x, y, w0, w1


# ---- an integrand that branches on the integration variable (a comparison of AD variables INSIDE the function handed to integrate():
# the reference takes the branch anew at every abscissa of the quadrature) ------------------------------------------------------------
def model_kinked_integrand(p, x):
    from gadfit_amd.ad import integrate

    def integrand(t, q):
        if t > q[1]:                                   # advar > advar, decided per abscissa of the quadrature
            return q[0] * exp(-((t - q[1]) / q[2]))
        return q[0] * (1.0 + 0.5 * (t - q[1]))
    return integrate(integrand, [p[0], p[1], p[2]], 0.0, x) + p[3]


KINKED_TRUTH = np.array([1.3, 1.2, 0.8, 0.1])


def kinked_numpy(p, x):
    """closed form of model_kinked_integrand"""
    x = np.asarray(x, dtype=np.float64)
    lo = np.minimum(x, p[1])
    left = p[0] * (lo + 0.25 * ((lo - p[1]) ** 2 - p[1] ** 2))
    right = np.where(x > p[1], p[0] * p[2] * (1.0 - np.exp(-(np.maximum(x, p[1]) - p[1]) / p[2])), 0.0)
    return left + right + p[3]


# ---- the same kinked function as the INNER integrand of a double integral (the outer integrand calls integrate() itself) -------------
def model_nested_kink(p, x):
    from gadfit_amd.ad import integrate

    def inner(u, q):
        if u > q[1]:
            return q[0] * exp(-((u - q[1]) / q[2]))
        return q[0] * (1.0 + 0.5 * (u - q[1]))

    def outer(t, q):
        return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, t)
    return integrate(outer, [p[0], p[1], p[2]], 0.0, x) + p[3]


# ---- a comparison in the OUTER integrand of a double integral (an integrand that compares AND calls integrate() itself) --------------
def model_outer_kink(p, x):
    from gadfit_amd.ad import integrate

    def inner(u, q):
        return q[0] * exp(-(q[1] * u))

    def outer(t, q):
        g = integrate(inner, [q[0], q[2]], 0.0, t)
        if t > q[1]:
            return g * exp(-((t - q[1]) / q[2]))
        return g * (1.0 + 0.5 * (t - q[1]))
    return integrate(outer, [p[0], p[1], p[2]], 0.0, x) + p[3]


# ---- an integrand that takes the data point's abscissa from the enclosing eval() WITHOUT passing it through pars(:) (a module variable
# in the Fortran program tests/fortran/fit_integrand_module_x.F90; here the symbolic x and an auxiliary column inside the closure):
# the reference evaluates the integrand afresh in eval()'s scope at every point (numerical_integration.F90:195-201) ---------------------
def model_integrand_module_x(p, x):
    from gadfit_amd.ad import integrate, aux

    def integrand(t, q):
        return q[0] * exp(-(q[1] * t * t)) * (1.0 + 0.1 * x) + aux(0) * t
    return integrate(integrand, [p[0], p[1]], 0.0, x) + p[2]


INTEGRAND_X_TRUTH = np.array([1.3, 0.7, 0.2])


def integrand_module_x_data(n=300):
    from scipy.special import erf
    i = np.arange(n, dtype=np.float64)
    x = 0.1 + 2.9 * i / (n - 1)
    A, b, c = INTEGRAND_X_TRUTH
    s = np.sin(0.3 * x)
    y = A * (1.0 + 0.1 * x) * 0.5 * np.sqrt(np.pi / b) * erf(x * np.sqrt(b)) + 0.5 * s * x * x + c + 1.0e-3 * np.sin(np.mod(37 * np.arange(n), 1000).astype(np.float64))
    return x, y, s


# ---- a real that eval() forms from the %val of a FITTED parameter (tests/fortran/fit_param_val.F90): the reference recomputes it whenever
# eval() runs and its AD never sees it (no derivative flows through %val); here value() = GFH_VAL ------------------------------------------
def model_param_val(p, x):
    from gadfit_amd.ad import value, sin
    s = sin(value(p[1]))
    return p[0] * exp(-(x / p[1])) * (1.0 + 0.05 * s * s) + p[2]


PARAM_VAL_TRUTH = np.array([5.0, 20.0, 1.0])


def param_val_data(n=400):
    i = np.arange(n, dtype=np.float64)
    x = 0.5 + 99.0 * i / (n - 1)
    A, tau, b = PARAM_VAL_TRUTH
    s = np.sin(tau)
    y = A * np.exp(-(x / tau)) * (1.0 + 0.05 * s * s) + b + 1.0e-3 * np.sin(np.mod(37 * np.arange(n), 1000).astype(np.float64))
    return x, y


# ---- ... formed from the %val of a fitted parameter TOGETHER with the abscissa (tests/fortran/fit_param_val_x.F90, round 5): the
# reference recomputes cos(rate%val * x) at every point of every pass; the Fortran layer tabulates it anew per pass, here value() --------
def model_param_val_x(p, x):
    from gadfit_amd.ad import value, cos
    s = cos(value(p[1]) * x)
    return p[0] * exp(-(p[1] * x)) * (1.0 + 0.1 * s) + p[2]


def model_param_x_plain(p, x):
    """the same function with nothing hidden from the differentiation: what a black-box eval() -- every operation in plain real
    arithmetic on %val -- is to the forward differences of use_ad = .false."""
    from gadfit_amd.ad import cos
    return p[0] * exp(-(p[1] * x)) * (1.0 + 0.1 * cos(p[1] * x)) + p[2]


PARAM_VAL_X_TRUTH = np.array([3.0, 0.8, 0.5])


def param_val_x_data(n=500):
    i = np.arange(n, dtype=np.float64)
    x = 5.0 * i / (n - 1)                   # (the first abscissa is 0: the real is 1 there whatever the parameter)
    A, b, c = PARAM_VAL_X_TRUTH
    y = A * np.exp(-(b * x)) * (1.0 + 0.1 * np.cos(b * x)) + c + 1.0e-3 * np.sin(np.mod(37 * np.arange(n), 1000).astype(np.float64))
    return x, y


# ---- ... the same INSIDE an integrand (tests/fortran/fit_integrand_param_val.F90, round 5): sin(pars(2)%val) formed by the function
# handed to integrate() from its own pars(:) -- a passive extra entry of the integrand's pars(:) on the device -------------------------
def model_integrand_param_val(p, x):
    from gadfit_amd.ad import integrate, value, sin

    def integrand(t, q):
        s = sin(value(q[1]))
        return q[0] * exp(-(q[1] * t * t)) * (1.0 + 0.05 * s * s)
    return integrate(integrand, [p[0], p[1]], 0.0, x) + p[2]


INTEGRAND_PVAL_TRUTH = np.array([1.3, 0.7, 0.2])


def integrand_param_val_data(n=300):
    from scipy.special import erf
    i = np.arange(n, dtype=np.float64)
    x = 0.1 + 2.9 * i / (n - 1)
    A, b, c = INTEGRAND_PVAL_TRUTH
    s = np.sin(b)
    y = A * (1.0 + 0.05 * s * s) * 0.5 * np.sqrt(np.pi / b) * erf(x * np.sqrt(b)) + c + 1.0e-3 * np.sin(np.mod(37 * np.arange(n), 1000).astype(np.float64))
    return x, y
