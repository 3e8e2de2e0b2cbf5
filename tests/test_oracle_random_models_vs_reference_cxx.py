"""The oracle's restated elementals against the REFERENCE'S OWN on random expressions over the whole operator set: the seeded random
fitting functions of tests/test_gpu_random_models.py (every elemental, every (advar, advar) / (advar, real) / (real, advar) variant the
operands' static types select, random active / passive parameter subsets, real sub-expressions of x) are traced to tapes; the oracle
evaluates them with its C restatement of automatic_differentiation.F90, oracle/_ref/libgadfit_refcxx.so evaluates THE SAME TAPES with
the reference's C++ AdVar operators, returnSweep and forward mode (oracle/ref_cxx_driver.cpp: tape_eval).  Value, reverse-mode gradient,
first and second directional derivative at 37 abscissas per seed.  The two sides of the reference differ in how they round (a / r is
a product with a reciprocal in Fortran, a quotient in C++; a ** n has an integer form of its own in Fortran), so: rounding, 1e-12."""
import ctypes as C
import os

import numpy as np
import pytest

from gadfit_amd.ad import trace_model
from oracle import binding as orc
from oracle import refcxx
from tests.test_gpu_random_models import NP_, _rand_expr

if not refcxx.available():
    if os.path.isdir('/root/reference/c++/gadfit'):
        raise RuntimeError('oracle/_ref/libgadfit_refcxx.so is missing although /root/reference is present: run `make -C oracle`')
    pytest.skip('oracle/_ref/libgadfit_refcxx.so not shipped and no reference to build it from', allow_module_level=True)

pytestmark = []          # (a CPU test: the module it borrows the generator from is marked gpu)


@pytest.mark.parametrize('block', range(8))
def test_random_operator_set_models_oracle_equals_reference_cxx(block):
    lib = refcxx.lib()
    worst = [0.0, 0.0, 0.0, 0.0]
    for seed in range(25 * block, 25 * (block + 1)):
        def model(p, x, seed=seed):
            r = np.random.default_rng(1000 + seed)
            return 1.0 * _rand_expr(r, p, x, 4)
        tape = trace_model(model, NP_)
        sub = np.random.default_rng(5000 + seed)
        pars = np.ascontiguousarray(sub.uniform(0.6, 1.8, size=NP_))
        mask = sub.random(NP_) < 0.6
        if not mask.any():
            mask[0] = True
        act = np.ascontiguousarray(mask.astype(np.int32))
        na = int(mask.sum())
        dseed = np.zeros(NP_); dseed[mask] = sub.uniform(-0.3, 0.3, size=na)
        for xv in sub.uniform(0.3, 1.6, size=37):
            val, grad = orc.eval_reverse(tape, xv, pars, act)
            fwd = orc.eval_forward(tape, xv, pars, act, dseed, np.zeros(NP_))
            rv = C.c_double(); rg = np.zeros(max(1, na)); rf = np.zeros(3)
            dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            ip = act.ctypes.data_as(C.POINTER(C.c_int))
            assert lib.refcxx_tape_reverse(C.byref(tape.c), C.c_double(xv), dp(pars), ip, C.byref(rv), dp(rg)) == 0
            assert lib.refcxx_tape_forward(C.byref(tape.c), C.c_double(xv), dp(pars), ip, dp(dseed), dp(rf)) == 0
            sc = lambda a: max(1.0, abs(a))
            devs = [abs(val - rv.value) / sc(rv.value), float(np.max(np.abs(grad[:na] - rg[:na]) / np.maximum(1.0, np.abs(rg[:na])))),
                    abs(fwd[1] - rf[1]) / sc(rf[1]), abs(fwd[2] - rf[2]) / sc(rf[2])]
            worst = [max(a, b) for a, b in zip(worst, devs)]
            assert max(devs) <= 1e-12, (seed, xv, devs)
    print('oracle against the reference C++ AD over seeds %d..%d: value %.1e, gradient %.1e, d %.1e, dd %.1e' % (25 * block, 25 * block + 24, *worst))


def _both(tape, x, pars, act, dseed):
    """(oracle, reference C++) x (value, gradient, d, dd) of the tape at x"""
    lib = refcxx.lib()
    pars = np.ascontiguousarray(pars, dtype=np.float64); act = np.ascontiguousarray(act, dtype=np.int32); dseed = np.ascontiguousarray(dseed, dtype=np.float64)
    na = int(act.sum())
    val, grad = orc.eval_reverse(tape, x, pars, act)
    fwd = orc.eval_forward(tape, x, pars, act, dseed, np.zeros(pars.size))
    rv = C.c_double(); rg = np.zeros(max(1, na)); rf = np.zeros(3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = act.ctypes.data_as(C.POINTER(C.c_int))
    assert lib.refcxx_tape_reverse(C.byref(tape.c), C.c_double(x), dp(pars), ip, C.byref(rv), dp(rg)) == 0
    assert lib.refcxx_tape_forward(C.byref(tape.c), C.c_double(x), dp(pars), ip, dp(dseed), dp(rf)) == 0
    return (val, grad[:na], fwd[1], fwd[2]), (rv.value, rg[:na], rf[1], rf[2])


def test_quadrature_with_active_bounds_and_nesting_oracle_equals_reference_cxx():
    """AD through the adaptive rule where the reference holds the fewest vectors: ACTIVE bounds (the Leibniz terms of
    numerical_integration.F90:377-630 -- reverse: INT_* tape operations; forward: the mixed term of the second derivative) and an
    integral inside an integrand whose upper bound follows the outer variable and a parameter (reference test 3's shape, finite
    outer range).  The oracle against the reference's own C++ integrate() overloads for AdVar bounds (numerical_integration.cpp), the
    same tapes through both, value / gradient / d / dd; rule GK15 (the C++ side's only one), tolerances handed over in the tape."""
    from gadfit_amd.ad import exp, integrate, log
    from tests.quadrature_cases import CASES

    def relerr(a, b):
        return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1e-3, np.abs(np.asarray(b)))))
    worst = 0.0
    for name in ('finite, passive bounds', 'finite, active upper bound', 'finite, both bounds active'):
        model, pv, dv = CASES[name]
        # (1.0 * ...: the function value must be the LAST recorded operation -- both sides of the reference seed the last forward value)
        t = trace_model(lambda p, x, m=model: 1.0 * m(p, x), len(pv)); t.set_integration(rel_error=1e-12, rule=15)
        o, r = _both(t, 1.0, pv, [1] * len(pv), dv)
        worst = max(worst, relerr(o[0], r[0]), relerr(o[1], r[1]), relerr(o[2], r[2]), relerr(o[3], r[3]))

    # bounds that follow the abscissa AND the parameters; the integrand's parameters active
    def moving(p, x):
        return p[2] * integrate(lambda tt, q: tt ** 2 * exp(-(q[0] * tt)) + q[1], [p[0], p[1]], 0.2 * p[3], x * p[3])
    t = trace_model(moving, 4); t.set_integration(rel_error=1e-12, rule=15)
    for xv in (0.7, 1.9, 3.3):
        o, r = _both(t, xv, [1.4, 0.3, 2.0, 0.9], [1, 1, 1, 1], [0.3, -0.2, 0.5, 0.4])
        worst = max(worst, relerr(o[0], r[0]), relerr(o[1], r[1]), relerr(o[2], r[2]), relerr(o[3], r[3]))

    # nested: the inner integral's upper bound is the outer variable over a parameter (3_integral_double.F90:27-61, finite outer range)
    def inner(tt, q):
        return log((exp(tt) - 1.0) * q[0] + 1.0) / tt

    def outer(yy, q):
        return yy ** 2 * exp(-yy) * integrate(inner, [q[0]], 1e-3, yy / q[1])

    def nested(p, x):
        return p[2] * integrate(outer, [p[0], p[1]], 0.1, x)
    t = trace_model(nested, 3); t.set_integration(rel_error=1e-9, rel_error_inner=1e-10, rule=15, dbl=True)
    for xv in (1.5, 4.0):
        o, r = _both(t, xv, [0.6, 1.7, 1.1], [1, 1, 1], [0.2, -0.3, 0.4])
        worst = max(worst, relerr(o[0], r[0]), relerr(o[1], r[1]), relerr(o[2], r[2]), relerr(o[3], r[3]))
    print('quadrature with active bounds / nesting, oracle against the reference C++: worst relative deviation %.2e' % worst)
    assert worst <= 1e-9
