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
