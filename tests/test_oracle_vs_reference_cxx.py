"""The oracle against OUTPUTS OF THE REFERENCE ITSELF on the hot path, at the bench models and seeded inputs no known-answer test
of the reference holds: oracle/_ref/libgadfit_refcxx.so is the reference's own C++ AD (automatic_differentiation.cpp,
fit_function.cpp, lapack_fallback.cpp, compiled from where they lie) under oracle/ref_cxx_driver.cpp, whose loop is
lm_solver.cpp:286-346.  The C++ side writes `x**2` as pow(x, 2.0) and divides where the Fortran side multiplies by a reciprocal
(SURVEY Appendix A), so the agreement is to rounding, not bitwise: residuals and Jacobian rows entrywise 1e-13 of the row's scale,
sums 1e-12 relative."""
import os

import numpy as np
import pytest

from gadfit_amd.ad import trace_model
from oracle import binding as orc
from oracle import refcxx
from tests import models as M

HAVE_REFERENCE = os.path.isdir('/root/reference/c++/gadfit')

if not refcxx.available():
    if HAVE_REFERENCE:
        raise RuntimeError('oracle/_ref/libgadfit_refcxx.so is missing although /root/reference is present: run `make -C oracle`')
    pytest.skip('oracle/_ref/libgadfit_refcxx.so not shipped and no reference to build it from', allow_module_level=True)


CASES = [('gauss8', refcxx.GAUSS8, M.model_gauss8, M.gauss8_numpy, M.gauss8_truth(), 32, (0.0, 100.0)),
         ('exp4', refcxx.EXP4, M.model_exp4, M.exp4_numpy, M.EXP4_TRUTH, 8, (0.0, 100.0))]


@pytest.mark.parametrize('name,mid,model,fn,truth,npar,span', CASES, ids=[c[0] for c in CASES])
def test_oracle_rows_and_sums_equal_the_reference_cxx_ad(name, mid, model, fn, truth, npar, span):
    n = 20000
    x, y, s = M.make_single(fn, truth, n, *span)
    start = M.start_values(truth)
    tape = trace_model(model, npar)
    act = list(range(npar))
    p = orc.OracleProblem(tape, [x], [y], [1.0 / s], [start], act, [0] * npar)
    JTJ, JTr, res, JT = p.sweep(want_J=True)
    chi, _ = p.chi2()
    rJTJ, rJTr, rres, rJ, _ = refcxx.sweep(mid, x, y, s, start)
    rchi, _ = refcxx.chi2(mid, x, y, s, start)
    scale = np.maximum(1.0, np.abs(rJ).max(axis=1, keepdims=True))
    assert np.max(np.abs(res - rres) / np.maximum(1.0, np.abs(rres))) <= 1e-13
    assert np.max(np.abs(JT - rJ) / scale) <= 1e-13
    d = np.sqrt(np.abs(np.diag(rJTJ)))
    assert np.max(np.abs(JTJ - rJTJ) / np.outer(d, d)) <= 1e-12
    assert np.max(np.abs(JTr - rJTr) / (d * np.sqrt(rchi))) <= 1e-12
    assert abs(chi - rchi) <= 1e-12 * rchi


def test_reference_cxx_threads_give_the_same_rows():
    n = 5000
    x, y, s = M.make_single(M.gauss8_numpy, M.gauss8_truth(), n, 0.0, 100.0)
    start = M.start_values(M.gauss8_truth())
    a = refcxx.sweep(refcxx.GAUSS8, x, y, s, start, threads=1)
    b = refcxx.sweep(refcxx.GAUSS8, x, y, s, start, threads=4)
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])
