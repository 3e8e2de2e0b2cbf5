"""gfh_set_pars_hook carrying a per-point column (C-ABI level): a real that a host model forms from a fitted parameter's value
TOGETHER with the abscissa -- cos(rate%val * x), tests/fortran/fit_param_val_x.F90 -- is uploaded by the hook (gfh_set_aux from
inside it) before every pass at new parameters.  The reference recomputes such a real at every point of every pass (gadfit.F90:679-690);
the oracle's model states it as value(p) * x on the tape."""
import ctypes as C

import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import branching as B

pytestmark = pytest.mark.gpu
HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double))


def model_with_column(p, x):
    from gadfit_amd.ad import aux, exp
    return p[0] * exp(-(p[1] * x)) * (1.0 + 0.1 * aux(0)) + p[2]


def test_columns_uploaded_by_the_parameter_hook_follow_every_pass():
    x, y = B.param_val_x_data()
    w = np.ones_like(x)
    start = np.array([[2.5, 0.9, 0.3]]); trial = np.array([[2.7, 0.85, 0.35]])
    active = [0, 1, 2]; glob = [0, 0, 0]
    calls = []

    ctx = _lib.Context(0)

    def hook(user, target, pars):
        col = np.ascontiguousarray(np.cos(pars[1] * x))
        calls.append(pars[1])
        return _lib.lib().gfh_set_aux(target, 1, col.ctypes.data_as(C.POINTER(C.c_double)))
    cb = HOOK(hook)
    try:
        ctx.set_model(trace_model(model_with_column, 3))
        ctx.set_data(x, y, w, [0, x.size])
        ctx.set_aux(np.cos(start[0, 1] * x))
        ctx.set_keep_jacobian(1)
        assert _lib.lib().gfh_set_pars_hook(ctx._h, C.cast(cb, C.c_void_p), None) == 0
        jac, dim = ctx.jacobian_indices(active, glob)
        # the oracle: the same model with the real on the tape as cos(value(p1) * x)
        t0 = trace_model(B.model_param_val_x, 3)
        p = orc.OracleProblem(t0, [x], [y], [w], start, active, glob)
        JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
        chi0, _ = p.chi2()
        JTJ, JTr, chi2 = ctx.sweep(start, active, jac, dim)
        assert np.max(np.abs(JTJ - JTJ0)) <= 1e-13 * np.max(np.abs(JTJ0)) and np.max(np.abs(JTr - JTr0)) <= 1e-13 * np.max(np.abs(JTr0))
        assert abs(chi2 - chi0) <= 1e-13 * chi0
        delta1 = orc.potr(JTJ0 + np.diag(np.diag(JTJ0)), JTr0)
        om0, jto0 = p.omega(delta1, JT0)
        jto_a = ctx.omega(start, delta1)
        assert np.max(np.abs(jto_a - jto0)) <= 2e-13 * np.max(np.abs(jto0))
        # a trial chi2() elsewhere: the hook uploads the column of THOSE parameters ...
        pt = orc.OracleProblem(t0, [x], [y], [w], trial, active, glob)
        chit, rest = pt.chi2()
        assert abs(ctx.chi2(trial) - chit) <= 1e-13 * chit
        # ... and STEP 3 back at the parameters of the sweep (a rejected trial: gadfit.F90:715-735 runs again with the next delta)
        # still stands on that sweep -- its active set and Jacobian are what they were, the column comes back through the hook
        jto_b = ctx.omega(start, delta1)
        assert np.array_equal(jto_a, jto_b)
        # (the convergence reduction of gadfit.F90:849: the sweep's Jacobian with the residuals of the latest chi2())
        g = ctx.aux(0, dim=dim)
        assert np.max(np.abs(g - JT0.T @ rest)) <= 2e-13 * np.max(np.abs(JTr0))
        assert calls[-1] == start[0, 1] and trial[0, 1] in calls
    finally:
        ctx.close()


def test_forward_differences_read_one_set_of_columns_per_evaluation():
    """use_ad = 0 over a column that follows the parameters (gfh_set_fd_column_sets): the reference's forward differences call eval()
    at p + step e_j, where the real has moved (fitfunction.F90:155-174), so the hook uploads 1 + n_active sets of the column -- set 0
    at p, set 1 + j at p + step e_j, the step formed as the device forms it -- and evaluation j reads its own.  Against the oracle's
    finite differences of the model with the real on the tape (value(p) * x follows the perturbed parameter there by itself); the
    tolerances of test_use_ad_false_finite_differences_vs_oracle.  Without the sets the Jacobian misses the real's share: asserted too."""
    x, y = B.param_val_x_data()
    w = np.ones_like(x)
    start = np.array([[2.5, 0.9, 0.3]])
    active = [0, 1]; glob = [0, 0, 0]               # (the third parameter passive: sets are counted by ACTIVE parameters)
    ctx = _lib.Context(0)

    def sets_at(pars):
        cols = [np.cos(pars[1] * x)]
        for j in active:
            q = [pars[0], pars[1], pars[2]]
            q[j] = q[j] + 2.0 ** -26 * q[j]
            cols.append(np.cos(q[1] * x))
        return np.ascontiguousarray(np.stack(cols))

    def hook(user, target, pars):
        tab = sets_at(pars)
        return _lib.lib().gfh_set_aux(target, tab.shape[0], tab.ctypes.data_as(C.POINTER(C.c_double)))
    cb = HOOK(hook)
    try:
        ctx.set_model(trace_model(model_with_column, 3))
        ctx.set_data(x, y, w, [0, x.size])
        ctx.set_keep_jacobian(1)
        ctx.set_use_ad(False)
        jac, dim = ctx.jacobian_indices(active, glob)
        t0 = trace_model(B.model_param_val_x, 3)
        p = orc.OracleProblem(t0, [x], [y], [w], start, active, glob, use_ad=False)
        JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
        sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
        # one set only, no hook: the differences see a frozen column
        ctx.set_aux(np.cos(start[0, 1] * x))
        JTJf = ctx.sweep(start, active, jac, dim)[0]
        assert np.max(np.abs(JTJf - JTJ0) / sc) > 1e-4
        # the sets announced but not supplied: refused, never read out of bounds
        ctx.set_fd_column_sets(True)
        with pytest.raises(_lib.GadfitHipError, match='gfh_set_aux must hold 3 columns'):
            ctx.sweep(start, active, jac, dim)
        assert _lib.lib().gfh_set_pars_hook(ctx._h, C.cast(cb, C.c_void_p), None) == 0
        ctx.set_aux(sets_at(start[0]))
        JTJ, JTr, chi2 = ctx.sweep(start, active, jac, dim)
        assert np.max(np.abs(JTJ - JTJ0) / sc) < 1e-6
        assert np.max(np.abs(ctx.jacobian(len(active)) - JT0)) < 1e-6 * np.max(np.abs(JT0))
        assert np.max(np.abs(ctx.residuals() - res0)) <= 1e-10 * np.max(np.abs(res0))
        chi0, _ = p.chi2()
        assert abs(ctx.chi2(start) - chi0) <= 1e-12 * chi0 and abs(chi2 - chi0) <= 1e-12 * chi0
        with pytest.raises(_lib.GadfitHipError, match='no column sets at p'):
            ctx.omega(start, np.ones(dim))
        # a whole fit
        q = orc.OracleProblem(t0, [x], [y], [w], start, active, glob, use_ad=False)
        r0 = q.fit(lambda_=np.float32(1.0), max_iter=6)
        out, r = ctx.fit(start, active, glob, lambda_=1.0, max_iter=6)
        assert r.iterations == r0.iterations
        assert np.max(np.abs(out - q.pars) / np.abs(q.pars)) < 1e-6 and abs(r.chi2 - r0.chi2) < 1e-8 * r0.chi2
    finally:
        ctx.close()
