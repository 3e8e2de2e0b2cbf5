"""GPU parity tests: the HIP path (through the C ABI of libgadfit_hip.so) against
(a) the reference's golden vectors and (b) the CPU oracle on identical seeded inputs.
Tolerances: north_star asks 1e-10 relative on fitted parameters.  What is asserted here is 10 x what was OBSERVED
(tools/parity_report.py -> profiles/parity_r02.json; GADFIT_PARITY_DUMP=<file> re-records the per-test maxima): per-pass
quantities (res, J, JTJ, JTres, chi2, omega, J^T omega) agree with the oracle to 1e-15 ... 2e-14 -- fp64 throughout, only FMA
contraction, shared reciprocals and libm differ -- and are held to 2e-13; fitted parameters after 4-6 LM iterations
differ by up to 2.5e-12 (the iteration amplifies the per-pass 1e-15 by the conditioning of the damped normal equations)
and are held to 3e-11."""
import itertools
import os

import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import models as M
from tests.golden import goldens as G

pytestmark = pytest.mark.gpu

# Asserted tolerances = about 10 x the maxima OBSERVED on MI355X (run with GADFIT_PARITY_DUMP=<file> to re-record them; the
# numbers in brackets are those maxima).  All relative.
TOL_FIT = 1e-12                # fitted parameters against the oracle after 3-20 LM iterations of the small test problems [7e-14]
TOL_LAMBDA = 3e-11             # final lambda under Nielsen's update: a function of a chi2 DIFFERENCE [2.6e-12]
# the reference's golden fits (its own tolerances: 1e-13, 1e-11, 1e-9, 1e-13 absolute): [1.1e-15, 2.7e-13, 3.6e-11, 4.1e-16].
# Test 3 (nested quadrature to rel 1e-5 / 1e-6) is pinned by the reference itself only to 1e-9 absolute: the value depends on
# the compiler's libm through the adaptive mesh (flang reproduces the gfortran golden to 3e-10).
TOL_GOLDEN_1, TOL_GOLDEN_2, TOL_GOLDEN_3, TOL_GOLDEN_4 = 2e-14, 3e-12, 4e-10, 1e-14
TOL_PASS = 2e-13               # one pass (JTJ, JTres, chi2, res, omega, J^T omega) outside _device_vs_oracle [1.3e-14]
TOL_LOSS = 3e-11               # the same under a robust loss: res and J carry sqrt(rho'), which the C++ side forms as sqrt(1 / (1 + r^2)) [2.6e-12]
TOL_CXX, TOL_CXX_SUMS, TOL_CXX_INTEGRAL, TOL_CXX_NESTED = 1e-13, 1e-13, 2e-14, 2e-13      # the C++ side's known answers [8.9e-15, 9.9e-15, 8.9e-16, 1.3e-14]


def rel(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300 + 1e-3 * np.max(np.abs(b))))


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _one_point(ctx, expr, values, n_pars, active_mask):
    t = trace_model(expr, n_pars)
    ctx.set_model(t)
    ctx.set_data([0.0], [0.0], [1.0], [0, 1])
    active = [i for i, a in enumerate(active_mask) if a]
    jac, dim = ctx.jacobian_indices(active, [0] * n_pars)
    JTJ, JTr, chi2 = ctx.sweep([values], active, jac, dim)
    J = ctx.jacobian(len(active))[0]
    res = ctx.residuals()[0]
    return t, -res, J, JTJ, JTr, chi2, active, jac, dim


@pytest.mark.parametrize('expr,values,ref,n', [
    (G.expr_basic_reverse, G.BASIC_VALUES, G.BASIC_REVERSE_REF, 3),
    (G.expr_power, G.POWER_VALUES, G.POWER_REVERSE_REF, 2),
    (G.expr_trig, G.TRIG_VALUES, G.TRIG_REVERSE_REF, 2)])
def test_reverse_goldens_on_device(ctx, expr, values, ref, n):
    """ad_reverse_mode.F90 goldens: the sweep kernel's Jacobian row of a single point."""
    row = 0
    for act in itertools.product([0, 1], repeat=n):
        if not any(act):
            continue
        t, f, J, JTJ, JTr, chi2, active, jac, dim = _one_point(ctx, expr, values, n, act)
        want = ref[row][:sum(act)]
        assert np.all(np.abs(J - want) <= 1e-13 * np.maximum(1.0, np.abs(want))), (act, J, want)
        # JTJ / JTres / chi2 of one point are outer products of that row
        assert rel(JTJ, np.outer(J, J)) < 1e-14 and rel(JTr, J * (-f)) < 1e-14 and abs(chi2 - f * f) <= 1e-14 * f * f
        row += 1


def test_erf_golden_on_device(ctx):
    t, f, J, *_ = _one_point(ctx, G.expr_erf, G.ERF_VALUE, 1, [1])
    assert abs(J[0] - G.ERF_REVERSE_REF) < 1e-15 and abs(f - G.ERF_FORWARD_REF[0]) < 1e-15


@pytest.mark.parametrize('expr,values,n', [(G.expr_basic_forward, G.BASIC_VALUES, 3), (G.expr_power, G.POWER_VALUES, 2),
                                           (G.expr_trig, G.TRIG_VALUES, 2), (G.expr_erf, G.ERF_VALUE, 1)])
def test_forward_mode_on_device_vs_oracle(ctx, expr, values, n):
    """omega kernel = forward-mode (val,d,dd) with d seeded by delta1, dd seed 0 (gadfit.F90:719)."""
    for act in itertools.product([0, 1], repeat=n):
        if not any(act):
            continue
        t, f, J, JTJ, JTr, chi2, active, jac, dim = _one_point(ctx, expr, values, n, act)
        delta = np.array([0.3 + 0.1 * k for k in range(dim)])
        ctx.omega([values], delta)
        om = ctx.omega_vector()[0]
        dseed = np.zeros(n); dseed[active] = delta
        want = orc.eval_forward(t, 0.0, values, act, dseed, np.zeros(n))
        assert abs(-om - want[2]) <= 1e-12 * max(1.0, abs(want[2])), (act, om, want)


_OBSERVED = {}      # GADFIT_PARITY_DUMP=<file>: the maxima actually seen, per test (tolerances below = these x 10, profiles/parity_r02.json)


def _observe(**kw):
    import os
    key = os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]
    d = _OBSERVED.setdefault(key, {})
    for k, v in kw.items():
        d[k] = max(d.get(k, 0.0), float(v))
    dst = os.environ.get('GADFIT_PARITY_DUMP')
    if dst:
        import json
        json.dump(_OBSERVED, open(dst, 'w'), indent=1)


def _close(label, got, want, tol, scale=None):
    """max |got - want| / scale <= tol (scale: |want| entry by entry unless given); the maximum seen is recorded (_observe)"""
    got = np.asarray(got, dtype=float); want = np.asarray(want, dtype=float)
    sc = np.abs(want) if scale is None else scale
    err = float(np.max(np.abs(got - want) / sc)) if want.size else 0.0
    _observe(**{label: err})
    assert err <= tol, (label, err, tol, got, want)


def _device_vs_oracle(ctx, tape, xs, ys, ws, pars, active, is_global, tol=1e-13, with_omega=True, jtol=7e-13, otol=1.5e-13):
    """tol: JTJ / JTres / chi2 [observed over all callers: 7e-15, 2.2e-15, 4.6e-15]; jtol: Jacobian entries -- relative to the
    entry, floored at 1e-6 of the column maximum, so cancellation in small entries shows -- and residuals [7.1e-14, 4.4e-15];
    otol: omega, J^T omega and the convergence reductions J^T res, cos(phi) sums [3e-15, 1.3e-15, 1.3e-14, 1.5e-15].
    See also profiles/parity_r02.json (the BASELINE configurations at N = 2e4)."""
    p = orc.OracleProblem(tape, xs, ys, ws, pars, active, is_global)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    ctx.set_model(tape)
    ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), p.dp)
    jac, dim = ctx.jacobian_indices(active, is_global)
    assert dim == p.dim and np.array_equal(jac, p.jac)
    JTJ, JTr, chi2 = ctx.sweep(p.pars, active, jac, dim)
    res = ctx.residuals()
    J = ctx.jacobian(len(active))
    # per-point quantities
    Jd = np.zeros_like(JT0)
    for d in range(p.nd):
        sl = slice(p.dp[d], p.dp[d + 1])
        Jd[sl][:, jac[d]] = J[sl]
    scale = np.maximum(np.abs(JT0), 1e-6 * np.max(np.abs(JT0), axis=0, keepdims=True) + 1e-300)
    dscale = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0))) + 1e-300
    chi_k = ctx.chi2(p.pars)
    _observe(J=np.max(np.abs(Jd - JT0) / scale), res=np.max(np.abs(res - res0)) / max(1.0, np.max(np.abs(res0))),
             JTJ=np.max(np.abs(JTJ - JTJ0) / dscale), JTres=np.max(np.abs(JTr - JTr0) / (np.sqrt(np.diag(JTJ0) * chi0) + 1e-300)),
             chi2=max(abs(chi2 - chi0), abs(chi_k - chi0)) / chi0)
    assert np.max(np.abs(Jd - JT0) / scale) < jtol, 'Jacobian entries'
    assert np.max(np.abs(res - res0)) <= jtol * max(1.0, np.max(np.abs(res0)))
    assert np.max(np.abs(JTJ - JTJ0) / dscale) < tol, 'JTJ'
    assert np.allclose(JTJ, JTJ.T, rtol=0, atol=0), 'JTJ must come back exactly symmetric'
    assert np.max(np.abs(JTr - JTr0) / (np.sqrt(np.diag(JTJ0) * chi0) + 1e-300)) < tol, 'JTres'
    assert abs(chi2 - chi0) <= tol * chi0
    assert abs(chi_k - chi0) <= tol * chi0
    if not with_omega:
        return p
    # STEP 3
    delta1 = orc.potr(JTJ0 + np.diag(np.diag(JTJ0)), JTr0)
    om0, jto0 = p.omega(delta1, JT0)
    jto = ctx.omega(p.pars, delta1)
    om = ctx.omega_vector()
    # convergence reductions (gadfit.F90:849, 865-873) with res from chi2 at shifted parameters
    g = ctx.aux(0, dim=dim)
    s3 = ctx.aux(1, delta1=delta1)
    jd = JT0 @ delta1
    _observe(omega=np.max(np.abs(om - om0)) / max(1e-300, np.max(np.abs(om0))), JTomega=np.max(np.abs(jto - jto0)) / np.max(np.abs(jto0)),
             grad=np.max(np.abs(g - JT0.T @ res0)) / np.max(np.abs(JTr0)), cosphi=rel(s3, [res0 @ jd, res0 @ res0, jd @ jd]))
    assert np.max(np.abs(om - om0)) <= otol * max(1e-300, np.max(np.abs(om0)))
    assert np.max(np.abs(jto - jto0)) <= otol * np.max(np.abs(jto0))
    assert np.max(np.abs(g - JT0.T @ res0)) <= otol * np.max(np.abs(JTr0))
    assert rel(s3, [res0 @ jd, res0 @ res0, jd @ jd]) < otol
    return p


def test_sweep_exp4_vs_oracle(ctx):
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 5003, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    _device_vs_oracle(ctx, t, [x], [y], [1.0 / s], [M.start_values(M.EXP4_TRUTH)], list(range(8)), [0] * 8)


def test_sweep_gauss8_vs_oracle(ctx):
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 4096 + 17, 0.0, 100.0)
    t = trace_model(M.model_gauss8, 32)
    _device_vs_oracle(ctx, t, [x], [y], [1.0 / s], [M.start_values(truth)], list(range(32)), [0] * 32)


def test_sweep_passive_subset(ctx):
    """some parameters passive (1_gaussian.F90 style): only active columns are produced."""
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 777, 0.0, 100.0)
    t = trace_model(M.model_gauss8, 32)
    active = [0, 2, 5, 6, 9, 17, 30]
    _device_vs_oracle(ctx, t, [x], [y], [1.0 / s], [M.start_values(truth)], active, [0] * 32)


def test_sweep_global_fit_ragged_vs_oracle(ctx):
    """global fit: ragged dataset sizes incl. a 1-point and an exactly-tile-sized dataset."""
    sizes = [1, 1024, 333, 2049, 57]
    xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
    t = trace_model(M.model_global7, 7)
    pars = np.array([M.start_values(tr) for tr in truths])
    pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)          # globals share one value
    p = _device_vs_oracle(ctx, t, xs, ys, [1.0 / s for s in ss], pars, list(range(7)), [0, 0, 0, 0, 1, 1, 1])
    assert p.dim == 3 + 4 * len(sizes)


def test_fit_1_gaussian_golden(ctx):
    from gadfit_amd import gadfit as gf

    class gaussian(gf.fitfunc):
        def init(self):
            self.allocate(4)
            for i, n in enumerate(['fmax', 'x0', 'a', 'bgr']):
                self.set(i + 1, n)

        def eval(self, x):
            return G.model_gaussian(self.pars, x)
    d = G.data()['1_gaussian']
    gf.gadf_init(gaussian())
    gf.gadf_add_dataset(d['x_data'], d['y_data'])
    gf.gadf_set('fmax', 1.0, True); gf.gadf_set('x0', 1e-12, False)
    gf.gadf_set('a', 1.0, True); gf.gadf_set('bgr', 1.0, True)
    gf.gadf_set_errors(gf.NONE)
    gf.gadf_set_verbosity(output='/dev/null')
    r = gf.gadf_fit(0.1, accth=0.9, max_iter=4)
    a = gf.fitfuncs[0].pars[2].val
    gf.gadf_close()
    assert r.iterations == 4
    _close('a', a, G.GAUSSIAN_A, TOL_GOLDEN_1)                  # the reference's own tolerance: 1e-13 absolute; north_star: 1e-10 relative


def test_fit_4_multiple_curves_golden(ctx):
    from gadfit_amd import gadfit as gf

    class exponential(gf.fitfunc):
        def init(self):
            self.allocate(3)

        def eval(self, x):
            return G.model_exponential(self.pars, x)
    d = G.data()['4_multiple_curves']
    gf.gadf_init(exponential(), 2)
    gf.gadf_add_dataset(d['x_data_1'], d['y_data_1'])
    gf.gadf_add_dataset(d['x_data_2'], d['y_data_2'])
    gf.gadf_set(1, 1, 1.0, True); gf.gadf_set(2, 1, 1.0, True)
    gf.gadf_set(1, 3, 1.0, True); gf.gadf_set(2, 3, 1.0, True)
    gf.gadf_set(2, 1.0, True)
    gf.gadf_set_errors(gf.SQRT_Y)
    gf.gadf_set_verbosity(output='/dev/null')
    r = gf.gadf_fit(lambda_=10.0, accth=0.9, max_iter=4)
    got = np.array([[p.val for p in f.pars] for f in gf.fitfuncs])
    gf.gadf_close()
    assert r.iterations == 4 and r.dim == 5
    _close('pars', got, G.MULTIPLE_CURVES, TOL_GOLDEN_4)        # the reference's own tolerance: 1e-13 absolute


@pytest.mark.parametrize('opts', [dict(lambda_=1.0, max_iter=5), dict(lambda_=1.0, accth=0.9, max_iter=5),
                                  dict(lambda_=0.01, lam_incs=4, max_iter=6, nielsen=1),
                                  dict(lambda_=1.0, max_iter=6, umnigh=1, uphill=1),
                                  dict(lambda_=1.0, max_iter=20, rel_error=1e-5, cos_phi=1e-3, grad_chi2=1e-3),
                                  dict(lambda_=1.0, max_iter=4, damp_max=0, chi2_rel=1e-9)])
def test_fit_vs_oracle_lm_options(ctx, opts):
    """The host LM logic (Appendix B) drives the same sequence of sweeps as the oracle."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 2000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH)
    o32 = {k: (np.float32(v) if k in ('lambda_', 'accth', 'rel_error', 'cos_phi', 'grad_chi2', 'chi2_rel') else v) for k, v in opts.items()}
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], list(range(8)), [0] * 8)
    r0 = p.fit(**o32)
    ctx.set_model(t)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], list(range(8)), [0] * 8, **{k: float(v) if isinstance(v, np.floating) else v for k, v in o32.items()})
    assert (r.iterations, r.n_sweeps, r.n_chi2, r.n_omega, r.exit_reason) == (r0.iterations, r0.n_sweeps, r0.n_chi2, r0.n_omega, r0.exit_reason)
    _close('pars', out, p.pars, TOL_FIT)
    _close('lambda', r.lambda_, r0.lambda_, TOL_LAMBDA)       # Nielsen: lambda depends on a chi2 difference


@pytest.mark.parametrize('pert,opts', [(0.05, dict(lambda_=1.0, max_iter=4)), (0.05, dict(lambda_=1.0, accth=0.9, max_iter=4)),
                                       (0.4, dict(lambda_=1e-6, lam_incs=8, max_iter=4)),       # rejected trials in between
                                       (0.4, dict(lambda_=1e-3, lam_incs=8, max_iter=5, nielsen=1))])
def test_lookahead_schedule_equals_reference_schedule(pert, opts):
    """gfh_set_lookahead: taking the first trial chi2 from a sweep at the trial point (and handing an
    accepted step's J^T J / J^T r to the next iteration) requests the same sequence of values as the
    reference's chi2()-then-sweep schedule; the fit does not move beyond rounding."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 3000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    # iteration counts stay short of convergence, where accept/reject is decided by rounding
    start = np.array(M.EXP4_TRUTH, dtype=float) * (1.0 + pert * np.where(np.arange(8) % 2 == 0, 1.0, -1.0))
    res = []
    for la in (0, 1):
        c = _lib.Context(0)
        c.set_lookahead(la)
        c.set_model(t)
        c.set_data(x, y, 1.0 / s, [0, x.size])
        out, r = c.fit([start], list(range(8)), [0] * 8, **opts)
        res.append((out.copy(), r, c.residuals().copy(), c.timers()))
        c.close()
    (p0, r0, res0, t0), (p1, r1, res1, t1) = res
    assert r0.n_lookahead == 0 and r1.n_lookahead >= 1
    if pert > 0.1:
        assert r0.n_chi2 > r0.iterations + 1, 'this case is meant to contain rejected trials'
    assert (r1.iterations, r1.n_sweeps, r1.n_chi2, r1.n_omega, r1.exit_reason) == (r0.iterations, r0.n_sweeps, r0.n_chi2, r0.n_omega, r0.exit_reason)
    # bitwise: the fused sweep sums r^2 in the partition and order of gfh_k_chi2, so both schedules see the same chi2 values
    assert np.array_equal(p1, p0) and r1.chi2 == r0.chi2 and r1.lambda_ == r0.lambda_
    assert np.array_equal(res1, res0)                                        # device res: last trial point in both
    # launches: reference schedule = n_sweeps sweeps + n_chi2 chi2 kernels; look-ahead moves
    # n_lookahead of the chi2 launches into sweeps of which the accepted ones replace a later sweep
    assert t0[6] == r0.n_sweeps and t0[7] == r0.n_chi2
    assert t1[7] == r1.n_chi2 - r1.n_lookahead and t1[6] <= r1.n_sweeps + r1.n_lookahead
    assert t1[6] + t1[7] < t0[6] + t0[7]


# ---- AD through adaptive Gauss-Kronrod quadrature on the device (BASELINE config 4) ----------
def test_integral_single_sweep_vs_oracle(ctx):
    """pi * int_0^x t^a exp(-b t^2) dt (2_integral_single.F90:27-46), GK15, rel 1e-12: same mesh
    decisions as the oracle, gradient w.r.t. the integrand parameters through the final pass."""
    d = G.data()['2_integral_single']
    x = np.array(d['x_data']); y = np.array(d['y_data'])
    for pars in ([10.0, 1.0], [7.5, 0.8]):
        t = trace_model(G.model_integral_single, 2)
        t.set_integration(rel_error=1e-12)
        _device_vs_oracle(ctx, t, [x], [y], [np.ones_like(y)], [pars], [0, 1], [0, 0])


def test_integral_double_sweep_vs_oracle(ctx):
    """nested integrals, infinite outer bound, ACTIVE inner upper bound, erf (3_integral_double.F90:27-61)."""
    d = G.data()['3_integral_double']
    x = np.array(d['x_data']); y = np.array(d['y_data']); s = np.array(d['weights'])
    t = trace_model(G.model_integral_double, 2)
    t.set_integration(rel_error=1e-5, rel_error_inner=1e-6, dbl=True)
    _device_vs_oracle(ctx, t, [x], [y], [1.0 / s], [[1.0, 1.0]], [0, 1], [0, 0])


def test_fit_2_integral_single_golden(ctx):
    """fortran/tests/2_integral_single.F90:56-74 through the driver API on the GPU: reverse-mode
    gradient AND forward-mode second directional derivative through the adaptive quadrature."""
    from gadfit_amd import gadfit as gf

    class integral_single(gf.fitfunc):
        def init(self):
            self.allocate(2); self.set(1, 'a'); self.set(2, 'b')

        def eval(self, x):
            return G.model_integral_single(self.pars, x)
    d = G.data()['2_integral_single']
    gf.gadf_init(integral_single(), rel_error=1e-12)
    gf.gadf_add_dataset(d['x_data'], d['y_data'])
    gf.gadf_set('a', 10.0, True); gf.gadf_set('b', 1.0, True)
    gf.gadf_set_errors(gf.NONE)
    gf.gadf_set_verbosity(output='/dev/null')
    gf.gadf_fit(10.0, accth=0.9, max_iter=6, rel_error=1e-6)
    a = gf.fitfuncs[0].pars[0].val
    gf.gadf_close()
    _close('a', a, G.INTEGRAL_SINGLE_A, TOL_GOLDEN_2)                            # reference tolerance: 1e-11 abs


def test_fit_3_integral_double_golden(ctx):
    """fortran/tests/3_integral_double.F90:74-96: nested integrals, infinite bound, active bound, USER errors."""
    from gadfit_amd import gadfit as gf

    class integral_double(gf.fitfunc):
        def init(self):
            self.allocate(2); self.set(1, 'a'); self.set(2, 'b')

        def eval(self, x):
            return G.model_integral_double(self.pars, x)
    d = G.data()['3_integral_double']
    gf.gadf_init(integral_double(), rel_error_inner=1e-6, rel_error=1e-5)
    gf.gadf_add_dataset(d['x_data'], d['y_data'], d['weights'])
    gf.gadf_set('a', 1.0, True); gf.gadf_set('b', 1.0, True)
    gf.gadf_set_errors(gf.USER)
    gf.gadf_set_verbosity(output='/dev/null')
    gf.gadf_fit(0.1, accth=0.9, max_iter=3)
    a = gf.fitfuncs[0].pars[0].val
    gf.gadf_close()
    _close('a', a, G.INTEGRAL_DOUBLE_A, TOL_GOLDEN_3)                            # the reference's own tolerance: 1e-9 abs


def test_quadrature_workspace_exhaustion_is_reported(ctx):
    """NI:282-283: too tight a tolerance for the workspace -> error, not a silent wrong answer."""
    d = G.data()['2_integral_single']
    x = np.array(d['x_data']); y = np.array(d['y_data'])
    t = trace_model(G.model_integral_single, 2)
    t.set_integration(rel_error=1e-30)
    ctx.set_model(t)
    ctx.set_data(x, y, np.ones_like(y), [0, x.size])
    with pytest.raises(_lib.GadfitHipError, match='Number of iterations was insufficient'):
        ctx.chi2([[10.0, 1.0]])


def test_rccl_communicator_single_rank(ctx):
    """The in-library RCCL path (ncclCommInitRank + all-reduce of the packed [JTJ|JTres|chi2], chi2,
    J^T v) with a 1-rank communicator must reproduce the no-communicator results."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 3000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH).reshape(1, 8)
    ctx.set_model(t); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    jac, dim = ctx.jacobian_indices(list(range(8)), [0] * 8)
    ref = ctx.sweep(start, list(range(8)), jac, dim); ref_chi = ctx.chi2(start)
    c2 = _lib.Context(0)
    c2.comm_init(1, 0, _lib.Context.unique_id())
    c2.set_model(t); c2.set_data(x, y, 1.0 / s, [0, x.size])
    got = c2.sweep(start, list(range(8)), jac, dim)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    assert c2.chi2(start) == ref_chi
    d1 = np.linspace(0.1, 0.8, dim)
    assert np.array_equal(c2.omega(start, d1), ctx.omega(start, d1))
    c2.close()


def test_bench_distributed_path_on_one_gpu():
    """bench.py through torch.distributed.run with one rank: process group (RCCL), unique-id
    broadcast, gfh_comm_init, all-reduces -- the path the 8-GPU scaling run takes."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', os.path.join(root, 'bench.py'),
                        '--gpus', '1', '--steps', '3', '--warmup', '1', '--points', '200000', '--cpu-sample', '0', '--min-timed', '0.05', '--multi', 'on'],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and d['final_chi2_per_dof'] < 1e3   # 4 LM iterations from the 5 % start
    assert d['kernels_ms']['allreduce'] > 0.0
    assert d['rccl_nranks'] == 1 and d['allreduces_in_main_leg'] >= 3      # ncclCommCount of the library's communicator
    # the self-check legs of an N > 1 line, rehearsed with the one rank behind its RCCL communicator (--multi on)
    from tests.test_cpu_bench_schema import check_multi_block
    mp = check_multi_block(d, 1)
    assert mp['ok'] and mp['cross_rank_sum_path'] == 'rccl' and 'ncclAllReduce' in d['allreduce_us']['path']
    assert mp['fit_vs_one_rank']['ok'] and mp['fit_vs_one_rank']['max_rel_dev_pars'] == 0.0 and mp['sums_vs_ordered_host_sum']['all_ranks_hold_the_same_bits']
    assert d['rccl_ms_per_step'] == d['ms_per_step'] and d['strong_leg']['points_total'] == 200000
    assert d['host_sum_ms_per_step'] > 0 and d['host_sum_leg']['rccl_nranks'] == 0, d['host_sum_leg']


def test_cfg1_two_exponential_200_points(ctx):
    """BASELINE config 1: 2-exponential decay, 200 points, 4 active parameters; full fit vs the oracle
    with the reference's defaults plus acceleration, run to the iteration limit."""
    x, y, s = M.make_single(M.exp2_numpy, M.EXP2_TRUTH, 200, 0.5, 100.0)
    t = trace_model(M.model_exp2, 4)
    start = M.start_values(M.EXP2_TRUTH)
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r0 = p.fit(lambda_=np.float32(1.0), accth=np.float32(0.9), max_iter=5)   # fixed count: at convergence accept/reject is decided by rounding
    ctx.set_model(t)
    ctx.set_data(x, y, 1.0 / s, [0, 200])
    out, r = ctx.fit([start], [0, 1, 2, 3], [0] * 4, lambda_=1.0, accth=float(np.float32(0.9)), max_iter=5)
    assert (r.iterations, r.n_sweeps, r.n_chi2, r.n_omega) == (r0.iterations, r0.n_sweeps, r0.n_chi2, r0.n_omega)
    _close('pars', out, p.pars, TOL_FIT)
    assert np.max(np.abs(out[0] - M.EXP2_TRUTH) / M.EXP2_TRUTH) < 0.05
    # launch-latency regime: report the wall time per LM iteration at this size
    st = np.array([1.0, -1.0, 0.0]); dtd = np.zeros(4); pr = np.array([start])
    import time
    ctx.lm_iterate(pr, [0, 1, 2, 3], [0] * 4, 3, st, dtd)
    t0 = time.perf_counter(); ctx.lm_iterate(pr, [0, 1, 2, 3], [0] * 4, 50, st, dtd); dt = (time.perf_counter() - t0) / 50
    print('cfg1: %.1f us per LM iteration at N=200' % (dt * 1e6))
    assert dt < 5e-3


@pytest.mark.parametrize('K,active', [(12, None), (16, None), (16, [5]), (3, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]),
                                      (17, None), (20, None), (24, None), (28, None), (32, None), (33, None), (20, list(range(3, 73))),
                                      (32, list(range(5, 110)))])
def test_gram_tile_counts_vs_oracle(ctx, K, active):
    """48 and 64 active parameters (3 and 4 sixteen-row tiles, 6 and 10 tile pairs), a single active
    parameter, and 12: every shape of the matrix-core path against the oracle.  68, 80 and 70 active parameters
    (5 tiles), 96, 112, 128 and 105 (6, 7, 8 and 7 tiles): the fused kernel in its workgroup-cooperative form (round 6: the waves
    share the tile pairs out and read each other's stages; STEP 3 reads the stored J there: gfh_k_omega_jt stops at 64).  132
    (9 tiles): beyond 128 STEP 1 and STEP 2 run as the plain sweep plus blocked Gram launches over the stored Jacobian (k_gram_block)."""
    truth = M.gaussK_truth(K)
    # 1501 points: no abscissa coincides with a start value of mu (at x == mu the reference's forward-mode
    # a**n formula divides by the base, AD:1051-1054, and yields NaN; the oracle and GADFIT_HIP_FAST_DIV=0 reproduce that,
    # the default device form is the polynomial identity and stays finite: test_forward_mode_square_at_zero_base)
    x, y, s = M.make_single(M.gaussK_numpy(K), truth, 1501, 0.0, 100.0)
    t = trace_model(M.make_model_gaussK(K), 4 * K)
    act = list(range(4 * K)) if active is None else active
    _device_vs_oracle(ctx, t, [x], [y], [1.0 / s], [M.start_values(truth)], act, [0] * (4 * K))


@pytest.mark.parametrize('which', ['gauss8', 'exp4', 'integral_single'])
def test_device_against_the_reference_cxx_directly(ctx, which):
    """The device against OUTPUTS OF THE REFERENCE ITSELF, without the oracle in between: oracle/_ref/libgadfit_refcxx.so (the reference's
    own C++ AD, quadrature and vendored linear algebra compiled from where they lie; loop = lm_solver.cpp:286-346, 513-529) on the
    headline model, BASELINE config 2's and config 4's at seeded inputs -- residuals, Jacobian rows, J^T J, J^T r, chi2.  Rounding
    only separates the two (the C++ side divides where the Fortran side multiplies by a reciprocal, writes x**2 as pow(x, 2.0))."""
    from oracle import refcxx
    assert refcxx.available(), 'oracle/_ref/libgadfit_refcxx.so did not travel with the snapshot (make -C oracle in the build container)'
    if which == 'integral_single':
        import ctypes
        n = 4000
        a, b = 7.5, 0.8
        x = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
        f0 = M.normal(n, M.SEED); y = 5.0 + f0; s = 0.5 + 0.1 * np.abs(f0)
        start = np.array([a * 1.05, b * 0.95]); mid = refcxx.INTEGRAL_SINGLE
        tape = trace_model(G.model_integral_single, 2); tape.set_integration(rel_error=1e-10)
        refcxx.lib().refcxx_set_rel_error.argtypes = [ctypes.c_double]; refcxx.lib().refcxx_set_rel_error(1e-10)
        tol = 1e-9                 # (the quadrature's own tolerance bounds how far two bisections may part)
    else:
        fn, truth, model, npar, mid = ((M.gauss8_numpy, M.gauss8_truth(), M.model_gauss8, 32, refcxx.GAUSS8) if which == 'gauss8' else
                                       (M.exp4_numpy, M.EXP4_TRUTH, M.model_exp4, 8, refcxx.EXP4))
        x, y, s = M.make_single(fn, truth, 20000, 0.0, 100.0)
        start = M.start_values(truth); tape = trace_model(model, npar); tol = 2e-12
    npar = start.size
    rJTJ, rJTr, rres, rJ, _ = refcxx.sweep(mid, x, y, s, start)
    rchi, _ = refcxx.chi2(mid, x, y, s, start)
    ctx.set_model(tape); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    act = list(range(npar))
    jac, dim = ctx.jacobian_indices(act, [0] * npar)
    JTJ, JTr, chi2 = ctx.sweep(start.reshape(1, npar), act, jac, dim)
    J = ctx.jacobian(npar); res = ctx.residuals()
    scale = np.maximum(1.0, np.abs(rJ).max(axis=1, keepdims=True))
    d = np.sqrt(np.abs(np.diag(rJTJ)))
    devs = dict(res=float(np.max(np.abs(res - rres) / np.maximum(1.0, np.abs(rres)))), J=float(np.max(np.abs(J - rJ) / scale)),
                JTJ=float(np.max(np.abs(JTJ - rJTJ) / np.outer(d, d))), JTres=float(np.max(np.abs(JTr - rJTr) / (d * np.sqrt(rchi)))),
                chi2=abs(chi2 - rchi) / rchi)
    _observe(**{'refcxx_' + k: v for k, v in devs.items()})
    print('device against the reference C++ (%s):' % which, ' '.join('%s %.1e' % kv for kv in devs.items()))
    assert max(devs.values()) <= tol, devs


def test_placement_of_the_jacobian_and_of_the_data_arrays_changes_no_bit(monkeypatch):
    """The placement search (context.cpp, place_jacobian_now; round 6: also x, y, w, res as a set) swaps the buffers under a context:
    the Jacobian buffer for the fastest of several allocations, the four data arrays for copies in other pages.  The sums of a sweep,
    the residuals, the weights and the abscissas behind the swap are bitwise those before it (1.1e6 points x 32 parameters: the
    smallest Jacobian the search is made for)."""
    truth = M.gauss8_truth()
    n = 1_100_000
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    t = trace_model(M.model_gauss8, 32)
    act = list(range(32)); start = M.start_values(truth).reshape(1, 32)
    c = _lib.Context(0)
    try:
        c.set_placement_after(1 << 30)
        c.set_model(t); c.set_data(x, y, s, [0, n]); c.init_weights(4)
        jac, dim = c.jacobian_indices(act, [0] * 32)
        JTJ0, JTr0, chi0 = c.sweep(start, act, jac, dim)
        res0 = c.residuals().copy(); w0 = c.weights().copy(); x0 = c.abscissas().copy()
        assert len(c.placement()) == 0                      # (no search yet)
        c.set_placement_after(0)
        JTJ1, JTr1, chi1 = c.sweep(start, act, jac, dim)    # the search runs at this sweep
        pl = c.placement()
        assert len(pl) >= 1 and pl[0] > 0.0, pl
        JTJ2, JTr2, chi2 = c.sweep(start, act, jac, dim)
        for JTJ, JTr, chi in ((JTJ1, JTr1, chi1), (JTJ2, JTr2, chi2)):
            assert np.array_equal(JTJ, JTJ0) and np.array_equal(JTr, JTr0) and chi == chi0
        assert np.array_equal(c.residuals(), res0) and np.array_equal(c.weights(), w0) and np.array_equal(c.abscissas(), x0)
        assert c.chi2(start) == chi0
    finally:
        c.close()


def test_device_meshes_hold_the_oracles_interval_counts(ctx):
    """gfh_debug_mesh_stats (round 6: the work count behind config 4's algorithmic roofline) against the oracle's own counters on the
    same inputs: the adaptive rule on the device makes, integral by integral, the bisections the restated reference algorithm makes
    (numerical_integration.F90:251-267) -- the totals agree EXACTLY, and the evaluation counts 15 (2n - 1) + 15 n follow from them."""
    n = 3000
    a, b = 7.5, 0.8
    x = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
    y = np.zeros(n); w = np.ones(n)
    t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
    start = np.array([[a * 1.05, b * 0.95]])
    p = orc.OracleProblem(t, [x], [y], [w], start, [0, 1], [0, 0])
    orc.quad_counters()
    p.sweep()
    c = orc.quad_counters()
    ctx.set_model(t); ctx.set_data(x, y, w, [0, n])
    jac, dim = ctx.jacobian_indices([0, 1], [0, 0])
    ctx.sweep(start, [0, 1], jac, dim)
    m = ctx.mesh_stats()
    assert m['integrals'] == n == c['calls'] and m['unrecorded'] == 0 and m['sites'] == 1
    assert m['bisections'] == c['intervals'] - c['calls']
    assert 15 * (m['integrals'] + 2 * m['bisections']) == c['evals_bisect'] and 15 * (m['integrals'] + m['bisections']) == c['evals_final']


def test_cooperative_fused_kernel_in_a_global_fit_and_a_device_group(ctx):
    """The workgroup-cooperative form of the fused kernel (81 ... 128 active parameters, round 6) where its gram blocks belong to
    different datasets and its sums cross members: two curves of 24 Gaussians with the 24 centres shared (96 active per dataset: 72
    local + 24 global columns, dim 168) against the oracle -- J, res, J^T J, J^T r, chi2, STEP 3 over the stored Jacobian --, a 4-iteration
    fit against the oracle's, and the same sweep as a three-member device group sharing the card (sums in rank order on the host)."""
    K = 24
    truth = M.gaussK_truth(K)
    t = trace_model(M.make_model_gaussK(K), 4 * K)
    xa, ya, sa = M.make_single(M.gaussK_numpy(K), truth, 1337, 0.0, 100.0)
    truth_b = truth.copy(); truth_b[0::4] *= 1.3                      # other amplitudes, the same centres
    xb, yb, sb = M.make_single(M.gaussK_numpy(K), truth_b, 911, 0.0, 100.0, seed=M.SEED + 5)
    act = list(range(4 * K))
    glob = [1 if k % 4 == 1 else 0 for k in range(4 * K)]            # the centres are global
    start = np.array([M.start_values(truth), M.start_values(truth_b)])
    start[1, 1::4] = start[0, 1::4]
    p = _device_vs_oracle(ctx, t, [xa, xb], [ya, yb], [1.0 / sa, 1.0 / sb], start, act, glob)
    assert p.dim == 2 * 3 * K + K
    out, r = ctx.fit(start, act, glob, lambda_=1.0, max_iter=4)
    r0 = p.fit(lambda_=1.0, max_iter=4)
    assert r.iterations == r0.iterations
    _close('pars_coop_global_fit', out, p.pars, TOL_FIT)
    os.environ['GADFIT_HIP_GROUP_WRAP'] = '1'; os.environ['GADFIT_HIP_GROUP_REDUCE'] = 'host'
    try:
        g = _lib.Context(devices=3)
        try:
            g.set_model(t)
            pos = np.array([0, xa.size, xa.size + xb.size], dtype=np.int64)
            g.set_data(np.concatenate([xa, xb]), np.concatenate([ya, yb]), np.concatenate([1.0 / sa, 1.0 / sb]), pos)
            jac, dim = g.jacobian_indices(act, glob)
            JTJg, JTrg, chig = g.sweep(start, act, jac, dim)
        finally:
            g.close()
    finally:
        os.environ.pop('GADFIT_HIP_GROUP_WRAP', None); os.environ.pop('GADFIT_HIP_GROUP_REDUCE', None)
    q = orc.OracleProblem(t, [xa, xb], [ya, yb], [1.0 / sa, 1.0 / sb], start, act, glob)
    JTJ0, JTr0, _, _ = q.sweep()
    chi0, _ = q.chi2()
    sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0))) + 1e-300
    assert np.max(np.abs(JTJg - JTJ0) / sc) < TOL_PASS and abs(chig - chi0) <= TOL_PASS * chi0
    assert np.max(np.abs(JTrg - JTr0) / (np.sqrt(np.diag(JTJ0) * chi0) + 1e-300)) < TOL_PASS


def test_gadf_fit_restart_and_changed_active_set(ctx):
    """gadf_fit may be called again and continues from the current parameters (gadfit.F90:563: x_data stays),
    also with a different active set; same sequence on the oracle."""
    from gadfit_amd import gadfit as gf

    class exp2(gf.fitfunc):
        def init(self):
            self.allocate(4)

        def eval(self, x):
            return M.model_exp2(self.pars, x)
    x, y, s = M.make_single(M.exp2_numpy, M.EXP2_TRUTH, 500, 0.5, 100.0)
    start = M.start_values(M.EXP2_TRUTH)
    t = trace_model(M.model_exp2, 4)
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    p.fit(lambda_=np.float32(1.0), max_iter=2)
    p.fit(lambda_=np.float32(0.5), accth=np.float32(0.9), max_iter=2)
    p2 = orc.OracleProblem(t, [x], [y], [1.0 / s], p.pars, [0, 2], [0] * 4)       # amplitudes only
    p2.fit(lambda_=np.float32(1.0), max_iter=2)
    gf.gadf_init(exp2())
    gf.gadf_add_dataset(x, y, s)
    for k in range(4):
        gf.gadf_set(k + 1, start[k], True)
    gf.gadf_set_errors(gf.USER)
    gf.gadf_set_verbosity(output='/dev/null')
    gf.gadf_fit(lambda_=1.0, max_iter=2)
    gf.gadf_fit(lambda_=0.5, accth=0.9, max_iter=2)
    got = np.array([q.val for q in gf.fitfuncs[0].pars])
    _close('pars_first_fit', got, p.pars[0], TOL_FIT)
    first = got.copy()
    gf.gadf_set(2, got[1], False); gf.gadf_set(4, got[3], False)                    # tau's passive now
    gf.gadf_fit(lambda_=1.0, max_iter=2)
    got = np.array([q.val for q in gf.fitfuncs[0].pars])
    gf.gadf_close()
    _close('pars_second_fit', got, p2.pars[0], TOL_FIT)
    assert got[1] == first[1] and got[3] == first[3]                                # a passive parameter is not touched


@pytest.mark.parametrize('nranks', [2, 3, 8])
def test_rank_sharding_on_one_gpu(ctx, nranks):
    """Every pseudo-rank gets the GLOBAL arrays (gfh_set_data) or only its slice (gfh_set_data_local) and
    shards them by the reference's rule; the per-rank [JTJ|JTres|chi2], chi2() and J^T omega sum to the
    single-image result.  Datasets of 1, 1024, 333, 2049 and 57 points: ranks straddle dataset boundaries
    and some ranks own nothing of some datasets."""
    sizes = [1, 1024, 333, 2049, 57]
    xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
    t = trace_model(M.model_global7, 7)
    pars = np.array([M.start_values(tr) for tr in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    active = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    X, Y, W = np.concatenate(xs), np.concatenate(ys), np.concatenate([1.0 / s for s in ss])
    pos = np.concatenate([[0], np.cumsum(sizes)])
    ctx.set_model(t); ctx.set_data(X, Y, W, pos)
    jac, dim = ctx.jacobian_indices(active, glob)
    JTJ, JTr, chi2 = ctx.sweep(pars, active, jac, dim)
    d1 = _lib.potr(JTJ + np.diag(np.diag(JTJ)), JTr)
    jto = ctx.omega(pars, d1)
    accJ = np.zeros_like(JTJ); accr = np.zeros_like(JTr); accc = 0.0; acco = np.zeros_like(jto); accchi = 0.0; total = 0
    for r in range(nranks):
        c = _lib.Context(0)
        c.debug_set_rank(nranks, r)
        c.set_model(t)
        if r % 2 == 0:
            c.set_data(X, Y, W, pos)
        else:
            b, n = _lib.partition(X.size, nranks, r)
            c.set_data_local(X.size, pos, b, X[b:b + n], Y[b:b + n], W[b:b + n])
        b, n = _lib.partition(X.size, nranks, r)
        assert (c.local_begin(), c.local_count()) == (b, n)
        total += n
        a, br, cc = c.sweep(pars, active, jac, dim)
        accJ += a; accr += br; accc += cc
        accchi += c.chi2(pars)
        acco += c.omega(pars, d1)
        res = c.residuals()
        assert res.shape == (n,)
        c.close()
    assert total == X.size
    sc = np.sqrt(np.outer(np.diag(JTJ), np.diag(JTJ)))
    assert np.max(np.abs(accJ - JTJ) / sc) < 1e-12
    _close('JTres_ranks', accr, JTr, TOL_PASS, np.max(np.abs(JTr)))
    assert abs(accc - chi2) <= 1e-12 * chi2 and abs(accchi - chi2) <= 1e-12 * chi2
    _close('JTomega_ranks', acco, jto, TOL_PASS, np.max(np.abs(jto)))


# ---- C++ side goldens of the reference (c++/tests/lm_solver.cpp) on the device ---------------------
from tests import cxx_goldens_common as CX


@pytest.mark.parametrize('k', [k for k in range(len(G.CXX_INDEXING)) if CX.representable(G.CXX_INDEXING[k][1])])
def test_cxx_indexing_scheme_goldens_on_device(ctx, k):
    t, xs, ys, ws, pars, act, exp = CX.case(k)
    ctx.set_model(t)
    ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), [0, 100, 200])
    out, r = ctx.fit(pars, CX.active_list(act), [0, 1, 0], lambda_=1.0, lam_incs=3, max_iter=4)
    assert r.iterations == 4
    chi2 = ctx.chi2(out)
    _close('chi2', chi2, exp['chi2'], TOL_CXX)
    _close('tau', out[0, 1], exp['tau'], TOL_CXX)
    assert out[1, 1] == out[0, 1]
    for d in range(2):
        for col, key in ((0, 'I0'), (2, 'bgr')):
            want = exp[key][d]
            if want is None:
                assert out[d, col] == pars[d, col]
            else:
                _close('pars', out[d, col], want, TOL_CXX)


def test_cxx_access_function_goldens_on_device(ctx):
    """sum(J), sum(residuals), sum(J^T r), the global parameter's JTJ row (lm_solver.cpp:230-241) from the
    device-resident Jacobian / residuals of the last sweep."""
    t, xs, ys, ws, pars, act, exp = CX.case(0)
    ctx.set_model(t)
    ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), [0, 100, 200])
    out, r = ctx.fit(pars, [0, 1, 2], [0, 1, 0], lambda_=1.0, lam_incs=3, max_iter=3)
    jac, dim = ctx.jacobian_indices([0, 1, 2], [0, 1, 0])
    JTJ, JTr, chi2 = ctx.sweep(out, [0, 1, 2], jac, dim)
    J = ctx.jacobian(3); res = ctx.residuals()
    _close('sumJ', J.sum(), G.CXX_SUM_JACOBIAN, TOL_CXX_SUMS)
    _close('sumres', res.sum(), G.CXX_SUM_RESIDUALS, TOL_CXX_SUMS)
    _close('sumJTr', JTr.sum(), G.CXX_SUM_RIGHT_SIDE, TOL_CXX_SUMS)
    _close('sumJTJrow', JTJ[1].sum(), G.CXX_JTJ_TAU_ROW_SUM, TOL_CXX_SUMS)


def test_keep_jacobian_modes():
    """gfh_set_keep_jacobian: without the J store the fused kernel returns bitwise the same J^T J / J^T r /
    chi2 / res; STEP 3 recomputes the Jacobian rows (gfh_k_omega_jt) and works without J; calls that read J
    back fail loudly; mode 2 lets gfh_fit decide per fit."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 4000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH)
    act = list(range(8))
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(x, y, 1.0 / s, [0, x.size])
        jac, dim = c.jacobian_indices(act, [0] * 8)
        JTJ1, JTr1, chi1 = c.sweep([start], act, jac, dim)
        res1 = c.residuals().copy(); J1 = c.jacobian(8).copy()
        om1 = c.omega([start], np.ones(dim))
        c.set_keep_jacobian(0)
        JTJ0, JTr0, chi0 = c.sweep([start], act, jac, dim)
        assert np.array_equal(JTJ0, JTJ1) and np.array_equal(JTr0, JTr1) and chi0 == chi1
        assert np.array_equal(c.residuals(), res1)
        assert np.array_equal(c.omega([start], np.ones(dim)), om1)          # STEP 3 does not read J
        for call in (lambda: c.jacobian(8), lambda: c.aux(0, dim=dim)):
            with pytest.raises(_lib.GadfitHipError, match='Jacobian was not kept'):
                call()
        with pytest.raises(_lib.GadfitHipError, match='Jacobian was not kept'):
            c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=20, cos_phi=1e-3)
        p0, r0 = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=4)
        p0a, r0a = c.fit([start], act, [0] * 8, lambda_=1.0, accth=0.9, max_iter=4)
        c.set_keep_jacobian(2)
        p2, r2 = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=4)               # plain: no J store
        with pytest.raises(_lib.GadfitHipError, match='Jacobian was not kept'):
            c.jacobian(8)
        p2a, r2a = c.fit([start], act, [0] * 8, lambda_=1.0, accth=0.9, max_iter=4)   # accelerated: still no J store
        with pytest.raises(_lib.GadfitHipError, match='Jacobian was not kept'):
            c.jacobian(8)
        p2c, r2c = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=4, cos_phi=1e-30)   # the cos(phi) test reads J
        assert c.jacobian(8).shape == J1.shape
        c.set_keep_jacobian(1)
        p1, r1 = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=4)
        p1a, r1a = c.fit([start], act, [0] * 8, lambda_=1.0, accth=0.9, max_iter=4)
        p1c, r1c = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=4, cos_phi=1e-30)
    finally:
        c.close()
    assert np.array_equal(p0, p1) and np.array_equal(p2, p1) and np.array_equal(p2a, p1a) and np.array_equal(p0a, p1a)
    assert np.array_equal(p2c, p1c)
    assert (r0.chi2, r2.chi2, r2a.chi2, r0a.chi2, r2c.chi2) == (r1.chi2, r1.chi2, r1a.chi2, r1a.chi2, r1c.chi2)


def test_five_tile_fused_kernel_modes_and_schedules():
    """65 ... 80 active parameters (round 5): the fused kernel on its half stage.  Without the Jacobian store it returns bitwise the
    stored form's sums; chi2() at the parameters of a sweep is bitwise that sweep's sum r^2 (so the look-ahead schedule applies and
    equals the reference schedule bit for bit); an accelerated fit under keep_jacobian mode 2 stores J (gfh_k_omega_jt stops at
    64 parameters) and equals the mode-1 fit; GADFIT_HIP_FUSED=0 (plain sweep + k_gram_block) agrees to rounding."""
    K = 20
    truth = M.gaussK_truth(K)
    x, y, s = M.make_single(M.gaussK_numpy(K), truth, 5 * 512 + 301, 0.0, 100.0)
    t = trace_model(M.make_model_gaussK(K), 4 * K)
    act = list(range(4 * K)); glob = [0] * (4 * K)
    start = M.start_values(truth).reshape(1, 4 * K)
    c = _lib.Context(0)
    try:
        c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size])
        jac, dim = c.jacobian_indices(act, glob)
        JTJ1, JTr1, chi1 = c.sweep(start, act, jac, dim)
        assert c.chi2(start) == chi1
        J1 = c.jacobian(4 * K).copy()
        c.set_keep_jacobian(0)
        JTJ0, JTr0, chi0 = c.sweep(start, act, jac, dim)
        assert np.array_equal(JTJ0, JTJ1) and np.array_equal(JTr0, JTr1) and chi0 == chi1 and c.chi2(start) == chi1
        with pytest.raises(_lib.GadfitHipError, match='Jacobian was not kept'):
            c.jacobian(4 * K)
        c.set_keep_jacobian(1)
        c.set_lookahead(True); pl, rl = c.fit(start, act, glob, lambda_=1.0, max_iter=5)
        c.set_lookahead(False); pr, rr = c.fit(start, act, glob, lambda_=1.0, max_iter=5)
        assert rl.n_lookahead > 0 and rr.n_lookahead == 0 and np.array_equal(pl, pr) and rl.chi2 == rr.chi2
        c.set_lookahead(True)
        p1a, r1a = c.fit(start, act, glob, lambda_=1.0, accth=0.9, max_iter=4)
        c.set_keep_jacobian(2)
        p2, r2 = c.fit(start, act, glob, lambda_=1.0, max_iter=5)
        assert np.array_equal(p2, pl) and r2.chi2 == rl.chi2
        p2a, r2a = c.fit(start, act, glob, lambda_=1.0, accth=0.9, max_iter=4)
        assert r2a.n_omega > 0 and np.array_equal(p2a, p1a) and c.jacobian(4 * K).shape == J1.shape
    finally:
        c.close()
    os.environ['GADFIT_HIP_FUSED'] = '0'
    try:
        c = _lib.Context(0)
        c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size])
        JTJu, JTru, chiu = c.sweep(start, act, jac, dim)
        c.close()
    finally:
        os.environ.pop('GADFIT_HIP_FUSED', None)
    sc = np.sqrt(np.outer(np.diag(JTJ1), np.diag(JTJ1)))
    assert np.max(np.abs(JTJu - JTJ1) / sc) < TOL_PASS and abs(chiu - chi1) <= TOL_PASS * chi1


@pytest.mark.parametrize('case', ['single', 'global'])
def test_step3_recomputing_kernel_equals_stored_jacobian_path(case, monkeypatch):
    """gfh_k_omega_jt (omega and J^T omega in one kernel, Jacobian rows recomputed in registers) against
    gfh_k_omega + k_jtv on the stored J (GADFIT_HIP_OMEGA_JT=0): bitwise the same omega and J^T omega,
    and both match the oracle."""
    if case == 'single':
        x, y, s = M.make_single(M.gauss8_numpy, M.gauss8_truth(), 30_000, 0.0, 100.0)
        t = trace_model(M.model_gauss8, 32)
        xs, ys, ws = [x], [y], [1.0 / s]
        pars = M.start_values(M.gauss8_truth()).reshape(1, 32); act = list(range(32)); glob = [0] * 32
    else:
        sizes = [1, 1024, 333, 2049, 57]
        xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
        ws = [1.0 / s for s in ss]
        t = trace_model(M.model_global7, 7)
        pars = np.array([M.start_values(tr) for tr in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
        act = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    pos = np.concatenate([[0], np.cumsum([a.size for a in xs])])
    p = orc.OracleProblem(t, xs, ys, ws, pars, act, glob)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    d1 = _lib.potr(JTJ0 + np.diag(np.diag(JTJ0)), JTr0)
    om0, jto0 = p.omega(d1, JT0)
    got = []
    for flag, merge in (('1', '1'), ('0', '1'), ('1', '0')):
        monkeypatch.setenv('GADFIT_HIP_OMEGA_JT', flag)
        monkeypatch.setenv('GADFIT_HIP_MERGE_SMALL', merge)      # reduce + assemble + publish of J^T omega as one launch, or three
        c = _lib.Context(0)
        try:
            c.set_model(t)
            c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
            jac, dim = c.jacobian_indices(act, glob)
            c.sweep(pars, act, jac, dim)
            got.append((c.omega(pars, d1).copy(), c.omega_vector().copy()))
        finally:
            c.close()
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert np.array_equal(got[0][0], got[2][0]) and np.array_equal(got[0][1], got[2][1])
    _close('JTomega', got[0][0], jto0, TOL_PASS, np.max(np.abs(jto0)))
    _close('omega', got[0][1], om0, TOL_PASS, np.max(np.abs(om0)))


@pytest.mark.parametrize('name', sorted(G.CXX_LOSS))
def test_cxx_loss_function_goldens_on_device(name):
    """c++/tests/lm_solver.cpp:499-565 on the device: Cauchy / Huber costs in the fused sweep kernel."""
    loss, iters, chi2_ref, tau, i00, b0, i01, b1 = G.CXX_LOSS[name]
    t, xs, ys, ws, pars, act, _ = CX.case(0)
    c = _lib.Context(0)
    try:
        c.set_loss(loss)
        c.set_model(t)
        c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), [0, 100, 200])
        out, r = c.fit(pars, CX.active_list(act), [0, 1, 0], lambda_=1.0, lam_incs=3, max_iter=iters)
        assert r.iterations == iters and (r.n_lookahead == 0 or loss == 0)
        chi2 = c.chi2(out)
    finally:
        c.close()
    want = np.array([[i00, tau, b0], [i01, tau, b1]])
    _close('chi2', chi2, chi2_ref, TOL_CXX)
    _close('pars', out, want, TOL_CXX)


@pytest.mark.parametrize('loss', [1, 2])
@pytest.mark.parametrize('fused', ['1', '0'])
def test_loss_sweep_vs_oracle(loss, fused, monkeypatch):
    """Scaled residuals, Jacobian, JTJ, JTres of one sweep against the oracle, fused and unfused kernels;
    residuals straddle |res| = 1 so both Huber branches are taken."""
    monkeypatch.setenv('GADFIT_HIP_FUSED', fused)
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 5000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH)
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], list(range(8)), [0] * 8, loss=loss)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    plain_res = p.chi2()[1]
    assert np.any(np.abs(plain_res) < 0.5) and np.any(np.abs(plain_res) > 1.0)
    c = _lib.Context(0)
    try:
        c.set_loss(loss)
        c.set_model(t)
        c.set_data(x, y, 1.0 / s, [0, x.size])
        jac, dim = c.jacobian_indices(list(range(8)), [0] * 8)
        JTJ, JTr, chi2 = c.sweep([start], list(range(8)), jac, dim)
        res = c.residuals(); J = c.jacobian(8)
        plain = c.chi2([start])
    finally:
        c.close()
    _observe(res=rel(res, res0), J=rel(J, JT0), JTJ=rel(JTJ, JTJ0), JTres=rel(JTr, JTr0))
    assert rel(res, res0) < TOL_LOSS and rel(J, JT0) < TOL_PASS
    assert rel(JTJ, JTJ0) < TOL_PASS and rel(JTr, JTr0) < TOL_LOSS
    assert abs(chi2 - np.sum(res0 * res0)) <= 1e-12 * chi2            # the sweep's sum is the robust one ...
    assert abs(plain - p.chi2()[0]) <= 1e-12 * plain and plain > chi2   # ... chi2() stays plain (lm_solver.cpp:513-529)


@pytest.mark.parametrize('name', sorted(G.CXX_SINGLE_INTEGRAL))
def test_cxx_single_integral_goldens_on_device(ctx, name):
    """c++/tests/numerical_integration.cpp 'Single integral' known answers on the GPU: active lower / upper /
    both bounds (reverse-mode Leibniz terms in the sweep, forward-mode ones in the omega kernel)."""
    model, fits = G.CXX_SINGLE_INTEGRAL[name]
    d = G.data()['cxx_lm_solver']
    x = np.array(d['x_data_single']); y = np.array(d['y_data_single'])
    t = trace_model(model, 2)
    ctx.set_model(t)
    ctx.set_data(x, y, np.ones_like(y), [0, x.size])
    pars = np.array([[10.0, 1.0]])
    for active, chi2_ref, a_ref, b_ref in fits:
        pars[0, 1] = 1.0
        out, r = ctx.fit(pars, active, [0, 0], lambda_=10.0, lam_incs=3, accth=0.9, max_iter=4)
        assert r.iterations == 4 and r.n_chi2 == 5 and r.n_omega == 4
        chi2 = ctx.chi2(out)
        _close('chi2', chi2, chi2_ref, TOL_CXX_INTEGRAL)
        _close('pars', out[0, :2], [a_ref, b_ref], TOL_CXX_INTEGRAL)
        pars = out.copy()


@pytest.mark.parametrize('name', sorted(G.CXX_NESTED))
def test_cxx_nested_integral_goldens_on_device(ctx, name):
    """c++/tests/numerical_integration.cpp 'Double integral (nested)' known answers on the GPU."""
    model, active, iters, chi2_ref, pars_ref = G.CXX_NESTED[name]
    d = G.data()['cxx_lm_solver']
    x = np.array(d['x_data_double']); y = np.array(d['y_data_double']); s = np.array(d['weights_double'])
    ctx.set_model(trace_model(model, 6))
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([G.CXX_NESTED_START], active, [0] * 6, lambda_=0.1, lam_incs=3, accth=0.9, max_iter=iters)
    assert r.iterations == iters and r.n_chi2 == iters + 1
    chi2 = ctx.chi2(out)
    _close('chi2', chi2, chi2_ref, TOL_CXX_NESTED)
    _close('pars', out[0], pars_ref, TOL_CXX_NESTED)


# ---- launch-path switches: same numbers whichever way the result reaches the host -------------------
@pytest.mark.parametrize('case', ['single', 'global', 'global20'])
def test_tail_and_kernarg_paths_are_bitwise_identical(case, monkeypatch):
    """GADFIT_HIP_TAIL (reduction + assembly + mailbox in the fused kernel's own tail instead of three
    more launches) and GADFIT_HIP_KERNARG (parameters as a kernel argument instead of an H2D copy) only
    change HOW the pass is launched: J^T J, J^T r, chi2, residuals and a whole fit are bitwise the same."""
    if case == 'single':
        x, y, s = M.make_single(M.gauss8_numpy, M.gauss8_truth(), 70_000, 0.0, 100.0)     # > 32 workgroups: every slice has members
        t = trace_model(M.model_gauss8, 32)
        xs, ys, ws = [x], [y], [1.0 / s]
        pars = M.start_values(M.gauss8_truth()).reshape(1, 32); act = list(range(32)); glob = [0] * 32
    elif case == 'global':
        sizes = [1, 1024, 333, 2049, 57, 5000]      # 7 active parameters: two workgroups per CU fit, so no tail here
        xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
        ws = [1.0 / s for s in ss]
        t = trace_model(M.model_global7, 7)
        pars = np.array([M.start_values(tr) for tr in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
        act = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    else:
        # 20 active parameters (2 tiles: one workgroup per CU, the tail is used), 3 datasets, the 5 widths global
        truth = M.gaussK_truth(5)
        x, y, s = M.make_single(M.gaussK_numpy(5), truth, 40_001, 0.0, 100.0)
        xs = [x[k::3] for k in range(3)]; ys = [y[k::3] for k in range(3)]     # every dataset covers all five peaks
        ws = [1.0 / s[k::3] for k in range(3)]
        t = trace_model(M.make_model_gaussK(5), 20)
        pars = np.array([M.start_values(truth)] * 3) * (1.0 + 0.01 * np.arange(3))[:, None]
        pars[:, 2::4] = M.start_values(truth)[2::4]                      # global parameters share their value
        act = list(range(20)); glob = [1 if k % 4 == 2 else 0 for k in range(20)]
    pos = np.concatenate([[0], np.cumsum([a.size for a in xs])])
    out = []
    for tail, karg in (('0', '0'), ('1', '1'), ('1', '0'), ('0', '1')):
        monkeypatch.setenv('GADFIT_HIP_TAIL', tail); monkeypatch.setenv('GADFIT_HIP_KERNARG', karg)
        c = _lib.Context(0)
        try:
            c.set_model(t)
            c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
            jac, dim = c.jacobian_indices(act, glob)
            a = c.sweep(pars, act, jac, dim)
            b = c.sweep(pars * 1.01, act, jac, dim)          # a second launch: the tail's counters were reset
            res = c.residuals().copy()
            p, r = c.fit(pars.copy(), act, glob, lambda_=1.0, max_iter=3)
            out.append((a, b, res, p.copy(), r.chi2))
        finally:
            c.close()
    for o in out[1:]:
        for k in (0, 1):
            assert np.array_equal(o[k][0], out[0][k][0]) and np.array_equal(o[k][1], out[0][k][1]) and o[k][2] == out[0][k][2]
        assert np.array_equal(o[2], out[0][2]) and np.array_equal(o[3], out[0][3]) and o[4] == out[0][4]


def test_more_ranks_than_points(ctx):
    """N = 5 points over 8 pseudo-ranks (gadfit.F90:978-983 gives the first 5 ranks one point each): ranks
    that own nothing return exact zeros and the sum over ranks is the single-image result."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 5, 1.0, 50.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH); act = list(range(8))
    ctx.set_model(t); ctx.set_data(x, y, 1.0 / s, [0, 5])
    jac, dim = ctx.jacobian_indices(act, [0] * 8)
    JTJ, JTr, chi2 = ctx.sweep([start], act, jac, dim)
    accJ = np.zeros_like(JTJ); accr = np.zeros_like(JTr); accc = 0.0; accchi = 0.0
    for r in range(8):
        c = _lib.Context(0)
        try:
            c.debug_set_rank(8, r)
            c.set_model(t); c.set_data(x, y, 1.0 / s, [0, 5])
            assert c.local_count() == (1 if r < 5 else 0)
            a, b, cc = c.sweep([start], act, jac, dim)
            if r >= 5:
                assert not a.any() and not b.any() and cc == 0.0 and c.chi2([start]) == 0.0
            accJ += a; accr += b; accc += cc; accchi += c.chi2([start])
        finally:
            c.close()
    sc = np.sqrt(np.outer(np.diag(JTJ), np.diag(JTJ)))
    assert np.max(np.abs(accJ - JTJ) / sc) < 1e-13 and np.max(np.abs(accr - JTr)) <= 1e-12 * np.max(np.abs(JTr))
    assert abs(accc - chi2) <= 1e-13 * chi2 and abs(accchi - chi2) <= 1e-13 * chi2


# ---- auxiliary per-point inputs (GFH_AUX) --------------------------------------------------------------
def _aux_models():
    from gadfit_amd.ad import aux, exp

    def m_sym(p, x):
        return p[0] + p[1] * x + p[2] * x ** 2 + p[3] * exp(-(x * x) / p[4])

    def m_aux(p, x):
        return p[0] + p[1] * x + p[2] * aux(0) + p[3] * exp(-aux(1) / p[4])
    return trace_model(m_sym, 5), trace_model(m_aux, 5)


def test_aux_columns_on_device_vs_oracle_and_vs_recorded_arithmetic(ctx):
    """Tabulated real functions of x (gfh_set_aux) through sweep, chi2, omega and a fit: equal to the oracle
    with the same columns, and to the model whose real arithmetic on x is recorded; ragged datasets
    (padding) and a pseudo-rank split with gfh_set_aux_local."""
    t_sym, t_aux = _aux_models()
    sizes = [700, 1, 1500, 333]
    rng = np.random.default_rng(5)
    xs = [np.sort(rng.uniform(-2.0, 3.0, n)) for n in sizes]
    ys = [np.cos(x) + 0.1 * x for x in xs]; ws = [0.5 + 0.1 * np.abs(x) for x in xs]
    X, Y, W = np.concatenate(xs), np.concatenate(ys), np.concatenate(ws)
    pos = np.concatenate([[0], np.cumsum(sizes)])
    cols = np.stack([X ** 2, X * X])
    pars = np.array([[0.3, 0.2, -0.1, 1.5, 2.0]] * 4) * (1.0 + 0.05 * np.arange(4))[:, None]
    act = [0, 1, 2, 3, 4]; glob = [0, 0, 1, 0, 1]
    p = orc.OracleProblem(t_aux, xs, ys, ws, pars, act, glob, aux=cols)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    ctx.set_model(t_aux); ctx.set_data(X, Y, W, pos)
    jac, dim = ctx.jacobian_indices(act, glob)
    with pytest.raises(_lib.GadfitHipError, match='auxiliary per-point column'):
        ctx.sweep(pars, act, jac, dim)                     # columns not set yet
    ctx.set_aux(cols)
    JTJ, JTr, chi2 = ctx.sweep(pars, act, jac, dim)
    sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
    _close('JTJ', JTJ, JTJ0, TOL_PASS, sc); _close('JTres', JTr, JTr0, TOL_PASS, np.max(np.abs(JTr0)))
    _close('chi2', [chi2, ctx.chi2(pars)], [chi0, chi0], TOL_PASS)
    _close('res', ctx.residuals(), res0, TOL_PASS, np.max(np.abs(res0)))
    d1 = _lib.potr(JTJ0 + np.diag(np.diag(JTJ0)), JTr0)
    om0, jto0 = p.omega(d1, JT0)
    _close('JTomega', ctx.omega(pars, d1), jto0, TOL_PASS, np.max(np.abs(jto0)))
    out, r = ctx.fit(pars.copy(), act, glob, lambda_=1.0, accth=0.9, max_iter=3)
    r0 = p.fit(lambda_=np.float32(1.0), accth=np.float32(0.9), max_iter=3)
    assert r.iterations == r0.iterations == 3
    _close('pars', out, p.pars, TOL_FIT)
    # the same model with its real arithmetic recorded (no columns): same fit
    ctx.set_model(t_sym); ctx.set_data(X, Y, W, pos)
    out2, r2 = ctx.fit(pars.copy(), act, glob, lambda_=1.0, accth=0.9, max_iter=3)
    assert np.max(np.abs(out2 - out) / np.abs(out)) < 1e-12
    # 3 pseudo-ranks, every other one with its local slice of the columns
    acc = np.zeros_like(JTJ); accc = 0.0
    for rk in range(3):
        c = _lib.Context(0)
        try:
            c.debug_set_rank(3, rk)
            c.set_model(t_aux); c.set_data(X, Y, W, pos)
            b, n = _lib.partition(X.size, 3, rk)
            if rk % 2:
                c.set_aux(cols[:, b:b + n], local=True)
            else:
                c.set_aux(cols)
            a, _, cc = c.sweep(pars, act, jac, dim)
            acc += a; accc += cc
        finally:
            c.close()
    assert np.max(np.abs(acc - JTJ0) / sc) < 1e-12 and abs(accc - chi0) <= 1e-12 * chi0


def test_pattern_only_transfer_of_global_fits_is_bitwise_the_dense_one(monkeypatch):
    """Global fit with 24 datasets (dim 99, beyond the in-kernel tail): the normal equations travel as their
    block-arrow pattern only (GADFIT_HIP_SPARSE, default) or densely -- same J^T J (both triangles, exact
    zeros off the pattern), J^T r, chi2 and fit; also through the public gfh_sweep into a dirty buffer."""
    sizes = [300 + 37 * k for k in range(24)]
    xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
    ws = [1.0 / s for s in ss]
    t = trace_model(M.model_global7, 7)
    pars = np.array([M.start_values(tr) for tr in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    act = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    pos = np.concatenate([[0], np.cumsum(sizes)])
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('GADFIT_HIP_SPARSE', flag)
        c = _lib.Context(0)
        try:
            c.set_model(t)
            c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
            jac, dim = c.jacobian_indices(act, glob)
            assert dim == 99
            a = c.sweep(pars, act, jac, dim)
            b = c.sweep(pars * 1.02, act, jac, dim)
            p, r = c.fit(pars.copy(), act, glob, lambda_=1.0, accth=0.9, max_iter=3)
            out.append((a, b, p.copy(), r.chi2, r.iterations))
        finally:
            c.close()
    (a1, b1, p1, c1, i1), (a0, b0, p0, c0, i0) = out
    for x1, x0 in ((a1, a0), (b1, b0)):
        assert np.array_equal(x1[0], x0[0]) and np.array_equal(x1[1], x0[1]) and x1[2] == x0[2]
        assert np.array_equal(x1[0], x1[0].T)
    assert np.array_equal(p1, p0) and c1 == c0 and i1 == i0 == 3
    # the pattern really is sparse: local blocks of different datasets are exact zeros
    jac = np.asarray(jac)
    assert a1[0][jac[3][0], jac[5][1]] == 0.0 and a1[0][jac[3][0], jac[3][1]] != 0.0
    p = orc.OracleProblem(t, xs, ys, ws, pars, act, glob)
    JTJ0, JTr0, _, _ = p.sweep()
    sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
    assert np.max(np.abs(a1[0] - JTJ0) / sc) < 1e-12
    # pattern-only layout under sharding: 5 pseudo-ranks, some owning nothing of most datasets; the layout is the same on
    # every rank (it is what RCCL would sum), so the per-rank results add up to the single-image one
    monkeypatch.setenv('GADFIT_HIP_SPARSE', '1')
    acc = np.zeros_like(a1[0]); accr = np.zeros_like(a1[1]); accc = 0.0
    X, Y, W = np.concatenate(xs), np.concatenate(ys), np.concatenate(ws)
    for rk in range(5):
        c = _lib.Context(0)
        try:
            c.debug_set_rank(5, rk)
            c.set_model(t); c.set_data(X, Y, W, pos)
            a, b, cc = c.sweep(pars, act, jac, 99)
            acc += a; accr += b; accc += cc
        finally:
            c.close()
    assert np.max(np.abs(acc - JTJ0) / sc) < 1e-12 and np.max(np.abs(accr - JTr0)) <= 1e-11 * np.max(np.abs(JTr0))
    assert abs(accc - a1[2]) <= 1e-12 * a1[2]


def test_tail_handoff_under_uneven_load_checks_every_word(monkeypatch):
    """The fence-free cross-workgroup hand-off of the fused kernel's tail, exercised the way the microarchitecture
    guide asks for: uneven load (datasets of very different sizes: workgroups finish at different times), a warm
    consumer (300 launches back to back, two alternating parameter sets), every word of every result compared
    with the three-launch chain's."""
    truth = M.gaussK_truth(5)
    x, y, s = M.make_single(M.gaussK_numpy(5), truth, 300_001, 0.0, 100.0)
    idx = np.arange(x.size)
    sel = [idx % 2 == 0, idx % 16 == 1, (idx % 2 == 1) & (idx % 16 != 1)]          # 150k, 19k, 131k points, all over the range
    xs = [x[m] for m in sel]; ys = [y[m] for m in sel]; ws = [1.0 / s[m] for m in sel]
    t = trace_model(M.make_model_gaussK(5), 20)
    pa = np.array([M.start_values(truth)] * 3) * (1.0 + 0.01 * np.arange(3))[:, None]
    pa[:, 2::4] = M.start_values(truth)[2::4]
    pb = pa * 1.013; pb[:, 2::4] = pa[0, 2::4] * 0.99
    act = list(range(20)); glob = [1 if k % 4 == 2 else 0 for k in range(20)]
    pos = np.concatenate([[0], np.cumsum([a.size for a in xs])])
    ref = None
    for tail in ('0', '1'):
        monkeypatch.setenv('GADFIT_HIP_TAIL', tail)
        c = _lib.Context(0)
        try:
            c.set_model(t)
            c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
            jac, dim = c.jacobian_indices(act, glob)
            if tail == '0':
                ref = (c.sweep(pa, act, jac, dim), c.sweep(pb, act, jac, dim))
                continue
            for it in range(300):
                got = c.sweep(pa if it % 2 == 0 else pb, act, jac, dim)
                want = ref[it % 2]
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2], it
        finally:
            c.close()


def test_small_problem_single_launch_finish_is_bitwise_the_chain(monkeypatch):
    """Few workgroups, few datasets, small dim (the sizes most fits have): the single-launch forms of the small
    reductions (k_jtv_finish, k_sum_publish; GADFIT_HIP_MERGE_SMALL) against the launch chains."""
    sizes = [700, 1, 1500, 333]
    xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
    ws = [1.0 / s for s in ss]
    t = trace_model(M.model_global7, 7)
    pars = np.array([M.start_values(tr) for tr in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    act = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    pos = np.concatenate([[0], np.cumsum(sizes)])
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('GADFIT_HIP_MERGE_SMALL', flag)
        c = _lib.Context(0)
        try:
            c.set_model(t)
            c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
            jac, dim = c.jacobian_indices(act, glob)
            a = c.sweep(pars, act, jac, dim); b = c.sweep(pars * 0.99, act, jac, dim)
            chi = c.chi2(pars)
            p, r = c.fit(pars.copy(), act, glob, lambda_=1.0, accth=0.9, max_iter=4)
            out.append((a, b, chi, p.copy(), r.chi2))
        finally:
            c.close()
    for k in (0, 1):
        assert np.array_equal(out[0][k][0], out[1][k][0]) and np.array_equal(out[0][k][1], out[1][k][1]) and out[0][k][2] == out[1][k][2]
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3]) and out[0][4] == out[1][4]


def test_use_ad_false_finite_differences_vs_oracle(ctx):
    """gadf_fit(use_ad=.false.) (gadfit.F90:583-584, 684-687, 721-728): STEP 1 by grad_finite and STEP 3 by
    dir_deriv_2nd_finite (fitfunction.F90:155-203) on the device against the oracle's restatement of the same
    formulas.  The reference has no known answer for this branch.  Tolerances: a forward difference divides a
    difference of two O(f) values by step = 1.5e-8 |p|, so the 1e-16 relative differences between device and
    host libm/FMA show up as ~1e-8 f/p in J (1e-6 asserted); the central second difference divides by 1.5e-8
    after cancelling to ~h^2 f'' with h = 1.2e-4, so its rounding noise is ~1e-8 |f| absolute."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 4000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    act = [0, 1, 2, 3, 5, 7]; start = M.start_values(M.EXP4_TRUTH)
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], act, [0] * 8, use_ad=False)
    pa = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], act, [0] * 8)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    JTJa = pa.sweep()[0]
    ctx.set_model(t); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    ctx.set_use_ad(False)
    try:
        jac, dim = ctx.jacobian_indices(act, [0] * 8)
        JTJ, JTr, chi2 = ctx.sweep([start], act, jac, dim)
        sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
        assert np.max(np.abs(JTJ - JTJ0) / sc) < 1e-6
        assert rel(ctx.residuals(), res0) < 1e-10      # values, no differencing: (y - f) w cancels to ~1e-2 of f w at the 1e-16 level of f
        # it IS the finite-difference Jacobian (differs from AD at the 1e-8..1e-6 level), not AD relabelled
        d_ad = np.max(np.abs(JTJ - JTJa) / sc)
        assert 1e-10 < d_ad < 1e-5
        J = ctx.jacobian(len(act))
        assert np.max(np.abs(J - JT0)) < 1e-6 * np.max(np.abs(JT0))
        d1 = _lib.potr(JTJ0 + np.diag(np.diag(JTJ0)), JTr0)
        om0, JTom0 = p.omega(d1, JT0)
        JTom = ctx.omega([start], d1)
        w = 1.0 / s
        assert np.max(np.abs(ctx.omega_vector() - om0)) < 1e-6 * np.max(np.abs(w * M.exp4_numpy(start, x)))     # ~1e-8 |f| w per point
        assert np.max(np.abs(JTom - JTom0)) < 1e-3 * np.max(np.abs(JTom0))
        # whole fits, fixed iteration count, with and without acceleration
        for accth in (0.0, 0.9):
            q = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], act, [0] * 8, use_ad=False)
            r0 = q.fit(lambda_=np.float32(1.0), accth=np.float32(accth), max_iter=6)
            out, r = ctx.fit([start], act, [0] * 8, lambda_=1.0, accth=float(np.float32(accth)), max_iter=6)
            assert r.iterations == r0.iterations == 6 and r.n_omega == r0.n_omega
            assert np.max(np.abs(out - q.pars) / np.abs(q.pars)) < 1e-6
            assert abs(r.chi2 - r0.chi2) < 1e-8 * r0.chi2
        # a parameter whose step underflows: the reference's error (fitfunction.F90:165-167)
        bad = start.copy(); bad[2] = 0.0
        with pytest.raises(_lib.GadfitHipError, match='Absolute value of parameter 3 is too small'):
            ctx.sweep([bad], act, jac, dim)
    finally:
        ctx.set_use_ad(True)
    # back on AD: bitwise the AD result again
    JTJb = ctx.sweep([start], act, jac, dim)[0]
    assert np.max(np.abs(JTJb - JTJa) / sc) < 1e-12


def test_use_ad_false_through_quadrature_and_global_fit(ctx):
    """Finite differences with every parameter passive also run through integrate() (value-only quadrature per
    evaluation) and through a global fit with per-dataset parameter blocks."""
    d = G.data()['2_integral_single']
    x = np.array(d['x_data']); y = np.array(d['y_data'])
    t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-12)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(y)], [[10.0, 1.0]], [0, 1], [0, 0], use_ad=False)
    r0 = p.fit(lambda_=np.float32(1.0), max_iter=4)
    ctx.set_model(t); ctx.set_data(x, y, np.ones_like(y), [0, x.size]); ctx.set_use_ad(False)
    try:
        out, r = ctx.fit([[10.0, 1.0]], [0, 1], [0, 0], lambda_=1.0, max_iter=4)
        assert r.iterations == r0.iterations and np.max(np.abs(out - p.pars) / np.abs(p.pars)) < 1e-5
        xs, ys, ss, truths = M.make_global7(3, 400)
        pars = np.array([M.start_values(tr) for tr in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
        t7 = trace_model(M.model_global7, 7); isg = [0, 0, 0, 0, 1, 1, 1]; act = list(range(7))
        q = orc.OracleProblem(t7, xs, ys, [1.0 / s for s in ss], [v.copy() for v in pars], act, isg, use_ad=False)
        r0 = q.fit(lambda_=np.float32(1.0), max_iter=5)
        ctx.set_model(t7)
        ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate([1.0 / s for s in ss]), np.arange(4) * 400)
        out, r = ctx.fit(pars.copy(), act, isg, lambda_=1.0, max_iter=5)
        assert r.iterations == r0.iterations and np.max(np.abs(out - q.pars) / np.maximum(np.abs(q.pars), 1e-3)) < 1e-5
    finally:
        ctx.set_use_ad(True)


def test_chi2_is_bitwise_the_sweeps_sum_of_squares(ctx):
    """gfh_k_chi2 uses the fused kernel's partition, thread-to-point map and order of additions: chi2() at the
    parameters of a sweep returns bitwise the sum r^2 that sweep returned (single curve, global fit, ragged sizes)."""
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 70001, 0.0, 100.0)
    t = trace_model(M.model_gauss8, 32)
    ctx.set_model(t)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    act = list(range(32))
    jac, dim = ctx.jacobian_indices(act, [0] * 32)
    for k in range(3):
        pars = M.start_values(truth) * (1.0 + 0.01 * k)
        _, _, chi_s = ctx.sweep([pars], act, jac, dim)
        assert ctx.chi2([pars]) == chi_s
    xs, ys, ss, truths = M.make_global7(5, [3000, 1, 2049, 777, 1024])
    t = trace_model(M.model_global7, 7)
    ctx.set_model(t)
    pos = np.concatenate([[0], np.cumsum([a.size for a in xs])])
    ctx.set_data(np.concatenate(xs), np.concatenate(ys), 1.0 / np.concatenate(ss), pos)
    act = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    jac, dim = ctx.jacobian_indices(act, glob)
    pars = truths * 1.03
    _, _, chi_s = ctx.sweep(pars, act, jac, dim)
    assert ctx.chi2(pars) == chi_s


def test_device_exp_special_values_and_accuracy(ctx):
    """The generated exp() (codegen.cpp, gfh_exp: the device library's operations with the range handled by a clamp):
    <= 1 ulp against correctly rounded values over the whole finite range, and the limits of the library's."""
    import mpmath
    t = trace_model(lambda p, x: p[0] * M.exp(p[1] * x), 2)
    xs = np.concatenate([np.linspace(-745.0, 709.0, 1501), np.array([-1e300, -1e6, -1075.5, -1074.9, -745.2, -708.5, -1e-300, 0.0, 1e-300, 709.7, 709.8, 1023.9,
                                                                    1024.1, 1e6, 1e300, -np.inf, np.inf, np.nan])])
    ctx.set_model(t)
    ctx.set_data(xs, np.zeros_like(xs), np.ones_like(xs), [0, xs.size])
    with np.errstate(all='ignore'):
        chi = ctx.chi2([[1.0, 1.0]])
    got = -ctx.residuals()
    mpmath.mp.prec = 200
    for xv, g in zip(xs, got):
        if np.isnan(xv):
            assert np.isnan(g)
        elif xv > 709.782712893384:
            assert g == np.inf, (xv, g)
        elif xv < -745.2:
            assert g == 0.0, (xv, g)
        else:
            want = mpmath.exp(mpmath.mpf(float(xv)))
            ulp = np.spacing(max(float(want), 2.2250738585072014e-308))
            assert abs(mpmath.mpf(float(g)) - want) <= 1.0 * ulp, (xv, g, float(want))
    assert np.isnan(chi)


def test_device_pow_accuracy(ctx):
    """The generated x**a (codegen.cpp, gfh_pow_ln: one extended-precision logarithm for x**a and ln x): against 60-digit
    references over x from 1e-300 to 1e300 and exponents whose a ln x spans +-690 -- value <= 3 ulp, the gradient
    <= 2 ulp of x**a ln x -- and the library's results (IEEE pow) for x = 0, negative x with an integral exponent, inf and NaN."""
    import mpmath
    mpmath.mp.prec = 200
    rng = np.random.default_rng(7)
    xs = np.concatenate([10.0 ** rng.uniform(-300, 300, 1500), rng.uniform(0.5, 2.0, 1500), 1.0 + rng.uniform(-1e-8, 1e-8, 300),
                         np.array([1.0, 2.0, 0.5, 2.2250738585072014e-308, 1.7976931348623157e308])])
    worst = 0.0
    for a in (7.5, -3.25, 0.5, 1e-3, 41.0, -0.8):
        t = trace_model(lambda p, x: x ** p[0], 1)            # real ** advar: value, and d/da = x**a ln x
        lim = 690.0 / abs(a)
        x = xs[np.abs(np.log(xs)) < lim]
        ctx.set_model(t)
        ctx.set_data(x, np.zeros_like(x), np.ones_like(x), [0, x.size])
        jac, dim = ctx.jacobian_indices([0], [0])
        ctx.sweep([[a]], [0], jac, dim)
        val = -ctx.residuals(); grad = ctx.jacobian(1)[:, 0]
        for xv, g, dg in zip(x, val, grad):
            want = mpmath.power(mpmath.mpf(float(xv)), mpmath.mpf(a))
            ulp = np.spacing(float(want))
            err = abs(mpmath.mpf(float(g)) - want) / ulp
            wantg = want * mpmath.log(mpmath.mpf(float(xv)))
            errg = abs(mpmath.mpf(float(dg)) - wantg) / max(np.spacing(abs(float(wantg))), 5e-324) if wantg != 0 else abs(dg)
            worst = max(worst, float(err), min(float(errg), 1e9) if abs(float(wantg)) > 1e-290 else 0.0)
            assert err <= 3.0 and (errg <= 4.0 or abs(float(wantg)) < 1e-290 or abs(float(mpmath.log(mpmath.mpf(float(xv))))) < 1e-7), (xv, a, g, float(want), float(err), float(errg))
    _observe(pow_ulp=worst)
    # special cases take the library's pow: IEEE results
    t = trace_model(lambda p, x: (x * p[1]) ** p[0], 2)        # advar ** advar
    x = np.array([0.0, -2.0, -2.0, np.inf, np.nan, 1e-320, 3.0])
    for a, want in ((2.0, [0.0, 4.0, 4.0, np.inf, np.nan, 0.0, 9.0]), (-1.0, [np.inf, -0.5, -0.5, 0.0, np.nan, np.inf, 1.0 / 3.0]), (0.5, [0.0, np.nan, np.nan, np.inf, np.nan, np.sqrt(1e-320), np.sqrt(3.0)])):
        ctx.set_model(t)
        ctx.set_data(x, np.zeros_like(x), np.ones_like(x), [0, x.size])
        with np.errstate(all='ignore'):
            ctx.chi2([[a, 1.0]])
        got = -ctx.residuals()
        for g, w_ in zip(got, want):
            assert (np.isnan(g) and np.isnan(w_)) or g == w_ or abs(g - w_) <= 4 * np.spacing(abs(w_)), (a, got, want)


def test_residuals_not_kept_under_mode_2_fail_loudly():
    """keep_jacobian mode 2: gfh_fit lets gfh_k_chi2 skip the residual store when nothing reads it; a read-back then
    raises instead of returning the residuals of an older pass, and the fit's numbers do not change."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 3000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH); act = list(range(8))
    c = _lib.Context(0)
    try:
        c.set_lookahead(False)                     # every trial chi2 comes from gfh_k_chi2
        c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size])
        p1, r1 = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=3)
        res1 = c.residuals().copy()
        c.set_keep_jacobian(2)
        p2, r2 = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=3)
        with pytest.raises(_lib.GadfitHipError, match='residual vector'):
            c.residuals()
        p3, r3 = c.fit([start], act, [0] * 8, lambda_=1.0, max_iter=3, grad_chi2=1e-30)     # the grad chi2 test reads res
        assert np.array_equal(c.residuals(), res1)
    finally:
        c.close()
    assert np.array_equal(p2, p1) and r2.chi2 == r1.chi2 and np.array_equal(p3, p1)


@pytest.mark.parametrize('error_type', [0, 1, 2, 3, 4])
def test_init_weights_every_mode_bit_exact(ctx, error_type):
    """init_weights (gadfit.F90:445-470) on the device, all five specifiers (NONE, SQRT_Y, PROPTO_Y, INVERSE_Y, USER), bit for bit
    the oracle's: incl. y = 0, |y| < 100*tiny (the reference's guard), the first value past it, negative y under SQRT_Y (NaN in
    both), huge and tiny magnitudes; two datasets so that pad slots sit between real points."""
    tiny = 2.2250738585072014e-308
    special = np.array([0.0, -0.0, 99.0 * tiny, -99.0 * tiny, 100.0 * tiny, np.nextafter(100.0 * tiny, 1.0), 101.0 * tiny, 5e-324, 1e-300,
                        1e300, 1.7976931348623157e308, -4.0, -1e-310, 1.0, 2.0, 3.0, 1e-8, 123456.789])
    rng = np.random.default_rng(7)
    y = np.concatenate([special, rng.lognormal(0.0, 3.0, 1500) * rng.choice([1.0, -1.0], 1500, p=[0.9, 0.1])])
    sigma = np.concatenate([np.full(special.size, 0.5), rng.lognormal(0.0, 2.0, 1500)])
    x = np.arange(y.size, dtype=float)
    t = trace_model(lambda p, x: p[0] * x, 1)
    ctx.set_model(t)
    ctx.set_data(x, y, sigma if error_type == 4 else np.ones_like(y), [0, 700, y.size])
    ctx.init_weights(error_type)
    with np.errstate(all='ignore'):
        want = orc.init_weights(error_type, y, sigma)
    got = ctx.weights()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    if error_type in (1, 2):
        assert np.all(got[:4] == 0.0) and got[4] != 0.0 and np.isfinite(got[4])          # the guard is a strict <
    if error_type == 1:
        assert np.isnan(got[11]) and np.isnan(want[11])                                   # sqrt of a negative y
    # and the weights are what the passes use: chi2 of a zero model = sum (y w)^2 over the finite weights
    if error_type in (0, 3, 4):
        chi = ctx.chi2([[0.0], [0.0]])
        with np.errstate(all='ignore'):
            ref = np.sum((y * want) ** 2)
        if np.isinf(ref):
            assert np.isinf(chi)
        else:
            assert abs(chi - ref) <= 1e-13 * ref


def test_init_weights_unknown_specifier_is_the_references_error(ctx):
    with pytest.raises(_lib.GadfitHipError, match='Unknown weight specifier'):
        ctx.init_weights(7)


@pytest.mark.parametrize('cfg', ['exp4', 'gauss8', 'global7'])
def test_reference_division_forms_agree_with_the_oracle_to_rounding(cfg, monkeypatch):
    """GADFIT_HIP_FAST_DIV=0: the generated code keeps the reference's expression shapes (a/r multiplies by 1/r, r/a divides,
    AD:814-913) instead of sharing one reciprocal per denominator; the Jacobian then agrees with the oracle to 2e-14 entry by
    entry [observed 1.3e-15, profiles/parity_r02.json], and the default (shared reciprocals) stays within 7e-13 of it."""
    monkeypatch.setenv('GADFIT_HIP_FAST_DIV', '0')
    c = _lib.Context(0)
    try:
        if cfg == 'exp4':
            x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 5003, 0.0, 100.0)
            _device_vs_oracle(c, trace_model(M.model_exp4, 8), [x], [y], [1.0 / s], [M.start_values(M.EXP4_TRUTH)], list(range(8)), [0] * 8,
                              tol=1e-13, jtol=2e-14, otol=1.5e-13)
        elif cfg == 'gauss8':
            truth = M.gauss8_truth()
            x, y, s = M.make_single(M.gauss8_numpy, truth, 4096 + 17, 0.0, 100.0)
            _device_vs_oracle(c, trace_model(M.model_gauss8, 32), [x], [y], [1.0 / s], [M.start_values(truth)], list(range(32)), [0] * 32,
                              tol=1e-13, jtol=2e-14, otol=1.5e-13)
        else:
            xs, ys, ss, truths = M.make_global7(5, [3000, 1, 2049, 777, 1024])
            pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
            _device_vs_oracle(c, trace_model(M.model_global7, 7), xs, ys, [1.0 / s for s in ss], pars, list(range(7)), [0, 0, 0, 0, 1, 1, 1],
                              tol=1e-13, jtol=2e-14, otol=1.5e-13)
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize('kernarg', ['1', '0'])
@pytest.mark.parametrize('n_datasets', [1, 3])
def test_step3_tangent_block_paths_vs_oracle(kernarg, n_datasets, monkeypatch):
    """STEP 3 with more than 16 parameters re-reads the tangent block (delta1 per parameter) through the scalar cache inside
    the point loop (codegen.cpp, GFH_TANGENTS): addressed through the kernarg segment when the blocks travel with the kernel
    arguments (one dataset: the block itself; several: dataset d's block at d * n_pars), through the device copy otherwise.
    Omega, J^T omega (gfh_k_omega_jt) and omega from gfh_k_omega (GADFIT_HIP_OMEGA_JT=0) against the oracle."""
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 6000 + 13, 0.0, 100.0)
    xs = [x[k::n_datasets] for k in range(n_datasets)]; ys = [y[k::n_datasets] for k in range(n_datasets)]
    ws = [1.0 / s[k::n_datasets] for k in range(n_datasets)]
    t = trace_model(M.model_gauss8, 32)
    pars = np.array([M.start_values(truth)] * n_datasets) * (1.0 + 0.01 * np.arange(n_datasets))[:, None]
    act = list(range(32)); glob = [0] * 32
    monkeypatch.setenv('GADFIT_HIP_KERNARG', kernarg)
    for jt in ('1', '0'):
        monkeypatch.setenv('GADFIT_HIP_OMEGA_JT', jt)
        c = _lib.Context(0)
        try:
            _device_vs_oracle(c, t, xs, ys, ws, pars, act, glob)
        finally:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize('K', [3, 8, 12, 16, 20])
def test_unfused_gram_kernels_every_tile_count_vs_oracle(K, monkeypatch):
    """GADFIT_HIP_FUSED=0: plain sweep + k_gram<T> over the stored Jacobian for T = 1 … 4 sixteen-row tiles (12, 32, 48, 64 active
    parameters; one and two tiles take the double-buffered loads, three and four the single buffer), and the one-launch 5 x 5 form
    of k_gram_block at 80 (which the fused kernel replaces by default since round 5)."""
    monkeypatch.setenv('GADFIT_HIP_FUSED', '0')
    truth = M.gaussK_truth(K)
    x, y, s = M.make_single(M.gaussK_numpy(K), truth, 3 * 1024 + 77, 0.0, 100.0)
    t = trace_model(M.make_model_gaussK(K), 4 * K)
    c = _lib.Context(0)
    try:
        _device_vs_oracle(c, t, [x], [y], [1.0 / s], [M.start_values(truth)], list(range(4 * K)), [0] * (4 * K))
    finally:
        c.close()


@pytest.mark.gpu
def test_forward_mode_square_at_zero_base(monkeypatch):
    """x**2 in forward mode at x == 0: the reference's formula (AD:1051-1054) is 0/0 there; GADFIT_HIP_FAST_DIV=0 and the oracle
    reproduce the NaN, the default device form (polynomial identity, codegen.cpp GFH_POWI) returns the derivative's value."""
    t = trace_model(lambda p, x: p[0] * (x - p[1]) ** 2, 2)
    xs = np.array([1.0, 2.0, 3.0]); ys = np.zeros(3); ws = np.ones(3)
    pars = np.array([[1.5, 2.0]])                       # the second point has x - p[1] == 0
    d1 = np.array([0.1, -0.2])
    out = {}
    for fd in ('1', '0'):
        monkeypatch.setenv('GADFIT_HIP_FAST_DIV', fd)
        c = _lib.Context(0)
        try:
            c.set_model(t); c.set_data(xs, ys, ws, [0, 3])
            jac, dim = c.jacobian_indices([0, 1], [0, 0])
            c.sweep(pars, [0, 1], jac, dim)
            c.omega(pars, d1)
            out[fd] = c.omega_vector().copy()
        finally:
            c.close()
    # f = A u^2, u = x - mu: second directional derivative along (dA, dmu) = 2 A dmu^2 - 4 u dA dmu; omega = -f'' w
    A, mu = pars[0]; dA, dmu = d1
    want = -(2 * A * dmu * dmu - 4 * (xs - mu) * dA * dmu)
    assert np.allclose(out['1'], want, rtol=1e-14, atol=0)
    assert np.isnan(out['0'][1]) and np.allclose(out['0'][[0, 2]], want[[0, 2]], rtol=1e-14, atol=0)
    om0 = orc.eval_forward(t, 2.0, pars[0], [1, 1], d1, np.zeros(2))
    assert np.isnan(om0[2])                               # the oracle follows the reference here


@pytest.mark.gpu
def test_timer_levels_sample_and_scale(ctx):
    """gfh_set_timer_detail: level 1 (default) brackets every 8th launch of a model kernel with HIP events and scales the sum to the
    launches, level 2 every launch, level 0 none; the launch counts are exact at every level."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 30011, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    ctx.set_model(t); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    act = list(range(8)); jac, dim = ctx.jacobian_indices(act, [0] * 8)
    pars = np.array([M.start_values(M.EXP4_TRUTH)])
    per_launch = {}
    for level, timed in ((1, 3), (2, 20), (0, 0)):
        ctx.set_timer_detail(level)
        ctx.reset_timers()
        for _ in range(20):
            ctx.sweep(pars, act, jac, dim)
        for _ in range(5):
            ctx.chi2(pars)
        tm = ctx.timers(); sp = ctx.timer_spread()
        assert tm[6] == 20 and tm[7] == 5 and sp[3] == timed            # launches 0, 8, 16 of 20 are the sampled ones
        if level:
            assert tm[0] > 0 and tm[4] > 0 and sp[0] <= tm[0] / 20 * 1.0000001 <= sp[1] * 1.0000001
            per_launch[level] = tm[0] / 20
        else:
            assert tm[0] == 0 and tm[4] == 0
    assert 0.5 < per_launch[1] / per_launch[2] < 2.0                    # the scaled sample is the same quantity as the full sum
    ctx.set_timer_detail(1)


def test_switching_keep_jacobian_after_a_fit_invalidates_the_sweep_state():
    """keep_jacobian 2 -> fit of a quadrature model (which keeps storing J, so only the residual store is switched off) -> keep_jacobian 1
    flips the residual store alone: the kernels of the old setting are gone, and the calls that build on "the last sweep" must say so
    instead of reaching through a stale kernel handle (round-2 advisor finding)."""
    d = G.data()['2_integral_single']
    x = np.array(d['x_data']); y = np.array(d['y_data'])
    t = trace_model(G.model_integral_single, 2)
    t.set_integration(rel_error=1e-10)
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(x, y, np.ones_like(y), [0, x.size])
        c.set_keep_jacobian(2)
        out, r = c.fit([[10.0, 1.0]], [0, 1], [0, 0], lambda_=10.0, max_iter=2)
        c.set_keep_jacobian(1)
        for call in (lambda: c.omega(out, np.zeros(2)), lambda: c.time_kernel(0, 1), lambda: c.time_kernel(2, 1)):
            with pytest.raises(_lib.GadfitHipError, match='gfh_sweep'):
                call()
        # and a new sweep puts everything back
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        c.sweep(out, [0, 1], jac, dim)
        assert c.omega(out, np.array([1e-3, 1e-3])).shape == (2,) and c.time_kernel(0, 1) > 0
    finally:
        c.close()


def test_abscissas_come_back_from_the_device(ctx):
    """gfh_get_abscissas: what a host layer reads instead of keeping its own copy of x -- ragged datasets (per-dataset padding on
    the device), also through a device group of three members (every member's range)"""
    rng = np.random.default_rng(5)
    sizes = [1, 700, 513, 0, 2049]
    xs = [np.sort(rng.uniform(0.0, 50.0, n)) for n in sizes]
    X = np.concatenate(xs); pos = np.concatenate([[0], np.cumsum(sizes)])
    ctx.set_model(trace_model(M.model_exp2, 4))
    ctx.set_data(X, np.ones_like(X), np.ones_like(X), pos)
    assert np.array_equal(ctx.abscissas(), X)
    import os
    os.environ['GADFIT_HIP_GROUP_WRAP'] = '1'
    try:
        g = _lib.Context(devices=3)
    finally:
        os.environ.pop('GADFIT_HIP_GROUP_WRAP', None)
    try:
        g.set_model(trace_model(M.model_exp2, 4))
        g.set_data(X, np.ones_like(X), np.ones_like(X), pos)
        assert np.array_equal(g.abscissas(), X)
    finally:
        g.close()


def test_upload_in_the_background_with_a_host_copy_beside_it(ctx):
    """gfh_set_data_begin returns at once (the next call waits for the upload); a host-to-host copy queued before it runs on a
    thread of its own beside the upload and is joined by gfh_wait_host_copy -- what the Fortran layer's first gadf_fit does with
    the user's arrays"""
    n = 300_000
    x, y, s = M.make_single(M.exp2_numpy, M.EXP2_TRUTH, n, 0.0, 50.0)
    mine = np.empty_like(x)
    ctx.set_model(trace_model(M.model_exp2, 4))
    ctx.queue_host_copy(mine, x)
    ctx.set_data_begin(x, y, 1.0 / s, [0, n])
    c1 = ctx.chi2([M.EXP2_TRUTH])                 # waits for the upload, not for the host copy
    ctx.wait_host_copy()
    assert np.array_equal(mine, x)
    ctx.set_data(x, y, 1.0 / s, [0, n])
    assert ctx.chi2([M.EXP2_TRUTH]) == c1 and np.array_equal(ctx.abscissas(), x)


@pytest.mark.gpu
def test_contexts_created_one_after_the_other_share_nothing_but_memory():
    """A destroyed context leaves its stream, events, pinned buffers and small device blocks to the next context of the device
    (context.cpp, BaseRes / dev_release).  Fits of different models and sizes through a row of short-lived contexts give bitwise
    what each gives in a process-fresh order: nothing of a previous context's state or data may show."""
    from gadfit_amd.ad import exp
    rng = np.random.default_rng(5)
    tape4 = trace_model(lambda p, x: p[0] * exp(-((x - p[1]) / p[2]) ** 2) + p[3], 4)
    tape2 = trace_model(lambda p, x: p[0] * exp(-p[1] * x), 2)
    jobs = []
    for k in range(8):
        n = int(rng.integers(50, 5000))
        x = np.sort(rng.uniform(0.0, 10.0, n))
        if k % 2:
            y = 3.0 * np.exp(-((x - 4.0 - 0.1 * k) / 0.8) ** 2) + 0.5 + 1e-3 * rng.standard_normal(n)
            jobs.append((tape4, x, y, np.array([[2.5, 4.3, 1.0, 0.3]]), [0, 1, 2, 3]))
        else:
            y = 2.0 * np.exp(-0.3 * x) + 1e-3 * rng.standard_normal(n)
            jobs.append((tape2, x, y, np.array([[1.5, 0.2]]), [0, 1]))

    def run(job):
        tape, x, y, start, act = job
        c = _lib.Context(0)
        try:
            c.set_model(tape); c.set_data(x, y, np.ones_like(x), [0, len(x)])
            p, r = c.fit(start.copy(), act, [0] * len(act), lambda_=1.0, max_iter=20, accth=0.9)
            jac, dim = c.jacobian_indices(act, [0] * len(act))
            JTJ, JTr, chi2 = c.sweep(p, act, jac, dim)
            return p.copy(), r.chi2, JTJ.copy(), JTr.copy(), chi2, c.residuals().copy()
        finally:
            c.close()
    forward = [run(j) for j in jobs]
    backward = [run(j) for j in reversed(jobs)][::-1]
    for a, b in zip(forward, backward):
        for u, v in zip(a, b):
            assert np.array_equal(np.asarray(u), np.asarray(v))


def test_value_of_an_advar_in_real_arithmetic_against_the_oracle():
    """value(p) = GFH_VAL: the VALUE of an AD variable taken into plain real arithmetic (what `p%val` is in a Fortran eval()): the
    real follows the parameter, no derivative flows through it -- residuals, Jacobian (without that dependence), J^T J, chi2() and
    STEP 3 against the oracle"""
    from tests import branching as B
    x, y = B.param_val_data()
    w = np.ones_like(x)
    t = trace_model(B.model_param_val, 3)
    pars = [[4.5, 22.0, 1.2]]
    p = orc.OracleProblem(t, [x], [y], [w], pars, [0, 1, 2], [0] * 3)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    c = _lib.Context(0)
    try:
        c.set_model(t); c.set_data(x, y, w, [0, x.size])
        jac, dim = c.jacobian_indices([0, 1, 2], [0] * 3)
        JTJ, JTr, chi2 = c.sweep(pars, [0, 1, 2], jac, dim)
        sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
        assert np.max(np.abs(JTJ - JTJ0) / sc) < 1e-13 and abs(chi2 - chi0) <= 1e-13 * chi0 and abs(c.chi2(pars) - chi0) <= 1e-13 * chi0
        assert np.max(np.abs(c.residuals() - res0)) <= 1e-13 * np.max(np.abs(res0))
        d1 = np.array([0.3, -0.05, 0.02])
        om0, jto0 = p.omega(d1, JT0)
        jto = c.omega(pars, d1)
        assert np.max(np.abs(c.omega_vector() - om0)) <= 1e-13 * np.max(np.abs(om0))
        out, r = c.fit(pars, [0, 1, 2], [0] * 3, lambda_=1.0, max_iter=8)
        r0 = p.fit(lambda_=1.0, max_iter=8)
        assert r.iterations == r0.iterations and np.max(np.abs(out - p.pars) / np.abs(p.pars)) < 1e-10
    finally:
        c.close()
