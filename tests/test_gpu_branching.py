"""GPU parity tests of branching eval() bodies (guards / variants; automatic_differentiation.F90:315-395, gadfit.F90:679-690):
the device walks the recorded decision tree per point at the current parameters.  Checked through the C ABI against the oracle
(which takes, per point, the recorded path whose comparisons hold) -- J / res 7e-13, sums 1e-13, fits 1e-12 with equal pass counts."""
import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd import tape as T
from oracle import binding as orc
from tests import branching as B
from tests import models as M
from tests.test_cpu_branching import reference_comparisons
from tests.test_gpu_parity import _device_vs_oracle, rel

pytestmark = pytest.mark.gpu

TOL_FIT = 1e-12


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _variants(model, n_pars, xs, pars):
    V = T.Variants(model, n_pars)
    V.explore(xs, pars)
    return V


@pytest.mark.parametrize('active', [[0, 1, 2, 3], [0, 2, 3]])
def test_two_segments_vs_oracle(ctx, active):
    """piecewise in x, breakpoint an active / a passive parameter: sweep, chi2, STEP 3 and the convergence reductions"""
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 5003)
    p0 = B.PIECEWISE2_TRUTH * np.array([1.03, 0.96, 1.05, 0.97])
    V = _variants(B.model_piecewise2, 4, [x[0], x[-1]], p0)
    assert len(V) == 2
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], active, [0] * 4)
    assert ctx.n_variants() == 2 and not ctx.unseen_log


def test_three_segments_vs_oracle(ctx):
    x, y, s = B.make_data(B.piecewise3_numpy, B.PIECEWISE3_TRUTH, 4099)
    p0 = B.PIECEWISE3_TRUTH * np.array([1.04, 0.97, 1.02, 0.95, 1.05, 0.96])
    V = _variants(B.model_piecewise3, 6, x[::97], p0)
    assert len(V) == 3
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], list(range(6)), [0] * 6)


def test_max_of_two_advars_vs_oracle(ctx):
    """max(p1, p2 x): advar > advar, the result IS one of the operands"""
    x, y, s = B.make_data(B.max_numpy, B.MAX_TRUTH, 3001)
    p0 = B.MAX_TRUTH * np.array([1.05, 0.95, 1.05, 0.95])
    V = _variants(B.model_max, 4, [x[0], x[-1]], p0)
    assert len(V) == 2
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4)


def test_nested_comparisons_vs_oracle(ctx):
    """a clipped term: two comparisons in a row, three of the four combinations occur in the data"""
    x, y, s = B.make_data(B.clip_numpy, B.CLIP_TRUTH, 2500)
    p0 = B.CLIP_TRUTH * np.array([1.05, 0.97, 1.04, 0.9])
    V = _variants(B.model_clip, 4, x[::50], p0)
    assert len(V) == 3
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4)


def test_comparison_goldens_on_device(ctx):
    """the fourteen comparisons of ad_forward_mode.F90:9-25 decided by the device: each adds its bit to the model value"""
    from tests.golden import goldens as G

    def model(p, x):
        y = p[0] * 0.0
        for k, c in enumerate(reference_comparisons(p[0], p[1])):
            if c:
                y = y + float(2 ** k)
        return y
    V = T.Variants(model, 2)
    V.add_point(0.0, [G.D(1), G.D(2)])
    ctx.set_model(V)
    ctx.set_data([0.0], [0.0], [1.0], [0, 1])
    ctx.init_weights(0)
    jac, dim = ctx.jacobian_indices([0, 1], [0, 0])
    ctx.sweep([[G.D(1), G.D(2)]], [0, 1], jac, dim)
    assert -ctx.residuals()[0] == 2.0 ** 14 - 1 and not ctx.unseen_log
    # with a and b swapped the two advar/advar comparisons (and most of the others) come out false: a path the device has not
    # been given -- it reports the point, the handler records it, the pass is repeated
    ctx.sweep([[G.D(2), G.D(1)]], [0, 1], jac, dim)
    want = sum(2 ** k for k, c in enumerate(reference_comparisons(G.D(2), G.D(1))) if c)
    assert -ctx.residuals()[0] == float(want) and ctx.unseen_log and ctx.n_variants() >= 2


def test_unrecorded_branch_is_reported_recorded_and_the_pass_repeated(ctx):
    """only the first segment is recorded; the device meets the second one, the handler adds it, results as with both from the start"""
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 3000)
    p0 = B.PIECEWISE2_TRUTH * np.array([1.03, 0.96, 1.05, 0.97])
    Vfull = _variants(B.model_piecewise2, 4, [x[0], x[-1]], p0)
    p = orc.OracleProblem(Vfull, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4)
    JTJ0, JTr0, res0, _ = p.sweep()
    V = _variants(B.model_piecewise2, 4, [x[0]], p0)
    assert len(V) == 1
    ctx.set_model(V)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    jac, dim = ctx.jacobian_indices([0, 1, 2, 3], [0] * 4)
    JTJ, JTr, chi2 = ctx.sweep([p0], [0, 1, 2, 3], jac, dim)
    assert ctx.n_variants() == 2 and len(V) == 2 and ctx.unseen_log
    assert all(script == [False] for (_, _, script, _) in ctx.unseen_log)
    sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
    assert np.max(np.abs(JTJ - JTJ0) / sc) < 1e-13 and rel(JTr, JTr0) < 1e-12
    assert np.max(np.abs(ctx.residuals() - res0)) <= 7e-13 * np.max(np.abs(res0))
    # chi2() first on a fresh model: the same recovery inside gfh_chi2
    V = _variants(B.model_piecewise2, 4, [x[-1]], p0)
    ctx.set_model(V)
    assert abs(ctx.chi2([p0]) - chi2) <= 1e-13 * chi2 and ctx.n_variants() == 2


def test_without_a_handler_an_unrecorded_branch_is_an_error(ctx):
    import ctypes as C
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 600)
    V = _variants(B.model_piecewise2, 4, [x[0]], B.PIECEWISE2_TRUTH)
    c2 = _lib.Context(0)
    try:
        n, arr = V.c_array
        c2.n_pars = 4; c2._tape = V
        c2._chk(_lib.lib().gfh_set_model_variants(c2._h, n, arr, -1))       # the C entry point alone: no handler installed
        c2.set_data(x, y, 1.0 / s, [0, x.size])
        with pytest.raises(_lib.GadfitHipError, match='none of the recorded variants covers'):
            c2.chi2([B.PIECEWISE2_TRUTH])
    finally:
        c2.close()


# (six iterations: the seventh already moves chi2 in the last digit only, where accept / reject is decided by rounding)
@pytest.mark.parametrize('opts', [dict(lambda_=1.0, max_iter=6), dict(lambda_=1.0, max_iter=6, accth=0.9),
                                  dict(lambda_=0.1, max_iter=5, nielsen=1)])
def test_fit_with_a_moving_breakpoint_vs_oracle(ctx, opts):
    """the breakpoint is an active parameter and starts 8 % off: points change segment from iteration to iteration (the guard
    flips at the current parameters, on the device); same passes and parameters as the oracle's fit"""
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 4000)
    start = B.PIECEWISE2_TRUTH * np.array([1.05, 0.92, 1.1, 0.93])
    V = _variants(B.model_piecewise2, 4, [x[0], x[-1]], start)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r0 = p.fit(**opts)
    ctx.set_model(V)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], [0, 1, 2, 3], [0] * 4, **opts)
    assert (r.iterations, r.n_chi2, r.n_omega) == (r0.iterations, r0.n_chi2, r0.n_omega)
    assert rel(out, p.pars) < TOL_FIT
    assert abs(out[0][1] - B.PIECEWISE2_TRUTH[1]) < 0.2            # the breakpoint was found
    # points between the start and the fitted breakpoint changed segment on the way
    lo, hi = sorted([start[1], out[0][1]])
    assert np.count_nonzero((x > lo) & (x < hi)) > 50


def test_path_met_for_the_first_time_inside_a_fit(ctx):
    """the clipped ramp with its upper clip level passive: at the start the slope is so small that no point reaches the clip, so
    that path does not exist yet in the recordings; the fit steepens the ramp, the device meets the clip inside gfh_fit (a trial
    chi2 or a sweep), the handler records it, the pass is repeated -- same passes and parameters as the oracle with all paths"""
    x, y, s = B.make_data(B.clip_numpy, B.CLIP_TRUTH, 3000)
    start = np.array([0.07, 17.0, 6.0, 1.3])
    assert np.max(start[0] * (x - start[1])) < start[2]
    active = [0, 1, 3]
    Vfull = T.Variants(B.model_clip, 4)
    Vfull.explore(x[::30], start); Vfull.explore(x[::30], B.CLIP_TRUTH)
    assert len(Vfull) == 3
    p = orc.OracleProblem(Vfull, [x], [y], [1.0 / s], [start], active, [0] * 4)
    r0 = p.fit(lambda_=1.0, max_iter=6)
    V = _variants(B.model_clip, 4, x[::30], start)
    assert len(V) == 2
    ctx.set_model(V)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], active, [0] * 4, lambda_=1.0, max_iter=6)
    assert (r.iterations, r.n_chi2) == (r0.iterations, r0.n_chi2)
    assert rel(out, p.pars) < TOL_FIT
    assert ctx.n_variants() == 3 and ctx.unseen_log           # the clip was met on the way
    assert abs(out[0][0] - B.CLIP_TRUTH[0]) < 0.01


@pytest.mark.parametrize('pars', [[2.6, 2.4, 11.0], [2.4, 2.6, 11.0]])
def test_guard_between_two_parameters(ctx, pars):
    """if (p0 > p1): advar > advar on parameters alone -- one outcome for every point"""
    truth = np.array([1.0, 4.0, 12.0])
    x, y, s = B.make_data(B.par_order_numpy, truth, 2000)
    V = T.Variants(B.model_par_order, 3)
    V.add_point(1.0, [2.6, 2.4, 11.0]); V.add_point(1.0, [2.4, 2.6, 11.0])
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [pars], [0, 1, 2], [0] * 3)


def test_branching_model_two_datasets_with_their_own_breakpoints(ctx):
    """global fit: the breakpoint is local, so the same point index falls on different sides in the two datasets"""
    t1 = B.PIECEWISE2_TRUTH.copy(); t2 = B.PIECEWISE2_TRUTH * np.array([0.8, 1.6, 1.0, 1.0])
    x1, y1, s1 = B.make_data(B.piecewise2_numpy, t1, 1500)
    x2, y2, s2 = B.make_data(B.piecewise2_numpy, t2, 1111, seed=M.SEED + 5)
    pars = np.array([t1 * [1.02, 0.97, 1.03, 0.98], t2 * [0.98, 1.02, 0.97, 1.03]])
    pars[1][3] = pars[0][3]                                    # the decay time is global
    V = _variants(B.model_piecewise2, 4, [x1[0], x1[-1]], pars[0])
    _device_vs_oracle(ctx, V, [x1, x2], [y1, y2], [1.0 / s1, 1.0 / s2], pars, [0, 1, 2, 3], [0, 0, 0, 1])


def test_branching_model_with_finite_differences(ctx):
    """use_ad = .false.: every evaluation of the difference quotients takes its own branch, as the reference's eval() would"""
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 1200)
    p0 = B.PIECEWISE2_TRUTH * np.array([1.03, 0.96, 1.05, 0.97])
    V = _variants(B.model_piecewise2, 4, [x[0], x[-1]], p0)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4, use_ad=False)
    JTJ0, JTr0, res0, _ = p.sweep()
    ctx.set_model(V)
    ctx.set_use_ad(False)
    try:
        ctx.set_data(x, y, 1.0 / s, [0, x.size])
        jac, dim = ctx.jacobian_indices([0, 1, 2, 3], [0] * 4)
        JTJ, JTr, chi2 = ctx.sweep([p0], [0, 1, 2, 3], jac, dim)
    finally:
        ctx.set_use_ad(True)
    sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
    assert np.max(np.abs(JTJ - JTJ0) / sc) < 1e-7 and rel(JTr, JTr0) < 1e-6      # difference quotients: step sqrt(eps)


def test_variants_without_a_comparison_follow_the_hint_column(ctx):
    """what a Fortran eval() that branches on the plain real x looks like to the device: two tapes with no guard between them and a
    per-point column naming the one each point takes"""
    from gadfit_amd.ad import exp, trace_model
    ta = trace_model(lambda p, x: p[0] * x + p[1], 2)
    tb = trace_model(lambda p, x: p[0] * exp(-(x / p[1])), 2)
    V = T.Variants(None, 2)
    V.tapes = [ta, tb]
    for t in V.tapes:
        t.n_aux = 1
    n = 1500
    x = np.linspace(0.1, 30.0, n)
    hint = (x > 11.0).astype(np.float64)
    truth = np.array([1.7, 6.0])
    f = np.where(hint > 0, truth[0] * np.exp(-x / truth[1]), truth[0] * x + truth[1])
    s = 0.01 * (1 + np.abs(f)); y = f + s * M.normal(n, M.SEED)
    p0 = truth * [1.03, 0.96]
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [p0], [0, 1], [0, 0], aux=hint[None, :], hint=hint)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    ctx.set_model(V, hint_aux=0)
    ctx.set_data(x, y, 1.0 / s, [0, n])
    ctx.set_aux(hint[None, :])
    jac, dim = ctx.jacobian_indices([0, 1], [0, 0])
    JTJ, JTr, chi2 = ctx.sweep([p0], [0, 1], jac, dim)
    assert np.max(np.abs(ctx.residuals() - res0)) <= 7e-13 * np.max(np.abs(res0))
    assert np.max(np.abs(ctx.jacobian(2) - JT0)) <= 7e-13 * np.max(np.abs(JT0))
    sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
    assert np.max(np.abs(JTJ - JTJ0) / sc) < 1e-13


def test_lookahead_schedule_equals_reference_schedule_for_a_branching_model(ctx):
    """chi2() stays bitwise the sweep's sum r^2 (one selector serves both), so the two schedules return the same bits"""
    x, y, s = B.make_data(B.piecewise3_numpy, B.PIECEWISE3_TRUTH, 6000)
    start = B.PIECEWISE3_TRUTH * np.array([1.04, 0.95, 1.03, 0.95, 1.05, 0.96])
    V = _variants(B.model_piecewise3, 6, x[::40], start)
    ctx.set_model(V)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    outs = []
    for la in (1, 0):
        ctx.set_lookahead(la)
        out, r = ctx.fit([start], list(range(6)), [0] * 6, lambda_=1.0, max_iter=7)
        outs.append((out.copy(), r.chi2, r.iterations))
    ctx.set_lookahead(1)
    assert outs[0][2] == outs[1][2] and outs[0][1] == outs[1][1] and np.array_equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize('n_active', [32, 33])
def test_branching_model_on_the_matrix_core_path(ctx, n_active):
    """32 / 33 active parameters: the variant bodies feed the fused kernel's LDS stage and FP64 matrix instructions (33: three row
    tiles); lanes of one wave sit on both sides of the saturation level"""
    truth = B.gauss8_saturating_truth()
    x, y, s = M.make_single(B.gauss8_saturating_numpy, truth, 6007, 0.0, 100.0)
    p0 = np.concatenate([M.start_values(truth[:32]), [3.3]])
    V = _variants(B.model_gauss8_saturating, 33, x[::61], p0)
    assert len(V) == 2
    sat = M.gauss8_numpy(p0, x) > p0[32]
    assert 200 < np.count_nonzero(sat) < x.size - 200
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], list(range(n_active)), [0] * 33)
    assert not ctx.unseen_log


def test_fit_of_a_branching_model_on_the_matrix_core_path(ctx):
    truth = B.gauss8_saturating_truth()
    x, y, s = M.make_single(B.gauss8_saturating_numpy, truth, 20011, 0.0, 100.0)
    start = np.concatenate([M.start_values(truth[:32]), [3.4]])
    V = _variants(B.model_gauss8_saturating, 33, x[::61], start)
    active = list(range(33))
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], active, [0] * 33)
    r0 = p.fit(lambda_=1.0, max_iter=5)
    ctx.set_model(V)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], active, [0] * 33, lambda_=1.0, max_iter=5)
    assert (r.iterations, r.n_chi2) == (r0.iterations, r0.n_chi2)
    assert rel(out, p.pars) < 1e-11
    assert r.chi2 < 0.5 * ctx.chi2([start])


def test_branching_model_with_a_global_breakpoint(ctx):
    """three datasets, breakpoint and decay time global, level and slope local: dim = 2 + 3 * 2"""
    xs, ys, ws, pars = [], [], [], []
    for d, (a, c) in enumerate([(4.0, 0.08), (2.5, 0.05), (6.0, 0.11)]):
        t = np.array([a, B.PIECEWISE2_TRUTH[1], c, B.PIECEWISE2_TRUTH[3]])
        x, y, s = B.make_data(B.piecewise2_numpy, t, 1200 + 311 * d, seed=M.SEED + d)
        xs.append(x); ys.append(y); ws.append(1.0 / s)
        pars.append(t * [1.03, 0.95, 0.96, 1.04])
    pars = np.array(pars)
    V = _variants(B.model_piecewise2, 4, [xs[0][0], xs[0][-1]], pars[0])
    p = _device_vs_oracle(ctx, V, xs, ys, ws, pars, [0, 1, 2, 3], [0, 1, 0, 1])
    assert p.dim == 8
    r0 = p.fit(lambda_=1.0, max_iter=5)
    out, r = ctx.fit(pars, [0, 1, 2, 3], [0, 1, 0, 1], lambda_=1.0, max_iter=5)
    assert (r.iterations, r.n_chi2) == (r0.iterations, r0.n_chi2)
    assert rel(out, p.pars) < TOL_FIT
    assert abs(out[0][1] - B.PIECEWISE2_TRUTH[1]) < 0.2 and out[0][1] == out[1][1] == out[2][1]


@pytest.mark.parametrize('loss', [1, 2])
def test_branching_model_with_a_robust_loss(loss):
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 3000)
    p0 = B.PIECEWISE2_TRUTH * np.array([1.03, 0.96, 1.05, 0.97])
    V = _variants(B.model_piecewise2, 4, [x[0], x[-1]], p0)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4, loss=loss)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    c = _lib.Context(0)
    try:
        c.set_loss(loss)
        c.set_model(V)
        c.set_data(x, y, 1.0 / s, [0, x.size])
        jac, dim = c.jacobian_indices([0, 1, 2, 3], [0] * 4)
        JTJ, JTr, chi2 = c.sweep([p0], [0, 1, 2, 3], jac, dim)
        res = c.residuals()
    finally:
        c.close()
    assert rel(res, res0) < 1e-12 and rel(JTJ, JTJ0) < 1e-12 and rel(JTr, JTr0) < 1e-12


def _integral_then_line(n):
    truth = B.INTEGRAL_THEN_LINE_TRUTH
    x = np.linspace(0.05, 4.0, n)
    f = B.integral_then_line_numpy(truth, x)
    s = 0.002 * (1.0 + np.abs(f))
    return x, f + s * M.normal(n, M.SEED + 3), s


def test_branching_model_with_quadrature_on_both_sides(ctx):
    """integrate() in both variants (one integrand sub-tape, two call sites, the second with an active upper bound): sweep, chi2,
    STEP 3 against the oracle; then the mesh hand-over between passes at the same parameters stays bitwise with variants"""
    x, y, s = _integral_then_line(1501)
    p0 = B.INTEGRAL_THEN_LINE_TRUTH * np.array([1.04, 0.95, 1.05, 0.9])
    V = T.Variants(B.model_integral_then_line, 4, configure=lambda t: t.set_integration(rel_error=1e-11))
    V.explore([x[0], x[-1]], p0)
    assert len(V) == 2
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4, tol=3e-12, jtol=5e-11, otol=3e-11)
    # the same passes again in the order of an LM iteration: chi2 -> sweep -> omega replay what the pass before them recorded
    jac, dim = ctx.jacobian_indices([0, 1, 2, 3], [0] * 4)
    import os
    got = []
    for mesh in ('1', '0'):
        os.environ['GADFIT_HIP_MESH'] = mesh
        c = _lib.Context(0)
        try:
            c.set_model(V); c.set_data(x, y, 1.0 / s, [0, x.size])
            chi = c.chi2([p0])
            JTJ, JTr, chi2 = c.sweep([p0], [0, 1, 2, 3], jac, dim)
            jto = c.omega([p0], np.linalg.solve(JTJ + np.diag(np.diag(JTJ)), JTr))
            got.append((chi, JTJ.copy(), JTr.copy(), jto.copy(), c.counters()['mesh_replays']))
        finally:
            c.close(); os.environ.pop('GADFIT_HIP_MESH', None)
    assert got[0][4] >= 1 and got[1][4] == 0
    for a, b in zip(got[0][:4], got[1][:4]):
        assert np.array_equal(a, b)


def test_fit_of_a_branching_model_with_quadrature(ctx):
    x, y, s = _integral_then_line(1200)
    start = B.INTEGRAL_THEN_LINE_TRUTH * np.array([1.1, 0.9, 1.12, 0.8])
    V = T.Variants(B.model_integral_then_line, 4, configure=lambda t: t.set_integration(rel_error=1e-10))
    V.explore([x[0], x[-1]], start)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r0 = p.fit(lambda_=1.0, max_iter=5, accth=0.9)
    ctx.set_model(V)
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], [0, 1, 2, 3], [0] * 4, lambda_=1.0, max_iter=5, accth=0.9)
    assert (r.iterations, r.n_chi2, r.n_omega) == (r0.iterations, r0.n_chi2, r0.n_omega)
    assert rel(out, p.pars) < 1e-9
    assert abs(out[0][2] - B.INTEGRAL_THEN_LINE_TRUTH[2]) < 0.05


def test_branching_eval_through_the_procedural_api():
    """gadf_init / gadf_add_dataset / gadf_set / gadf_fit with an eval() that compares AD variables: the layer records the paths over the
    data (tape.Variants) where the reference would simply call eval() per point; two datasets, global decay time, the fit of the
    oracle with the same options"""
    from gadfit_amd import gadfit as gf

    class piecewise(gf.fitfunc):
        def init(self):
            self.allocate(4); self.set(1, 'top'); self.set(2, 'break'); self.set(3, 'slope'); self.set(4, 'tau')

        def eval(self, x):
            return B.model_piecewise2(self.pars, x)
    t1 = B.PIECEWISE2_TRUTH.copy(); t2 = B.PIECEWISE2_TRUTH * np.array([0.8, 1.6, 1.0, 1.0])
    x1, y1, s1 = B.make_data(B.piecewise2_numpy, t1, 1500)
    x2, y2, s2 = B.make_data(B.piecewise2_numpy, t2, 1111, seed=M.SEED + 5)
    start = np.array([t1 * [1.04, 0.93, 1.05, 0.95], t2 * [0.97, 1.04, 0.96, 0.95]])
    V = T.Variants(B.model_piecewise2, 4)
    V.explore([x1[0], x1[-1]], start[0])
    p = orc.OracleProblem(V, [x1, x2], [y1, y2], [1.0 / s1, 1.0 / s2], start, [0, 1, 2, 3], [0, 0, 0, 1])
    r0 = p.fit(lambda_=1.0, max_iter=6, accth=0.9)
    gf.gadf_init(piecewise(), 2)
    try:
        gf.gadf_add_dataset(x1, y1, s1); gf.gadf_add_dataset(x2, y2, s2)
        for d in (1, 2):
            for k, name in enumerate(('top', 'break', 'slope')):
                gf.gadf_set(d, name, start[d - 1][k], True)
        gf.gadf_set('tau', start[0][3], True)                # global
        gf.gadf_set_errors(gf.USER)
        gf.gadf_set_verbosity(output='/dev/null')
        gf.gadf_fit(1.0, accth=0.9, max_iter=6)
        out = np.array([[q.val for q in f.pars] for f in gf.fitfuncs])
    finally:
        gf.gadf_close()
    assert rel(out, p.pars) < TOL_FIT
    assert out[0][3] == out[1][3] and abs(out[1][1] - t2[1]) < 0.5


def _kinked(n, rel):
    truth = B.KINKED_TRUTH
    x = np.linspace(0.05, 4.0, n)
    f = B.kinked_numpy(truth, x)
    s = 0.002 * (1.0 + np.abs(f))
    V = T.Variants(B.model_kinked_integrand, 4, configure=lambda t: t.set_integration(rel_error=rel))
    return x, f + s * M.normal(n, M.SEED + 9), s, V


def test_integrand_that_compares_ad_variables(ctx):
    """a comparison INSIDE the function handed to integrate() (t > q(2): the integrand has a kink at a fitted position): the reference
    takes the branch anew at every abscissa of the quadrature, and so does the device -- the recordings of the integrand's two paths are
    pooled into one call site and every evaluation picks its own (codegen.cpp emit_family; oracle eval_integrand).  Sweep, chi2, STEP 3,
    mesh hand-over and a fit that moves the kink, against the oracle"""
    x, y, s, V = _kinked(1201, 1e-10)
    p0 = B.KINKED_TRUTH * np.array([1.05, 0.93, 1.06, 0.8])
    V.explore(x[::40], p0)
    assert len(V) == 2 and all(t.has_integrand_guards() for t in V.tapes)
    _device_vs_oracle(ctx, V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4, tol=3e-12, jtol=5e-11, otol=3e-11)
    assert ctx.n_variants() == 1                       # ONE path through eval(); the integrand's paths live in its call site
    c = ctx.counters()
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [p0], [0, 1, 2, 3], [0] * 4)
    r0 = p.fit(lambda_=1.0, max_iter=5, accth=0.9)
    ctx.set_model(V); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([p0], [0, 1, 2, 3], [0] * 4, lambda_=1.0, max_iter=5, accth=0.9)
    assert (r.iterations, r.n_chi2, r.n_omega) == (r0.iterations, r0.n_chi2, r0.n_omega)
    assert rel(out, p.pars) < 1e-9
    assert abs(out[0][1] - B.KINKED_TRUTH[1]) < 0.05 and ctx.counters()['mesh_replays'] > c['mesh_replays']


def test_integrand_path_first_met_on_the_device(ctx):
    """recordings at small x only see the integrand below its kink; at larger x the device meets the other side (status 2): the
    handler records eval() over its sample of the data again at the parameters of the pass, the model gains the path, the pass is
    repeated -- results as with both paths known from the start.  Without a handler: a loud error (the oracle raises the same), not
    a silently wrong integral"""
    x, y, s, V = _kinked(300, 1e-8)
    V.explore(x[:20], B.KINKED_TRUTH)                  # all x < the kink at 1.2
    assert len(V) == 1 and V.tapes[0].has_integrand_guards()
    with pytest.raises(Exception, match='none of the recordings covers'):
        orc.OracleProblem(V, [x], [y], [1.0 / s], [B.KINKED_TRUTH], [0, 1, 2, 3], [0] * 4).chi2()
    c2 = _lib.Context(0)
    try:
        n, arr = V.c_array
        c2.n_pars = 4; c2._tape = V
        c2._chk(_lib.lib().gfh_set_model_variants(c2._h, n, arr, -1))       # the C entry point alone: no handler installed
        c2.set_data(x, y, 1.0 / s, [0, x.size])
        with pytest.raises(_lib.GadfitHipError, match='an integrand took a path'):
            c2.chi2([B.KINKED_TRUTH])
        c2.set_data(x[:20], y[:20], 1.0 / s[:20], [0, 20])      # where the recording holds, it is fine
        assert np.isfinite(c2.chi2([B.KINKED_TRUTH]))
    finally:
        c2.close()
    ctx.set_model(V)                                   # installs the handler
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    chi = ctx.chi2([B.KINKED_TRUTH])
    assert len(V) == 2 and any(e[2] == 'integrand' for e in ctx.unseen_log)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [B.KINKED_TRUTH], [0, 1, 2, 3], [0] * 4)
    assert abs(chi - p.chi2()[0]) <= 1e-11 * chi
    # and inside a fit: the kink starts beyond every range of integration (one path), the fit pulls it in
    x, y, s, V = _kinked(600, 1e-9)
    start = B.KINKED_TRUTH * np.array([1.02, 3.6, 1.0, 0.9])       # kink at 4.3 > max x
    V.explore(x[::25], start)
    assert len(V) == 1
    Vfull = T.Variants(B.model_kinked_integrand, 4, configure=lambda t: t.set_integration(rel_error=1e-9))
    Vfull.explore(x[::25], start); Vfull.explore(x[::25], B.KINKED_TRUTH)
    active = [0, 1, 3]                                 # (the decay time has no influence while the kink lies outside: passive)
    p = orc.OracleProblem(Vfull, [x], [y], [1.0 / s], [start], active, [0] * 4)
    r0 = p.fit(lambda_=1.0, max_iter=6)
    ctx.set_model(V); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], active, [0] * 4, lambda_=1.0, max_iter=6)
    assert (r.iterations, r.n_chi2) == (r0.iterations, r0.n_chi2) and rel(out, p.pars) < 1e-8
    assert len(V) == 2


def test_inner_integrand_of_a_double_integral_that_compares_ad_variables(ctx):
    """the kinked function as the INNER integrand of a nested integral (the outer integrand calls integrate() itself): the recordings
    differ in the inner integrand's path only; the library compares the outer integrands node by node and pools the inner recordings
    at the inner call site; sweep, chi2, STEP 3 against the oracle"""
    truth = B.KINKED_TRUTH
    x = np.linspace(0.3, 4.0, 301)
    V = T.Variants(B.model_nested_kink, 4, configure=lambda t: t.set_integration(rel_error=1e-5, rel_error_inner=1e-8, dbl=True))
    V.explore(x[::20], truth)
    assert len(V) == 2
    p0 = truth * np.array([1.04, 0.95, 1.05, 0.9])
    V.explore(x[::20], p0)
    y = np.zeros_like(x); s = np.ones_like(x)
    p = orc.OracleProblem(V, [x], [y], [s], [truth], [0, 1, 2, 3], [0] * 4)
    y = -p.chi2()[1] + 0.003 * M.normal(x.size, M.SEED + 4)              # the oracle's own values at the truth, plus noise
    _device_vs_oracle(ctx, V, [x], [y], [s / 0.003], [p0], [0, 1, 2, 3], [0] * 4, tol=1e-11, jtol=1e-9, otol=1e-9)
    assert ctx.n_variants() == 1 and _lib.lib().gfh_model_n_tapes(ctx._h) == len(V)


def test_outer_integrand_of_a_double_integral_that_compares_ad_variables(ctx):
    """an integrand that compares AD variables AND calls integrate() itself: its recordings share the inner call site; the call sites
    are generated in the order of their dependencies"""
    truth = B.KINKED_TRUTH
    x = np.linspace(0.3, 4.0, 201)
    V = T.Variants(B.model_outer_kink, 4, configure=lambda t: t.set_integration(rel_error=1e-5, rel_error_inner=1e-8, dbl=True))
    p0 = truth * np.array([1.04, 0.95, 1.05, 0.9])
    V.explore(x[::20], truth); V.explore(x[::20], p0)
    assert len(V) == 2
    s = np.ones_like(x)
    p = orc.OracleProblem(V, [x], [np.zeros_like(x)], [s], [truth], [0, 1, 2, 3], [0] * 4)
    y = -p.chi2()[1] + 0.003 * M.normal(x.size, M.SEED + 6)
    _device_vs_oracle(ctx, V, [x], [y], [s / 0.003], [p0], [0, 1, 2, 3], [0] * 4, tol=1e-11, jtol=1e-9, otol=1e-9)
    assert ctx.n_variants() == 1


def test_integrand_that_compares_through_the_procedural_api():
    """gadf_init / gadf_fit with an eval() whose integrand compares AD variables: the Python layer records it over the data like the
    Fortran one"""
    from gadfit_amd import gadfit as gf

    class kinked(gf.fitfunc):
        def init(self):
            self.allocate(4); self.set(1, 'amp'); self.set(2, 'kink'); self.set(3, 'tau'); self.set(4, 'bgr')

        def eval(self, x):
            return B.model_kinked_integrand(self.pars, x)
    x, y, s, V = _kinked(400, 1e-10)
    start = B.KINKED_TRUTH * np.array([1.05, 0.93, 1.06, 0.8])
    V.explore(x[::10], start); V.explore(x[::10], B.KINKED_TRUTH)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r0 = p.fit(lambda_=1.0, max_iter=5, accth=0.9)
    gf.gadf_init(kinked(), rel_error=1e-10)
    try:
        gf.gadf_add_dataset(x, y, s)
        for k, name in enumerate(('amp', 'kink', 'tau', 'bgr')):
            gf.gadf_set(name, start[k], True)
        gf.gadf_set_errors(gf.USER)
        gf.gadf_set_verbosity(output='/dev/null')
        gf.gadf_fit(1.0, accth=0.9, max_iter=5)
        out = np.array([q.val for q in gf.fitfuncs[0].pars])
    finally:
        gf.gadf_close()
    assert rel(out, p.pars[0]) < 1e-9


@pytest.mark.parametrize('branching', [False, True])
def test_global_changed_between_two_fits_takes_effect_in_the_procedural_api(branching):
    """The reference calls eval() afresh at every point of every fit (gadfit.F90:679-690), so a global the model reads and the
    program changes between two gadf_fit calls takes effect in the second; here the recorded model must notice and be recorded
    again (straight-line model: traced again; branching model: every variant recorded again where it was first met)."""
    from gadfit_amd import gadfit as gf
    from gadfit_amd.ad import exp
    knob = {'stretch': 1.0}

    class stretched(gf.fitfunc):
        def init(self):
            self.allocate(2); self.set(1, 'amp'); self.set(2, 'rate')

        def eval(self, x):
            y = self.pars[0] * exp(-self.pars[1] * (knob['stretch'] * x))
            if branching:
                return y if x > 1.0 else y * 1.0
            return y
    x = 0.01 * np.arange(1, 501); y = 3.0 * np.exp(-0.5 * x)
    gf.gadf_init(stretched())
    try:
        gf.gadf_add_dataset(x, y)
        gf.gadf_set('amp', 2.0, True); gf.gadf_set('rate', 0.3, True)
        gf.gadf_set_errors(gf.NONE)
        gf.gadf_set_verbosity(output='/dev/null')
        gf.gadf_fit(1.0, max_iter=50)
        first = [q.val for q in gf.fitfuncs[0].pars]
        knob['stretch'] = 2.0
        gf.gadf_fit(1.0, max_iter=50)
        second = [q.val for q in gf.fitfuncs[0].pars]
    finally:
        gf.gadf_close()
    assert abs(first[0] - 3.0) < 1e-8 and abs(first[1] - 0.5) < 1e-8
    assert abs(second[0] - 3.0) < 1e-8 and abs(second[1] - 0.25) < 1e-8
