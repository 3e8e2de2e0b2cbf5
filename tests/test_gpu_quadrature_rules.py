"""AD through adaptive Gauss-Kronrod quadrature on the device: every rule 15 ... 61
(gauss_kronrod_parameters.F90:74-617) and every kind of bound -- finite, (a, inf), (-inf, b), (-inf, inf), passive and
active (numerical_integration.F90:291-369, 377-630) -- value, reverse-mode gradient (AD:1637-1654) and forward-mode second
directional derivative (NI:425-437) against the oracle (same mesh decisions) and against the closed forms; plus the
reference's nested model (3_integral_double.F90) with the 31-point rule on both levels."""
import json
import os

import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import quadrature_cases as Q
from tests.golden import goldens as G

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'quadrature_closed_forms.json')))
# about 10 x the maxima observed on MI355X over the 6 rules x 8 cases (GADFIT_PARITY_DUMP=<file> re-records them):
TOL_ORACLE = 1e-14         # device against the oracle: same intervals, libm / FMA differences only [value 3.4e-16, gradient 6.1e-16, dd 7.4e-16]
TOL_CLOSED = 1e-14         # device against the closed forms [4.3e-16, 4.9e-16, 4.6e-16]
TOL_NESTED = dict(res=5e-15, J=2e-14, chi2=5e-15, omega=1e-14, JTomega=2e-14)      # [2.8e-16, 1.6e-15, 0, 7.9e-16, 1.2e-15]
_SEEN = {}


def _see(key, err, tol):
    _SEEN[key] = max(_SEEN.get(key, 0.0), float(err))
    if os.environ.get('GADFIT_PARITY_DUMP'):
        json.dump(_SEEN, open(os.environ['GADFIT_PARITY_DUMP'] + '.quadrature', 'w'), indent=1)
    assert err <= tol, (key, err, tol)


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _device_point(ctx, t, values, direction, active_mask):
    """value, Jacobian row (reverse mode) and second directional derivative (forward mode) of one data point"""
    n = len(values)
    ctx.set_model(t)
    ctx.set_data([0.0], [0.0], [1.0], [0, 1])
    active = [i for i, a in enumerate(active_mask) if a]
    jac, dim = ctx.jacobian_indices(active, [0] * n)
    ctx.sweep([values], active, jac, dim)
    f = -ctx.residuals()[0]
    J = ctx.jacobian(len(active))[0]
    ctx.omega([values], np.asarray(direction)[active])
    dd = -ctx.omega_vector()[0]
    return f, J, dd, ctx.chi2([values])


@pytest.mark.parametrize('rule', Q.RULES)
@pytest.mark.parametrize('name', sorted(Q.CASES))
def test_device_quadrature_rules_and_bounds(ctx, name, rule):
    g = GOLD[name]
    n = len(g['values'])
    t = trace_model(Q.CASES[name][0], n)
    t.set_integration(rel_error=1e-12, rule=rule)
    masks = [[1] * n] if n == 1 else [[1] * n, [1] + [0] * (n - 1), [0] + [1] * (n - 1)]     # everything / integrand parameter only / bounds only
    for mask in masks:
        f, J, dd, chi2 = _device_point(ctx, t, g['values'], g['direction'], mask)
        f0, grad0 = orc.eval_reverse(t, 0.0, g['values'], mask)
        dseed = np.where(mask, g['direction'], 0.0)
        _, d0, dd0 = orc.eval_forward(t, 0.0, g['values'], mask, dseed, np.zeros(n))
        na = sum(mask)
        scale = max(abs(v) for v in g['grad'])
        _see('value vs oracle', abs(f - f0) / abs(f0), TOL_ORACLE)
        _see('chi2 vs oracle', abs(chi2 - f0 * f0) / (f0 * f0), 4 * TOL_ORACLE)
        _see('gradient vs oracle', np.max(np.abs(J - grad0[:na])) / scale, TOL_ORACLE)
        _see('dd vs oracle', abs(dd - dd0) / max(abs(dd0), scale), TOL_ORACLE)
        if all(mask):
            _see('value vs closed form', abs(f - g['F']) / abs(g['F']), TOL_CLOSED)
            _see('gradient vs closed form', np.max(np.abs(J - g['grad'])) / scale, TOL_CLOSED)
            _see('dd vs closed form', abs(dd - g['dd']) / max(abs(g['dd']), scale), TOL_CLOSED)


@pytest.mark.parametrize('rule', [31, 61])
def test_device_nested_integral_with_higher_rule(ctx, rule):
    """3_integral_double.F90's model (outer (0, inf), inner finite with an ACTIVE upper bound, erf in the inner binding)
    with the 31- and 61-point rules on both levels: sweep, chi2 and STEP 3 against the oracle."""
    t = trace_model(G.model_integral_double, 2)
    t.set_integration(rel_error=1e-9, rel_error_inner=1e-10, rule=rule, dbl=True)
    x = np.array([0.4, 1.1, 2.5, 4.0])
    pars = np.array([[8.5, 1.2]])
    p = orc.OracleProblem(t, [x], [np.ones_like(x)], [np.ones_like(x)], pars, [0, 1], [0, 0])
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    ctx.set_model(t)
    ctx.set_data(x, np.ones_like(x), np.ones_like(x), [0, x.size])
    jac, dim = ctx.jacobian_indices([0, 1], [0, 0])
    JTJ, JTr, chi2 = ctx.sweep(pars, [0, 1], jac, dim)
    J = ctx.jacobian(2)
    _see('nested res', np.max(np.abs(ctx.residuals() - res0)) / np.max(np.abs(res0)), TOL_NESTED['res'])
    _see('nested J', np.max(np.abs(J - JT0) / np.max(np.abs(JT0), axis=0)), TOL_NESTED['J'])
    _see('nested chi2', max(abs(chi2 - chi0), abs(ctx.chi2(pars) - chi0)) / chi0, TOL_NESTED['chi2'])
    delta1 = np.array([0.3, -0.05])
    om0, jto0 = p.omega(delta1, JT0)
    jto = ctx.omega(pars, delta1)
    _see('nested omega', np.max(np.abs(ctx.omega_vector() - om0)) / np.max(np.abs(om0)), TOL_NESTED['omega'])
    _see('nested JTomega', np.max(np.abs(jto - jto0)) / np.max(np.abs(jto0)), TOL_NESTED['JTomega'])


# ---- the quadrature workspace is the user's (numerical_integration.F90:40, 84-98, 114-135, 251, 282-283) --------------------------
TOL_PEAKS_DD = 3e-8       # second directional derivative of the narrow-peaks model against the oracle (cancellation-bound, see below)


def _peaks_integrand(t, q):
    """six narrow Lorentzians: the mesh refines around each, 300-500 intervals at rel 1e-13"""
    y = q[0] / ((t - q[1]) ** 2 + 1.0e-10)
    for k in range(1, 6):
        y = y + q[0] / ((t - (q[1] + 0.13 * k)) ** 2 + 1.0e-10)
    return y


def _peaks_model(p, x):
    from gadfit_amd.ad import integrate
    return integrate(_peaks_integrand, [p[0], p[1]], 0.0, x) * 1.0e-5


@pytest.mark.parametrize('ws', [None, 500, 300, 50])
def test_workspace_size_is_the_users(ws):
    """an integrand that needs between 300 and 500 intervals: with the default workspace (1000, NI:40) and with ws_size = 500 the
    device meets the oracle (the kernels first carry 100 intervals, exhaust them, and the pass is repeated with the user's size);
    with ws_size = 300 or 50 both raise the reference's error (NI:282-283)"""
    from gadfit_amd import _lib
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    x = np.array([0.5, 0.8, 1.0] * 40) + 1e-3 * np.arange(120); y = np.ones(120); w = np.ones(120)
    t = trace_model(_peaks_model, 2)
    t.set_integration(rel_error=1e-13, ws_size=ws)
    pars = [[1.0, 0.111]]
    p = orc.OracleProblem(t, [x], [y], [w], pars, [0, 1], [0, 0])
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(x, y, w, [0, x.size])
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        if ws in (None, 500):
            JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
            chi0, _ = p.chi2()
            assert abs(c.chi2(pars) - chi0) <= 1e-12 * chi0
            JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
            sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
            # (Lorentzians of width 1e-5 and height 1e10 next to a target of 1e-13: the gradient's sum carries the cancellation; observed 2.6e-11)
            assert np.max(np.abs(JTJ - JTJ0) / sc) < 3e-10 and abs(chi2 - chi0) <= 1e-12 * chi0
            assert np.max(np.abs(c.residuals() - res0)) <= 1e-11 * np.max(np.abs(res0))
            delta1 = np.array([0.3, -0.05])
            om0, jto0 = p.omega(delta1, JT0)
            jto = c.omega(pars, delta1)
            # (the second derivative of a Lorentzian of width 1e-5 peaks at 2e20 and integrates to O(1): ten more digits of cancellation
            # than the gradient; observed 2.4e-9)
            assert np.max(np.abs(c.omega_vector() - om0)) <= TOL_PEAKS_DD * np.max(np.abs(om0))
            assert np.all(np.abs(jto - jto0) <= TOL_PEAKS_DD * np.max(np.abs(om0)) * np.sum(np.abs(JT0), axis=0))      # (J^T omega: omega's bound through the sum)
        else:
            with pytest.raises(RuntimeError):
                p.chi2()
            with pytest.raises(_lib.GadfitHipError, match='Number of iterations was insufficient'):
                c.chi2(pars)
            with pytest.raises(_lib.GadfitHipError, match='Number of iterations was insufficient'):
                c.sweep(pars, [0, 1], jac, dim)
    finally:
        c.close()


def test_workspace_beyond_any_scratch_lives_in_the_pool_and_goes_with_the_context():
    """ws_size = 5000 (160 KB per lane: beyond any private scratch) is the user's to ask for (NI:128-134: a heap array there).  The
    fast form carries 100 intervals in scratch; the pass that exhausts them is repeated with kernels whose workspaces are the
    context's pool in global memory (codegen.cpp GFH_WSG; context.cpp wsg_grid): every pass meets the oracle, the pool is an
    allocation the library owns -- reported by gfh_device_memory, gone after gfh_destroy -- and an absurd size is an error."""
    from gadfit_amd import _lib
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    x = np.array([0.5, 0.8, 1.0] * 40) + 1e-3 * np.arange(120); y = np.ones(120); w = np.ones(120)
    t = trace_model(_peaks_model, 2)
    t.set_integration(rel_error=1e-13, ws_size=5000)
    pars = [[1.0, 0.111]]
    p = orc.OracleProblem(t, [x], [y], [w], pars, [0, 1], [0, 0])
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    c = _lib.Context(0)
    try:
        free0 = c.device_memory()['free']
        c.set_model(t)
        c.set_data(x, y, w, [0, x.size])
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        assert c.counters()['ws_size'] == 100 and c.device_memory()['workspace_pool'] == 0
        assert abs(c.chi2(pars) - chi0) <= 1e-12 * chi0
        assert c.counters()['ws_size'] == 5000
        pool = c.device_memory()['workspace_pool']
        # one slot of 5000 intervals x 4 fields x 64 lanes per wave of the launch: 120 points = one gram block of 8 waves
        assert pool >= 8 * 5000 * 4 * 64 * 8 and pool % (5000 * 4 * 64 * 8) == 0
        JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
        sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
        assert np.max(np.abs(JTJ - JTJ0) / sc) < 3e-10 and abs(chi2 - chi0) <= 1e-12 * chi0
        assert np.max(np.abs(c.residuals() - res0)) <= 1e-11 * np.max(np.abs(res0))
        delta1 = np.array([0.3, -0.05])
        om0, jto0 = p.omega(delta1, JT0)
        jto = c.omega(pars, delta1)
        assert np.max(np.abs(c.omega_vector() - om0)) <= TOL_PEAKS_DD * np.max(np.abs(om0))
        assert np.all(np.abs(jto - jto0) <= TOL_PEAKS_DD * np.max(np.abs(om0)) * np.sum(np.abs(JT0), axis=0))      # (J^T omega: omega's bound through the sum)
        assert c.device_memory()['workspace_pool'] == pool          # one pool serves the three kernels
    finally:
        c.close()
    c = _lib.Context(0)
    try:
        assert c.device_memory()['free'] >= free0 - (8 << 20)        # (small blocks parked for the next context: GADFIT_HIP_POOL)
        t.set_integration(ws_size=1 << 23)
        with pytest.raises(_lib.GadfitHipError, match='beyond'):
            c.set_model(t)
    finally:
        c.close()


def _peaks_inside(p, x):
    """the narrow peaks as the INNER integrand of a double integral: the inner workspace is the one that must grow"""
    from gadfit_amd.ad import integrate

    def outer(t, q):
        return integrate(_peaks_integrand, [q[0], q[1]], 0.0, t) * (1.0 + 0.1 * t)
    return integrate(outer, [p[0], p[1]], 0.5, x) * 1.0e-5


def test_inner_workspace_of_a_double_integral_grows_into_the_pool():
    """nested integrals: the outer level keeps its few intervals, the inner one needs 300-500 (ws_size_inner, NI:70, 84-98): both levels
    move to the pool (level 2 behind level 1 in the wave's slot) and sweep, chi2() and STEP 3 meet the oracle"""
    from gadfit_amd import _lib
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    x = np.array([0.8, 0.62]); y = np.ones(2); w = np.ones(2)             # (the oracle takes 12 s over these two points)
    t = trace_model(_peaks_inside, 2)
    t.set_integration(rel_error=1e-4, rel_error_inner=1e-12, dbl=True)
    pars = [[1.0, 0.111]]
    p = orc.OracleProblem(t, [x], [y], [w], pars, [0, 1], [0, 0])
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(x, y, w, [0, x.size])
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        fast = c.counters()
        assert fast['ws_size'] <= 100 and fast['ws_size_inner'] <= 100
        JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
        assert c.counters()['ws_size_inner'] == 1000 and c.device_memory()['workspace_pool'] > 0
        sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
        assert np.max(np.abs(JTJ - JTJ0) / sc) < 3e-10 and abs(chi2 - chi0) <= 1e-11 * chi0
        assert abs(c.chi2(pars) - chi0) <= 1e-11 * chi0
        delta1 = np.array([0.3, -0.05])
        om0, jto0 = p.omega(delta1, JT0)
        jto = c.omega(pars, delta1)
        assert np.max(np.abs(c.omega_vector() - om0)) <= TOL_PEAKS_DD * np.max(np.abs(om0))
        assert np.all(np.abs(jto - jto0) <= TOL_PEAKS_DD * np.max(np.abs(om0)) * np.sum(np.abs(JT0), axis=0))      # (J^T omega: omega's bound through the sum)
    finally:
        c.close()


@pytest.mark.parametrize('n', [3000, 70000])
def test_pool_kernels_stride_over_their_work_and_change_no_bit(n):
    """with the workspaces in the pool the grid is capped at the pool's slots and a workgroup takes several tiles / gram blocks:
    the sums are defined on the partition, so a context that carries the user's size in the pool from the start (GADFIT_HIP_WS_FAST=0)
    returns bitwise what the scratch form returns where that suffices, and the oracle's numbers where it is compared"""
    from gadfit_amd import _lib
    t, x, y, w, pars = _single_integral_problem(n)
    got = {}
    for fast in ('100', '0'):
        old = os.environ.get('GADFIT_HIP_WS_FAST')
        os.environ['GADFIT_HIP_WS_FAST'] = fast
        try:
            c = _lib.Context(0)
        finally:
            if old is None:
                del os.environ['GADFIT_HIP_WS_FAST']
            else:
                os.environ['GADFIT_HIP_WS_FAST'] = old
        try:
            c.set_model(t)
            c.set_data(x, y, w, [0, x.size])
            jac, dim = c.jacobian_indices([0, 1], [0, 0])
            chi = c.chi2(pars)
            JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
            res = c.residuals(); J = c.jacobian(2)
            jto = c.omega(pars, np.array([0.01, -0.02])); om = c.omega_vector()
            got[fast] = (chi, JTJ, JTr, chi2, res, J, jto, om)
            assert (c.device_memory()['workspace_pool'] > 0) == (fast == '0') and c.counters()['ws_size'] == (100 if fast == '100' else 1000)
        finally:
            c.close()
    for k in range(8):
        assert np.array_equal(np.asarray(got['100'][k]), np.asarray(got['0'][k])), k


def test_integrand_reads_the_abscissa_and_a_column_of_the_enclosing_eval():
    """an integrand that takes x (and an auxiliary per-point column) from the enclosing eval() without passing them through pars(:) --
    valid under the reference, whose integrand runs in eval()'s scope at every point (numerical_integration.F90:195-201): GFH_X / GFH_AUX
    leaves inside the integrand's sub-tape, read on the device from the lane's stash; sweep, chi2() and STEP 3 against the oracle"""
    from gadfit_amd import ad

    def model(p, x):
        def f(t, q):
            return q[0] * ad.exp(-q[1] * t * t) * (1.0 + 0.1 * x) + ad.aux(0) * t
        return ad.integrate(f, [p[0], p[1]], 0.0, x)
    t = trace_model(model, 2)
    t.set_integration(rel_error=1e-10)
    x = np.linspace(0.1, 3.0, 700); y = np.ones(700); w = np.ones(700); cols = np.sin(x)[None, :]
    pars = [[1.3, 0.7]]
    p = orc.OracleProblem(t, [x], [y], [w], pars, [0, 1], [0, 0], aux=cols)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    c = _lib.Context(0)
    try:
        c.set_model(t); c.set_data(x, y, w, [0, 700]); c.set_aux(cols)
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
        _see('integrand x/aux JTJ', np.max(np.abs(JTJ - JTJ0) / np.abs(JTJ0)), TOL_NESTED['J'])
        _see('integrand x/aux chi2', max(abs(chi2 - chi0), abs(c.chi2(pars) - chi0)) / chi0, TOL_NESTED['chi2'])
        d1 = np.array([0.3, -0.05])
        om0, jto0 = p.omega(d1, JT0)
        jto = c.omega(pars, d1)
        _see('integrand x/aux omega', np.max(np.abs(c.omega_vector() - om0)) / np.max(np.abs(om0)), TOL_NESTED['omega'])
        _see('integrand x/aux JTomega', np.max(np.abs(jto - jto0) / np.abs(jto0)), TOL_NESTED['JTomega'])
    finally:
        c.close()


def test_value_of_an_advar_inside_an_integrand():
    """value() = GFH_VAL inside the function handed to integrate(): the VALUE of a bound parameter in plain real arithmetic (a weight
    that follows the parameter, no derivative through it), on the device against the oracle"""
    from gadfit_amd import ad

    def model(p, x):
        def f(t, q):
            s = ad.cos(ad.value(q[1]))
            return q[0] * ad.exp(-q[1] * t) * (1.0 + 0.2 * s * s)
        return ad.integrate(f, [p[0], p[1]], 0.0, x)
    t = trace_model(model, 2)
    t.set_integration(rel_error=1e-10)
    x = np.linspace(0.2, 4.0, 300); y = np.ones(300); w = np.ones(300)
    pars = [[1.3, 0.7]]
    p = orc.OracleProblem(t, [x], [y], [w], pars, [0, 1], [0, 0])
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    c = _lib.Context(0)
    try:
        c.set_model(t); c.set_data(x, y, w, [0, 300])
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
        _see('integrand value() JTJ', np.max(np.abs(JTJ - JTJ0) / np.abs(JTJ0)), TOL_NESTED['J'])
        _see('integrand value() chi2', max(abs(chi2 - chi0), abs(c.chi2(pars) - chi0)) / chi0, TOL_NESTED['chi2'])
        d1 = np.array([0.3, -0.05])
        om0, jto0 = p.omega(d1, JT0)
        jto = c.omega(pars, d1)
        _see('integrand value() omega', np.max(np.abs(c.omega_vector() - om0)) / np.max(np.abs(om0)), TOL_NESTED['omega'])
    finally:
        c.close()


# ---- mesh hand-over between passes at the same parameters (codegen.cpp mesh_build; context.cpp mesh_mode_for) ------------------------
def _fresh_context(mesh):
    old = os.environ.get('GADFIT_HIP_MESH')
    os.environ['GADFIT_HIP_MESH'] = '1' if mesh else '0'
    try:
        return _lib.Context(0)
    finally:
        if old is None:
            del os.environ['GADFIT_HIP_MESH']
        else:
            os.environ['GADFIT_HIP_MESH'] = old


def _single_integral_problem(n=6000):
    a, b = 7.5, 0.8
    x = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
    from scipy.special import gammainc, gamma
    f = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * x * x)
    s = 0.01 * (1 + np.abs(f))
    y = f + s * np.sin(37.0 * np.arange(n))
    t = trace_model(G.model_integral_single, 2)
    t.set_integration(rel_error=1e-10)
    return t, x, y, 1.0 / s, np.array([[a * 1.05, b * 0.95]])


@pytest.mark.parametrize('which', ['single', 'nested'])
def test_replayed_mesh_gives_bitwise_the_fresh_bisection(which):
    """chi2() at p records every point's bisections; the sweep at the same p replays them (no integrand evaluation until the final
    pass), and STEP 3 after it does too: residuals, Jacobian, J^T J, omega and J^T omega are BITWISE those of a context that bisects
    in every pass"""
    if which == 'single':
        t, x, y, w, pars = _single_integral_problem()
    else:
        d = G.data()['3_integral_double']
        x = np.array(d['x_data']); y = np.array(d['y_data']); w = 1.0 / np.array(d['weights'])
        t = trace_model(G.model_integral_double, 2)
        t.set_integration(rel_error=1e-5, rel_error_inner=1e-6, dbl=True)
        pars = np.array([[1.0, 1.0]])
    got = {}
    for mesh in (True, False):
        c = _fresh_context(mesh)
        try:
            c.set_model(t)
            c.set_data(x, y, w, [0, x.size])
            jac, dim = c.jacobian_indices([0, 1], [0, 0])
            chi = c.chi2(pars)
            JTJ, JTr, chi2 = c.sweep(pars, [0, 1], jac, dim)
            res = c.residuals(); J = c.jacobian(2)
            jto = c.omega(pars, np.array([0.01, -0.02])); om = c.omega_vector()
            got[mesh] = (chi, JTJ, JTr, chi2, res, J, jto, om, c.counters()['mesh_replays'])
        finally:
            c.close()
    assert got[True][8] == 2 and got[False][8] == 0          # the sweep and STEP 3 replayed
    for k in range(8):
        assert np.array_equal(np.asarray(got[True][k]), np.asarray(got[False][k])), k


@pytest.mark.parametrize('lookahead', [1, 0])
def test_fit_of_a_quadrature_model_with_and_without_mesh_hand_over(lookahead):
    """whole accelerated fits: same bits with and without the hand-over.  Reference schedule (trial chi2 -> accepted -> sweep at the
    same parameters -> STEP 3): the sweep and STEP 3 of every iteration after the first replay; look-ahead schedule (the trial chi2
    IS a sweep at the trial point): STEP 3 replays that sweep's meshes"""
    t, x, y, w, pars = _single_integral_problem(3000)
    out = {}
    for mesh in (True, False):
        c = _fresh_context(mesh)
        try:
            c.set_model(t)
            c.set_data(x, y, w, [0, x.size])
            c.set_lookahead(lookahead)
            p, r = c.fit(pars, [0, 1], [0, 0], lambda_=1.0, accth=0.9, max_iter=5)
            out[mesh] = (p.copy(), r.chi2, r.iterations, c.counters()['mesh_replays'])
        finally:
            c.close()
    assert np.array_equal(out[True][0], out[False][0]) and out[True][1] == out[False][1] and out[True][2] == out[False][2] == 5
    assert out[True][3] >= (5 if lookahead else 8) and out[False][3] == 0


@pytest.mark.parametrize('which', ['single', 'nested'])
def test_chi2_is_bitwise_the_sweeps_sum_of_squares_for_quadrature_models(which):
    """what the look-ahead schedule builds on, on the two-kernel path these models take: the residuals of the value-only pass and of
    the sweep agree bit for bit and k_gram_small / gfh_k_chi2 add their squares in the same order"""
    if which == 'single':
        t, x, y, w, pars = _single_integral_problem(5000)
    else:
        d = G.data()['3_integral_double']
        x = np.array(d['x_data']); y = np.array(d['y_data']); w = 1.0 / np.array(d['weights'])
        t = trace_model(G.model_integral_double, 2)
        t.set_integration(rel_error=1e-5, rel_error_inner=1e-6, dbl=True)
        pars = np.array([[1.0, 1.0]])
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(x, y, w, [0, x.size])
        jac, dim = c.jacobian_indices([0, 1], [0, 0])
        for p in (pars, pars * [1.02, 0.97]):
            JTJ, JTr, chi2 = c.sweep(p, [0, 1], jac, dim)
            res_s = c.residuals()
            assert c.chi2(p) == chi2
            assert np.array_equal(c.residuals(), res_s)
    finally:
        c.close()


def test_lookahead_schedule_equals_reference_schedule_for_a_quadrature_model():
    """the first trial chi2 of an iteration taken from a sweep at the trial point (handed to the next iteration when the step is
    accepted) against the reference's schedule of passes: same bits, one N-sized pass per accepted iteration instead of two"""
    t, x, y, w, pars = _single_integral_problem(4000)
    out = {}
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(x, y, w, [0, x.size])
        for la in (1, 0):
            c.set_lookahead(la)
            for opts in (dict(lambda_=1.0, max_iter=5), dict(lambda_=1.0, max_iter=4, accth=0.9)):
                c.reset_timers()
                p, r = c.fit(pars, [0, 1], [0, 0], **opts)
                out[(la, 'accth' in opts)] = (p.copy(), r.chi2, r.iterations, r.n_lookahead, int(c.timers()[7]))
    finally:
        c.close()
    for acc in (False, True):
        a, b = out[(1, acc)], out[(0, acc)]
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]
        assert a[3] >= a[2] and b[3] == 0 and a[4] < b[4]          # look-ahead sweeps replaced chi2() launches


def test_order_of_dispatch_changes_no_bit():
    """quadrature models: after the first sweep the workgroups take their tiles / gram blocks in the order of measured cost,
    expensive first (context.cpp build_orders; GADFIT_HIP_ORDER=0 keeps the index order).  Every sum is defined on the fixed
    partition, so sweep, chi2, STEP 3 and a whole fit return the same bits either way -- here on 60 000 x-sorted points in two
    datasets (235 tiles), with the second and later passes running ordered"""
    t, x, y, w, pars = _single_integral_problem(60000)
    pos = [0, 25000, 60000]
    P = np.array([pars[0], pars[0] * [0.98, 1.03]])
    got = {}
    for order in ('1', '0'):
        old = os.environ.get('GADFIT_HIP_ORDER'); os.environ['GADFIT_HIP_ORDER'] = order
        try:
            c = _lib.Context(0)
        finally:
            if old is None:
                del os.environ['GADFIT_HIP_ORDER']
            else:
                os.environ['GADFIT_HIP_ORDER'] = old
        try:
            c.set_model(t)
            c.set_data(x, y, w, pos)
            jac, dim = c.jacobian_indices([0, 1], [0, 1])
            first = c.sweep(P, [0, 1], jac, dim)                       # measures
            P2 = P * [1.01, 0.99]
            chi = c.chi2(P2)                                           # ordered (gram blocks)
            JTJ, JTr, chi2 = c.sweep(P2, [0, 1], jac, dim)             # ordered (tiles), replaying chi2's meshes
            res = c.residuals(); J = c.jacobian(2)
            jto = c.omega(P2, np.array([0.01, -0.02, 0.005])); om = c.omega_vector()
            JTJ3, JTr3, chi3 = c.sweep(P * [0.99, 1.02], [0, 1], jac, dim)   # ordered, bisecting
            out, r = c.fit(P, [0, 1], [0, 1], lambda_=1.0, max_iter=4, accth=0.9)
            got[order] = (first[0], first[1], first[2], chi, JTJ, JTr, chi2, res, J, jto, om, JTJ3, JTr3, chi3, out, r.chi2, r.iterations)
        finally:
            c.close()
    for k, (a, b) in enumerate(zip(got['1'], got['0'])):
        assert np.array_equal(np.asarray(a), np.asarray(b)), k


def test_iterated_integral_from_zero(tmp_path):
    """int_0^x w(t) int_0^t f(u) du dt: the inner upper bound IS the outer integration variable and both integrals start at the same
    literal 0.  In the gradient pass f is evaluated at a bound only where the bound carries an adjoint (NI:413-417) -- evaluated
    unconditionally, the outer integrand at t = 0 asks for the inner integral over [0, 0], whose error test 0/0 never passes."""
    from gadfit_amd.ad import integrate, exp
    from tests.test_gpu_parity import _device_vs_oracle

    def model(p, x):
        def inner(u, q):
            return q[0] * (1.0 + 0.5 * (u - q[1])) * exp(-(q[2] * u))

        def outer(t, q):
            return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, t)
        return integrate(outer, [p[0], p[1], p[2]], 0.0, x) + p[3]
    t = trace_model(model, 4)
    t.set_integration(rel_error=1e-6, rel_error_inner=1e-9, dbl=True)
    x = np.linspace(0.2, 4.0, 77)
    pars = np.array([1.3, 1.2, 0.8, 0.1])
    y = 0.5 + 0.1 * np.sin(3.0 * x)
    c = _lib.Context(0)
    try:
        _device_vs_oracle(c, t, [x], [y], [np.ones_like(x)], [pars], [0, 1, 2, 3], [0] * 4, tol=1e-12, jtol=1e-11, otol=1e-11)
    finally:
        c.close()
