"""CPU tests of branching eval() support (guards / variants): the Python recorder, the oracle's choice of the recorded path per
point (against closed forms), the library's decision tree (no GPU: compile-only contexts)."""
import numpy as np
import pytest

from gadfit_amd import _lib, ad
from gadfit_amd import tape as T
from oracle import binding as orc
from tests import branching as B


def test_recorder_records_comparisons_as_guards():
    V = T.Variants(B.model_piecewise2, 4)
    idx = V.explore([1.0, 36.0, 38.0, 99.0], B.PIECEWISE2_TRUTH)
    assert idx == [0, 0, 1, 1] and len(V) == 2
    g0 = [n for n in V.tapes[0].subtapes[0][0] if n[0] in (T.GUARD_GT, T.GUARD_LT)]
    g1 = [n for n in V.tapes[1].subtapes[0][0] if n[0] in (T.GUARD_GT, T.GUARD_LT)]
    assert len(g0) == len(g1) == 1 and g0[0][0] == T.GUARD_LT
    assert g0[0][:3] == g1[0][:3] and g0[0][3] == T.F_TAKEN and g1[0][3] == 0      # same comparison, the two outcomes
    # a forced outcome sends the recording down the other path at the same point
    assert V.add_point(1.0, B.PIECEWISE2_TRUTH, script=[False]) == 1 and len(V) == 2
    # symbolic recordings cannot decide a comparison
    with pytest.raises(TypeError):
        ad.trace_model(B.model_piecewise2, 4)


def reference_comparisons(a, b):
    """the fourteen comparisons of ad_forward_mode.F90:9-25 (a%val = fix_d(1), b%val = fix_d(2)); every one must hold"""
    from tests.golden import goldens as G
    f = lambda i: float(np.float32(G.D(i)))      # fix_f = real(fix_d, real32)
    d = q = G.D                                   # fix_q holds the same numbers to quad precision: compared as real(kp) values here
    return [a > b, b < a, a > f(2), f(2) < a, a < f(3), f(3) > a, a > d(2), d(2) < a, a < d(3), d(3) > a,
            a > q(2), q(2) < a, a < q(3), q(3) > a]


def test_comparison_goldens_of_the_reference():
    """ad_forward_mode.F90:9-25 through the recorder: advar/advar, advar against real32 / dp / qp on either side, `>` and `<`:
    values only (AD:315-395), all true for the fixture; each comparison leaves one guard node with its outcome"""
    from tests.golden import goldens as G
    outcomes = []

    def probe(p, x):
        outcomes.extend(reference_comparisons(p[0], p[1]))
        return p[0] + p[1]
    t = ad.trace_model(probe, 2, x=0.0, pars=[G.D(1), G.D(2)])
    assert outcomes == [True] * 14
    assert t.guard_outcomes() == outcomes
    # swapped fixture: every comparison with b in a's place and vice versa is false where it involves both
    outcomes.clear()
    t = ad.trace_model(probe, 2, x=0.0, pars=[G.D(2), G.D(1)])
    assert outcomes[:2] == [False, False] and t.guard_outcomes() == outcomes


@pytest.mark.parametrize('active', [[0, 1, 2, 3], [0, 2, 3]])
def test_oracle_takes_the_branch_per_point(active):
    """res and J of the oracle against the closed form of the piecewise model, breakpoint active and passive"""
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 400)
    p0 = B.PIECEWISE2_TRUTH * np.array([1.03, 0.96, 1.05, 0.97])
    V = T.Variants(B.model_piecewise2, 4)
    V.explore([x[0], x[-1]], p0)
    prob = orc.OracleProblem(V, [x], [y], [1.0 / s], [p0], active, [0] * 4)
    JTJ, JTr, res, JT = prob.sweep(want_J=True)
    want_res = (y - B.piecewise2_numpy(p0, x)) / s
    want_J = B.piecewise2_grad_numpy(p0, x)[:, active] / s[:, None]
    assert np.max(np.abs(res - want_res)) <= 1e-13 * np.max(np.abs(want_res))
    assert np.max(np.abs(JT - want_J)) <= 1e-13 * np.max(np.abs(want_J))
    chi2, _ = prob.chi2()
    assert abs(chi2 - want_res @ want_res) <= 1e-13 * chi2


def test_oracle_reports_a_point_no_variant_covers():
    x, y, s = B.make_data(B.piecewise2_numpy, B.PIECEWISE2_TRUTH, 50)
    V = T.Variants(B.model_piecewise2, 4)
    V.explore([x[0]], B.PIECEWISE2_TRUTH)          # only the first segment was recorded
    prob = orc.OracleProblem(V, [x], [y], [1.0 / s], [B.PIECEWISE2_TRUTH], [0, 1, 2, 3], [0] * 4)
    with pytest.raises(RuntimeError, match='none of the recorded variants'):
        prob.sweep()


def test_library_builds_the_decision_tree():
    V = T.Variants(B.model_piecewise3, 6)
    V.explore([1.0, 30.0, 90.0], B.PIECEWISE3_TRUTH)
    assert len(V) == 3
    c = _lib.Context(-1)
    try:
        c.set_model(V)
        assert c.n_variants() == 3 and c.model_needs_hint() == 0
        src = c.model_source([0, 1, 2, 3, 4, 5])
        assert 'gfh_select' in src and 'gfh_point_grad_v2' in src and 'gfh_report_unseen' in src
        c.model_prepare([0, 1, 2, 3, 4, 5])          # compiles for gfx950 through hiprtc
        # a straight-line model keeps its plain source: no selector, no slot argument
        c.set_model(ad.trace_model(lambda p, x: p[0] * ad.exp(-(x / p[1])), 2))
        src = c.model_source([0, 1])
        assert 'gfh_select' not in src and '#define GFH_SLOT_DECL\n' in src
    finally:
        c.close()


def test_variants_that_fork_without_a_comparison_need_the_hint_column():
    """two recordings whose operations differ with no guard in between (what a Fortran eval() branching on the plain real x
    looks like): only a per-point column can tell them apart"""
    class Two(T.Variants):
        pass
    V = Two(None, 2)
    V.tapes = [ad.trace_model(lambda p, x: p[0] * x + p[1], 2), ad.trace_model(lambda p, x: p[0] * ad.exp(-(x / p[1])), 2)]
    c = _lib.Context(-1)
    try:
        c.set_model(V)
        assert c.model_needs_hint() == 1
        with pytest.raises(_lib.GadfitHipError, match='per-point variant column'):
            c.model_prepare([0, 1])
        for t in V.tapes:
            t.n_aux = 1; t._c = None
        V._c = None
        c.set_model(V, hint_aux=0)
        c.model_prepare([0, 1])
        assert 'h0 == 0' in c.model_source([0, 1]) or 'h1 == 0' in c.model_source([0, 1]) or ' == 0)' in c.model_source([0, 1])
    finally:
        c.close()


def test_one_variant_column_per_set_of_outcomes():
    """round 5 (gfh_set_variant_hint_columns): four recordings -- behind `x < p1` a plain-real fork (two bodies), behind its other
    outcome another one -- and one column per set of outcomes: the walk reads, kid by kid, the column of the kid's own outcomes
    and, at a leaf, the column of the leaf's outcomes; a kid it cannot walk to such a leaf is left for the next (gfh_next labels)"""
    import ctypes as C
    import numpy as np

    class Four(T.Variants):
        pass

    V = Four(None, 2)
    bodies = [lambda p, x: p[0] * x, lambda p, x: p[0] * ad.exp(-x), lambda p, x: p[0] + x, lambda p, x: p[0] * ad.sqrt(x)]
    tapes = []
    for k, body in enumerate(bodies):
        tv = T.Variants(lambda p, x, body=body: body(p, x) if x < p[1] else body(p, x), 2)
        # (the comparison comes out True for the first two recordings, False for the others: x and p1 chosen accordingly)
        tv.add_point(1.0, [2.0, 5.0] if k < 2 else [2.0, 0.5])
        assert len(tv) == 1
        tapes.append(tv.tapes[0])
    V.tapes = tapes
    for t in V.tapes:
        t.n_aux = 3; t._c = None
    V._c = None
    c = _lib.Context(-1)
    try:
        cols = np.array([1, 1, 2, 2], dtype=np.int32)        # outcomes T: column 1, outcomes F: column 2 (column 0: the natural one)
        assert _lib.lib().gfh_set_variant_hint_columns(c._h, 4, cols.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        c.set_model(V, hint_aux=0)
        assert c.n_variants() == 4 and c.model_needs_hint() == 1
        c.model_prepare([0, 1])
        src = c.model_source([0, 1])
        sel = src[src.index('gfh_select'):src.index('gfh_point_grad')] if 'gfh_point_grad' in src else src
        assert 'AXP[(i64)1 * LDA]' in sel and 'AXP[(i64)2 * LDA]' in sel and 'AXP[(i64)0 * LDA]' not in sel
        assert sel.count('gfh_next') >= 8 and 'hl == ' in sel
        # without the call: the one natural column for every fork and leaf
        c.set_model(V, hint_aux=0)
        src0 = c.model_source([0, 1])
        sel0 = src0[src0.index('gfh_select'):src0.index('gfh_point_grad')] if 'gfh_point_grad' in src0 else src0
        assert 'AXP[(i64)0 * LDA]' in sel0 and 'AXP[(i64)1 * LDA]' not in sel0
    finally:
        c.close()


def test_identical_variants_are_refused():
    V = T.Variants(None, 2)
    t = ad.trace_model(lambda p, x: p[0] * x + p[1], 2)
    V.tapes = [t, ad.trace_model(lambda p, x: p[0] * x + p[1], 2)]
    c = _lib.Context(-1)
    try:
        with pytest.raises(_lib.GadfitHipError, match='repeats an earlier one'):
            c.set_model(V)
    finally:
        c.close()


def test_integrand_that_compares_ad_variables_oracle_and_pooling():
    """a comparison INSIDE the function handed to integrate(): the recorder places the integration variable at several points of its
    range, so both paths through the integrand are met; the oracle evaluates every abscissa of the quadrature through the recording
    whose comparisons hold there (closed form: tests/branching.py kinked_numpy); the library pools the recordings into ONE variant of
    eval() whose call site picks its integrand per evaluation"""
    from gadfit_amd import _lib
    truth = B.KINKED_TRUTH
    x = np.concatenate([np.linspace(0.05, 2.2, 40), np.linspace(2.6, 4.0, 20)])       # (not 2.4 = twice the kink: there the first
    V = T.Variants(B.model_kinked_integrand, 4, configure=lambda t: t.set_integration(rel_error=1e-12))   # Gauss-Kronrod estimate is blind)
    idx = V.explore(x, truth)
    assert len(V) == 2 and all(t.has_integrand_guards() for t in V.tapes)
    y = B.kinked_numpy(truth, x)
    p = orc.OracleProblem(V, [x], [y], [np.ones_like(x)], [truth], [0, 1, 2, 3], [0] * 4)
    chi, res = p.chi2()
    # (a kink costs the adaptive rule its accuracy where an interval boundary falls close to it: the reference's own limitation)
    assert np.median(np.abs(res)) < 1e-12 and np.max(np.abs(res)) < 1e-8
    _, _, _, JT = p.sweep(want_J=True)
    g = np.zeros((x.size, 4))
    for k in range(4):
        h = 1e-6 * max(1.0, abs(truth[k])); a = truth.copy(); a[k] += h; b = truth.copy(); b[k] -= h
        g[:, k] = (B.kinked_numpy(a, x) - B.kinked_numpy(b, x)) / (2 * h)
    # (the derivative with respect to the kink's position sees the quadrature's error at the kink: looser)
    assert np.max(np.abs(JT - g)[:, [0, 2, 3]]) < 1e-6 and np.max(np.abs(JT - g)[:, 1]) < 1e-3
    # one recording alone does not cover the abscissas beyond the kink: the oracle says so
    V1 = T.Variants(B.model_kinked_integrand, 4)
    V1.explore(x[:5], truth)
    assert len(V1) == 1
    with pytest.raises(Exception, match='none of the recordings covers'):
        orc.OracleProblem(V1, [x], [y], [np.ones_like(x)], [truth], [0, 1, 2, 3], [0] * 4).chi2()
    ctx = _lib.Context(-1)
    try:
        ctx.set_model(V)
        assert ctx.n_variants() == 1 and not ctx.model_needs_hint()
        src = ctx.model_source([0, 1, 2, 3])
        assert 'gfh_sf0_sel' in src and 'gfh_sf0_grad' in src and 'GFH_RAISE(STATUS, 2)' in src
        ctx.model_prepare([0, 1, 2, 3])
    finally:
        ctx.close()


def test_literal_formed_from_the_integration_variables_value_is_refused():
    """Reading .val inside an integrand forms a plain number; a recording would freeze it at the one place its integration
    variable sat while it was recorded (the reference calls the integrand afresh at every abscissa, numerical_integration.F90:
    238-275).  Recorded once more with the variable elsewhere, the recording differs: loud, never a wrong integral."""
    import math
    from gadfit_amd.ad import integrate, exp

    def bad(p, x):
        m = p[0] if x > p[1] else 2.0 * p[0]
        return m * integrate(lambda t, q: exp(-q[0] * t) * math.cos(t.val), [p[2]], 0.0, 1.0)

    def good(p, x):
        m = p[0] if x > p[1] else 2.0 * p[0]
        return m * integrate(lambda t, q: exp(-q[0] * t) * 0.5, [p[2]], 0.0, 1.0)

    v = T.Variants(bad, 3)
    with pytest.raises(TypeError, match='integration variable'):
        v.add_point(1.0, [1.0, 2.0, 0.5])
    g = T.Variants(good, 3)
    assert g.add_point(1.0, [1.0, 2.0, 0.5]) == 0 and g.add_point(3.0, [1.0, 2.0, 0.5]) == 1 and len(g) == 2
