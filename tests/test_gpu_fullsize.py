"""BASELINE.json configurations at (or near) full size on the GPU.  The oracle is too slow for
1e7 points, so parity at size is shown through size-independent properties plus oracle checks
on sub-samples of the SAME inputs:
  * chi2 from the fused sweep (matrix-core path) == chi2 from the value-only kernel,
  * J^T res recomputed by the separate J^T v kernel from the stored J == JTres of the sweep,
  * JTJ symmetric, and equal to the sum of the oracle's JTJ over disjoint sub-samples when the
    device is given exactly those sub-samples,
  * weights scaling: w -> 2w scales JTJ, JTres, chi2 by 4,
  * additivity over a partition of the points (what the multi-GPU all-reduce relies on)."""
import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import models as M
from tests.golden import goldens as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _props(ctx, tape, xs, ys, ws, pars, active, is_global, sample=1500, seed=3, chi2_bitwise=True):
    pos = np.zeros(len(xs) + 1, dtype=np.int64)
    for i, a in enumerate(xs):
        pos[i + 1] = pos[i] + len(a)
    X, Y, W = np.concatenate(xs), np.concatenate(ys), np.concatenate(ws)
    ctx.set_model(tape)
    ctx.set_data(X, Y, W, pos)
    jac, dim = ctx.jacobian_indices(active, is_global)
    JTJ, JTr, chi2 = ctx.sweep(pars, active, jac, dim)
    assert np.array_equal(JTJ, JTJ.T)
    assert np.all(np.diag(JTJ) > 0)
    # chi2: matrix-core path vs value-only kernel (different code, same points)
    c2 = ctx.chi2(pars)
    if chi2_bitwise:       # the fused kernel sums r^2 in gfh_k_chi2's partition and order: the same bits at any size
        assert c2 == chi2
    else:                  # models with integrate(): separate sweep and Gram kernels, another order of additions
        assert abs(c2 - chi2) <= 1e-12 * chi2
    # J^T res from the stored Jacobian by the VALU J^T v kernel
    g = ctx.aux(0, dim=dim)
    assert np.max(np.abs(g - JTr)) <= 1e-11 * np.max(np.abs(JTr))
    # weights x2 -> everything x4 (exact in binary floating point)
    ctx.set_data(X, Y, 2.0 * W, pos)
    JTJ4, JTr4, chi4 = ctx.sweep(pars, active, jac, dim)
    assert np.array_equal(JTJ4, 4.0 * JTJ) and np.array_equal(JTr4, 4.0 * JTr) and chi4 == 4.0 * chi2
    # additivity over a 3-way contiguous partition (gfh_partition rule)
    acc = np.zeros_like(JTJ); accr = np.zeros_like(JTr); accc = 0.0
    N = X.size
    for r in range(3):
        b, cnt = _lib.partition(N, 3, r)
        lx, ly, lw, lp = [], [], [], [0]
        for d in range(len(xs)):
            lo, hi = max(b, pos[d]), min(b + cnt, pos[d + 1])
            sl = slice(lo, max(lo, hi))
            lx.append(X[sl]); ly.append(Y[sl]); lw.append(W[sl]); lp.append(lp[-1] + max(0, hi - lo))
        ctx.set_data(np.concatenate(lx), np.concatenate(ly), np.concatenate(lw), lp)
        a, b_, c_ = ctx.sweep(pars, active, jac, dim)
        acc += a; accr += b_; accc += c_
    sc = np.sqrt(np.outer(np.diag(JTJ), np.diag(JTJ)))
    assert np.max(np.abs(acc - JTJ) / sc) < 1e-12 and abs(accc - chi2) <= 1e-12 * chi2
    assert np.max(np.abs(accr - JTr)) <= 1e-11 * np.max(np.abs(JTr))
    # oracle on a random sub-sample of the same inputs (per dataset, so the block structure holds)
    rng = np.random.default_rng(seed)
    sx, sy, sw = [], [], []
    for d in range(len(xs)):
        k = max(1, min(len(xs[d]), sample // len(xs)))
        idx = np.sort(rng.choice(len(xs[d]), size=k, replace=False))
        sx.append(xs[d][idx]); sy.append(ys[d][idx]); sw.append(ws[d][idx])
    p = orc.OracleProblem(tape, sx, sy, sw, pars, active, is_global)
    J0, r0, _, _ = p.sweep(); c0, _ = p.chi2()
    ctx.set_data(np.concatenate(sx), np.concatenate(sy), np.concatenate(sw), p.dp)
    J1, r1, c1 = ctx.sweep(pars, active, jac, dim)
    sc = np.sqrt(np.outer(np.diag(J0), np.diag(J0)))
    assert np.max(np.abs(J1 - J0) / sc) < 1e-11 and abs(c1 - c0) <= 1e-11 * c0
    assert np.max(np.abs(r1 - r0)) <= 1e-10 * np.max(np.abs(r0))
    return JTJ, JTr, chi2


def test_cfg2_single_curve_1e7_points_8_params(ctx):
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 10_000_000, 0.0, 100.0)
    _props(ctx, trace_model(M.model_exp4, 8), [x], [y], [1.0 / s], M.start_values(M.EXP4_TRUTH).reshape(1, 8), list(range(8)), [0] * 8)


def test_cfg3_global_fit_64_datasets_x_1e5(ctx):
    xs, ys, ss, truths = M.make_global7(64, 100_000)
    pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    JTJ, JTr, chi2 = _props(ctx, trace_model(M.model_global7, 7), xs, ys, [1.0 / s for s in ss], pars, list(range(7)),
                            [0, 0, 0, 0, 1, 1, 1], sample=6400)
    assert JTJ.shape == (259, 259)
    # block structure (user_guide.tex:222-235): local blocks of different datasets do not couple
    jac = np.array([[0, 1, 2, 3, 4, 5, 6]] + [[7 + 4 * (d - 1) + k for k in range(4)] + [4, 5, 6] for d in range(1, 64)])
    loc0, loc1 = jac[0][:4], jac[1][:4]
    assert np.all(JTJ[np.ix_(loc0, loc1)] == 0.0)
    # and a complete LM fit from the 5 % start values lands on the generating parameters
    X, Y, W = np.concatenate(xs), np.concatenate(ys), np.concatenate([1.0 / s for s in ss])
    ctx.set_data(X, Y, W, np.arange(65) * 100_000)
    out, r = ctx.fit(pars, list(range(7)), [0, 0, 0, 0, 1, 1, 1], lambda_=1.0, max_iter=12)
    assert r.dim == 259 and r.chi2 / r.dof < 1.1
    assert np.max(np.abs(out[0, 4:] - M.GLOBAL7_TAUS) / M.GLOBAL7_TAUS) < 2e-3


def test_cfg4_integral_model_1e6_points(ctx):
    """AD through Gauss-Kronrod quadrature at N = 1e6 (model of 2_integral_single.F90), rel 1e-10."""
    n = 1_000_000
    a, b = 7.5, 0.8
    x = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
    t = trace_model(G.model_integral_single, 2)
    t.set_integration(rel_error=1e-10)
    # exact data from the oracle's own integral would need 1e6 quadratures on the CPU; use the closed
    # form pi/2 * b^(-(a+1)/2) * gamma_lower((a+1)/2, b x^2) instead
    from scipy.special import gammainc, gamma
    f = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * x * x)
    sig = 0.01 * (1 + np.abs(f))
    y = f + sig * M.normal(n, M.SEED)
    pars = np.array([[a * 1.05, b * 0.95]])
    _props(ctx, t, [x], [y], [1.0 / sig], pars, [0, 1], [0, 0], sample=600)      # (two-kernel path, <= 8 parameters: k_gram_small sums r^2 in gfh_k_chi2's order)
    X = x; ctx.set_data(X, y, 1.0 / sig, [0, n])
    # the quadrature reproduces the closed form: chi2/N ~ 1 at the generating parameters
    assert abs(ctx.chi2(np.array([[a, b]])) / n - 1.0) < 0.01
    out, r = ctx.fit(pars, [0, 1], [0, 0], lambda_=1.0, accth=0.9, max_iter=8)
    assert abs(out[0, 0] - a) < 5e-3 and abs(out[0, 1] - b) < 5e-4 and r.n_omega == r.iterations


def test_cfg5_headline_1e7_points_32_params(ctx):
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 10_000_000, 0.0, 100.0)
    _props(ctx, trace_model(M.model_gauss8, 32), [x], [y], [1.0 / s], M.start_values(truth).reshape(1, 32), list(range(32)), [0] * 32)


def test_branching_model_1e7_points_33_params(ctx):
    """a branching eval() at the headline size: 8 Gaussians whose sum saturates at a fitted level (advar > advar; tests/branching.py),
    33 active parameters -- per-lane variant bodies in front of the fused kernel's matrix stage; ~12 % of the points saturate"""
    from gadfit_amd import tape as T
    from tests import branching as B
    truth = B.gauss8_saturating_truth()
    x, y, s = M.make_single(B.gauss8_saturating_numpy, truth, 10_000_000, 0.0, 100.0)
    start = np.concatenate([M.start_values(truth[:32]), [3.3]])
    V = T.Variants(B.model_gauss8_saturating, 33)
    V.explore(x[::200_003], start)
    assert len(V) == 2
    sat = np.count_nonzero(M.gauss8_numpy(start, x[::1000]) > start[32])
    assert 300 < sat < 5000
    _props(ctx, V, [x], [y], [1.0 / s], start.reshape(1, 33), list(range(33)), [0] * 33)
    assert ctx.n_variants() == 2
    ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit(start.reshape(1, 33), list(range(33)), [0] * 33, lambda_=1.0, max_iter=12)
    assert r.chi2 / r.dof < 1.05 and abs(out[0][32] - truth[32]) < 1e-3


@pytest.mark.gpu
def test_jacobian_placement_changes_nothing_but_the_address():
    """gfh_set_placement_tries / _after: once a Jacobian buffer of 256 MB or more has been written by `after` sweeps, the next sweep
    times several allocations with the kernel itself and keeps the fastest.  Whatever buffer is kept and
    whenever it is chosen, every number is the same: sums, residuals, the Jacobian read back, a short fit -- bitwise against a
    context that takes its first allocation."""
    n = 1_100_000                                     # 32 columns x 1.1e6 points x 8 B = 282 MB
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    act = list(range(32)); glob = [0] * 32
    start = M.start_values(truth).reshape(1, 32)
    out = []
    for tries, after in ((1, 0), (4, 0), (4, 2)):
        c = _lib.Context(0)
        try:
            c.set_placement_tries(tries); c.set_placement_after(after)
            c.set_model(tape); c.set_data(x, y, 1.0 / s, [0, n])
            jac, dim = c.jacobian_indices(act, glob)
            for k in range(after):                    # the sweeps of a job that is still short: no search yet
                first = c.sweep(start, act, jac, dim)
                assert c.placement() == []
            JTJ, JTr, chi2 = c.sweep(start, act, jac, dim)
            if after:
                assert np.array_equal(first[0], JTJ) and np.array_equal(first[1], JTr) and first[2] == chi2
            placed = c.placement()
            assert (len(placed) == 0) if tries == 1 else (1 <= len(placed) <= 4 and placed[0] == min(placed))
            J = c.jacobian(32)[::997].copy(); res = c.residuals()[::997].copy()
            p, r = c.fit(start.copy(), act, glob, lambda_=1.0, max_iter=3)
            out.append((JTJ, JTr, chi2, J, res, p.copy(), r.chi2))
        finally:
            c.close()
    for b in out[1:]:
        for k in range(7):
            assert np.array_equal(out[0][k], b[k]), k
    with pytest.raises(_lib.GadfitHipError):
        c2 = _lib.Context(0)
        try:
            c2.set_placement_tries(0)
        finally:
            c2.close()
    with pytest.raises(_lib.GadfitHipError):
        c2 = _lib.Context(0)
        try:
            c2.set_placement_after(-1)
        finally:
            c2.close()


def test_cfg5_total_size_1e8_points_one_gpu():
    """BASELINE config 5's TOTAL size on one card: N = 1e8 points x 32 active parameters (25.6 GB of Jacobian, offsets far beyond
    2^32 bytes, 2^31 elements).  Size-independent properties at full size, the oracle on a 1500-point sub-sample of the same inputs:
    chi2() bitwise the sweep's sum r^2; w -> 2 w scales everything by exactly 4; additivity over a 3-way contiguous partition;
    residuals and Jacobian rows read back at strided points over the whole range (the last ones at byte offsets ~25 GB) against
    the oracle's at the same abscissas."""
    n = 100_000_000
    truth = M.gauss8_truth()
    x, y, s = M.make_single_slice(M.gauss8_numpy, truth, n, 0, n, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    active = list(range(32)); start = M.start_values(truth).reshape(1, 32)
    c = _lib.Context(0)
    try:
        c.set_model(tape)
        c.set_data(x, y, s, [0, n])
        c.init_weights(4)
        jac, dim = c.jacobian_indices(active, [0] * 32)
        JTJ, JTr, chi2 = c.sweep(start, active, jac, dim)
        assert np.array_equal(JTJ, JTJ.T) and np.all(np.diag(JTJ) > 0)
        assert c.chi2(start) == chi2
        # strided sample over the whole range, both ends included
        idx = np.unique(np.concatenate([np.arange(0, n, n // 1499), [n - 1]])).astype(np.int64)
        res, J = c.points(idx, 32)
        p = orc.OracleProblem(tape, [x[idx]], [y[idx]], [1.0 / s[idx]], start, active, [0] * 32)
        _, _, res0, JT0 = p.sweep(want_J=True)
        assert np.max(np.abs(res - res0)) <= 7e-13 * np.max(np.abs(res0))
        scale = np.maximum(np.abs(JT0), 1e-6 * np.max(np.abs(JT0), axis=0, keepdims=True) + 1e-300)
        assert np.max(np.abs(J - JT0) / scale) < 7e-13
        # J^T res recomputed from the stored Jacobian by the J^T v kernel (reads all 25.6 GB)
        g = c.aux(0, dim=dim)
        assert np.max(np.abs(g - JTr)) <= 1e-10 * np.max(np.abs(JTr))
        # a fit at this size: the device's chi2 / dof sits at 1 for data drawn with the given sigma
        out, r = c.fit(start, active, [0] * 32, lambda_=1.0, max_iter=10)
        assert r.iterations == 10 and abs(r.chi2 / (n - 32) - 1.0) < 5e-3
        assert abs(out[0][0] - truth[0]) < 1e-3 * truth[0]      # (positions and skews of a peak are correlated and still on their way after 10 iterations)
        # weights x 2 -> everything x 4, exactly
        c.set_data(x, y, 0.5 * s, [0, n])
        c.init_weights(4)
        JTJ4, JTr4, chi4 = c.sweep(start, active, jac, dim)
        assert np.array_equal(JTJ4, 4.0 * JTJ) and np.array_equal(JTr4, 4.0 * JTr) and chi4 == 4.0 * chi2
        # additivity over the reference's 3-way contiguous partition
        acc = np.zeros_like(JTJ); accr = np.zeros_like(JTr); accc = 0.0
        for rk in range(3):
            b, cnt = _lib.partition(n, 3, rk)
            c.set_data(x[b:b + cnt], y[b:b + cnt], s[b:b + cnt], [0, cnt])
            c.init_weights(4)
            A, g3, c3 = c.sweep(start, active, jac, dim)
            acc += A; accr += g3; accc += c3
        sc = np.sqrt(np.outer(np.diag(JTJ), np.diag(JTJ)))
        assert np.max(np.abs(acc - JTJ) / sc) < 1e-12 and np.max(np.abs(accr - JTr)) <= 1e-11 * np.max(np.abs(JTr))
        assert abs(accc - chi2) <= 1e-12 * chi2
    finally:
        c.close()
