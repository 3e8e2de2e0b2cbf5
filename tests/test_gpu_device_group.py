"""Single-process device group (gfh_create_group): the reference's coarray images as one member context
and host thread per GPU behind one handle.  The round's GPU boxes have one card, so the members share
device 0 (separate streams and buffers): partition (gadfit.F90:977-983), the fan-out of every C-ABI
call, the ordered host sum of the members' result mailboxes (co_sum, misc.F90:133-170), the replicated
LM loop and the error path are the same code as on N cards.  Checked against a plain one-context run
and the CPU oracle."""
import os

import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import models as M
from tests.golden import goldens as G

pytestmark = pytest.mark.gpu


def _scaled(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize('members', [2, 3])
def test_group_passes_equal_single_context(members):
    """gfh_sweep / gfh_chi2 / gfh_omega / gfh_aux and the read-backs on a group of `members` contexts."""
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 20011, 0.0, 100.0)
    t = trace_model(M.model_gauss8, 32)
    act = list(range(32)); start = M.start_values(truth).reshape(1, 32)
    one = _lib.Context(0); grp = _lib.Context(devices=[0] * members)
    assert grp.group_size() == members and one.group_size() == 1
    out = []
    for c in (one, grp):
        c.set_model(t); c.set_data(x, y, s, [0, x.size]); c.init_weights(4)
        jac, dim = c.jacobian_indices(act, [0] * 32)
        JTJ, JTr, chi2 = c.sweep(start, act, jac, dim)
        d1 = np.linspace(-0.3, 0.4, dim)
        out.append(dict(JTJ=JTJ, JTr=JTr, chi2=chi2, chi2b=c.chi2(start), om=c.omega(start, d1), g=c.aux(0, dim=dim),
                        a1=c.aux(1, d1), res=c.residuals(), J=c.jacobian(32), omv=None, n=c.local_count(), b=c.local_begin()))
    a, b = out
    assert b['n'] == x.size and b['b'] == 0
    # per-point quantities do not depend on the split: bitwise
    assert np.array_equal(a['res'], b['res']) and np.array_equal(a['J'], b['J'])
    # sums: another order of additions (per-member partials, then rank order)
    sc = np.sqrt(np.outer(np.diag(a['JTJ']), np.diag(a['JTJ'])))
    assert np.max(np.abs(a['JTJ'] - b['JTJ']) / sc) < 1e-13
    assert _scaled(b['JTr'], a['JTr']) < 1e-12 and abs(a['chi2'] - b['chi2']) < 1e-13 * a['chi2']
    assert abs(a['chi2b'] - b['chi2b']) < 1e-13 * a['chi2b']
    assert _scaled(b['om'], a['om']) < 1e-11 and _scaled(b['g'], a['g']) < 1e-12 and _scaled(b['a1'], a['a1']) < 1e-12
    one.close(); grp.close()


def test_group_of_one_is_bitwise_the_plain_context():
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 5000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH).reshape(1, 8); act = list(range(8))
    res = []
    for c in (_lib.Context(0), _lib.Context(devices=[0])):
        c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size])
        out, r = c.fit(start.copy(), act, [0] * 8, lambda_=1.0, accth=0.9, max_iter=6)
        res.append((out, r.chi2, r.iterations, r.n_sweeps, r.n_chi2, r.n_omega)); c.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1:] == res[1][1:]


def test_group_fit_equals_single_context_and_oracle():
    """gfh_fit on a group: the LM loop runs on every member's thread with identical sums; result = member 0's."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 30000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH); act = list(range(8))
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], act, [0] * 8)
    r0 = p.fit(lambda_=np.float32(1.0), accth=np.float32(0.9), max_iter=6)
    one = _lib.Context(0); grp = _lib.Context(devices=[0, 0, 0])
    got = []
    for c in (one, grp):
        c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size])
        out, r = c.fit([start], act, [0] * 8, lambda_=1.0, accth=float(np.float32(0.9)), max_iter=6)
        got.append((out, r))
        assert (r.iterations, r.n_sweeps, r.n_chi2, r.n_omega) == (r0.iterations, r0.n_sweeps, r0.n_chi2, r0.n_omega)
        assert np.max(np.abs(out - p.pars) / np.abs(p.pars)) < 1e-10
    assert np.max(np.abs(got[0][0] - got[1][0]) / np.abs(got[0][0])) < 1e-11
    # lm_iterate (bench driver) through the group
    st = np.array([1.0, -1.0, 0.0]); dtd = np.zeros(8); pr = np.array([start])
    grp.lm_iterate(pr, act, [0] * 8, 4, st, dtd)
    st1 = np.array([1.0, -1.0, 0.0]); dtd1 = np.zeros(8); pr1 = np.array([start])
    one.lm_iterate(pr1, act, [0] * 8, 4, st1, dtd1)
    assert np.max(np.abs(pr - pr1) / np.abs(pr1)) < 1e-10 and st[2] == st1[2] and abs(st[1] - st1[1]) < 1e-11 * st1[1]
    one.close(); grp.close()


def test_group_global_fit_members_span_dataset_boundaries():
    """Reference test 4 (two curves, one global parameter) and a 5-dataset global fit on 3 members: a member's
    contiguous range crosses dataset boundaries; the packed block-arrow image is summed over the members."""
    xs, ys, ss, truths = M.make_global7(5, 700)
    pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    t = trace_model(M.model_global7, 7)
    X = np.concatenate(xs); Y = np.concatenate(ys); W = np.concatenate([1.0 / s for s in ss])
    pos = np.arange(6) * 700
    isg = [0, 0, 0, 0, 1, 1, 1]; act = list(range(7))
    p = orc.OracleProblem(t, xs, ys, [1.0 / s for s in ss], [q.copy() for q in pars], act, isg)
    r0 = p.fit(lambda_=np.float32(1.0), max_iter=5)
    grp = _lib.Context(devices=[0, 0, 0])
    grp.set_model(t); grp.set_data(X, Y, W, pos)
    out, r = grp.fit(pars.copy(), act, isg, lambda_=1.0, max_iter=5)
    assert r.dim == 5 * 4 + 3 and r.iterations == r0.iterations
    assert np.max(np.abs(out - p.pars) / np.abs(p.pars)) < 1e-9
    grp.close()


def test_group_error_raised_by_one_member_reaches_the_caller():
    """A quadrature that exhausts its workspace (status word raised by some members' kernels): the status
    travels with the host sum, every member stops with the reference's message (NI:282-283)."""
    t = trace_model(G.model_integral_single, 2)
    t.set_integration(rel_error=1e-30)
    # two points on three members (gadfit.F90:978-983: one each to the first two): the third member launches
    # nothing and learns of the failure only through the sum
    n = 2
    x = np.array([1.0, 2.0])
    grp = _lib.Context(devices=[0, 0, 0])
    grp.set_model(t); grp.set_data(x, np.ones(n), np.ones(n), [0, n])
    with pytest.raises(_lib.GadfitHipError, match='Number of iterations was insufficient'):
        grp.chi2([[7.5, 0.8]])
    # the group stays usable
    t2 = trace_model(G.model_integral_single, 2); t2.set_integration(rel_error=1e-8)
    grp.set_model(t2)
    assert np.isfinite(grp.chi2([[7.5, 0.8]]))
    # calls that make no sense on a group handle are refused
    with pytest.raises(_lib.GadfitHipError, match='not available on a device-group handle'):
        grp.debug_set_rank(2, 0)
    grp.close()


def test_group_rccl_reduction_single_member(monkeypatch):
    """RCCL is how a device group sums wherever every member has a card of its own (ncclCommInitAll, all-reduces on the
    members' streams).  One card here, so one member; members sharing a card fall back to the ordered host sum, and
    asking for RCCL explicitly then is refused."""
    g2 = _lib.Context(devices=[0, 0])
    assert g2.comm_info()[0] == 0                  # host sum: no communicator
    g2.close()
    monkeypatch.setenv('GADFIT_HIP_GROUP_REDUCE', 'rccl')
    with pytest.raises(_lib.GadfitHipError, match='one device per member'):
        _lib.Context(devices=[0, 0])
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 3000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH).reshape(1, 8); act = list(range(8))
    grp = _lib.Context(devices=[0]); one = _lib.Context(0)
    assert grp.comm_info()[0] == 1 and one.comm_info()[0] == 0
    res = []
    for c in (one, grp):
        c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size])
        jac, dim = c.jacobian_indices(act, [0] * 8)
        res.append(c.sweep(start, act, jac, dim) + (c.chi2(start),))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]
    assert grp.comm_info()[1] >= 2                 # the sweep's and the chi2's all-reduce
    one.close(); grp.close()


def test_group_with_finite_differences_and_losses():
    """Options set on the handle reach every member: use_ad = .false. (gfh_set_use_ad), a robust loss, keep_jacobian
    mode 2 and the reference schedule; each against the plain context with the same option."""
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 6000, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH).reshape(1, 8); act = list(range(8))
    for setup, tol in [(lambda c: c.set_use_ad(False), 1e-6), (lambda c: c.set_loss(1), 1e-10),
                       (lambda c: c.set_keep_jacobian(2), 1e-10), (lambda c: c.set_lookahead(False), 1e-10)]:
        res = []
        for c in (_lib.Context(0), _lib.Context(devices=[0, 0])):
            c.set_model(t); c.set_data(x, y, 1.0 / s, [0, x.size]); setup(c)
            out, r = c.fit(start.copy(), act, [0] * 8, lambda_=1.0, accth=0.9, max_iter=5)
            res.append((out, r)); c.close()
        assert res[0][1].iterations == res[1][1].iterations == 5 and res[0][1].n_sweeps == res[1][1].n_sweeps
        assert np.max(np.abs(res[0][0] - res[1][0]) / np.abs(res[0][0])) < tol


def test_bench_launcherless_multi_gpu_path_on_one_card():
    """`python bench.py --gpus 2` without torch.distributed.run drives the GPUs from one process through the device
    group, whose sums travel by RCCL.  On a one-card box that run must FAIL LOUDLY (no second device), and the rehearsal
    with both members on the one card (GADFIT_HIP_GROUP_WRAP) is refused too unless the host sum is asked for by name."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--pre-roll', '0',
            '--points', '150000', '--cpu-sample', '0', '--min-timed', '0', '--legs', 'main']
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'GADFIT_HIP_GROUP_WRAP', 'GADFIT_HIP_GROUP_REDUCE'):
        env.pop(k, None)
    p = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode != 0 and 'device index out of range' in p.stderr, p.stderr[-2000:]
    p = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, GADFIT_HIP_GROUP_WRAP='1'))
    assert p.returncode != 0 and 'does not sum through RCCL' in p.stderr, p.stderr[-2000:]
    p = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root,
                       env=dict(env, GADFIT_HIP_GROUP_WRAP='1', GADFIT_HIP_GROUP_REDUCE='host'))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['config']['points_total'] == 300000 and d['value'] > 0
    assert 'device group' in d['config']['parallelism'] and d['final_chi2_per_dof'] < 1e3
    assert d['rccl_nranks'] == 0 and 'host sum' in d['cross_rank_sum']


def test_group_load_balancing_recuts_the_ranges():
    """load_balancing of gadf_fit (adaptive parallelism, gadfit.F90:672-673, 935-983) on a device group: the cost of a
    point of the quadrature model grows with x, so an even contiguous split is uneven work; gfh_repartition cuts the
    ranges for given image weights, gfh_rebalance derives them from the measured device times by the reference's
    update, gfh_fit does that before every iteration.  Sums and fits must not depend on the cut."""
    n = 24000
    a, b = 7.5, 0.8
    x = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
    from scipy.special import gammainc, gamma
    f = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * x * x)
    sig = 0.01 * (1 + np.abs(f))
    y = f + sig * M.normal(n, M.SEED)
    t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
    pars = np.array([[a * 1.05, b * 0.95]])
    one = _lib.Context(0)
    one.set_model(t); one.set_data(x, y, sig, [0, n]); one.init_weights(4)
    jac, dim = one.jacobian_indices([0, 1], [0, 0])
    ref = one.sweep(pars, [0, 1], jac, dim)
    out1, r1 = one.fit(pars.copy(), [0, 1], [0, 0], lambda_=1.0, accth=0.9, max_iter=5)
    one.close()
    g = _lib.Context(devices=[0, 0, 0])
    g.set_load_balancing(True)
    g.set_model(t); g.set_data(x, y, sig, [0, n]); g.init_weights(4)
    even = g.sweep(pars, [0, 1], jac, dim)
    # an explicit cut: 50 % / 30 % / 20 %
    g.repartition([0.5, 0.3, 0.2])
    assert list(g.group_ranges()[1]) == [12000, 7200, 4800]
    g.repartition([5.0, 3.0, 2.0])                     # weights are normalised
    assert list(g.group_ranges()[1]) == [12000, 7200, 4800]
    cut = g.sweep(pars, [0, 1], jac, dim)
    res = g.residuals()
    assert res.size == n and g.local_count() == n
    for got in (even, cut):
        assert np.max(np.abs(got[0] - ref[0]) / np.abs(ref[0])) < 1e-12 and abs(got[2] - ref[2]) < 1e-12 * ref[2]
    # measured cut: the members with the cheap (small x) ranges must end up with more points than an even share
    g.repartition([1 / 3, 1 / 3, 1 / 3])
    g.reset_timers()
    moved = False
    for _ in range(6):
        for _ in range(3):
            g.sweep(pars, [0, 1], jac, dim)
        moved = g.rebalance() or moved
    assert moved
    b, cnt = g.group_ranges()
    print('ranges after balancing:', b, cnt)
    assert b[0] == 0 and np.all(b[1:] == np.cumsum(cnt)[:-1]) and cnt.sum() == n
    # (which way the cut moves is not asserted: the three members share one card here, so their kernels overlap and the
    # measured times do not order by work; on separate cards the cheap end of the array gets the longer range)
    final = g.sweep(pars, [0, 1], jac, dim)
    assert np.max(np.abs(final[0] - ref[0]) / np.abs(ref[0])) < 1e-12
    # whole fit with balancing switched on: same iterations and parameters as the single context
    out, r = g.fit(pars.copy(), [0, 1], [0, 0], lambda_=1.0, accth=0.9, max_iter=5)
    assert r.iterations == r1.iterations == 5 and r.n_lookahead == 0
    assert np.max(np.abs(out - out1) / np.abs(out1)) < 1e-9
    g.close()


def test_large_workspaces_from_several_contexts_of_one_card():
    """Kernels compiled at the user's workspace size used to carry 32 KB of private scratch per lane, which the runtime provides per
    hardware queue for a whole device (15.8 GB): two queues asking for it at once ended the process (HSA_STATUS_ERROR_OUT_OF_RESOURCES,
    no HIP error to catch).  The user-sized workspaces now live in a pool in global memory that each context allocates itself
    (codegen.cpp GFH_WSG, context.cpp wsg_grid): plain contexts on two host threads and device groups of two and three members
    sharing the card escalate side by side, each stops with the reference's message, and a fit that NEEDS the large workspace gives
    the same numbers from two threads at once as alone."""
    import threading
    x = np.array([1.0, 2.0])

    def exhausted(make):
        t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-30)
        c = make()
        try:
            c.set_model(t); c.set_data(x, np.ones(2), np.ones(2), [0, 2])
            with pytest.raises(_lib.GadfitHipError, match='Number of iterations was insufficient'):
                c.chi2([[7.5, 0.8]])
        finally:
            c.close()
    for make in (lambda: _lib.Context(0), lambda: _lib.Context(devices=[0, 0]), lambda: _lib.Context(devices=[0, 0, 0])):
        exhausted(make)
    # a sum of narrow peaks that needs ~300-500 intervals: beyond the fast form's 100, within the default 1000
    from gadfit_amd.ad import integrate

    def peaks(p, xx):
        def kern(tt, q):
            y = q[0] / ((tt - q[1]) ** 2 + 1e-10)
            for k in range(1, 6):
                y = y + q[0] / ((tt - (q[1] + 0.13 * k)) ** 2 + 1e-10)
            return y
        return integrate(kern, [p[0], p[1]], 0.0, xx) * 1e-5
    tp = trace_model(peaks, 2); tp.set_integration(rel_error=1e-13)
    xs = np.array([0.5, 0.8, 1.0, 0.62, 0.93])
    results = {}

    def run(tag):
        c = _lib.Context(0)
        try:
            c.set_model(tp); c.set_data(xs, np.ones(5), np.ones(5), [0, 5])
            results[tag] = [c.chi2([[1.0, 0.111]]) for _ in range(3)]
        finally:
            c.close()
    run('alone')
    th = [threading.Thread(target=run, args=(k,)) for k in ('a', 'b')]
    for t_ in th: t_.start()
    for t_ in th: t_.join()
    assert results['a'] == results['alone'] and results['b'] == results['alone'] and np.isfinite(results['alone'][0])


def test_bench_multi_gpu_line_proves_itself_on_one_card():
    """Eight members on the one card (GADFIT_HIP_GROUP_WRAP, ordered host sum): the N > 1 line carries its own cross-rank checks --
    the summed [JTJ | JTres | chi2] against the rank-ordered sum of the members' partials, a 10-iteration fit against a one-rank
    fit, the measured latency of one sum, a strong-scaling leg -- and the run fails when they fail (tests/test_cpu_bench_schema.py
    pins the schema on dry members)."""
    import json, subprocess, sys
    from tests.test_cpu_bench_schema import check_multi_block
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '1', '--pre-roll', '0',
            '--points', '120000', '--cpu-sample', '0', '--min-timed', '0', '--legs', 'main']
    env = dict(os.environ, GADFIT_HIP_GROUP_WRAP='1', GADFIT_HIP_GROUP_REDUCE='host')
    for k in ('WORLD_SIZE', 'RANK'):
        env.pop(k, None)
    p = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    mp = check_multi_block(d, 8)
    assert mp['ok'] and mp['cross_rank_sum_path'] == 'host'
    assert max(mp['sums_vs_ordered_host_sum']['max_dev'].values()) == 0.0        # the host sum IS the rank-ordered sum: bitwise
    f = mp['fit_vs_one_rank']
    assert f['iterations'] == [6, 6] and f['skews_fixed']['iterations'] == [6, 6] and f['ok'] and f['max_dev_over_bound'] <= 1.0 and f['ranks_agree_bitwise'] and f['skews_fixed']['max_rel_dev_pars'] <= 1e-10
    assert d['strong_leg']['points_total'] == 120000 and d['strong_leg']['ms_per_step'] > 0
    assert d['host_sum_ms_per_step'] == d['ms_per_step'] and d['rccl_ms_per_step'] is None
