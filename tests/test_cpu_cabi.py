"""CPU-only checks of the C ABI library: it loads, exports every symbol include/gadfit_hip.h
declares, generates + compiles model kernels without a GPU, and refuses to compute without one."""
import os
import re

import numpy as np
import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import models as M
from tests.golden import goldens as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'gadfit_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gfh_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported_and_bound():
    import ctypes
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), 'libgadfit_hip.so does not export ' + n
        assert n in _lib.SYMBOLS, 'gadfit_amd/_lib.py does not bind ' + n
    assert sorted(_lib.SYMBOLS) == names
    # ... and the header is the library's WHOLE dynamic symbol table (-fvisibility=hidden + gadfit_amd/csrc/exports.map): no C++
    # internals, no kernel handles
    import shutil
    import subprocess
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    out = subprocess.run([nm, '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == names, sorted(set(exported) ^ set(names))


def test_version_and_partition_rule():
    assert _lib.lib().gfh_version() >= 100
    # gadfit.F90:978-983: int(N/G) each, remainder +1 to the first ranks
    for n, g in [(80, 3), (10, 4), (7, 8), (10_000_019, 8), (0, 2)]:
        tot = 0
        for r in range(g):
            b, c = _lib.partition(n, g, r)
            assert b == tot and c == n // g + (1 if r < n % g else 0)
            tot += c
        assert tot == n


def test_partition_matches_oracle_img_bounds():
    import ctypes as C
    dp = np.array([0, 50, 80, 80, 200], dtype=np.int64)
    for g in (1, 2, 3, 5):
        for r in range(g):
            b = np.zeros(5, dtype=np.int64)
            orc.lib().orc_img_bounds(g, r, 4, dp.ctypes.data_as(C.POINTER(C.c_int64)), b.ctypes.data_as(C.POINTER(C.c_int64)))
            begin, count = _lib.partition(200, g, r)
            assert b[0] == begin and b[-1] == begin + count
            for d in range(4):   # per-dataset sub-range = intersection with the dataset
                lo, hi = max(begin, dp[d]), min(begin + count, dp[d + 1])
                assert b[d + 1] - b[d] == max(0, hi - lo)


def test_jacobian_indices_and_potr_match_oracle():
    ctx = _lib.Context(-1)
    ctx.nd = 3
    jac, dim = ctx.jacobian_indices([0, 1, 3, 4], [0, 1, 0, 1, 1])
    p_active = np.array([0, 1, 3, 4], dtype=np.int32); g = np.array([0, 1, 0, 1, 1], dtype=np.int32)
    j2 = np.zeros((3, 4), dtype=np.int32)
    d2 = orc.lib().orc_jacobian_indices(3, 4, orc._ip(p_active), orc._ip(g), orc._ip(j2))
    assert dim == d2 == 3 + 1 * 3 and np.array_equal(jac, j2)   # 3 globals + 1 local x 3 datasets
    rng = np.random.default_rng(1)
    a = rng.normal(size=(9, 9)); a = a @ a.T + 9 * np.eye(9); b = rng.normal(size=9)
    x = _lib.potr(a, b)
    assert np.array_equal(x, orc.potr(a, b))
    assert np.allclose(a @ x, b, rtol=1e-12, atol=1e-12)
    with pytest.raises(_lib.GadfitHipError):
        _lib.potr(-np.eye(3), np.ones(3))
    ctx.close()


@pytest.mark.parametrize('model,n,active', [(M.model_gauss8, 32, list(range(32))), (M.model_exp4, 8, [0, 1, 2, 5]),
                                            (G.expr_trig, 2, [0, 1]), (G.expr_power, 2, [1]), (G.expr_basic_forward, 3, [0, 2])])
def test_codegen_compiles_for_gfx950_without_gpu(model, n, active):
    ctx = _lib.Context(-1)
    ctx.set_model(trace_model(model, n))
    src = ctx.model_source(active)
    assert 'gfh_k_sweep' in src and 'gfh_k_chi2' in src and 'gfh_k_omega' in src
    assert '#define GFH_NA %d' % len(active) in src
    ctx.model_prepare(active)        # hiprtc --offload-arch=gfx950
    ctx.close()


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import ctypes
    if os.path.exists('/dev/kfd'):
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.GadfitHipError):
        _lib.Context(0)
    with pytest.raises(_lib.GadfitHipError, match='no HIP device'):      # single-process device group: same rule
        _lib.Context(devices=[0, 1])
    with pytest.raises(_lib.GadfitHipError, match='no HIP device'):
        _lib.Context(devices='all')
    ctx = _lib.Context(-1)
    ctx.set_model(trace_model(M.model_exp4, 8))
    with pytest.raises(_lib.GadfitHipError, match='no GPU'):
        ctx.set_data([0.0], [0.0], [1.0], [0, 1])
    with pytest.raises(_lib.GadfitHipError, match='no GPU'):
        ctx.chi2(np.ones((1, 8)))
    ctx.close()


def test_malformed_tapes_are_rejected():
    from gadfit_amd import tape as T
    ctx = _lib.Context(-1)
    t = T.Tape(1)
    t.subtapes.append(([(T.PARAM, 0, -1, 0, 0.0), (T.EXP, 5, -1, 0, 0.0)], 1))   # operand refers forward
    with pytest.raises(_lib.GadfitHipError):
        ctx.set_model(t)
    t = T.Tape(1)
    t.subtapes.append(([(T.PARAM, 3, -1, 0, 0.0)], 0))                             # parameter out of range
    with pytest.raises(_lib.GadfitHipError):
        ctx.set_model(t)
    ctx.close()


def test_committed_generated_examples_are_current():
    """gadfit_amd/csrc/generated_examples/*.hip are what the code generator emits today."""
    ctx = _lib.Context(-1)
    ctx.set_model(trace_model(M.model_gauss8, 32))
    src = ctx.model_source(list(range(32)))
    text = open(os.path.join(ROOT, 'gadfit_amd', 'csrc', 'generated_examples', 'gauss8_32active.hip')).read()
    assert text.endswith(src), 'run tools/dump_generated.py'
    t = trace_model(G.model_integral_single, 2)
    t.set_integration(rel_error=1e-12)
    ctx.set_model(t)
    text = open(os.path.join(ROOT, 'gadfit_amd', 'csrc', 'generated_examples', 'integral_single_2active.hip')).read()
    assert text.endswith(ctx.model_source([0, 1])), 'run tools/dump_generated.py'
    ctx.close()


def test_aux_column_tapes_validate_and_compile_without_gpu():
    """GFH_AUX nodes (tabulated real functions of x): accepted in eval() and, since round 4, inside an integrand (a real the integrand
    takes from the enclosing eval() past pars(:): read back from the lane's stash), refused when the column index exceeds
    gfh_tape.n_aux; the generated kernels read them through the aux pointer."""
    from gadfit_amd import tape as T
    from gadfit_amd.ad import aux, exp, integrate
    t = trace_model(lambda p, x: p[0] * aux(0) + p[1] * exp(-aux(1) * p[2]), 3)
    assert t.n_aux == 2
    ctx = _lib.Context(-1)
    try:
        ctx.set_model(t)
        src = ctx.model_source([0, 1, 2])
        assert 'AXP[(i64)0 * LDA]' in src and 'AXP[(i64)1 * LDA]' in src
        ctx.model_prepare([0, 1, 2])                      # hiprtc compile for gfx950
        t.n_aux = 1; t._c = None                          # column 1 now out of range
        with pytest.raises(_lib.GadfitHipError, match='auxiliary column out of range'):
            ctx.set_model(t)
        t2 = trace_model(lambda p, x: integrate(lambda u, q: q[0] * u * aux(0) * x, [p[0]], 0.0, x), 1)
        ctx.set_model(t2)
        src = ctx.model_source([0])
        assert 'gfh_lane_axp[threadIdx.x][(i64)0 * gfh_lane_lda]' in src and 'gfh_lane_x[threadIdx.x]' in src and 'GFH_LANE_STASH gfh_lane_x' in src
        ctx.model_prepare([0])
    finally:
        ctx.close()


def test_finite_differences_over_column_sets_offset_the_aux_pointer_per_evaluation():
    """gfh_set_fd_column_sets (use_ad = 0 over columns that follow the parameters, fitfunction.F90:155-174): evaluation j of the forward
    differences reads set 1 + j of the model's columns -- in the generated source an offset of (1 + j) * n_aux columns on the aux
    pointer; without the switch, and under AD, the pointer is passed as it is.  Host code: source and hiprtc compile without a GPU."""
    from gadfit_amd.ad import aux, exp
    t = trace_model(lambda p, x: p[0] * aux(0) + p[1] * exp(-aux(1) * p[2]), 3)
    ctx = _lib.Context(-1)
    try:
        ctx.set_model(t)
        ctx.set_use_ad(False)
        plain = ctx.model_source([0, 2])
        assert 'AXP + (i64)' not in plain
        ctx.set_fd_column_sets(True)
        src = ctx.model_source([0, 2])
        assert 'STATUS, AXP + (i64)2 * LDA, LDA' in src and 'STATUS, AXP + (i64)4 * LDA, LDA' in src and 'AXP + (i64)6' not in src
        ctx.model_prepare([0, 2])
        ctx.set_use_ad(True)
        assert 'AXP + (i64)' not in ctx.model_source([0, 2])
    finally:
        ctx.close()


@pytest.mark.parametrize('nd,nl,ng', [(64, 4, 3), (2, 2, 1), (5, 3, 0), (3, 0, 4), (40, 6, 5), (1, 8, 0), (7, 1, 2)])
def test_damped_solve_block_arrow_equals_dense(nd, nl, ng):
    """gfh_solve_damped: the structure-exploiting solve of a global fit's normal equations (local blocks one by
    one, dense Schur complement of the global parameters) against the dense Cholesky and numpy, on matrices
    with the reference's column map (Jacobian_indices, gadfit.F90:615-628); also the blocked dense path > 64."""
    rng = np.random.default_rng(nd * 100 + nl * 10 + ng)
    na = nl + ng
    act = np.arange(na, dtype=np.int32); glob = np.array([0] * nl + [1] * ng, dtype=np.int32)
    jac = np.zeros((nd, na), dtype=np.int32)
    dim = _lib.lib().gfh_jacobian_indices(nd, na, act.ctypes.data_as(_lib._ip), glob.ctypes.data_as(_lib._ip), jac.ctypes.data_as(_lib._ip))
    assert dim == ng + nl * nd
    J = np.zeros((30 * nd, dim))
    for d in range(nd):
        J[30 * d:30 * (d + 1), jac[d]] = rng.standard_normal((30, na))
    JTJ = J.T @ J; rhs = rng.standard_normal(dim); DTD = np.diag(JTJ).copy(); lam = 0.37
    ref = np.linalg.solve(JTJ + lam * np.diag(DTD), rhs)
    xa = _lib.solve_damped(jac, dim, JTJ, DTD, lam, rhs, True)
    xd = _lib.solve_damped(jac, dim, JTJ, DTD, lam, rhs, False)
    sc = np.max(np.abs(ref))
    assert np.max(np.abs(xa - ref)) <= 1e-12 * sc and np.max(np.abs(xd - ref)) <= 1e-12 * sc
    # a matrix that is not positive definite is reported, not solved
    bad = JTJ.copy(); bad[0, 0] = -1.0
    for use in (True, False):
        with pytest.raises(_lib.GadfitHipError, match='Cholesky factorization failed'):
            _lib.solve_damped(jac, dim, bad, np.zeros(dim), 0.0, rhs, use)


def test_device_group_fan_out_and_ordered_host_sum_without_gpu():
    """The single-process device group (gfh_create_group) with compile-only members: every call runs on all member
    threads, the host sum over the members is taken in rank order (bitwise the sequential sum, identical on all
    members), the status word travels as a maximum, and a member that fails releases the others from the barrier."""
    for members in (1, 2, 5, 8):
        g = _lib.Context(devices=[-1] * members)
        assert g.group_size() == members
        rng = np.random.default_rng(members)
        for rep in range(50):                          # many rounds: the barrier's phases, back to back
            n = int(rng.integers(1, 1200))
            bufs = rng.normal(size=(members, n)) * 10.0 ** rng.integers(-8, 8, size=(members, 1))
            want = bufs[0].copy()
            for r in range(1, members):
                want = want + bufs[r]                  # rank order
            status = np.zeros(members, dtype=np.int32); status[rep % members] = rep % 3
            work = bufs.copy()
            g.debug_group_allreduce(work, status)
            assert all(np.array_equal(work[r], want) for r in range(members))
            assert np.all(status == rep % 3)
        if members > 1:
            work = np.ones((members, 7)); status = np.zeros(members, dtype=np.int32)
            with pytest.raises(_lib.GadfitHipError, match='failed on purpose'):
                g.debug_group_allreduce(work, status, fail_member=members - 1)
            g.debug_group_allreduce(work, status)      # usable again after a failed call
            assert np.all(work == members)
        # fan-out of ordinary calls: the model reaches every member (compiled once, the others load the cache) ...
        g.set_model(trace_model(M.model_exp4, 8))
        g.model_prepare([0, 1, 2, 5])
        assert 'gfh_k_sweep' in g.model_source([0, 1, 2, 5])
        # ... and a call that needs a GPU fails on all of them with the plain context's message
        with pytest.raises(_lib.GadfitHipError, match='no GPU'):
            g.set_data([0.0, 1.0], [0.0, 1.0], [1.0, 1.0], [0, 2])
        g.close()


def test_fit_arguments_are_validated_before_anything_reads_through_them():
    """gfh_fit / Context.fit: active[] indexes is_global[] and the parameter block, DTD_min is read for dim entries, pars is
    written back -- lengths and ranges are refused up front (no GPU needed: validation precedes the first device call)."""
    import pytest
    from gadfit_amd import _lib
    from gadfit_amd.ad import trace_model
    from tests import models as M
    c = _lib.Context(-1)
    c.set_model(trace_model(M.model_exp4, 8))
    c.nd = 1
    start = np.ones((1, 8))
    with pytest.raises(_lib.GadfitHipError, match='active parameter indices'):
        c.fit(start, [0, 8], [0] * 8, max_iter=1)
    with pytest.raises(_lib.GadfitHipError, match='active parameter indices'):
        c.fit(start, [-1], [0] * 8, max_iter=1)
    with pytest.raises(_lib.GadfitHipError, match='n_datasets x n_pars'):
        c.fit(np.ones((1, 7)), [0, 1], [0] * 8, max_iter=1)
    with pytest.raises(_lib.GadfitHipError, match='DTD_min must hold dim = 2'):
        c.fit(start, [0, 1], [0] * 8, DTD_min=[1.0], max_iter=1)
    with pytest.raises(_lib.GadfitHipError, match='one flag per parameter'):
        c.fit(start, [0, 1], [0] * 3, max_iter=1)
    with pytest.raises(_lib.GadfitHipError, match='no GPU bound'):          # a well-formed call gets as far as the device
        c.fit(start, [0, 1], [0] * 8, max_iter=1)
    c.close()


def test_a_queued_host_copy_is_made_whatever_becomes_of_the_upload():
    """gfh_queue_host_copy: the copy is made by the next gfh_set_data_begin on every path out of it -- here a context without a GPU,
    which refuses the upload -- and nothing stays queued for a later call (the Fortran layer points its abscissas at the
    destination after the fit)"""
    import numpy as np
    c = _lib.Context(-1)
    try:
        src = np.arange(1000, dtype=np.float64); dst = np.zeros(1000)
        c.queue_host_copy(dst, src)
        with pytest.raises(_lib.GadfitHipError, match='no GPU bound'):
            c.set_data_begin(src, src, src, [0, 1000])
        c.wait_host_copy()
        assert np.array_equal(dst, src)
        dst[:] = 0.0
        with pytest.raises(_lib.GadfitHipError, match='no GPU bound'):
            c.set_data_begin(src, src, src, [0, 1000])
        c.wait_host_copy()
        assert not dst.any()
    finally:
        c.close()


def test_create_begin_reports_a_missing_device_at_the_first_call_that_needs_it():
    """gfh_create_begin (round 5) returns at once and sets the device up on a thread of the context; without a GPU the first call that
    needs the device fails with the message gfh_create would have given, and so does every later one (no call ever runs on a
    half-made context); gfh_destroy cleans up.  device < 0 is plain gfh_create."""
    import ctypes as C
    if os.path.exists('/dev/kfd'):
        pytest.skip('a GPU is present')
    L = _lib.lib()
    h = C.c_void_p()
    assert L.gfh_create_begin(0, C.byref(h)) == 0 and h.value
    for _ in range(2):
        assert L.gfh_sync(h) != 0
        assert 'no HIP device available' in L.gfh_last_error(h).decode()
    # host-only calls still work on it (what gadf_init makes before the first gadf_fit)
    assert L.gfh_set_keep_jacobian(h, 2) == 0 and L.gfh_comm_init_from_env(h) == 0
    L.gfh_destroy(h)
    h2 = C.c_void_p()
    assert L.gfh_create_begin(-1, C.byref(h2)) == 0 and h2.value
    L.gfh_destroy(h2)
