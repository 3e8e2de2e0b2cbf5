"""The Fortran drop-in layer (modules ad / fitfunction / gadfit over ISO_C_BINDING).
CPU: it builds with amdflang, captures a user eval() into a tape the library accepts, and --
there being no CPU fallback -- stops loudly at the first device call.
GPU: the reference's golden fits are reproduced through the Fortran API."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'tests', 'fortran', 'build')
GOLD = os.path.join(ROOT, 'tests', 'golden')

needs_flang = pytest.mark.skipif(shutil.which('amdflang') is None and not os.path.exists('/opt/rocm/bin/amdflang'),
                                 reason='amdflang not available')


_built = []


def _build():
    """once per test session; build.py itself skips whatever is newer than its sources (build() made it ahead of time)"""
    if not _built:
        subprocess.check_call(['python3', os.path.join(ROOT, 'gadfit_amd', 'fortran', 'build.py')])
        _built.append(True)


@needs_flang
def test_fortran_layer_builds_and_fails_loudly_without_gpu():
    _build()
    exe = os.path.join(BUILD, 'fit_gaussian')
    assert os.path.exists(exe) and os.path.exists(os.path.join(BUILD, 'fit_two_curves'))
    if os.path.exists('/dev/kfd'):
        pytest.skip('a GPU is present')
    # compile-only context: model capture + gfh_set_model succeed, the first device call stops
    p = subprocess.run([exe, os.path.join(GOLD, 'gaussian_xy.txt')], env=dict(os.environ, GADFIT_HIP_DEVICE='-1'),
                       capture_output=True, text=True)
    assert p.returncode != 0
    assert 'no GPU bound to this context' in p.stderr and 'gadfit.F90' in p.stderr
    # default device 0: creation itself fails
    p = subprocess.run([exe, os.path.join(GOLD, 'gaussian_xy.txt')], capture_output=True, text=True)
    assert p.returncode != 0 and 'no HIP device' in p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_fit_gaussian_golden():
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_gaussian'), os.path.join(GOLD, 'gaussian_xy.txt')],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_fit_two_curves_golden():
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_two_curves'), os.path.join(GOLD, 'curve1_xy.txt'),
                        os.path.join(GOLD, 'curve2_xy.txt')], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_fit_integral_single_golden():
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_integral_single'), os.path.join(GOLD, 'integral_single_xy.txt')],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_fit_integral_double_golden():
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_integral_double'), os.path.join(GOLD, 'integral_double_xys.txt')],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cauchy', 'huber'])
def test_fortran_robust_loss_goldens(name):
    """gadf_set_loss through the Fortran layer against c++/tests/lm_solver.cpp:499-565."""
    from tests.golden import goldens as G
    _build()
    loss, iters, chi2_ref, tau, i00, b0, i01, b1 = G.CXX_LOSS[name]
    fd = G.cxx_fix_d()
    start = [fd[0], fd[1], fd[4], fd[5], fd[3]]
    p = subprocess.run([os.path.join(BUILD, 'fit_robust_loss'), os.path.join(GOLD, 'curve1_xy.txt'), os.path.join(GOLD, 'curve2_xy.txt'),
                        str(loss), str(iters)] + ['%.17g' % v for v in start], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
    got = {}
    for ln in p.stdout.splitlines():
        if ln.startswith('PAR'):
            _, i, j, v = ln.split()
            got[(int(i), int(j))] = float(v)
    want = {(1, 1): i00, (1, 2): tau, (1, 3): b0, (2, 1): i01, (2, 2): tau, (2, 3): b1}
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-10 * abs(v), (k, got[k], v)


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('prog,files', [('fit_two_curves', ['curve1_xy.txt', 'curve2_xy.txt']), ('fit_gaussian', ['gaussian_xy.txt']),
                                        ('fit_integral_single', ['integral_single_xy.txt'])])
def test_fortran_device_group_reproduces_goldens(prog, files):
    """GADFIT_HIP_DEVICES=3: the unchanged Fortran programs on a single-process device group of three images
    (threads; the three share the one card of this box, GADFIT_HIP_GROUP_WRAP) still meet the reference's
    known answers -- as the reference's own tests do under `cafrun -n 3`."""
    _build()
    env = dict(os.environ, GADFIT_HIP_DEVICES='3', GADFIT_HIP_GROUP_WRAP='1')
    p = subprocess.run([os.path.join(BUILD, prog)] + [os.path.join(GOLD, f) for f in files],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    # and with adaptive parallelism (load_balancing=.true. of gadf_fit, here through the environment): the ranges of
    # the three images are re-cut between iterations from their device times; the known answers still hold
    p = subprocess.run([os.path.join(BUILD, prog)] + [os.path.join(GOLD, f) for f in files],
                       capture_output=True, text=True, timeout=600, env=dict(env, GADFIT_HIP_LOAD_BALANCING='1'))
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_env_communicator_single_rank(tmp_path):
    """GADFIT_HIP_NRANKS/_RANK/_IDFILE bootstrap (file rendezvous + ncclCommInitRank) with one rank."""
    _build()
    env = dict(os.environ, GADFIT_HIP_NRANKS='1', GADFIT_HIP_RANK='0', GADFIT_HIP_IDFILE=str(tmp_path / 'rccl_id'))
    p = subprocess.run([os.path.join(BUILD, 'fit_two_curves'), os.path.join(GOLD, 'curve1_xy.txt'),
                        os.path.join(GOLD, 'curve2_xy.txt')], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    assert not (tmp_path / 'rccl_id').exists()       # (removed once the communicator stands: the next run may name the same path)


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
def test_fortran_eval_with_plain_real_arithmetic_on_x(images):
    """eval() computes x**2 and sin(0.05*x) in real(kp) arithmetic: the recorder tabulates them as auxiliary
    per-point columns (gfh_set_aux).  The Fortran fit equals the Python-API fit of the same model, where the
    arithmetic on the symbolic x is recorded and evaluated on the device.  images = 3: the same program on a
    single-process device group (every member takes its range of the auxiliary columns)."""
    import numpy as np
    from gadfit_amd import _lib
    from gadfit_amd.ad import trace_model, exp, sin
    _build()
    path = os.path.join(GOLD, 'gaussian_xy.txt')
    env = dict(os.environ) if images == 1 else dict(os.environ, GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    p = subprocess.run([os.path.join(BUILD, 'fit_real_x_functions'), path], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
    got = np.array([float(l.split('=')[1]) for l in p.stdout.splitlines() if l.startswith('par ')])
    assert 'iterations = 4' in p.stdout and got.size == 6
    # gadf_print after the fit: curve on 11 points over the data range, parameter table, log with the device timings
    curve = np.loadtxt('/tmp/gadfit_real_x_print')
    assert curve.shape == (11, 2) and curve[0, 0] == -100.0 and curve[-1, 0] == 100.0
    assert 'fmax' in open('/tmp/gadfit_real_x_print_parameters').read() and 'Jacobian' in open('/tmp/gadfit_real_x_print_log').read()

    def model(q, x):
        return q[0] * exp(-((x - q[1]) / q[2]) ** 2) + q[3] + q[4] * (x ** 2 * 1.0e-4) + q[5] * sin(0.05 * x)
    xy = np.loadtxt(path)
    t = trace_model(model, 6)
    c = _lib.Context(0)
    try:
        c.set_model(t)
        c.set_data(xy[:, 0], xy[:, 1], np.ones(xy.shape[0]), [0, xy.shape[0]])
        start = np.array([[1.0, 1e-12, 1.0, 1.0, float(np.float32(0.1)), float(np.float32(0.1))]])
        out, r = c.fit(start, [0, 2, 3, 4, 5], [0] * 6, lambda_=float(np.float32(0.1)), accth=float(np.float32(0.9)), max_iter=4)
    finally:
        c.close()
    assert r.iterations == 4
    assert np.max(np.abs(got - out[0]) / np.maximum(np.abs(out[0]), 1e-300)) < 1e-9, (got, out[0])


@needs_flang
@pytest.mark.gpu
def test_fortran_literal_that_looks_constant_at_the_probes_is_verified_against_the_data():
    """A narrow bump computed in plain real(kp) arithmetic is 0 at the three abscissas capture_model probes; the verification
    pass over the data (verify_capture) promotes it to a per-point column, so the fit returns the bump's amplitude instead of
    silently fitting a constant (tests/fortran/fit_narrow_bump.F90)."""
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_narrow_bump')], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
def test_fortran_gadf_print_curves_without_gpu(tmp_path):
    """gadf_print (gadfit.F90:1255-1395) before any fit: curves of two datasets on a grid, one file with a column
    per dataset or one file per dataset (grouped=.false.), linear and logarithmic spacing; evaluated on the host,
    so it runs without a GPU."""
    import numpy as np
    _build()
    prefix = str(tmp_path / 'pc')
    p = subprocess.run([os.path.join(BUILD, 'print_curves'), prefix], env=dict(os.environ, GADFIT_HIP_DEVICE='-1'),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr

    def f(x, i0, bgr):
        return i0 * np.exp(-x / 4.0) + bgr
    a = np.loadtxt(prefix + '_a')
    x = np.linspace(0.0, 10.0, 5)
    assert a.shape == (5, 3) and np.allclose(a[:, 0], x, rtol=0, atol=1e-15)
    assert np.allclose(a[:, 1], f(x, 5.0, 1.0), rtol=1e-15) and np.allclose(a[:, 2], f(x, 7.0, 2.0), rtol=1e-15)
    b1, b2 = np.loadtxt(prefix + '_b1'), np.loadtxt(prefix + '_b2')
    assert np.array_equal(b1, a[:, [0, 1]]) and np.array_equal(b2, a[:, [0, 2]])
    c = np.loadtxt(prefix + '_c')
    assert np.allclose(c[:, 0], [1.0, 10.0, 100.0], rtol=1e-14) and np.allclose(c[:, 1], f(c[:, 0], 5.0, 1.0), rtol=1e-15)
    assert not os.path.exists(prefix + '_a_parameters')          # no fit has run
    # fitfunc%grad_finite / dir_deriv_2nd_finite / info / destroy used directly (fitfunction.F90:155-231)
    fd = [float(v) for v in [l for l in p.stdout.splitlines() if l.startswith('fd ')][0].split()[1:]]
    e = np.exp(-0.5)
    assert np.allclose(fd[:3], [e, 5.0 * e * 2.0 / 16.0, 1.0], rtol=1e-6)
    assert abs(fd[3] - 5.0 * e * (4.0 / 256.0 - 4.0 / 64.0)) < 1e-6
    assert 'Passive  tau  4.' in p.stdout


@needs_flang
def test_fortran_host_forward_mode_equals_oracle():
    """Module ad computes (val, d, dd) of active operands on the host with the reference's forward-mode formulas
    (AD:454-1459): one expression over every elemental, against the oracle's forward mode of the same tape."""
    import numpy as np
    from gadfit_amd import ad as A
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    _build()
    p = subprocess.run([os.path.join(BUILD, 'forward_mode')], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.array([float(v) for v in [l for l in p.stdout.splitlines() if l.startswith('fwd ')][0].split()[1:4]])
    assert [l for l in p.stdout.splitlines() if l.startswith('pas ')][0].split()[2:] == ['0.00000000000000000E+00'] * 2 + ['0']

    def model(q, x):
        a, b, c = q
        return (A.sin(a * b) / A.sqrt(b) + A.exp(-a) * A.log(b) + a ** b + b ** 3 + 2.0 ** a + a ** 1.5 + A.atan(a / b) + A.tanh(a)
                + A.erf(b) + abs(-a) + A.cos(a + c) + A.tan(0.3 * a) + A.asin(a / 3.0) + A.acos(b / 3.0) + A.sinh(a - b)
                + A.cosh(b * c) + A.asinh(a) + A.acosh(b + 1.0) + A.atanh(a / 4.0) + (a + 2.0) / (b - 0.5) + 3.0 / a - b / 2.0 + c ** a)
    t = trace_model(model, 3)
    want = orc.eval_forward(t, 0.0, [1.3, 2.1, 0.8], [1, 1, 0], [0.7, -0.4, 0.0], [0.2, 0.1, 0.0])     # activity flags and seeds per parameter
    assert np.all(np.abs(got - want) <= 1e-13 * np.maximum(1.0, np.abs(want))), (got, want)


@needs_flang
def test_fortran_host_reverse_mode_equals_oracle():
    """Module ad keeps the reference's host-side reverse mode under its public names (ad_init_reverse, forward_values,
    index_count, ad_grad, adjoints, trace, ad_memory_report, ad_close; AD:233-313, 1476-1690).  The trace sizes derived
    from a memory string are the reference test's known answers (fortran/tests/ad_reverse_mode.F90:9-21); value and
    gradient of one expression over every elemental are compared with the oracle's reverse tape, which is pinned to
    the reference's ad_reverse_mode goldens (tests/test_oracle_goldens.py)."""
    import numpy as np
    from gadfit_amd import ad as A
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    _build()
    p = subprocess.run([os.path.join(BUILD, 'reverse_mode')], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.splitlines()
    assert [int(l.split()[1]) for l in lines if l.startswith('trace_size ')] == [4, 108, 108, 1108]
    assert [l for l in lines if l.startswith('sizes ')][0].split()[1:] == ['1000', '1000', '100', 'T']

    def model(q, x):
        a, b, c = q
        return (A.sin(a * b) / A.sqrt(b) + A.exp(-a) * A.log(b) + a ** b + b ** 3 + 2.0 ** a + a ** 1.5 + A.atan(a / b) + A.tanh(a)
                + A.erf(b) + abs(-a) + A.cos(a + c) + A.tan(0.3 * a) + A.asin(a / 3.0) + A.acos(b / 3.0) + A.sinh(a - b)
                + A.cosh(b * c) + A.asinh(a) + A.acosh(b + 1.0) + A.atanh(a / 4.0) + (a + 2.0) / (b - 0.5) + 3.0 / a - b / 2.0 + c ** a
                + 2 * a - b * 3 + a / c + c / b)
    t = trace_model(model, 3)
    val, grad = orc.eval_reverse(t, 0.0, [1.3, 2.1, 0.8], [1, 1, 0])      # activity flags per parameter
    revs = [[float(v) for v in l.split()[1:4]] + [int(v) for v in l.split()[4:]] for l in lines if l.startswith('rev ')]
    assert len(revs) == 2 and revs[0] == revs[1]                    # ad_grad reset the counters: the second evaluation repeats the first
    assert revs[0][3:] == [2, 0, 0]                                 # index_count = num_pars, trace and constants rewound (AD:1658)
    got = np.array(revs[0][:3]); want = np.array([val, grad[0], grad[1]])
    assert np.all(np.abs(got - want) <= 1e-13 * np.maximum(1.0, np.abs(want))), (got, want)
    vb, gb = orc.eval_reverse(t, 0.0, [1.3, 2.1, 0.8], [0, 1, 0])
    revb = [float(v) for v in [l for l in lines if l.startswith('revb ')][0].split()[1:]]
    assert abs(revb[0] - vb) <= 1e-13 * abs(vb) and abs(revb[1] - gb[0]) <= 1e-13 * abs(gb[0])
    assert [l for l in lines if l.startswith('pas ')][0].split()[2:] == ['0', '0', '0']
    assert 'AD memory usage' in p.stdout and '(2x73)' in p.stdout


@needs_flang
@pytest.mark.gpu
def test_fortran_headline_workload_end_to_end():
    """The 32-parameter headline model through the Fortran API (gadf_init ... gadf_fit) at a reduced size: captured from a
    loop over eight peaks in eval(), fitted back to the generating parameters; the program prints the host-clock phases."""
    _build()
    p = subprocess.run([os.path.join(BUILD, 'bench_headline'), '200000', '8'], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
    assert 'iterations = 8' in p.stdout and 'ms per LM iteration' in p.stdout


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
@pytest.mark.parametrize('prog,data', [('fit_piecewise', 'piecewise_aux_xys.txt'), ('fit_hidden_branch', 'piecewise2_xys.txt'),
                                       ('fit_clip_unseen', 'clip_xys.txt'), ('fit_integral_branch', 'integral_branch_xys.txt'),
                                       ('fit_kinked_integrand', 'kinked_integrand_xys.txt')])
def test_fortran_branching_eval(prog, data, images):
    """eval() bodies that branch -- on a comparison of x with a fitted parameter (plus an auxiliary column on one branch), on the
    plain real x, through two comparisons of AD variables one of whose outcomes is first met inside gadf_fit, and with an integrate()
    call site on either side of a fitted breakpoint -- land on the
    oracle's fits (tests/golden/make_branching_goldens.py) through the Fortran API; alone and as a device group of three images."""
    _build()
    env = dict(os.environ) if images == 1 else dict(os.environ, GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    p = subprocess.run([os.path.join(BUILD, prog), os.path.join(GOLD, data)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
def test_fortran_integrand_path_first_met_inside_the_fit(images):
    """fit_kinked_integrand far: the kink of the integrand starts beyond every range of integration, so the recordings hold one path
    through the integrand; the fit pulls the kink in, the device reports the unrecorded path (status 2), the layer records the
    integrands again at the parameters of that pass and the pass is repeated -- the oracle's fit with both paths known"""
    _build()
    env = dict(os.environ) if images == 1 else dict(os.environ, GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    p = subprocess.run([os.path.join(BUILD, 'fit_kinked_integrand'), os.path.join(GOLD, 'kinked_integrand_xys.txt'), 'far'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
def test_fortran_branch_the_sampled_recordings_miss(images):
    """400001 points under the sampled capture (GADFIT_HIP_VERIFY=sample: eval() recorded at every 4th abscissa): a window holding two
    points between samples is a path no recording contains -- its bounds are PARAMETERS, so the comparison is the device's to decide:
    it reports the points in the first pass, the layer records them, the fit lands on the oracle's with all paths known
    (tests/golden/make_branching_goldens.py, case rare_branch); alone and as a device group of three images.  Under the default
    capture (every abscissa) the path is found before the first pass: the same fit."""
    _build()
    env = dict(os.environ) if images == 1 else dict(os.environ, GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    for verify in ('sample', 'all'):
        p = subprocess.run([os.path.join(BUILD, 'fit_rare_branch')], capture_output=True, text=True, timeout=600, env=dict(env, GADFIT_HIP_VERIFY=verify))
        assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
def test_fortran_integrand_takes_x_from_a_module_variable(images):
    """the function handed to integrate() reads the data point's abscissa from a module variable that eval() sets -- past pars(:),
    which the reference allows (its integrand runs in eval()'s scope at every point, NI:195-201): an affine and a non-affine real
    function of x inside the integrand's sub-tape, as X-node expression and as tabulated per-point column; the fit lands on the
    oracle's (tests/golden/make_branching_goldens.py, case integrand_module_x); alone and as a device group of three images"""
    _build()
    env = dict(os.environ) if images == 1 else dict(os.environ, GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    p = subprocess.run([os.path.join(BUILD, 'fit_integrand_module_x')], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    if images == 1:
        # 40000 points: the per-point column of a path that calls integrate() is tabulated on threads (the call sites' bookkeeping
        # in thread-local storage) -- the same column as the serial tabulation, so the same fit to the last bit
        outs = []
        for threads in ('16', '1'):
            p = subprocess.run([os.path.join(BUILD, 'fit_integrand_module_x'), '40000'], capture_output=True, text=True, timeout=600,
                               env=dict(env, GADFIT_HIP_RECORD_THREADS=threads, GADFIT_HIP_SETUP_TIMES='3', GADFIT_HIP_THREADS_FROM='4096'))
            assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
            outs.append(([l.split('rel. dev.')[0] for l in p.stdout.splitlines() if l.startswith('par ')], p.stderr))
        assert len(outs[0][0]) == 3 and outs[0][0] == outs[1][0]
        assert 'threaded tabulation' in outs[0][1] and 'threaded tabulation' not in outs[1][1]


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
def test_fortran_real_formed_from_the_val_of_a_fitted_parameter(images):
    """s = sin(tau%val) in plain real arithmetic inside eval(), tau fitted: the reference recomputes it whenever eval() runs and
    differentiates around it; here a passive pseudo-parameter that the layer recomputes on the host before every pass
    (gfh_set_pars_hook).  The fit lands on the oracle's fit of the same model written with value() = GFH_VAL (case param_val); alone
    and as a device group of three images"""
    _build()
    env = dict(os.environ) if images == 1 else dict(os.environ, GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    p = subprocess.run([os.path.join(BUILD, 'fit_param_val')], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_val_of_a_parameter_inside_an_integrand():
    """tests/fortran/fit_integrand_param_val.F90 (round 5): the function handed to integrate() forms sin(pars(2)%val) in plain real
    arithmetic.  The literal becomes one more, passive entry of the integrand's pars(:), bound at the call site to a pseudo-parameter
    the layer recomputes before every pass; the fit lands on the oracle's (case integrand_param_val: value() inside the integrand)."""
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_integrand_param_val')], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
def test_fortran_capture_tells_reals_that_follow_the_parameters_and_the_abscissa():
    """host side of tests/fortran/fit_param_val_x.F90 (compile-only context: the capture runs, the first device call stops): the real
    cos(rate%val*x) is 1 at the data's first abscissa x = 0 whatever the rate -- the parameter probe at the path's SECOND abscissa is
    what finds that it follows the fitted parameters; it becomes a per-point column flagged as such.  Under use_ad=.false. the
    black-box eval() is ONE literal node that follows them."""
    _build()
    exe = os.path.join(BUILD, 'fit_param_val_x')
    env = dict(os.environ, GADFIT_HIP_DEVICE='-1', GADFIT_HIP_TRACE_PATHS='1')
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    err = ' '.join(p.stderr.split())
    assert p.returncode != 0 and 'no GPU bound to this context' in err, p.stderr
    assert 'model: 1 path(s), 1 column(s)' in err and 'pseudo-parameters 0' in err, p.stderr
    assert 'class 3' in err and 'follows the fitted parameters T' in err, p.stderr
    p = subprocess.run([exe, '500', 'blackbox'], capture_output=True, text=True, timeout=600, env=env)
    err = ' '.join(p.stderr.split())
    assert 'model: 1 path(s), 1 column(s)' in err and 'follows the fitted parameters T' in err, p.stderr
    # 'branch': a comparison with a fitted parameter on top -- two paths, the column on both
    p = subprocess.run([exe, '500', 'branch'], capture_output=True, text=True, timeout=600, env=env)
    err = ' '.join(p.stderr.split())
    assert 'model: 2 path(s), 2 column(s)' in err and err.count('follows the fitted parameters T') == 2, p.stderr


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('how', ['serial', 'threads', 'group', 'accel', 'accel_group', 'fd', 'fd_group', 'blackbox', 'branch', 'branch_group'])
def test_fortran_val_of_a_parameter_together_with_the_abscissa(how):
    """tests/fortran/fit_param_val_x.F90 (round 5): eval() forms cos(rate%val * x) in plain real arithmetic -- a real with another value
    at every point AND every pass.  The reference recomputes it whenever eval() runs (gadfit.F90:679-690); here it is a per-point
    column that the layer tabulates anew (eval() at every data point) before every pass at new parameters (on_pars).  500 points: the
    fit lands on the oracle's (case param_val_x, value() = GFH_VAL) to 1e-10; 40000 points with the columns read off threaded
    recordings; a device group of three images (each member's hook finds the table of the pass and uploads its share)."""
    _build()
    env = dict(os.environ, GADFIT_HIP_SETUP_TIMES='1')
    args = []
    if how == 'threads':
        env.update(GADFIT_HIP_THREADS_FROM='16384'); args = ['40000']
    if how.endswith('group'):
        env.update(GADFIT_HIP_DEVICES='3', GADFIT_HIP_GROUP_WRAP='1')
    if how.startswith('accel'):       # (geodesic acceleration: STEP 3 at the parameters of the sweep, after a trial chi2() elsewhere)
        args = ['500', 'accel']
    # use_ad=.false.: the forward differences evaluate eval() at p + step e_j, where the real has moved -- 1 + n_active sets of the
    # columns, one per evaluation; 'blackbox': eval() entirely in plain real arithmetic on %val (what use_ad=.false. exists for)
    if how.startswith('fd'):
        args = ['500', 'fd']
    if how == 'blackbox':
        args = ['500', 'blackbox']
    # eval() also compares x with a fitted parameter (points change sides during the fit): two paths, the column on both
    if how.startswith('branch'):
        args = ['500', 'branch']
    p = subprocess.run([os.path.join(BUILD, 'fit_param_val_x')] + args, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    assert 'per-point column(s) follow the fitted parameters' in p.stderr, p.stderr


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('how', ['intermediate', 'stateful'])
def test_fortran_refreshed_columns_at_2e5_points_are_the_serial_ones(how):
    """Round-5 advisor finding (high): on_pars refreshes the columns that follow the parameters before every pass, and from 1e5
    points on it did so on threads with no check at all.  'intermediate': s = cos(t%val) with t = rate*x an intermediate AD variable
    -- the threaded check skipped the values of its nodes, so every refresh uploaded cos(0) = 1; 'stateful': the real waits in a
    module variable between two statements -- detected and tabulated serially the first time, then called concurrently again on
    every refresh.  Both must give, at 2e5 points on 16 threads, the bits of the fit made with GADFIT_HIP_RECORD_THREADS=1 (eval()
    called from one thread, as the reference calls it), and at 500 points the oracle's fit of the same numbers (case param_val_x)."""
    _build()
    exe = os.path.join(BUILD, 'fit_param_val_x')
    p = subprocess.run([exe, '500', how], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    outs = []
    for threads in ('16', '1'):
        p = subprocess.run([exe, '200000', how], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_RECORD_THREADS=threads, OMP_NUM_THREADS='16', GADFIT_HIP_SETUP_TIMES='1'))
        assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
        assert 'per-point column(s) follow the fitted parameters' in p.stderr, p.stderr
        outs.append(([l for l in p.stdout.splitlines() if l.startswith('par ')], p.stderr))
    assert len(outs[0][0]) == 3 and outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    if how == 'stateful':       # where the race shows it is reported ONCE (the verdict is remembered: every refresh after it runs on one thread)
        assert outs[0][1].count('several threads') <= 1, outs[0][1]


@needs_flang
@pytest.mark.gpu
def test_fortran_plain_real_window_narrower_than_any_sample():
    """a window three points wide of 400001 whose bounds are plain reals of eval()'s module: no comparison of an AD variable for the
    device to decide, no sampled abscissa inside.  The reference sees every point (gadfit.F90:679-690); so does gadf_fit's capture by
    default, and the fit lands on the oracle's (case narrow_window).  Under GADFIT_HIP_VERIFY=sample, the capture of rounds 1-3, the
    model has no node for the step inside the window: its Jacobian column is zero and the fit cannot be made -- what the default is
    there to prevent."""
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_narrow_window')], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    p = subprocess.run([os.path.join(BUILD, 'fit_narrow_window')], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GADFIT_HIP_VERIFY='sample'))
    assert p.returncode != 0 and 'Cholesky' in p.stderr, p.stdout + p.stderr
    # ... and the same choice made in the program's source (round 6): gadf_init(f, record_every_abscissa=.false.)
    p = subprocess.run([os.path.join(BUILD, 'fit_narrow_window'), 'sample'], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'Cholesky' in p.stderr, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_per_point_columns_tabulated_on_threads_equal_the_serial_ones():
    """real(kp) arithmetic on x inside eval() (sin(0.05 x)**2, x**2) becomes per-point columns that gadf_fit tabulates over all data
    points: on several OpenMP threads when eval() is one straight-line path (the values read off recordings made in checking mode),
    serially otherwise -- the same columns, so the same fit to the last bit"""
    _build()
    # bench_hidden_branch: eval() branches on the plain real x -- the column of per-point paths and the column of a real factor on
    # one side, both read off the threads' recordings (two known paths side by side)
    # bench_guard_aux: a comparison with a fitted parameter and a real factor on one side -- every point recorded along its own path
    # and, with the comparison forced (thread-local script), along the other
    for prog, npar in (('bench_real_x', 5), ('bench_hidden_branch', 3), ('bench_guard_aux', 4)):
        outs = []
        for threads in ('8', '1'):
            p = subprocess.run([os.path.join(BUILD, prog), '100000', '6'], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, GADFIT_HIP_RECORD_THREADS=threads))
            assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
            outs.append([l for l in p.stdout.splitlines() if l.startswith('par ')])
        assert len(outs[0]) == npar and outs[0] == outs[1], prog


@needs_flang
@pytest.mark.gpu
def test_fortran_later_fits_under_load_balancing_see_the_abscissas():
    """load_balancing = .true. (here through the environment) takes gfh_set_data_begin through gfh_set_data: the layer's own copy of
    the abscissas, queued beside the upload, must be made on that path too -- a second gadf_fit tabulates the per-point columns from
    it.  Three fits with and without load balancing end at the same parameters."""
    _build()
    outs = []
    for lb in ('1', '0'):
        p = subprocess.run([os.path.join(BUILD, 'bench_real_x'), '50000', '6', '3'], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_LOAD_BALANCING=lb))
        assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
        outs.append([float(l.split('=')[1]) for l in p.stdout.splitlines() if l.startswith('par ')])
    assert len(outs[0]) == 5 and np.allclose(outs[0], outs[1], rtol=1e-9, atol=0)


@needs_flang
@pytest.mark.gpu
def test_fortran_eval_that_is_not_thread_safe_is_noticed():
    """an eval() that parks an intermediate in a module variable: called from several threads its per-point column changes from one
    pass of the tabulation to the next; the layer warns, records serially, and the fit is the serial one to the bit"""
    _build()
    outs = []
    for threads in ('16', '1'):
        p = subprocess.run([os.path.join(BUILD, 'fit_stateful_eval'), '200000'], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_RECORD_THREADS=threads))
        assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
        outs.append(([l for l in p.stdout.splitlines() if l.startswith('par ')], p.stderr))
    assert len(outs[0][0]) == 4 and outs[0][0] == outs[1][0]
    assert 'several threads' not in outs[1][1]


@needs_flang
@pytest.mark.gpu
def test_fortran_gadf_init_keyword_calls_eval_from_one_thread():
    """gadf_init(f, eval_is_thread_safe=.false., force_outcomes=.false.) (round 6): the reference-faithful capture stated in the
    program's source instead of its environment -- eval() is called from one thread even where the environment asks for 16, no
    threaded attempt, no warning, the bits of the GADFIT_HIP_RECORD_THREADS=1 fit"""
    _build()
    exe = os.path.join(BUILD, 'fit_stateful_eval')
    ref = subprocess.run([exe, '200000'], capture_output=True, text=True, timeout=600, env=dict(os.environ, GADFIT_HIP_RECORD_THREADS='1'))
    assert ref.returncode == 0 and 'PASS' in ref.stdout, ref.stdout + ref.stderr
    p = subprocess.run([exe, '200000', 'keyword'], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GADFIT_HIP_RECORD_THREADS='16', OMP_NUM_THREADS='16', GADFIT_HIP_SETUP_TIMES='3'))
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr
    assert 'several threads' not in p.stderr and 'threaded tabulation' not in p.stderr, p.stderr
    pars = lambda out: [l for l in out.splitlines() if l.startswith('par ')]
    assert pars(p.stdout) == pars(ref.stdout) and len(pars(p.stdout)) == 4


@needs_flang
@pytest.mark.gpu
def test_fortran_arrays_are_copied_when_they_are_added_whatever_their_size():
    """gadf_add_dataset(x, y) then x = -1; y = 0 then gadf_fit: the fit is of the contents at the call, on either side of the 2^20
    points where rounds 4-5 changed from copying to borrowing (round-4 advisor finding)"""
    _build()
    for n in (2 ** 20 - 1, 2 ** 20 + 1):
        p = subprocess.run([os.path.join(BUILD, 'fit_mutated_arrays'), str(n)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and 'PASS' in p.stdout, (n, p.stdout + p.stderr)


@needs_flang
@pytest.mark.gpu
def test_fortran_racy_memo_cache_in_eval_never_gives_a_wrong_fit():
    """an eval() with a memo cache in module variables (correct under the reference's one-image-at-a-time calls, racy under the
    layer's threads, and only sporadically so): 100 fits on 16 threads, each either notices -- the second threaded pass or the
    serial re-verification (64 fixed points + 1 % drawn afresh per fit) disagrees: warning, serial tabulation -- or is right anyway:
    every one prints the bits of the fit made with GADFIT_HIP_RECORD_THREADS=1, the reference-faithful setting"""
    _build()
    serial = subprocess.run([os.path.join(BUILD, 'fit_racy_cache'), '20000', '1'], capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, GADFIT_HIP_RECORD_THREADS='1'))
    assert serial.returncode == 0 and 'PASS' in serial.stdout and 'several threads' not in serial.stderr, serial.stdout + serial.stderr
    want = [l.split()[2:] for l in serial.stdout.splitlines() if l.startswith('cycle ')]
    assert len(want) == 1 and len(want[0]) == 4
    p = subprocess.run([os.path.join(BUILD, 'fit_racy_cache'), '20000', '100'], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GADFIT_HIP_RECORD_THREADS='16', OMP_NUM_THREADS='16', GADFIT_HIP_THREADS_FROM='4096'))
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    got = [l.split()[2:] for l in p.stdout.splitlines() if l.startswith('cycle ')]
    assert len(got) == 100 and all(g == want[0] for g in got), [k for k, g in enumerate(got) if g != want[0]]


@needs_flang
@pytest.mark.gpu
def test_fortran_val_of_a_passive_parameter_in_real_arithmetic():
    """eval() reads %val of a PASSIVE parameter (an integer exponent): a constant of the captured model, which gadf_fit captures again
    when the passive value has changed between two fits on the same gadf_init"""
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_passive_val')], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'PASS' in p.stdout, p.stdout + p.stderr


@needs_flang
def test_fortran_branching_eval_is_captured_without_gpu():
    """the recordings over the data, the variants and (for the plain-real branch) the need for the per-point column are all host
    work: a compile-only context accepts the model and only the first device call stops"""
    _build()
    for prog, data in [('fit_piecewise', 'piecewise_aux_xys.txt'), ('fit_hidden_branch', 'piecewise2_xys.txt'), ('fit_clip_unseen', 'clip_xys.txt'),
                       ('fit_integral_branch', 'integral_branch_xys.txt'), ('fit_kinked_integrand', 'kinked_integrand_xys.txt')]:
        p = subprocess.run([os.path.join(BUILD, prog), os.path.join(GOLD, data)], env=dict(os.environ, GADFIT_HIP_DEVICE='-1'),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and 'no GPU bound to this context' in p.stderr, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_workspace_size_is_the_users():
    """gadf_init(ws_size=...) reaches the device: the default (1000) and 500 fit an integrand that needs 300-500 intervals and agree;
    300 and 50 stop with the reference's message (numerical_integration.F90:282-283), where the reference would"""
    _build()
    exe = os.path.join(BUILD, 'ws_size')
    chi = []
    for ws in ('0', '500'):
        p = subprocess.run([exe, ws], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
        chi.append(float([l for l in p.stdout.splitlines() if l.startswith('chi2')][0].split('=')[1]))
    assert chi[0] == chi[1] and chi[0] > 0
    for ws in ('300', '50'):
        p = subprocess.run([exe, ws], capture_output=True, text=True, timeout=600)
        assert p.returncode != 0 and 'Number of iterations was insufficient' in p.stderr, p.stdout + p.stderr


@needs_flang
def test_fortran_literals_the_recorder_cannot_capture_stop_loudly():
    """tests/fortran/refused_literals.F90: a real number formed from the %val of an integration variable (over a range that follows x,
    and over a fixed range, where only a second recording with the variable elsewhere shows it), or -- inside an integrand -- of a
    fitted parameter TOGETHER with the abscissa, cannot follow its source on the device; model capture (host code: runs on a
    compile-only context too) stops and names it.  (A real formed from a fitted parameter's %val alone is carried since round 4:
    fit_param_val.F90; together with x, in eval() itself, since round 5: fit_param_val_x.F90 -- not under use_ad=.false.)"""
    _build()
    exe = os.path.join(BUILD, 'refused_literals')
    env = dict(os.environ) if os.path.exists('/dev/kfd') else dict(os.environ, GADFIT_HIP_DEVICE='-1')
    for mode, what in (('tval', 'integration variable'), ('tfix', 'value of its integration variable (%val)'),
                       ('ipvx', 'An integrand forms a real number from parameter values (%val) AND the abscissa'),
                       ('fdival', 'use_ad=.false. with a real number that an integrand forms from the %val of a fitted parameter'),
                       ('fdacc', 'use_ad=.false. with geodesic acceleration')):
        p = subprocess.run([exe, mode], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode != 0 and what in ' '.join(p.stderr.split()), mode + ': ' + p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_same_models_in_advar_arithmetic_fit():
    _build()
    p = subprocess.run([os.path.join(BUILD, 'refused_literals'), 'good'], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr


@needs_flang
@pytest.mark.gpu
def test_fortran_a_constant_of_eval_changed_between_two_fits_takes_effect():
    """tests/fortran/fit_changed_constant.F90: eval() reads a module variable that the program changes between two gadf_fit calls; the
    reference would simply evaluate the new function (gadfit.F90:679-690), so the captured model must notice and be captured again"""
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_changed_constant')], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
    fit = {l.split(':')[0]: [float(v) for v in l.split(':')[1].split()] for l in p.stdout.splitlines() if l.startswith('fit ')}
    assert abs(fit['fit 1'][0] - 3.0) < 1e-8 and abs(fit['fit 1'][1] - 0.5) < 1e-8
    assert abs(fit['fit 2'][0] - 3.0) < 1e-8 and abs(fit['fit 2'][1] - 0.25) < 1e-8


@needs_flang
@pytest.mark.gpu
def test_fortran_global_fit_of_many_curves():
    """tests/fortran/bench_global.F90 (BASELINE config 3's shape: amplitudes and background per curve, three decay times shared) at a
    small size: the datasets are laid side by side on several threads, the global fit finds the shared decay times"""
    _build()
    out = []
    for extra in ([], ['files']):                      # from arrays; from text files handed over by path (read side by side)
        p = subprocess.run([os.path.join(BUILD, 'bench_global'), '8', '3000', '30'] + extra, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
        out.append([l for l in p.stdout.splitlines() if l.startswith('tau')][0])
    assert out[0] == out[1]


@needs_flang
@pytest.mark.gpu
def test_fortran_batch_of_small_fits_in_one_process():
    """tests/fortran/bench_many_small_fits.F90: gadf_init ... gadf_close per spectrum, several in one process -- each context adopts
    what the one before it left behind (stream, events, pinned buffers, small device blocks); the fits must not notice, with the
    pool and without it (GADFIT_HIP_POOL=0)"""
    _build()
    for pool in ('1', '0'):
        p = subprocess.run([os.path.join(BUILD, 'bench_many_small_fits'), '6', '1500'], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_POOL=pool))
        assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr


@needs_flang
def test_fortran_integrate_outside_gadf_fit_runs_on_the_host():
    """gadf_print draws the fitted curve by calling eval() itself, and a program may do the same after the fit (the reference's
    2_integral_single / 3_integral_double do the former): outside a recording integrate() is the reference's adaptive
    Gauss-Kronrod rule on the host through module ad's arithmetic (numerical_integration.F90, host_integral) -- finite and infinite
    ranges, reversed bounds, a nested integral, another rule, forward-mode derivatives through the integrand and through an
    active bound -- against closed forms.  No GPU involved: gadf_fit's own passes never take this route."""
    import numpy as np
    from scipy import integrate, special
    _build()
    r = subprocess.run([os.path.join(BUILD, 'host_integrate')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'DONE' in r.stdout, r.stdout + r.stderr
    got = {ln.split()[0]: float(ln.split()[1]) for ln in r.stdout.splitlines() if len(ln.split()) == 2}
    p = 1.3
    pg = integrate.quad(lambda t: t**1.5 * np.exp(-0.7 * t * t), 0, 2, epsabs=0, epsrel=1e-13)[0]
    want = {'power_gauss_0_2': pg, 'power_gauss_41': pg, 'decay_0_inf': 1 / p, 'decay_1_inf': np.exp(-p) / p,
            'bell_inf_inf': np.sqrt(np.pi / p), 'bell_inf_half': np.sqrt(np.pi / p) * 0.5 * (1 + special.erf(np.sqrt(p) * 0.5)),
            'bell_reversed': -np.sqrt(np.pi / p) * 0.5 * special.erfc(np.sqrt(p) * 0.5), 'nested': (2 - (1 - np.exp(-2 * p)) / p) / p,
            'forward_value': (1 - np.exp(-p)) / p, 'forward_d': (np.exp(-p) * p - (1 - np.exp(-p))) / p**2,
            'bound_value': (1 - np.exp(-0.8 * p)) / p, 'bound_d': np.exp(-0.8 * p), 'lower_bound_d': -np.exp(-0.2 * p)}
    for k, v in want.items():
        assert abs(got[k] - v) <= 2e-12 * abs(v), (k, got[k], v)         # observed <= 1e-15; the loosest bound asked of the rule is 1e-12


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize('cross', ['1', '0'])
def test_fortran_branch_on_plain_x_hidden_behind_a_comparison_of_ad_variables(cross):
    """tests/fortran/fit_fork_behind_guard.F90: `if (x < brk) then; if (x < 20) A else B; else C` with brk fitted from 17 to 27.3.  At
    the capture no point with x >= 20 has x < brk, so the data only show A behind the comparison; the capture's cross-check forces
    the recorded outcomes on eval() at every point, meets B, and the fit equals the one through the Python API (which records both
    comparisons).  GADFIT_HIP_CROSS_CHECK=0 shows what it prevents: the points that enter x < brk beyond 20 are sent into A and the
    fit ends elsewhere -- silently."""
    import numpy as np
    from gadfit_amd import _lib, tape as T
    from gadfit_amd.ad import exp
    _build()
    p = subprocess.run([os.path.join(BUILD, 'fit_fork_behind_guard'), '2000'], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GADFIT_HIP_CROSS_CHECK=cross))
    assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
    got = np.array([float(l.split('=')[1]) for l in p.stdout.splitlines() if l.startswith('par ')])

    def model(q, x):
        if x < q[3]:
            if x < 20.0:
                return q[0] + q[1] * (x - 20.0)
            return q[0] + 3.0 * q[1] * (x - 20.0)
        return (q[0] + 3.0 * q[1] * (q[3] - 20.0)) * exp(-((x - q[3]) / q[2]))
    n = 2000
    i = np.arange(1, n + 1, dtype=np.float64)
    xs = 60.0 * (i - 0.5) / n
    t = np.where(xs < 27.3, 4.0 + np.where(xs < 20.0, 0.08, 0.24) * (xs - 20.0), (4.0 + 0.24 * 7.3) * np.exp(-(xs - 27.3) / 11.0))
    ys = t * (1.0 + 0.01 * np.sin(12.9898 * i))
    start = np.array([[4.2, 0.07, 10.0, 17.0]])
    V = T.Variants(model, 4)
    V.explore(xs, start[0])
    c = _lib.Context(0)
    try:
        c.set_model(V)
        c.set_data(xs, ys, np.ones(n), [0, n])
        out, r = c.fit(start, [0, 1, 2, 3], [0] * 4, lambda_=1.0, max_iter=12)
    finally:
        c.close()
    dev = np.max(np.abs(got - out[0]) / np.abs(out[0]))
    if cross == '1':
        assert dev < 1e-7, (got, out[0])           # (the data are formed with libm on one side, numpy on the other: 1e-16 apart, 12 iterations on)
        assert abs(out[0, 3] - 27.3) < 0.5         # ... and the breakpoint has moved past the hidden fork
    else:
        assert dev > 1e-4, (got, out[0])


@needs_flang
@pytest.mark.gpu
def test_fortran_points_that_change_sides_in_front_of_a_fork_need_no_host():
    """the same program, 200000 points: the fitted breakpoint moves from 17 to 27.3 and thousands of points cross `x < brk` during the
    fit, in front of the plain-real fork `x < 20`.  With one variant column per set of outcomes (round 5: the path each point takes
    when the comparisons are GIVEN is a function of x alone, which cross_check has seen at every point) the device finds their leaves by
    itself: no report reaches the host after the capture.  GADFIT_HIP_HINT_SETS=0 is round 4's scheme -- every such pass is reported,
    the column tabulated anew -- and ends on the same bits."""
    _build()
    exe = os.path.join(BUILD, 'fit_fork_behind_guard')
    outs = {}
    for sets in ('1', '0'):
        p = subprocess.run([exe, '200000'], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_HINT_SETS=sets, GADFIT_HIP_TRACE_PATHS='1'))
        assert p.returncode == 0 and 'DONE' in p.stdout, p.stdout + p.stderr
        outs[sets] = ([l for l in p.stdout.splitlines() if l.startswith('par ') or l.startswith('chi2')], p.stderr.count('on_unseen:'))
    assert outs['1'][0] == outs['0'][0] and len(outs['1'][0]) >= 4
    assert outs['1'][1] == 0 and outs['0'][1] >= 1, (outs['1'][1], outs['0'][1])


@needs_flang
def test_fortran_capture_meets_the_branch_hidden_behind_a_comparison_without_a_gpu():
    """the capture half of the test above on a compile-only context (GADFIT_HIP_DEVICE=-1; host code): with the cross-check the model
    holds three paths -- the third met only by forcing "x < brk" on eval() at abscissas beyond 20 -- and the per-point variant column;
    without it two paths and no column (GADFIT_HIP_TRACE_PATHS prints what the capture holds)"""
    _build()
    exe = os.path.join(BUILD, 'fit_fork_behind_guard')
    for cross, want in (('1', 'model: 3 path(s), 0 column(s), hint column 0'), ('0', 'model: 2 path(s), 0 column(s), hint column -1')):
        p = subprocess.run([exe, '2000'], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_DEVICE='-1', GADFIT_HIP_TRACE_PATHS='1', GADFIT_HIP_CROSS_CHECK=cross))
        assert want in p.stderr, p.stdout + p.stderr
        if cross == '1':
            assert 'first x  2.24850E+01, comparisons 1 outcomes T' in p.stderr, p.stderr


@needs_flang
def test_fortran_capture_of_a_local_parameters_val_literal_without_a_gpu(tmp_path):
    """seed 1110 of the layout kind of tests/fortran_fuzz.py (two datasets, every parameter local with another value per dataset, eval()
    forms sin(pars(k)%val)): the capture used to take the difference between the datasets for a dependence on x and refuse the
    program; it now runs through to the first device call (compile-only context: host code)."""
    import numpy as np
    from tests import fortran_fuzz as FZ
    _build()
    c = FZ.make_layout_case(1110)
    assert c['nd'] == 2 and not any(c['is_global']) and '%val' in c['root'].f90
    files = []
    for d in range(c['nd']):
        x = np.linspace(0.3, 1.6, 120)
        path = tmp_path / ('d%d.txt' % d)
        cols = [x, 1.0 + 0.1 * x] + ([np.ones_like(x)] if c['mode'] == 'USER' else [])
        np.savetxt(path, np.column_stack(cols), fmt='%.17e')
        files.append(str(path))
    src = tmp_path / 'case.F90'
    src.write_text(FZ.fortran_source_layout(c))
    exe = tmp_path / 'case'
    mods = os.path.join(ROOT, 'gadfit_amd', 'fortran', 'build')
    libdir = os.path.join(ROOT, 'gadfit_amd', 'lib')
    fc = shutil.which('amdflang') or '/opt/rocm/bin/amdflang'
    subprocess.run([fc, '-O2', '-cpp', '-fopenmp', '-I', mods, '-module-dir', str(tmp_path), str(src), os.path.join(mods, 'libgadfit_f.a'),
                    '-L' + libdir, '-lgadfit_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath,/opt/rocm/lib/llvm/lib',
                    '-o', str(exe)], check=True, capture_output=True, timeout=600)
    p = subprocess.run([str(exe)] + files, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GADFIT_HIP_DEVICE='-1', GADFIT_HIP_TRACE_PATHS='1'))
    out = p.stdout + p.stderr
    assert 'AND the abscissa' not in out and 'no GPU bound to this context' in out, out
    assert 'pseudo-parameters 1' in out or 'pseudo-parameters 2' in out, out       # (the literal follows the parameters: a pseudo-parameter)
