"""The reference's OWN Fortran programs, unchanged, against this repository's modules (gadfit_amd/fortran): what "drop-in" means
for the user-facing API.  Compiled from where they lie under /root/reference/fortran/tests into a scratch directory -- nothing of
the reference is copied into the repository, and the tests are skipped where the reference is absent (the GPU box).

  * ad_forward_mode.F90, ad_reverse_mode.F90 (with the reference's testing.F90 fixture module): link and RUN -- every known
    answer of the reference's AD tests at the reference's own tolerance (absolute 10 epsilon) from the repository's host-side
    module ad (forward mode, the reverse sweep, comparisons, assignments, safe_deallocate);
  * 1_gaussian, 2_integral_single, 3_integral_double, 4_multiple_curves: semantic analysis (-fsyntax-only) of the user modules
    and main programs against modules ad, fitfunction, gadf_constants, numerical_integration, gadfit -- every name, generic,
    keyword argument and type they use resolves (flang cannot lower their this_image(), with any library);
  * example.F90: link, and run up to the first device call of a context without a GPU;
  * ON THE GPU (-m gpu): the four fit programs themselves, unchanged, as oracle/build_ref_programs.py built them into oracle/_ref/
    in the build container (`-Dthis_image()=1` is the one compile-time mapping): each fits its data on the device and holds the
    result against the constant the reference keeps, at the reference's own tolerance, or `error stop`s; and example.F90 (the user
    guide's worked example, default options), whose fitted parameters are held against the CPU oracle's.
The same fits with the same data also run from tests/fortran/fit_*.F90 (test_fortran_binding.py)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/fortran/tests'
FC = shutil.which('amdflang') or ('/opt/rocm/bin/amdflang' if os.path.exists('/opt/rocm/bin/amdflang') else None)
MODS = os.path.join(ROOT, 'gadfit_amd', 'fortran', 'build')
LIBDIR = os.path.join(ROOT, 'gadfit_amd', 'lib')

needs_reference = [pytest.mark.skipif(not os.path.isdir(REF), reason='the reference is not on this machine'),
                   pytest.mark.skipif(FC is None, reason='no Fortran compiler')]


def _marked(f):
    for m in needs_reference:
        f = m(f)
    return f


@pytest.fixture(scope='module')
def built():
    import sys
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'gadfit_amd', 'build.py')])
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'gadfit_amd', 'fortran', 'build.py')])
    return [os.path.join(MODS, 'libgadfit_f.a'), '-L' + LIBDIR, '-lgadfit_hip', '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,/opt/rocm/lib',
            '-Wl,-rpath,/opt/rocm/lib/llvm/lib']


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    return r.returncode, r.stdout + r.stderr


@_marked
@pytest.mark.parametrize('name', ['ad_forward_mode', 'ad_reverse_mode'])
def test_reference_ad_known_answers_run_unchanged(built, tmp_path, name):
    d = str(tmp_path)
    rc, out = _run([FC, '-O2', '-cpp', '-I', MODS, '-module-dir', d, '-c', os.path.join(REF, 'testing.F90'), '-o', os.path.join(d, 'testing.o')])
    assert rc == 0, out
    exe = os.path.join(d, name)
    rc, out = _run([FC, '-O2', '-cpp', '-fopenmp', '-I', MODS, '-I', d, '-module-dir', d, os.path.join(REF, name + '.F90'),
                    os.path.join(d, 'testing.o')] + built + ['-o', exe])
    assert rc == 0, out
    rc, out = _run([exe])
    assert rc == 0, out            # (the programs `error stop` at the first value off by more than 10 epsilon)


@_marked
@pytest.mark.parametrize('name', ['1_gaussian', '2_integral_single', '3_integral_double', '4_multiple_curves'])
def test_reference_fit_programs_pass_semantic_analysis_unchanged(built, tmp_path, name):
    d = str(tmp_path)
    rc, out = _run([FC, '-O2', '-cpp', '-I', MODS, '-module-dir', d, '-c', os.path.join(REF, name + '_data.F90'), '-o', os.path.join(d, 'data.o')])
    assert rc == 0, out
    rc, out = _run([FC, '-fsyntax-only', '-cpp', '-I', MODS, '-I', d, '-module-dir', d, os.path.join(REF, name + '.F90')])
    assert rc == 0 and 'error' not in out.lower(), out


@_marked
def test_reference_example_links_and_reaches_the_device(built, tmp_path):
    d = str(tmp_path)
    exe = os.path.join(d, 'example')
    rc, out = _run([FC, '-O2', '-cpp', '-fopenmp', "-DDATA_DIR='%s'" % REF, '-I', MODS, '-module-dir', d, os.path.join(REF, 'example.F90')] + built + ['-o', exe])
    assert rc == 0, out
    rc, out = _run([exe], env=dict(os.environ, GADFIT_HIP_DEVICE='-1'))
    assert rc != 0 and 'no GPU bound to this context' in out, out      # gadf_init ... gadf_set, the data files read, the model captured


REF_BIN = os.path.join(ROOT, 'oracle', '_ref')


@pytest.mark.gpu
@pytest.mark.parametrize('images', [1, 3])
@pytest.mark.parametrize('name', ['1_gaussian', '2_integral_single', '3_integral_double', '4_multiple_curves'])
def test_reference_fit_programs_run_unchanged_on_the_device(tmp_path, name, images):
    """fortran/tests/{1_gaussian,2_integral_single,3_integral_double,4_multiple_curves}.F90 as the reference wrote them, linked with
    this repository's library: exit code 0 = the program's own check of the fitted parameter against the reference-held constant
    passed (1e-13 / 1e-11 / 1e-9 / 1e-13 absolute), and gadf_print wrote its results file.  images = 3: the same executable as a
    single-process device group of three members (here sharing the one card, sums in rank order on the host) -- the data split by
    the reference's rule, every pass on all members, and still inside the reference's tolerance."""
    exe = os.path.join(REF_BIN, name)
    # (a missing executable FAILS: under -m gpu the library is there, so these were meant to run -- a skip would read as green)
    assert os.path.exists(exe), 'oracle/_ref/%s was not built (oracle/build_ref_programs.py in the build container; the directory travels with the snapshot)' % name
    env = dict(os.environ)
    if images > 1:
        env.update(GADFIT_HIP_DEVICES=str(images), GADFIT_HIP_GROUP_WRAP='1')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'Error at' not in r.stdout
    if name != '4_multiple_curves':                 # (the one that does not call gadf_print(output='<name>_results'))
        assert any(f.startswith(name) for f in os.listdir(str(tmp_path))), os.listdir(str(tmp_path))


@pytest.mark.gpu
def test_reference_example_runs_unchanged_on_the_device(tmp_path):
    """fortran/tests/example.F90 (the user guide's worked example: two decay curves from files, shot-noise weights, a global
    lifetime, `gadf_fit(lambda=10.0)` with every other argument at its default, `gadf_print`): the parameters it ends with are the
    CPU oracle's for the same fit, to the fit tolerance of the parity suite."""
    import numpy as np
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    from tests.golden import goldens as G
    exe = os.path.join(REF_BIN, 'example')
    assert os.path.exists(exe), 'oracle/_ref/example was not built (oracle/build_ref_programs.py in the build container; the directory travels with the snapshot)'
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.zeros((2, 3))
    for ln in open(os.path.join(str(tmp_path), 'example_results_parameters')):
        f = ln.split()
        if len(f) >= 3 and f[0] in ('1', '2') and f[1].isdigit():       # dataset, parameter, [name,] value
            got[int(f[0]) - 1, int(f[1]) - 1] = float(f[-1])
    d = G.data()['4_multiple_curves']                   # (the example's two files hold the records of reference test 4)
    xs = [np.array(d['x_data_1']), np.array(d['x_data_2'])]
    ys = [np.array(d['y_data_1']), np.array(d['y_data_2'])]
    p = orc.OracleProblem(trace_model(G.model_exponential, 3), xs, ys, [orc.init_weights(orc.SQRT_Y, y) for y in ys],
                          [[1.0] * 3, [1.0] * 3], [0, 1, 2], [0, 1, 0])
    p.fit(lambda_=np.float32(10.0))
    assert np.all(np.abs(got - p.pars) <= 1e-9 * np.abs(p.pars)), (got, p.pars)
    assert all(os.path.exists(os.path.join(str(tmp_path), 'example_results' + sfx)) for sfx in ('', '_parameters', '_log'))
