"""Is a large first-pass deviation of a 'pvx_fd' fuzz case (reals from %val, use_ad=.false.) finite-difference noise or a wrong number?
For each seed: the device's first pass (forward differences over column sets), the oracle's (forward differences, value() on the tape)
and the EXACT sums (the same function with value() taken as the identity, differentiated by AD): both finite-difference results are
held against the exact one.    python tools/probes/fuzz_pvx_fd_noise.py 108 124 160 193"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from gadfit_amd import ad
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import fortran_fuzz as FZ
from tests import test_gpu_fortran_fuzz as T

w = tempfile.mkdtemp()
for seed in [int(a) for a in sys.argv[1:]]:
    got = {}
    orig = T.first_pass_deviation

    def grab(path, first, record=0):
        lines = open(path).read().splitlines()
        head = lines[0].split(); dim = int(head[2])
        got['JTJ'] = np.array(lines[2].split()[1:], dtype=float).reshape(dim, dim); got['oracle'] = first['JTJ']
        return 0.0
    T.first_pass_deviation = grab
    T.TOL_PARS = 1.0; T.TOL_CHI2 = 1.0
    prep = T.prepare_case(seed, 300, w, max_iter=3, pvx=True, use_ad=False)
    try:
        T.run_case(seed, 300, w, max_iter=3, pvx=True, use_ad=False, tol=1.0)
    finally:
        T.first_pass_deviation = orig
    root, active, start = prep['root'], prep['active'], prep['start']
    x, y = np.loadtxt(prep['data'], unpack=True)
    keep = ad.value
    ad.value = lambda v: v
    try:
        tape = trace_model(lambda p, xx: root.fn(p, xx), FZ.NP_)
    finally:
        ad.value = keep
    exact = T.exact_first_pass(orc.OracleProblem(tape, [x], [y], [np.ones_like(x)], [start], active, [0] * FZ.NP_))['JTJ']
    d = np.sqrt(np.abs(np.diag(exact)))
    sc = np.outer(d, d)
    print('seed %d: device FD vs exact %.2e   oracle FD vs exact %.2e   device FD vs oracle FD %.2e   (J^T J in units of sqrt(JTJ_ii JTJ_jj); diag %s)'
          % (seed, np.max(np.abs(got['JTJ'] - exact) / sc), np.max(np.abs(got['oracle'] - exact) / sc), np.max(np.abs(got['JTJ'] - got['oracle']) / sc),
             ' '.join('%.1e' % v for v in np.diag(exact))), flush=True)
    if os.environ.get('NOISE_VERBOSE'):
        print('  active', active, 'start', start)
        print('  device - exact\n', (got['JTJ'] - exact) / sc, '\n  oracle - exact\n', (got['oracle'] - exact) / sc, '\n  exact\n', exact)
