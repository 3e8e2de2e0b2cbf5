#!/usr/bin/env python3
"""Soak beyond the seeds of tests/test_gpu_fortran_fuzz.py: python tools/probes/soak_fortran_fuzz.py 200 260 [n_points [branching|integral|integral_branching|integral_nested|layout|layout_branching|layout_big|layout_branching_big|sessions]]
(random Fortran eval() bodies, compiled and fitted on the GPU through the Fortran API, against the CPU oracle)"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GADFIT_HIP_CACHE', '/tmp/gadfit_soak_kcache')
from tests import test_gpu_fortran_fuzz as T       # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 300
branching = len(sys.argv) > 4 and sys.argv[4] == 'branching'
integral = len(sys.argv) > 4 and sys.argv[4].startswith('integral')
if integral and 'branching' in sys.argv[4]: branching = True
nested = integral and 'nested' in sys.argv[4]
layout = len(sys.argv) > 4 and sys.argv[4].startswith('layout')
sessions = len(sys.argv) > 4 and sys.argv[4] == 'sessions'
layout_branching = len(sys.argv) > 4 and 'branching' in sys.argv[4] and layout
layout_big = len(sys.argv) > 4 and 'big' in sys.argv[4] and layout
work = tempfile.mkdtemp(prefix='fzsoak')
worst = [0.0, 0.0]; skipped = 0; failed = []
for seed in range(lo, hi):
    try:
        out = T.run_two_sessions(seed, seed + 7919, work, branching_a=bool(seed & 1), branching_b=bool(seed & 2)) if sessions else T.run_layout_case(seed, work, branching=layout_branching, big=layout_big) if layout else T.run_case(seed, npts, work, branching=branching, integral=integral, tol=(1e-6 if integral and (branching or nested) else None), nested=nested)
    except AssertionError as e:
        failed.append(seed)
        print('seed %d FAILED: %s' % (seed, str(e)[:1500]), flush=True)
        continue
    if out is None:
        skipped += 1
        continue
    worst = [max(worst[0], out[0]), max(worst[1], out[1])]
    if (seed - lo) % 10 == 9:
        print('... seed %d, worst so far: parameters %.2e, chi2 %.2e' % (seed, *worst), flush=True)
print('seeds %d..%d (N = %d): %d failures %s, %d skipped, worst deviation: parameters %.3e, chi2 %.3e' % (lo, hi - 1, npts, len(failed), failed, skipped, *worst))
