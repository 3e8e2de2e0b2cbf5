#!/usr/bin/env python3
"""Soak beyond the seeds of tests/test_gpu_fortran_fuzz.py: python tools/probes/soak_fortran_fuzz.py 200 260 [n_points [branching|integral|integral_branching|integral_nested|layout|layout_branching|layout_big|layout_branching_big|sessions|pvx|pvx_fd|layout_pvx|layout_branching_pvx|integral_pv]]
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
integral = len(sys.argv) > 4 and sys.argv[4].startswith('integral') and sys.argv[4] != 'integral_pv'
if integral and 'branching' in sys.argv[4]: branching = True
nested = integral and 'nested' in sys.argv[4]
layout = len(sys.argv) > 4 and sys.argv[4].startswith('layout')
sessions = len(sys.argv) > 4 and sys.argv[4] == 'sessions'
layout_pvx = layout and 'pvx' in sys.argv[4]                           # layout_pvx / layout_branching_pvx: ... whose leaves form reals from %val and x
pvx = len(sys.argv) > 4 and sys.argv[4].startswith('pvx')          # reals formed from %val and x; 'pvx_fd': under use_ad=.false.
pvx_fd = pvx and sys.argv[4].endswith('fd')
ipv = len(sys.argv) > 4 and sys.argv[4] == 'integral_pv'              # an integrand that forms a real from the %val of one of its parameters
layout_branching = len(sys.argv) > 4 and 'branching' in sys.argv[4] and layout
layout_big = len(sys.argv) > 4 and 'big' in sys.argv[4] and layout
work = tempfile.mkdtemp(prefix='fzsoak')
worst = [0.0, 0.0]; skipped = 0; failed = []
by_kind = {}      # kind (incl. use_ad) -> [cases, worst parameter deviation, worst chi2 deviation, worst first-pass deviation]
n_logged = 0
mode = sys.argv[4] if len(sys.argv) > 4 else 'straight-line'
for seed in range(lo, hi):
    try:
        out = T.run_two_sessions(seed, seed + 7919, work, branching_a=bool(seed & 1), branching_b=bool(seed & 2)) if sessions else T.run_layout_case(seed, work, branching=layout_branching, big=layout_big, pvx=layout_pvx) if layout else T.run_case(seed, npts, work, max_iter=3, pvx=True, use_ad=not pvx_fd) if pvx else T.run_case(seed, npts, work, integral=True, ipv=True) if ipv else T.run_case(seed, npts, work, branching=branching, integral=integral, tol=(1e-6 if integral and (branching or nested) else None), nested=nested)
    except AssertionError as e:
        failed.append(seed)
        print('seed %d FAILED: %s' % (seed, str(e)[:1500]), flush=True)
        continue
    if out is None:
        skipped += 1
        continue
    worst = [max(worst[0], out[0]), max(worst[1], out[1])]
    print('seed %d: parameters %.2e, chi2 %.2e  %s' % (seed, out[0], out[1], ' | '.join('[%s] %.1e' % (c[0], c[1]) for c in T.CASE_LOG[n_logged:])), flush=True)
    n_logged = len(T.CASE_LOG)
for c in T.CASE_LOG:
    e = by_kind.setdefault(c[0], [0, 0.0, 0.0, 0.0]); e[0] += 1; e[1] = max(e[1], c[1]); e[2] = max(e[2], c[2]); e[3] = max(e[3], c[3])
print('seeds %d..%d (N = %d, %s): %d failures %s, %d skipped, worst deviation: parameters %.3e, chi2 %.3e' % (lo, hi - 1, npts, mode, len(failed), failed, skipped, *worst))
for kind in sorted(by_kind):
    print('  kind %-40s %4d cases   worst: fitted parameters %.2e   chi2 %.2e   first pass (JTJ / JTres / chi2 at the start parameters) %.2e'
          % (kind, by_kind[kind][0], by_kind[kind][1], by_kind[kind][2], by_kind[kind][3]))
