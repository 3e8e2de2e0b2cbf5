#!/usr/bin/env python3
"""Soak beyond the 12 seeds of test_random_branching_model: python tools/probes/soak_random_branching.py 12 200"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GADFIT_HIP_CACHE', '/tmp/gadfit_soak_kcache')
from tests import test_gpu_random_models as T
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for seed in range(first, last):
    try:
        T.test_random_branching_model(seed)
    except Exception as e:
        bad.append(seed)
        print('seed', seed, 'FAILED', type(e).__name__, str(e)[:400], flush=True)
    if seed % 10 == 0:
        print('seed', seed, 'done, %.0f s' % (time.time() - t0), flush=True)
print('seeds %d..%d: %d failures %s' % (first, last - 1, len(bad), bad))
sys.exit(1 if bad else 0)
