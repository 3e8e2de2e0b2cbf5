#!/usr/bin/env python3
"""One-off soak beyond the 16 seeds of tests/test_gpu_random_models.py: the same randomised parity check (value, gradient, second
directional derivative of random fitting functions over the whole operator set, device against oracle) for seeds
[first, last).    python tools/probes/soak_random_models.py 16 200 [layouts]       (needs the GPU; ~1.5 s per seed: one hiprtc compile each)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GADFIT_HIP_CACHE', '/tmp/gadfit_soak_kcache')
from tests import test_gpu_random_models as T

first, last = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else 'models'       # 'layouts': test_random_layouts_vs_oracle (ragged global fits) instead
fn = T.test_random_layouts_vs_oracle if which == 'layouts' else T.test_random_model_value_gradient_dd
bad = []
t0 = time.time()
for seed in range(first, last):
    try:
        fn(seed)
    except AssertionError as e:
        bad.append(seed)
        print('seed', seed, 'FAILED', str(e)[:300], flush=True)
    if seed % 10 == 0:
        print('seed', seed, 'done, %.0f s' % (time.time() - t0), flush=True)
print('seeds %d..%d: %d failures %s' % (first, last - 1, len(bad), bad))
sys.exit(1 if bad else 0)
