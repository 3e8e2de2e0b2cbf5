#!/usr/bin/env python3
"""The fused STEP 1 + STEP 2 kernel over the tile counts (K skewed Gaussians = 4K active parameters), stored and not stored, under
the environment's GADFIT_HIP_* variant: HIP-event ms per launch and a digest of the sums.
  python3 tools/probes/fused_matrix_variants.py [N] [K ...]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
Ks = [int(k) for k in sys.argv[2:]] or [3, 4, 8, 12, 16, 20, 24]
out = {'variant': {k: os.environ[k] for k in sorted(os.environ) if k.startswith('GADFIT_HIP_')}, 'points': n, 'ms': {}}
for K in Ks:
    truth = M.gaussK_truth(K)
    x, y, s = M.make_single(M.gaussK_numpy(K), truth, n, 0.0, 100.0)
    ctx = _lib.Context(0)
    ctx.set_placement_after(0)
    ctx.set_model(trace_model(M.make_model_gaussK(K), 4 * K))
    ctx.set_data(x, y, 1.0 / s, [0, n])
    act = list(range(4 * K)); jac, dim = ctx.jacobian_indices(act, [0] * (4 * K))
    start = M.start_values(truth).reshape(1, 4 * K)
    e = {}
    for label, mode in (('stored', 1), ('nostore', 0)):
        ctx.set_keep_jacobian(mode)
        JTJ, JTr, chi2 = ctx.sweep(start, act, jac, dim)
        e[label + '_sha'] = hashlib.sha256(JTJ.tobytes() + JTr.tobytes() + np.float64(chi2).tobytes()).hexdigest()[:12]
        ctx.time_kernel(5, 60)
        e[label] = round(min(ctx.time_kernel(5, 100) for _ in range(3)), 5)
    e['chi2'] = round(min(ctx.time_kernel(2, 100) for _ in range(2)), 5)
    out['ms'][4 * K] = e
    ctx.close()
print(json.dumps(out), flush=True)
