import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
n = 4_000_000
for K in (20, 24):
    truth = M.gaussK_truth(K)
    x, y, s = M.make_single(M.gaussK_numpy(K), truth, n, 0.0, 100.0)
    ctx = _lib.Context(0); ctx.set_placement_after(0)
    ctx.set_model(trace_model(M.make_model_gaussK(K), 4 * K)); ctx.set_data(x, y, 1.0 / s, [0, n])
    act = list(range(4 * K)); jac, dim = ctx.jacobian_indices(act, [0] * (4 * K))
    start = M.start_values(truth).reshape(1, 4 * K)
    ctx.sweep(start, act, jac, dim)
    ctx.time_kernel(4, 60); a = min(ctx.time_kernel(4, 60) for _ in range(3))
    ctx.time_kernel(1, 60); b = min(ctx.time_kernel(1, 60) for _ in range(3))
    import time
    ctx.set_keep_jacobian(2)
    ctx.fit(start, act, [0] * (4 * K), lambda_=1.0, max_iter=3)
    t0 = time.perf_counter(); _, r = ctx.fit(start, act, [0] * (4 * K), lambda_=1.0, max_iter=10); dt = time.perf_counter() - t0
    print(json.dumps({'params': 4 * K, 'fused_env': os.environ.get('GADFIT_HIP_FUSED'), 'sweep_ms': a, 'gram_ms': b, 'fit_ms_per_iteration': 1e3 * dt / r.iterations, 'iterations': r.iterations,
                      'n_sweeps': r.n_sweeps, 'n_chi2': r.n_chi2, 'n_lookahead': r.n_lookahead}), flush=True)
    ctx.close()
