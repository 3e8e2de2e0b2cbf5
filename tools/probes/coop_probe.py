#!/usr/bin/env python3
"""The fused kernel from 5 tiles on (65 ... 128 active parameters): the workgroup-cooperative form (round 6, GADFIT_HIP_COOP=1) against round
5's per-wave form (GADFIT_HIP_COOP=0: up to 80 parameters; the two-kernel path beyond), in one process.
usage: coop_probe.py [N]   prints per K: stored / not stored kernel ms, LM iteration ms."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
for K in (20, 24, 28, 32):
    truth = M.gaussK_truth(K)
    x, y, s = M.make_single(M.gaussK_numpy(K), truth, n, 0.0, 100.0)
    tape = trace_model(M.make_model_gaussK(K), 4 * K)
    act = list(range(4 * K)); glob = [0] * (4 * K); start = M.start_values(truth).reshape(1, 4 * K)
    for coop in ('1', '0'):
        os.environ['GADFIT_HIP_COOP'] = coop
        ctx = _lib.Context(0)
        ctx.set_placement_after(0)
        ctx.set_model(tape); ctx.set_data(x, y, 1.0 / s, [0, n])
        jac, dim = ctx.jacobian_indices(act, glob)
        out = {'K': K, 'active': 4 * K, 'coop': coop, 'points': n}
        JTJ, JTr, chi2 = ctx.sweep(start, act, jac, dim)
        out['chi2'] = chi2
        try:
            ctx.time_kernel(5, 20); out['fused_stored_ms'] = round(ctx.time_kernel(5, 30), 4)
        except _lib.GadfitHipError:
            ctx.time_kernel(4, 20); a = ctx.time_kernel(4, 30); b = ctx.time_kernel(1, 30)
            out['sweep_ms'] = round(a, 4); out['gram_block_ms'] = round(b, 4)
        ctx.set_keep_jacobian(0)
        try:
            ctx.sweep(start, act, jac, dim); ctx.time_kernel(5, 20); out['fused_nostore_ms'] = round(ctx.time_kernel(5, 30), 4)
        except _lib.GadfitHipError as e:
            out['fused_nostore_ms'] = None
        ctx.set_keep_jacobian(2)
        ctx.fit(start, act, glob, lambda_=1.0, max_iter=2)
        t0 = time.perf_counter(); _, r = ctx.fit(start, act, glob, lambda_=1.0, max_iter=6)
        out['lm_iteration_ms'] = round(1e3 * (time.perf_counter() - t0) / max(1, r.iterations), 4)
        ctx.close()
        print(json.dumps(out), flush=True)
