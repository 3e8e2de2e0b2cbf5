#!/usr/bin/env python3
"""Timing-only ablations of the cooperative fused kernel (GADFIT_HIP_ABLATE: 4 no matrix instructions, 8 the AD body replaced by a few cheap
instructions; results are wrong).  usage: coop_ablate.py [K] [N]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
truth = M.gaussK_truth(K)
x, y, s = M.make_single(M.gaussK_numpy(K), truth, n, 0.0, 100.0)
tape = trace_model(M.make_model_gaussK(K), 4 * K)
act = list(range(4 * K)); start = M.start_values(truth).reshape(1, 4 * K)
for ab in ('0', '4', '8', '12'):
    os.environ['GADFIT_HIP_ABLATE'] = ab
    ctx = _lib.Context(0)
    ctx.set_model(tape); ctx.set_data(x, y, 1.0 / s, [0, n])
    jac, dim = ctx.jacobian_indices(act, [0] * (4 * K))
    ctx.set_keep_jacobian(0)
    ctx.sweep(start, act, jac, dim); ctx.time_kernel(5, 20)
    print(json.dumps({'K': K, 'ablate': ab, 'nostore_ms': round(ctx.time_kernel(5, 30), 4)}), flush=True)
    ctx.close()
