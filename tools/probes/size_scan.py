#!/usr/bin/env python3
"""Headline kernels against the number of points on one card (1e7 ... 1e8): fused sweep + Gram with and without the Jacobian store,
plain sweep, chi2 -- ns per 1000 points and the fused kernel's fraction of 8 TB/s; at the largest size also with the placement of
the Jacobian buffer switched off.  Explains the drop from 77 % at 1e7 to 67 % at 1e8 seen in round 1 (tools/probes/big_single_gpu.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

truth = M.gauss8_truth()
tape = trace_model(M.model_gauss8, 32)
act = list(range(32)); start = M.start_values(truth).reshape(1, 32)
nmax = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
X, Y, S = M.make_single_slice(M.gauss8_numpy, truth, nmax, 0, nmax, 0.0, 100.0)
for n, tries in [(10_000_000, 12), (25_000_000, 12), (50_000_000, 12), (nmax, 12), (nmax, 1)]:
    # (a prefix of the big array: the same density of points per unit of x does not matter to these kernels)
    ctx = _lib.Context(0)
    ctx.set_placement_tries(tries); ctx.set_placement_after(0)
    ctx.set_model(tape)
    ctx.set_data(X[:n], Y[:n], S[:n], [0, n]); ctx.init_weights(4)
    jac, dim = ctx.jacobian_indices(act, [0] * 32)
    ctx.sweep(start, act, jac, dim)
    row = {'n': n, 'placement_tries': tries, 'placement_ms': [round(v, 3) for v in ctx.placement()], 'copy_GBps': round(ctx.placement_copy_GBps())}
    for label, which in [('fused', 5), ('plain_sweep', 4), ('chi2', 2)]:
        ctx.time_kernel(which, 60)
        ms = ctx.time_kernel(which, 40)
        row[label + '_ms'] = round(ms, 4); row[label + '_ns_per_kpt'] = round(1e6 * ms / (n / 1000.0), 3)
    row['fused_frac_of_8TBps'] = round(288.0 * n / (row['fused_ms'] * 1e-3) / 8e12, 4)
    ctx.set_keep_jacobian(0); ctx.sweep(start, act, jac, dim)
    ctx.time_kernel(5, 60); ms = ctx.time_kernel(5, 40)
    row['fused_nostore_ms'] = round(ms, 4); row['fused_nostore_ns_per_kpt'] = round(1e6 * ms / (n / 1000.0), 3)
    print(row, flush=True)
    ctx.close()
