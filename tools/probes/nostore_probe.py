#!/usr/bin/env python3
"""The fused STEP 1 + STEP 2 kernel WITHOUT the Jacobian store (gfh_k_sweep_gram_nostore: what gadf_fit's own mode runs) on the
headline workload, back to back: `python3 tools/probes/nostore_probe.py [N] [launches] [rounds]`.  Prints one JSON line with the
HIP-event average per launch of every round.  The program rocprofv3 is pointed at by tools/pmc_nostore.sh."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    ctx = _lib.Context(0)
    ctx.set_model(trace_model(M.model_gauss8, 32))
    ctx.set_data(x, y, 1.0 / s, [0, n])
    active = list(range(32))
    jac, dim = ctx.jacobian_indices(active, [0] * 32)
    ctx.set_keep_jacobian(0)
    JTJ, JTr, chi2 = ctx.sweep(M.start_values(truth).reshape(1, 32), active, jac, dim)
    if os.environ.get('NOSTORE_AS_FIT', '1') != '0':      # (round 6: the kernel exactly as gfh_fit configures it under mode 2 -- what bench.py's leg times)
        ctx.set_keep_jacobian(2)
        ctx.fit(M.start_values(truth).reshape(1, 32), active, [0] * 32, lambda_=1.0, max_iter=2)
    import hashlib
    digest = hashlib.sha256(JTJ.tobytes() + JTr.tobytes() + np.float64(chi2).tobytes()).hexdigest()[:16]      # (bitwise identity of the sums across kernel variants)
    ctx.time_kernel(5, 60)          # (the first ~40 launches after an idle gap run in the power-management transient)
    ms = [round(ctx.time_kernel(5, launches), 5) for _ in range(rounds)]
    chi = [round(ctx.time_kernel(2, launches), 5) for _ in range(2)]
    src_sha1 = hashlib.sha1(ctx.model_source(active).encode()).hexdigest()      # (the kernel the counts belong to: bench.py refuses another)
    print(json.dumps({'kernel': 'gfh_k_sweep_gram_nostore', 'source_sha1': src_sha1, 'points': n, 'launches_per_round': launches, 'ms_per_launch': ms, 'chi2_ms_per_launch': chi, 'sums_sha256': digest,
                      'variant': {k: os.environ[k] for k in sorted(os.environ) if k.startswith('GADFIT_HIP_')}}), flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
