"""How far do the fitted parameters of an N-rank fit (ordered host sum, members on one card) lie from the one-rank fit of the same points,
per parameter and per iteration count?  (bench.py multi_gpu_parity.fit_vs_one_rank: which bound is conditioning-free?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['GADFIT_HIP_GROUP_WRAP'] = '1'; os.environ['GADFIT_HIP_GROUP_REDUCE'] = 'host'
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
truth = M.gauss8_truth(); tape = trace_model(M.model_gauss8, 32); active = list(range(32)); glob = [0] * 32
start = M.start_values(truth).reshape(1, 32)
for nf in (200_000, 2_000_000):
    x, y, s = M.make_single_slice(M.gauss8_numpy, truth, nf, 0, nf, 0.0, 100.0)
    for iters in (10, 20, 40):
        res = []
        for world in (1, 8):
            c = _lib.Context(0) if world == 1 else _lib.Context(devices=world)
            c.set_model(tape); c.set_data(x, y, s, [0, nf]); c.init_weights(4)
            p, r = c.fit(start, active, glob, lambda_=1.0, max_iter=iters)
            res.append((p.ravel().copy(), r.chi2, r.iterations, r.exit_reason, r.lambda_)); c.close()
        d = np.abs(res[1][0] - res[0][0]) / np.abs(res[0][0])
        k = int(np.argmax(d))
        print('N=%d iters=%d (%d/%d, exit %d/%d, lambda %.1e): max rel dev %.2e at par %d (kind %d); by kind A %.1e mu %.1e w %.1e s %.1e; chi2 dev %.1e; dist to truth %.1e'
              % (nf, iters, res[0][2], res[1][2], res[0][3], res[1][3], res[0][4], d[k], k, k % 4, d[0::4].max(), d[1::4].max(), d[2::4].max(), d[3::4].max(),
                 abs(res[1][1] - res[0][1]) / res[0][1], np.max(np.abs(res[0][0] - truth) / truth)), flush=True)
