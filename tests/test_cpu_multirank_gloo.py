"""world_size-2 (and 3) gloo test of the multi-GPU data path logic on CPU: each rank takes
its gfh_partition slice, forms its partial [JTJ | JTres | chi2] (here with the oracle as the
stand-in for the device kernels -- tests may use it), the packed buffer is all-reduced, and
the result must equal the single-image result.  This is the co_sum replacement of
gadfit.F90:700-701 / misc.F90:133-170."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import models as M

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
sizes = [700, 1, 1300]
xs, ys, ss, truths = M.make_global7(len(sizes), sizes)
pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
tape = trace_model(M.model_global7, 7)
active = list(range(7)); is_global = [0, 0, 0, 0, 1, 1, 1]
full = orc.OracleProblem(tape, xs, ys, [1 / s for s in ss], pars, active, is_global)
JTJ0, JTr0, _, _ = full.sweep(); chi0, _ = full.chi2()
# this rank's contiguous slice of the concatenated arrays
N = full.N
begin, count = _lib.partition(N, world, rank)
X = np.concatenate(xs); Y = np.concatenate(ys); W = np.concatenate([1 / s for s in ss])
lx, ly, lw = [], [], []
for d in range(len(sizes)):
    lo, hi = max(begin, full.dp[d]), min(begin + count, full.dp[d + 1])
    sl = slice(lo, max(lo, hi))
    lx.append(X[sl]); ly.append(Y[sl]); lw.append(W[sl])
loc = orc.OracleProblem(tape, lx, ly, lw, pars, active, is_global)
JTJ, JTr, _, _ = loc.sweep(); chi, _ = loc.chi2()
packed = torch.from_numpy(np.concatenate([JTJ.ravel(), JTr, [chi]]))
dist.all_reduce(packed)                       # one fused all-reduce per sweep (SURVEY section 2 table)
got = packed.numpy(); dim = full.dim
sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
assert np.max(np.abs(got[:dim * dim].reshape(dim, dim) - JTJ0) / sc) < 1e-13
assert np.max(np.abs(got[dim * dim:dim * dim + dim] - JTr0)) <= 1e-12 * np.max(np.abs(JTr0))
assert abs(got[-1] - chi0) <= 1e-13 * chi0
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world', [2, 3])
def test_partition_allreduce_equals_single_image(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
