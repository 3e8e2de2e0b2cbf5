"""world_size-2 (and 3) gloo test of the multi-GPU data path logic on CPU: each rank takes
its gfh_partition slice, forms its partial [JTJ | JTres | chi2] (here with the oracle as the
stand-in for the device kernels -- tests may use it), packs it in the layout the LIBRARY derives for
that rank (gfh_debug_packed_layout: dense, or pattern-only for the larger global fit), the packed
buffer is all-reduced, and the unpacked result must equal the single-image result.  This is the
co_sum replacement of gadfit.F90:700-701 / misc.F90:133-170."""
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
sizes = %(sizes)r
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
dim = full.dim
# the layout of the all-reduced image as the library derives it for THIS rank (no GPU needed)
L = _lib.debug_packed_layout(world, rank, N, full.dp, full.jac, dim)
assert (L['begin'], L['count']) == (begin, count)
everyone = [None] * world
dist.all_gather_object(everyone, (L['packed_n'], L['pattern_only'], L['nnz'], L['hash']))
assert all(e == everyone[0] for e in everyone), everyone       # what ncclAllReduce silently requires
assert L['pattern_only'] == %(pattern)d
if L['pattern_only']:
    vals = JTJ[L['nz_row'], L['nz_col']]
else:
    vals = JTJ.T.ravel()                      # column-major, as the device image
packed = torch.from_numpy(np.concatenate([vals, JTr, [chi]]))
assert packed.numel() == L['packed_n']
dist.all_reduce(packed)                       # one fused all-reduce per sweep (SURVEY section 2 table)
got = packed.numpy()
if L['pattern_only']:
    G = np.zeros((dim, dim)); G[L['nz_row'], L['nz_col']] = got[:L['nnz']]; G[L['nz_col'], L['nz_row']] = got[:L['nnz']]
    off = L['nnz']
    assert np.count_nonzero(JTJ0) <= 2 * L['nnz'], 'the full result has entries outside the pattern'
else:
    G = got[:dim * dim].reshape(dim, dim).T; off = dim * dim
sc = np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0)))
assert np.max(np.abs(G - JTJ0) / sc) < 1e-13
assert np.max(np.abs(got[off:off + dim] - JTr0)) <= 1e-12 * np.max(np.abs(JTr0))
assert abs(got[-1] - chi0) <= 1e-13 * chi0
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world,sizes,pattern', [(2, [700, 1, 1300], 0), (3, [700, 1, 1300], 0),
                                                 (2, [60 + 7 * (k % 4) for k in range(39)] + [1], 1)])      # dim 163: pattern-only
def test_partition_allreduce_equals_single_image(tmp_path, world, sizes, pattern):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT, 'sizes': sizes, 'pattern': pattern})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + world + 10 * pattern), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
