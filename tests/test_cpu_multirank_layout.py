"""The library's own multi-rank bookkeeping, driven without a GPU (gfh_debug_packed_layout): every rank must derive the
same layout and length of the all-reduced image [JTJ | JTres | chi2] -- dense, or pattern-only for global fits -- from
what the ranks share (column map, dim, number of datasets), whatever share of the points it holds; ncclAllReduce silently
requires that (co_sum's replacement, misc.F90:133-170, call sites gadfit.F90:700-701)."""
import numpy as np
import pytest

from gadfit_amd import _lib


def _jac(nd, active, is_global):
    c = _lib.Context(-1)
    c.nd = nd
    jac, dim = c.jacobian_indices(active, is_global)
    c.close()
    return jac, dim


CASES = {
    # name: (dataset sizes, n_act, is_global)
    'single_32': ([100003], 32, [0] * 32),
    'global_64x7_ragged': ([1500 + 37 * (k % 5) for k in range(60)] + [1, 2, 40000, 3], 7, [0, 0, 0, 0, 1, 1, 1]),      # dim 259: pattern-only
    'global_3x7_small': ([700, 1, 1300], 7, [0, 0, 0, 0, 1, 1, 1]),                                                    # dense (tail's reach)
    'global_20x7': ([300] * 20, 7, [0, 0, 0, 0, 1, 1, 1]),
    'fewer_points_than_ranks': ([2, 1], 3, [0, 1, 0]),
}


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('nranks', [1, 2, 3, 8])
def test_every_rank_derives_the_same_packed_layout(name, nranks):
    sizes, na, is_global = CASES[name]
    nd = len(sizes)
    pos = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    jac, dim = _jac(nd, list(range(na)), is_global)
    lay = [_lib.debug_packed_layout(nranks, r, int(pos[-1]), pos, jac, dim) for r in range(nranks)]
    ref = _lib.debug_packed_layout(1, 0, int(pos[-1]), pos, jac, dim)
    for r, L in enumerate(lay):
        for key in ('packed_n', 'pattern_only', 'nnz', 'hash'):
            assert L[key] == ref[key], (r, key)
        assert np.array_equal(L['nz_row'], ref['nz_row']) and np.array_equal(L['nz_col'], ref['nz_col'])
    # the shares: contiguous, in rank order, the reference's sizes (gadfit.F90:978-983)
    assert lay[0]['begin'] == 0 and sum(L['count'] for L in lay) == pos[-1]
    for r in range(1, nranks):
        assert lay[r]['begin'] == lay[r - 1]['begin'] + lay[r - 1]['count']
    base = int(pos[-1]) // nranks
    assert all(L['count'] in (base, base + 1) for L in lay)
    if nranks > 1 and nd > 1:
        assert any(L['datasets_held'] < nd for L in lay), 'the case is meant to leave some rank without points of some dataset'
    if name == 'fewer_points_than_ranks' and nranks == 8:
        assert sum(L['count'] == 0 for L in lay) == 5 and all(L['gram_blocks'] == 0 for L in lay if L['count'] == 0)
    # the layout itself against an independent statement of it
    dense_n = dim * dim + dim + 1
    pat = sorted({(min(a, b), max(a, b)) for d in range(nd) for a in jac[d] for b in jac[d]}, key=lambda rc: (rc[1], rc[0]))
    if nd > 1:
        assert ref['nnz'] == len(pat) and [tuple(p) for p in zip(ref['nz_row'], ref['nz_col'])] == pat
    want_pattern = nd > 1 and 4 * (len(pat) + dim + 1) < dense_n and dim * dim * nd > 65536
    assert ref['pattern_only'] == int(want_pattern)
    assert ref['packed_n'] == (len(pat) + dim + 1 if want_pattern else dense_n)


def test_pattern_switch_off_gives_the_dense_image_on_every_rank():
    sizes, na, is_global = CASES['global_64x7_ragged']
    pos = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    jac, dim = _jac(len(sizes), list(range(na)), is_global)
    for r in range(3):
        L = _lib.debug_packed_layout(3, r, int(pos[-1]), pos, jac, dim, sparse_ok=False)
        assert L['pattern_only'] == 0 and L['packed_n'] == dim * dim + dim + 1
