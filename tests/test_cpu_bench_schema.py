"""The N > 1 bench line must prove itself (cross-rank parity, measured all-reduce latency, host-sum leg, strong leg): its schema
is pinned here on a device group of compile-only members, which runs without a GPU (`bench.py --dry`).  The GPU tests
(test_gpu_device_group.py, test_gpu_parity.py) fill the same keys with measured values on one card."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MULTI_KEYS = {'multi_gpu_parity', 'allreduce_us', 'strong_leg', 'rccl_ms_per_step', 'host_sum_ms_per_step', 'host_sum_leg'}
LINE_KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
             'rccl_nranks'}


def check_multi_block(d, ranks):
    """shared with the GPU tests: every key of the self-check legs is there and says what it measured"""
    assert MULTI_KEYS <= set(d), sorted(MULTI_KEYS - set(d))
    mp = d['multi_gpu_parity']
    assert mp['ranks'] == ranks and mp['cross_rank_sum_path'] in ('rccl', 'host')
    sums = mp['sums_vs_ordered_host_sum']
    assert set(sums['max_dev']) == {'JTJ', 'JTres', 'chi2'} and sums['tol'] == 1e-13
    assert sums['ok'] == (max(sums['max_dev'].values()) <= sums['tol'])
    au = d['allreduce_us']
    assert {'median', 'p95', 'min', 'max', 'host_round_trip_median', 'doubles', 'rounds', 'ranks_counted', 'path'} <= set(au)
    assert au['ranks_counted'] == ranks and au['doubles'] == 32 * 32 + 32 + 1 and 0 < au['min'] <= au['median'] <= au['p95'] <= au['max']
    return mp


def test_dry_line_has_every_key_of_the_multi_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry', '--gpus', '4', '--steps', '7', '--warmup', '2'],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert LINE_KEYS <= set(d) and d['dry'] is True and d['n_gpus'] == 4 and d['steps'] == 7 and d['warmup'] == 2
    mp = check_multi_block(d, 4)
    # the dry members' sum went through the group's real host sum: bitwise the rank-ordered sum
    assert mp['cross_rank_sum_path'] == 'host' and mp['ok'] and max(mp['sums_vs_ordered_host_sum']['max_dev'].values()) == 0.0
    assert d['value'] is None and d['strong_leg'] is None and d['host_sum_ms_per_step'] is None      # nothing that needs a kernel is made up


def test_parity_verdict_follows_the_tolerances():
    sys.path.insert(0, ROOT)
    import bench
    lat = dict(median_us=1.0, p95_us=2.0, min_us=0.5, max_us=3.0, host_round_trip_median_us=4.0, nranks=2)
    good = {'JTJ': 1e-15, 'JTres': 2e-15, 'chi2': 0.0}
    fit = {'max_rel_dev_pars': 3e-12, 'ranks_agree_bitwise': True, 'ok': True}
    assert bench._multi_gpu_block(2, 'rccl', good, True, fit, lat, 1057, None, None, 0.5)['multi_gpu_parity']['ok']
    assert not bench._multi_gpu_block(2, 'rccl', dict(good, JTJ=2e-13), True, fit, lat, 1057, None, None, 0.5)['multi_gpu_parity']['ok']
    assert not bench._multi_gpu_block(2, 'rccl', good, True, dict(fit, ok=False), lat, 1057, None, None, 0.5)['multi_gpu_parity']['ok']
    assert not bench._multi_gpu_block(2, 'rccl', good, False, fit, lat, 1057, None, None, 0.5)['multi_gpu_parity']['ok']
