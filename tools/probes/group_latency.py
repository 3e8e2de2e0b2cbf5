#!/usr/bin/env python3
"""Cost of the single-process device group (gfh_create_group) measured on ONE card: the members share
device 0, so what shows is the host side -- the fan-out of a call to the member threads and the ordered host
sum of their result mailboxes -- not bandwidth scaling.
(a) latency regime: LM iterations of the 2-exponential fit at N = 200 (BASELINE config 1) inside one gfh_lm_iterate
    call (the loop runs on the member threads; per iteration only the host sum is added) and as one gfh_sweep
    call per pass (adds the fan-out of every call);
(b) headline size: N = 1e7 x 32 active parameters split over the members of one card (the kernels of the
    members run concurrently on it), against one context.
usage: group_latency.py [members ...]   (default 1 2 4 8)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def make(members):
    return _lib.Context(0) if members == 0 else _lib.Context(devices=[0] * members)


def small(members):
    x, y, s = M.make_single(M.exp2_numpy, M.EXP2_TRUTH, 200, 0.5, 100.0)
    t = trace_model(M.model_exp2, 4)
    start = M.start_values(M.EXP2_TRUTH)
    c = make(members)
    c.set_model(t); c.set_data(x, y, 1.0 / s, [0, 200])
    act = [0, 1, 2, 3]
    st = np.array([1.0, -1.0, 0.0]); dtd = np.zeros(4); pr = np.array([start])
    c.lm_iterate(pr, act, [0] * 4, 20, st, dtd)
    t0 = time.perf_counter(); c.lm_iterate(pr, act, [0] * 4, 400, st, dtd); it_us = (time.perf_counter() - t0) / 400 * 1e6
    jac, dim = c.jacobian_indices(act, [0] * 4)
    p0 = np.array([start])
    for _ in range(50):
        c.sweep(p0, act, jac, dim)
    t0 = time.perf_counter()
    for _ in range(400):
        c.sweep(p0, act, jac, dim)
    call_us = (time.perf_counter() - t0) / 400 * 1e6
    c.close()
    return it_us, call_us


def big(members, n):
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    t = trace_model(M.model_gauss8, 32)
    start = M.start_values(truth).reshape(1, 32)
    c = make(members)
    c.set_model(t); c.set_data(x, y, 1.0 / s, [0, n])
    act = list(range(32))
    c.fit(start, act, [0] * 32, lambda_=1.0, max_iter=10)
    for _ in range(4):
        c.fit(start, act, [0] * 32, lambda_=1.0, max_iter=10)
    t0 = time.perf_counter()
    for _ in range(5):
        out, r = c.fit(start, act, [0] * 32, lambda_=1.0, max_iter=10)
    ms = (time.perf_counter() - t0) / 50 * 1e3
    c.close()
    return ms, r.chi2


def host_only(members):
    """the group's own machinery with compile-only members (no card involved): microseconds per host sum of the headline's packed
    image (32 x 32 + 32 + 2 doubles) inside one task, and per fan-out of an empty call"""
    c = _lib.Context(devices=[-1] * members)
    c.debug_group_latency(1090, 2000)
    best = min(c.debug_group_latency(1090, 20000) for _ in range(3))
    c.close()
    return best


def main():
    members = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    for m in members:
        s_us, f_us = host_only(m)
        print(json.dumps({'members': m, 'compile_only_members': True, 'us_per_host_sum_of_1090_doubles': round(s_us, 2), 'us_per_fan_out': round(f_us, 2)}), flush=True)
    if os.environ.get('GROUP_HOST_ONLY'):
        return
    n = int(os.environ.get('GROUP_POINTS', '10000000'))
    it0, call0 = small(0)
    ms0, chi0 = big(0, n)
    print(json.dumps({'members': 'plain context', 'cfg1_us_per_lm_iteration': round(it0, 1), 'cfg1_us_per_sweep_call': round(call0, 1),
                      'headline_ms_per_lm_iteration': round(ms0, 4)}), flush=True)
    for m in members:
        it, call = small(m)
        ms, chi = big(m, n)
        print(json.dumps({'members': m, 'cfg1_us_per_lm_iteration': round(it, 1), 'cfg1_us_per_sweep_call': round(call, 1),
                          'headline_ms_per_lm_iteration': round(ms, 4), 'headline_chi2_rel_diff_vs_plain': abs(chi - chi0) / chi0}), flush=True)


if __name__ == '__main__':
    main()
