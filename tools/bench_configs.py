#!/usr/bin/env python3
"""Per-kernel device times (HIP events on the context stream) for the BASELINE.json configurations.
Prints one JSON line per configuration; numbers go into DESIGN.md section 3."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
from tests.golden import goldens as G


def run(name, tape, xs, ys, ws, pars, active, is_global, reps=100, extra=None, fit_iters=10):
    reps = int(os.environ.get('BENCH_REPS', reps))
    ctx = _lib.Context(0)
    ctx.set_placement_after(0)          # per-kernel rates of a long job: the Jacobian buffer is placed at the first sweep
    pos = np.zeros(len(xs) + 1, dtype=np.int64)
    for i, a in enumerate(xs):
        pos[i + 1] = pos[i] + len(a)
    ctx.set_model(tape)
    ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
    jac, dim = ctx.jacobian_indices(active, is_global)
    JTJ, JTr, chi2 = ctx.sweep(pars, active, jac, dim)
    d1 = _lib.potr(JTJ + np.diag(np.diag(JTJ)), JTr)
    ctx.omega(pars, d1)
    n = int(pos[-1]); na = len(active)
    out = {'config': name, 'points': n, 'n_active': na, 'dim': dim}
    for label, which, bytes_pp in [('fused_sweep_gram', 5, 32 + 8 * na), ('sweep_only', 4, 32 + 8 * na), ('gram_only', 1, 8 * na + 8),
                                   ('chi2', 2, 32), ('omega', 3, 24), ('omega_jt', 6, 32), ('jtv_stored_J', 7, 8 * na + 8),
                                   ('sweep_gram_no_store', -5, 32)]:
        if which == -5:      # the fused kernel without the Jacobian store (what gfh_fit launches under keep_jacobian 0 / 2)
            try:
                ctx.set_keep_jacobian(0); ctx.sweep(pars, active, jac, dim); which = 5
            except _lib.GadfitHipError as e:
                out[label] = {'skipped': str(e)[:60]}
                continue
        try:
            ctx.time_kernel(which, max(40, reps))      # untimed: the first ~40 launches after an idle gap run in the power-management transient
            ms = ctx.time_kernel(which, reps)
        except _lib.GadfitHipError as e:
            out[label] = {'skipped': str(e)[:60]}
            continue
        out[label] = {'ms': round(ms, 4), 'GBps': round(bytes_pp * n / (ms * 1e-3) / 1e9, 1)}
    ctx.set_keep_jacobian(1); ctx.sweep(pars, active, jac, dim)
    # quadrature models: STEP 1 and STEP 3 replaying the meshes the pass before them recorded at the same parameters (what an accepted LM
    # step runs: trial chi2 -> sweep -> STEP 3), against the bisecting forms above
    for label, which in [('sweep_mesh_replay', 8), ('omega_mesh_replay', 9)]:
        try:
            ctx.chi2(pars); ctx.sweep(pars, active, jac, dim); ctx.omega(pars, d1)
            ctx.time_kernel(which, max(40, reps))
            out[label] = {'ms': round(ctx.time_kernel(which, reps), 4)}
        except _lib.GadfitHipError as e:
            out[label] = {'skipped': str(e)[:60]}
    # chi2 without the residual store (what gfh_fit asks for under keep_jacobian mode 2: nothing reads res back)
    try:
        ctx.set_keep_jacobian(2)
        ctx.fit(pars, active, is_global, lambda_=1.0, max_iter=1)
        ctx.time_kernel(2, max(40, reps))
        ms = ctx.time_kernel(2, reps)
        out['chi2_no_res_store'] = {'ms': round(ms, 4), 'GBps': round(24 * n / (ms * 1e-3) / 1e9, 1)}
    except _lib.GadfitHipError as e:
        out['chi2_no_res_store'] = {'skipped': str(e)[:60]}
    ctx.set_keep_jacobian(1)
    ctx.sweep(pars, active, jac, dim)
    # use_ad = .false.: the same STEP 1(+2) kernel with the reference's forward differences (n_active extra value evaluations per point)
    try:
        ctx.set_use_ad(False)
        ctx.sweep(pars, active, jac, dim)
        ctx.time_kernel(0, max(40, reps))
        out['sweep_finite_differences'] = {'ms': round(ctx.time_kernel(0, reps), 4)}
    except _lib.GadfitHipError as e:
        out['sweep_finite_differences'] = {'skipped': str(e)[:60]}
    ctx.set_use_ad(True)
    ctx.sweep(pars, active, jac, dim)
    # whole LM iterations through gfh_fit (look-ahead schedule, plain lambda x/÷10): wall time per iteration
    import time
    try:
        ctx.fit(pars, active, is_global, lambda_=1.0, max_iter=2)
        t0 = time.perf_counter()
        _, r = ctx.fit(pars, active, is_global, lambda_=1.0, max_iter=fit_iters)
        out['fit'] = {'iterations': r.iterations, 'ms_per_iteration': round(1e3 * (time.perf_counter() - t0) / max(1, r.iterations), 4)}
    except _lib.GadfitHipError as e:      # cfg 4 is timed on y = 1 data: a fit from there may leave the quadrature's domain
        out['fit'] = {'skipped': str(e)[:80]}
    if extra:
        out.update(extra)
    ctx.close()
    print(json.dumps(out), flush=True)


def main():
    only = os.environ.get('BENCH_CFG')      # e.g. BENCH_CFG=4: that configuration only
    if only in (None, '2'):
      x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 10_000_000, 0.0, 100.0)
      run('cfg2: 4-exponential, 8 active, N=1e7', trace_model(M.model_exp4, 8), [x], [y], [1 / s],
        M.start_values(M.EXP4_TRUTH).reshape(1, 8), list(range(8)), [0] * 8)
    if only in (None, '3'):
      xs, ys, ss, truths = M.make_global7(64, 100_000)
      pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
      run('cfg3: global fit 64 x 1e5, 4 local + 3 global', trace_model(M.model_global7, 7), xs, ys, [1 / s for s in ss], pars,
        list(range(7)), [0, 0, 0, 0, 1, 1, 1])
    if only in (None, '4'):
      n = 1_000_000
      a, b = 7.5, 0.8
      xq = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
      # data from the closed form pi/2 b^(-(a+1)/2) gamma_lower((a+1)/2, b x^2) + noise (1e6 host quadratures would take minutes)
      from scipy.special import gammainc, gamma
      fq = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * xq * xq)
      sq = 0.01 * (1 + np.abs(fq))
      yq = fq + sq * M.normal(n, M.SEED)
      t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
      run('cfg4: pi*int_0^x t^a exp(-b t^2) dt (GK15, rel 1e-10), 2 active, N=1e6', t, [xq], [yq], [1.0 / sq],
        np.array([[a * 1.05, b * 0.95]]), [0, 1], [0, 0], reps=3, fit_iters=6)
    if only == '80':      # beyond 64 active parameters there is no fused kernel: gfh_k_sweep writes J, k_gram_block re-reads it (on record, not a BASELINE config)
      K = 20
      truth = M.gaussK_truth(K)
      n = 4_000_000
      x, y, s = M.make_single(M.gaussK_numpy(K), truth, n, 0.0, 100.0)
      run('p80: 20 skewed Gaussians, 80 active, N=4e6 (two-kernel path: J written, then re-read)', trace_model(M.make_model_gaussK(K), 4 * K), [x], [y], [1 / s],
        M.start_values(truth).reshape(1, 4 * K), list(range(4 * K)), [0] * (4 * K), reps=30, fit_iters=6)
    if only in (None, '5'):
      truth = M.gauss8_truth()
      x, y, s = M.make_single(M.gauss8_numpy, truth, 10_000_000, 0.0, 100.0)
      run('cfg5: 8 skewed Gaussians, 32 active, N=1e7', trace_model(M.model_gauss8, 32), [x], [y], [1 / s],
        M.start_values(truth).reshape(1, 32), list(range(32)), [0] * 32)


if __name__ == '__main__':
    main()
