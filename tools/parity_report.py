#!/usr/bin/env python3
"""Observed parity of the device path against the CPU oracle on the BASELINE.json configurations at reduced N:
maximum relative error of res, J, JTJ, JTres, chi2, omega, J^T omega and of the fitted parameters after a fixed number of
LM iterations, with GADFIT_HIP_FAST_DIV = 1 (default: shared reciprocals) and = 0 (the reference's expression shapes).
Writes profiles/parity_r02.json; the tolerances asserted in tests/ are these numbers x 10 (rounded up), see the tests.

  python tools/parity_report.py [out.json]        (needs the GPU)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import models as M
from tests.golden import goldens as G


def configs():
    out = []
    x, y, s = M.make_single(M.exp2_numpy, M.EXP2_TRUTH, 200, 0.5, 100.0)
    out.append(('cfg1: 2-exponential, 200 points, 4 active', trace_model(M.model_exp2, 4), [x], [y], [1 / s], [M.start_values(M.EXP2_TRUTH)],
                list(range(4)), [0] * 4, dict(lambda_=1.0, accth=0.9, max_iter=5)))
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 20011, 0.0, 100.0)
    out.append(('cfg2: 4-exponential, 8 active (N = 20011)', trace_model(M.model_exp4, 8), [x], [y], [1 / s], [M.start_values(M.EXP4_TRUTH)],
                list(range(8)), [0] * 8, dict(lambda_=1.0, max_iter=5)))
    xs, ys, ss, truths = M.make_global7(8, [2500, 1777, 2500, 1, 3000, 2049, 2500, 1024])
    pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    out.append(('cfg3: global fit 8 datasets, 4 local + 3 global', trace_model(M.model_global7, 7), xs, ys, [1 / s for s in ss], pars,
                list(range(7)), [0, 0, 0, 0, 1, 1, 1], dict(lambda_=1.0, max_iter=4)))
    d = G.data()['2_integral_single']
    t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
    x = np.array(d['x_data']); y = np.array(d['y_data'])
    out.append(('cfg4: pi int_0^x t^a exp(-b t^2) dt, GK15 rel 1e-10, 2 active (150 points)', t, [x], [y], [np.ones_like(y)], [[10.0, 1.0]],
                [0, 1], [0, 0], dict(lambda_=10.0, accth=0.9, max_iter=4)))
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, 20011, 0.0, 100.0)
    out.append(('cfg5: 8 skewed Gaussians, 32 active (N = 20011)', trace_model(M.model_gauss8, 32), [x], [y], [1 / s], [M.start_values(truth)],
                list(range(32)), [0] * 32, dict(lambda_=1.0, max_iter=5)))
    return out


def measure(name, tape, xs, ys, ws, pars, active, is_global, fit_opts):
    p = orc.OracleProblem(tape, xs, ys, ws, pars, active, is_global)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    chi0, _ = p.chi2()
    c = _lib.Context(0)
    c.set_model(tape)
    c.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), p.dp)
    jac, dim = c.jacobian_indices(active, is_global)
    JTJ, JTr, chi2 = c.sweep(p.pars, active, jac, dim)
    res = c.residuals(); J = c.jacobian(len(active))
    Jd = np.zeros_like(JT0)
    for d in range(p.nd):
        sl = slice(p.dp[d], p.dp[d + 1])
        Jd[sl][:, jac[d]] = J[sl]
    colmax = np.max(np.abs(JT0), axis=0, keepdims=True) + 1e-300
    out = {
        'res_rel_to_max': float(np.max(np.abs(res - res0)) / np.max(np.abs(res0))),
        'J_rel_to_column_max': float(np.max(np.abs(Jd - JT0) / colmax)),
        'J_rel_entrywise_floor_1e-6_of_column_max': float(np.max(np.abs(Jd - JT0) / np.maximum(np.abs(JT0), 1e-6 * colmax))),
        'JTJ_rel_to_sqrt_diag_product': float(np.max(np.abs(JTJ - JTJ0) / (np.sqrt(np.outer(np.diag(JTJ0), np.diag(JTJ0))) + 1e-300))),
        'JTres_rel': float(np.max(np.abs(JTr - JTr0) / (np.sqrt(np.diag(JTJ0) * chi0) + 1e-300))),
        'chi2_sweep_rel': float(abs(chi2 - chi0) / chi0),
        'chi2_kernel_rel': float(abs(c.chi2(p.pars) - chi0) / chi0),
    }
    delta1 = orc.potr(JTJ0 + np.diag(np.diag(JTJ0)), JTr0)
    om0, jto0 = p.omega(delta1, JT0)
    jto = c.omega(p.pars, delta1)
    om = c.omega_vector()
    out['omega_rel_to_max'] = float(np.max(np.abs(om - om0)) / max(1e-300, np.max(np.abs(om0))))
    out['JTomega_rel_to_max'] = float(np.max(np.abs(jto - jto0)) / np.max(np.abs(jto0)))
    # fitted parameters after a fixed number of iterations (not "converged")
    opts32 = {k: (np.float32(v) if k in ('lambda_', 'accth') else v) for k, v in fit_opts.items()}
    r0 = p.fit(**opts32)
    got, r = c.fit(np.array(pars, dtype=float).reshape(p.nd, -1), active, is_global,
                   **{k: (float(np.float32(v)) if k in ('lambda_', 'accth') else v) for k, v in fit_opts.items()})
    out['fit'] = {'iterations': r.iterations, 'same_pass_counts_as_oracle': bool((r.iterations, r.n_sweeps, r.n_chi2, r.n_omega) ==
                                                                             (r0.iterations, r0.n_sweeps, r0.n_chi2, r0.n_omega)),
                  'fitted_parameters_rel': float(np.max(np.abs(got - p.pars) / np.abs(p.pars))),
                  'final_chi2_rel': float(abs(r.chi2 - r0.chi2) / r0.chi2)}
    c.close()
    return out


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'parity_r02.json')
    report = {'what': 'device (libgadfit_hip.so, MI355X) against oracle/gadfit_oracle.c on identical inputs; maxima over all points / entries',
              'north_star_tolerance_on_fitted_parameters': 1e-10,
              'reading': 'per-pass quantities agree to 1e-15 ... 2e-14 (fp64 throughout; FMA contraction, shared reciprocals and libm '
                         'differ); the fitted parameters after 4-5 LM iterations differ by up to 2.5e-12 at cfg 5: every iteration solves '
                         '(JTJ + lambda DTD) delta = JTres, whose conditioning (1e3 ... 1e4 for the 32 correlated Gaussian parameters) '
                         'amplifies the 1e-15 perturbation of JTJ / JTres, and the next sweep starts from the perturbed parameters',
              'results': {}}
    for fd in ('1', '0'):
        os.environ['GADFIT_HIP_FAST_DIV'] = fd
        key = 'FAST_DIV=%s (%s)' % (fd, 'default: one reciprocal per denominator' if fd == '1' else "the reference's division forms")
        report['results'][key] = {}
        for cfg in configs():
            report['results'][key][cfg[0]] = measure(*cfg)
            print(key, cfg[0], json.dumps(report['results'][key][cfg[0]]), flush=True)
    os.environ.pop('GADFIT_HIP_FAST_DIV', None)
    worst = {}
    for key, per in report['results'].items():
        w = {}
        for m in per.values():
            for k, v in m.items():
                if k == 'fit':
                    w['fitted_parameters_rel'] = max(w.get('fitted_parameters_rel', 0.0), v['fitted_parameters_rel'])
                else:
                    w[k] = max(w.get(k, 0.0), v)
        worst[key] = w
    report['worst_over_configs'] = worst
    with open(dst, 'w') as f:
        json.dump(report, f, indent=1)
    print('wrote', dst)


if __name__ == '__main__':
    main()
