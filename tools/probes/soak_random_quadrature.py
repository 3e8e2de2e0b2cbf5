#!/usr/bin/env python3
"""Random integrands through the adaptive quadrature on the device against the oracle: integrand = envelope(t) x a random expression
over the operator set in (t, q), random kind of bounds (finite with the upper bound following x, active bounds, (a, inf), (-inf, b),
(-inf, inf)), random Gauss-Kronrod rule; residuals, Jacobian, omega at 23 abscissas.
   python tools/probes/soak_random_quadrature.py 0 100"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GADFIT_HIP_CACHE', '/tmp/gadfit_soak_kcache')
import numpy as np
from gadfit_amd import _lib, ad
from gadfit_amd.ad import trace_model, integrate, exp, INFINITY
from oracle import binding as orc
from tests import test_gpu_random_models as R

RULES = [15, 21, 31, 41, 51, 61]


def run(seed):
    rng0 = np.random.default_rng(20000 + seed)
    kind = int(rng0.integers(0, 6))
    rule = RULES[int(rng0.integers(0, 6))]

    def model(p, x):
        r = np.random.default_rng(30000 + seed)

        def integrand(t, q):
            rr = np.random.default_rng(40000 + seed)
            body = R._rand_expr(rr, list(q) + [q[0]] * (R.NP_ - len(q)), t, 2)      # random expression in (t, q)
            return exp(-(q[0] * t * t)) * (1.0 + 0.3 * ad.tanh(body))               # envelope: integrable on every range
        q = [p[0], p[1], p[2]]
        if kind == 0:
            return integrate(integrand, q, 0.1, x) + p[3]
        if kind == 1:
            return integrate(integrand, q, p[3] * 0.2, x * p[4])
        if kind == 2:
            return integrate(integrand, q, x * 0.5, INFINITY)
        if kind == 3:
            return integrate(integrand, q, -INFINITY, x - p[3])
        if kind == 4:
            return integrate(integrand, q, -INFINITY, INFINITY) * x
        return p[4] * integrate(integrand, q, p[3] * 0.1, INFINITY)
    tape = trace_model(model, R.NP_)
    tape.set_integration(rel_error=1e-9, rule=rule)
    sub = np.random.default_rng(50000 + seed)
    pars = sub.uniform(0.7, 1.6, size=(1, R.NP_))
    mask = sub.random(R.NP_) < 0.7
    if not mask.any():
        mask[0] = True
    active = [int(i) for i in np.nonzero(mask)[0]]
    xs = np.sort(sub.uniform(0.4, 2.5, size=23)); ys = sub.uniform(-1, 1, size=23); ws = sub.uniform(0.5, 2.0, size=23)
    p = orc.OracleProblem(tape, [xs], [ys], [ws], pars, active, [0] * R.NP_)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    ctx = _lib.Context(0)
    try:
        ctx.set_model(tape); ctx.set_data(xs, ys, ws, [0, xs.size])
        jac, dim = ctx.jacobian_indices(active, [0] * R.NP_)
        JTJ, JTr, chi2 = ctx.sweep(pars, active, jac, dim)
        J = ctx.jacobian(len(active)); res = ctx.residuals()
        delta = sub.uniform(-0.2, 0.2, size=dim)
        ctx.omega(pars, delta); om = ctx.omega_vector()
        chi_k = ctx.chi2(pars)
    finally:
        ctx.close()
    om0, _ = p.omega(delta, JT0)
    e = [np.max(np.abs(res - res0) / np.maximum(1.0, np.abs(res0))), np.max(np.abs(J - JT0[:, jac[0]]) / np.maximum(1.0, np.abs(JT0[:, jac[0]]))),
         np.max(np.abs(om - om0) / np.maximum(1.0, np.abs(om0)))]
    assert np.all(np.isfinite(res0)) and np.all(np.isfinite(JT0)), 'oracle not finite'
    assert e[0] < 1e-12 and e[1] < 2e-11 and e[2] < 2e-10, (kind, rule, e)
    assert abs(chi_k - chi2) <= 1e-12 * chi2
    return kind, rule, e


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []; worst = [0.0, 0.0, 0.0]
t0 = time.time()
for seed in range(first, last):
    try:
        k, r, e = run(seed)
        worst = [max(a, b) for a, b in zip(worst, e)]
    except Exception as ex:
        bad.append(seed); print('seed', seed, 'FAILED', type(ex).__name__, str(ex)[:300], flush=True)
    if seed % 10 == 0:
        print('seed', seed, 'done, %.0f s' % (time.time() - t0), flush=True)
print('seeds %d..%d: %d failures %s; worst res / J / omega deviations %.1e %.1e %.1e' % (first, last - 1, len(bad), bad, *worst))
sys.exit(1 if bad else 0)
