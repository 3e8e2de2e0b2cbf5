#!/usr/bin/env python3
"""Random integrands that BRANCH on comparisons of AD variables (up to two nested comparisons in (t, q), each side its own random
expression) under random kinds of bounds, device against oracle: residuals, Jacobian, omega at 17 abscissas.
   python tools/probes/soak_random_integrand_branches.py 0 60"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GADFIT_HIP_CACHE', '/tmp/gadfit_soak_kcache')
import numpy as np
from gadfit_amd import _lib, ad, tape as T
from gadfit_amd.ad import integrate, exp, INFINITY
from oracle import binding as orc
from tests import test_gpu_random_models as R


def run(seed):
    rng0 = np.random.default_rng(60000 + seed)
    kind = int(rng0.integers(0, 4))

    def model(p, x):
        def integrand(t, q):
            rr = np.random.default_rng(70000 + seed)
            qq = list(q) + [q[0]] * (R.NP_ - len(q))
            body = R._rand_branching(rr, qq, t, 2)
            return exp(-(q[0] * t * t)) * (1.0 + 0.3 * ad.tanh(body))
        q = [p[0], p[1], p[2]]
        if kind == 0:
            return integrate(integrand, q, 0.1, x) + p[3]
        if kind == 1:
            return integrate(integrand, q, p[3] * 0.2, x * p[4])
        if kind == 2:
            return integrate(integrand, q, x * 0.5, INFINITY)
        return integrate(integrand, q, -INFINITY, x - p[3])
    sub = np.random.default_rng(80000 + seed)
    pars = sub.uniform(0.7, 1.6, size=(1, R.NP_))
    mask = sub.random(R.NP_) < 0.7
    if not mask.any():
        mask[0] = True
    active = [int(i) for i in np.nonzero(mask)[0]]
    xs = np.sort(sub.uniform(0.4, 2.5, size=17)); ys = sub.uniform(-1, 1, size=17); ws = sub.uniform(0.5, 2.0, size=17)
    V = T.Variants(model, R.NP_, configure=lambda t: t.set_integration(rel_error=1e-9))
    V.THETAS = tuple(np.linspace(0.01, 0.99, 41))          # a finer net than the default: the soak wants every path
    V.explore(xs, pars[0])
    p = orc.OracleProblem(V, [xs], [ys], [ws], pars, active, [0] * R.NP_)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    ctx = _lib.Context(0)
    try:
        ctx.set_model(V); ctx.set_data(xs, ys, ws, [0, xs.size])
        jac, dim = ctx.jacobian_indices(active, [0] * R.NP_)
        JTJ, JTr, chi2 = ctx.sweep(pars, active, jac, dim)
        J = ctx.jacobian(len(active)); res = ctx.residuals()
        delta = sub.uniform(-0.2, 0.2, size=dim)
        ctx.omega(pars, delta); om = ctx.omega_vector()
        nv = ctx.n_variants()
    finally:
        ctx.close()
    om0, _ = p.omega(delta, JT0)
    e = [np.max(np.abs(res - res0) / np.maximum(1.0, np.abs(res0))), np.max(np.abs(J - JT0[:, jac[0]]) / np.maximum(1.0, np.abs(JT0[:, jac[0]]))),
         np.max(np.abs(om - om0) / np.maximum(1.0, np.abs(om0)))]
    assert e[0] < 1e-11 and e[1] < 1e-9 and e[2] < 1e-8, (kind, len(V), e)
    return len(V), nv, e


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []; skipped = []; worst = [0.0, 0.0, 0.0]; tapes = []
t0 = time.time()
for seed in range(first, last):
    try:
        n, nv, e = run(seed)
        tapes.append(n); worst = [max(a, b) for a, b in zip(worst, e)]
    except Exception as ex:
        msg = str(ex)
        if 'none of the recordings covers' in msg or 'an integrand took a path' in msg:
            skipped.append(seed)                               # the sampling net missed a path: reported loudly by both, not a parity failure
        else:
            bad.append(seed); print('seed', seed, 'FAILED', type(ex).__name__, msg[:300], flush=True)
    if seed % 10 == 0:
        print('seed', seed, 'done, %.0f s' % (time.time() - t0), flush=True)
print('seeds %d..%d: %d failures %s; %d seeds whose integrands took a path the recordings missed (reported as such by oracle and device) %s; '
      'recordings per model %s; worst res / J / omega deviations %.1e %.1e %.1e' % (first, last - 1, len(bad), bad, len(skipped), skipped,
      sorted(set(tapes)), *worst))
sys.exit(1 if bad else 0)
