import os, sys, tempfile, numpy as np
sys.path.insert(0, '/root/repo')
from tests import fortran_fuzz as FZ
from tests import test_gpu_fortran_fuzz as T
from gadfit_amd.ad import trace_model
from gadfit_amd import _lib
from oracle import binding as orc
seed = int(sys.argv[1])
c = FZ.make_layout_case(seed)
root, nd = c['root'], c['nd']
tape = trace_model(lambda p, x: root.fn(p, x), FZ.NP_)
rng = np.random.default_rng(88000 + seed)
xs, ys, ss = [], [], []
for d in range(nd):
    n = int(rng.integers(50, 300))
    x = np.sort(rng.uniform(0.3, 1.6, size=n))
    y = np.array([orc.eval_reverse(tape, float(v), c['truth'][d], [0] * FZ.NP_)[0] for v in x])
    y = (np.abs(y) + 1.0) * (1.0 + 0.01 * rng.standard_normal(n))
    sg = rng.uniform(0.5, 2.0, size=n)
    path = '/tmp/dbg_%d.txt' % d
    with open(path, 'w') as fh:
        for k in range(n):
            fh.write(('%.17e %.17e %.17e\n' % (x[k], y[k], sg[k])) if c['mode'] == 'USER' else ('%.17e %.17e\n' % (x[k], y[k])))
    cols = np.loadtxt(path, unpack=True)
    xs.append(cols[0]); ys.append(cols[1]); ss.append(cols[2] if c['mode'] == 'USER' else None)
ws = [orc.init_weights(getattr(orc, c['mode']), y, s) if s is not None else orc.init_weights(getattr(orc, c['mode']), y) for y, s in zip(ys, ss)]
more = dict(c['more']); use_ad = more.pop('use_ad', True)
kw = dict(lambda_=np.float32(c['lam']), max_iter=c['max_iter'])
if c['accth'] is not None: kw['accth'] = np.float32(c['accth'])
for k, v in more.items(): kw[k] = int(v) if isinstance(v, (bool, int)) else np.float32(v)
print('case', {k: c[k] for k in ('nd','mode','is_global','active','accth','lam','max_iter','more')})
N = sum(len(x) for x in xs)
for mi in range(1, kw['max_iter'] + 1):
    p = orc.OracleProblem(tape, xs, ys, ws, c['start'], c['active'], c['is_global'], use_ad=use_ad)
    k2 = dict(kw); k2['max_iter'] = mi
    r0 = p.fit(**k2)
    print('oracle max_iter', mi, 'iterations', r0.iterations, 'exit', r0.exit_reason, 'chi2 %.17g' % r0.chi2, 'chi2/dof %.6g' % (r0.chi2 / r0.dof), 'lambda', r0.lambda_, 'pars', p.pars[0][c['active']])
ctx = _lib.Context(0)
ctx.set_model(tape)
X = np.concatenate(xs); Y = np.concatenate(ys); W = np.concatenate(ws)
dp = np.concatenate([[0], np.cumsum([len(x) for x in xs])])
ctx.set_data(X, Y, W, list(dp))
if not use_ad: ctx.set_use_ad(False)
for mi in range(1, kw['max_iter'] + 1):
    k2 = {k: (float(v) if isinstance(v, np.floating) else v) for k, v in kw.items()}; k2['max_iter'] = mi
    out, r = ctx.fit(c['start'], c['active'], c['is_global'], **k2)
    print('device max_iter', mi, 'iterations', r.iterations, 'exit', r.exit_reason, 'chi2 %.17g' % r.chi2, 'lambda', r.lambda_, 'pars', out[0][c['active']])
ctx.close()
