"""debug: device vs oracle fit of a branching model, iteration by iteration"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib, tape as T
from oracle import binding as orc
from tests import branching as B

truth = np.array([1.0, 4.0, 12.0])
x, y, s = B.make_data(B.par_order_numpy, truth, 2000)
start = np.array([2.6, 2.4, 11.0])
Vfull = T.Variants(B.model_par_order, 3); Vfull.add_point(1.0, start); Vfull.add_point(1.0, truth)
ctx = _lib.Context(0)
for it in range(1, 13):
    p = orc.OracleProblem(Vfull, [x], [y], [1.0 / s], [start], [0, 1, 2], [0] * 3)
    r0 = p.fit(lambda_=1.0, max_iter=it)
    V = T.Variants(B.model_par_order, 3); V.add_point(1.0, start)
    ctx.set_model(V); ctx.set_data(x, y, 1.0 / s, [0, x.size])
    out, r = ctx.fit([start], [0, 1, 2], [0] * 3, lambda_=1.0, max_iter=it)
    print(it, (r0.iterations, r0.n_chi2, r0.exit_reason), (r.iterations, r.n_chi2, r.exit_reason),
          'chi2', r0.chi2, r.chi2, 'lambda', r0.lambda_, r.lambda_, 'dp', np.max(np.abs(out - p.pars) / np.abs(p.pars)), ctx.n_variants(), len(ctx.unseen_log))
    print('   oracle', p.pars[0], '\n   device', out[0])
