#!/usr/bin/env python3
"""Writes the data files and the expected parameters of the Fortran tests of branching eval() bodies
(tests/fortran/fit_piecewise.F90, fit_hidden_branch.F90, fit_clip_unseen.F90, fit_integral_branch.F90, fit_rare_branch.F90, fit_kinked_integrand.F90).  The expected values are fits of the CPU oracle
(oracle/gadfit_oracle.c, which takes the branch per point as the reference's eval() does) to the same data with the same options;
the Fortran programs reach the device through the recorder of gadfit_amd/fortran/ad.F90 and must land on them.
Run from the repository root:  python tests/golden/make_branching_goldens.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gadfit_amd import tape as T          # noqa: E402
from oracle import binding as orc         # noqa: E402
from tests import branching as B          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def write(name, x, y, s):
    with open(os.path.join(HERE, name), 'w') as f:
        for a, b, c in zip(x, y, s):
            f.write('%.17g %.17g %.17g\n' % (a, b, c))


def main():
    out = {}
    # 1. piecewise with an active breakpoint and a plain-real factor on the second segment; geodesic acceleration as the reference's tests
    truth = B.PIECEWISE2_TRUTH
    x, y, s = B.make_data(B.piecewise_aux_numpy, truth, 600)
    write('piecewise_aux_xys.txt', x, y, s)
    start = np.array([4.2, 34.0, 0.088, 10.0])
    V = T.Variants(B.model_piecewise_aux, 4); V.explore(x, start); V.explore(x, truth)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r = p.fit(lambda_=1.0, max_iter=6, accth=0.9)
    out['piecewise_aux'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 2. the same shape with the breakpoint a LITERAL of eval() (control flow on the plain real x): parameter 2 passive at 37.3
    x, y, s = B.make_data(B.piecewise2_numpy, truth, 500)
    write('piecewise2_xys.txt', x, y, s)
    start = np.array([4.2, 37.3, 0.088, 10.0])
    V = T.Variants(B.model_piecewise2, 4); V.explore(x, start)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 2, 3], [0] * 4)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['hidden_branch'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 3. the clipped ramp whose upper clip is met for the first time inside the fit
    x, y, s = B.make_data(B.clip_numpy, B.CLIP_TRUTH, 800)
    write('clip_xys.txt', x, y, s)
    start = np.array([0.07, 17.0, 6.0, 1.3])
    V = T.Variants(B.model_clip, 4); V.explore(x, start); V.explore(x, B.CLIP_TRUTH)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 3], [0] * 4)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['clip_unseen'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 4. a quadrature on either side of a fitted breakpoint (the second call site with an active upper bound)
    truth = B.INTEGRAL_THEN_LINE_TRUTH
    x = np.linspace(0.05, 4.0, 400)
    f = B.integral_then_line_numpy(truth, x)
    s = 0.002 * (1.0 + np.abs(f))
    from tests import models as M
    y = f + s * M.normal(x.size, M.SEED + 3)
    write('integral_branch_xys.txt', x, y, s)
    start = np.array([2.2, 0.8, 1.8, -0.04])
    V = T.Variants(B.model_integral_then_line, 4, configure=lambda t: t.set_integration(rel_error=1e-10))
    V.explore(x, start); V.explore(x, truth)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r = p.fit(lambda_=1.0, max_iter=6, accth=0.9)
    out['integral_branch'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 5. a window of two points among 400001 that the sampled recordings of gadf_fit miss (data by formula, no file)
    x, y, w0, w1 = B.rare_data()
    assert np.count_nonzero((x > w0) & (x < w1)) == 2
    start = np.array([4.5, 22.0, 1.2, w0, w1, 0.1])
    V = T.Variants(B.model_rare, 6); V.explore([x[0], x[200001], x[-1]], start)
    assert len(V) == 3
    p = orc.OracleProblem(V, [x], [y], [np.ones_like(x)], [start], [0, 1, 2, 5], [0] * 6)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['rare_branch'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 6. an integrand that compares AD variables: a kink at a fitted position inside the function handed to integrate()
    truth = B.KINKED_TRUTH
    x = np.linspace(0.05, 4.0, 400)
    f = B.kinked_numpy(truth, x)
    s = 0.002 * (1.0 + np.abs(f))
    from tests import models as M2
    y = f + s * M2.normal(x.size, M2.SEED + 9)
    write('kinked_integrand_xys.txt', x, y, s)
    start = truth * np.array([1.05, 0.93, 1.06, 0.8])
    V = T.Variants(B.model_kinked_integrand, 4, configure=lambda t: t.set_integration(rel_error=1e-10))
    V.explore(x[::10], start); V.explore(x[::10], truth)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 2, 3], [0] * 4)
    r = p.fit(lambda_=1.0, max_iter=6, accth=0.9)
    out['kinked_integrand'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 7. the same data, the kink starting BEYOND every range of integration (one path through the integrand at the start; the other
    # is first met on the device inside the fit); the decay time has no influence while the kink lies outside: passive
    start = truth * np.array([1.02, 3.6, 1.0, 0.9])
    V = T.Variants(B.model_kinked_integrand, 4, configure=lambda t: t.set_integration(rel_error=1e-10))
    V.explore(x[::10], start); V.explore(x[::10], truth)
    p = orc.OracleProblem(V, [x], [y], [1.0 / s], [start], [0, 1, 3], [0] * 4)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['kinked_far'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 8. a window three points wide whose bounds are PLAIN REALS of eval() (tests/fortran/fit_narrow_window.F90): invisible to operator
    # overloading, and no abscissa of a 2^17-point sample of the 400001 falls inside -- only a recorder that visits every point finds the
    # path.  The oracle's arithmetic is the same with the bounds as passive parameters.
    x, y, w0, w1 = B.rare_data(inside=3)
    assert np.count_nonzero((x > w0) & (x < w1)) == 3
    start = np.array([4.5, 22.0, 1.2, w0, w1, 0.1])
    V = T.Variants(B.model_rare, 6); V.explore([x[0], x[200002], x[-1]], start)
    assert len(V) == 3
    p = orc.OracleProblem(V, [x], [y], [np.ones_like(x)], [start], [0, 1, 2, 5], [0] * 6)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['narrow_window'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 9. an integrand that takes x (affine: 1 + 0.1 x) and a real function of x (sin(0.3 x): a per-point column) from the enclosing
    # eval() without passing them through pars(:)
    from gadfit_amd.ad import trace_model
    x, y, saux = B.integrand_module_x_data()
    start = np.array([1.1, 0.8, 0.0])
    t = trace_model(B.model_integrand_module_x, 3); t.set_integration(rel_error=1e-10)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3, aux=saux[None, :])
    r = p.fit(lambda_=1.0, max_iter=6, accth=0.9)
    out['integrand_module_x'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 10. a real formed from the %val of a fitted parameter (no derivative through it: the Jacobian is the reference's, inexact in the
    # same way; the residuals are exact, so the fit still converges on the minimum)
    x, y = B.param_val_data()
    start = np.array([4.5, 22.0, 1.2])
    t = trace_model(B.model_param_val, 3)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3)
    r = p.fit(lambda_=1.0, max_iter=8)
    out['param_val'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 11. ... formed INSIDE an integrand from the integrand's own pars(:) (tests/fortran/fit_integrand_param_val.F90)
    x, y = B.integrand_param_val_data()
    start = np.array([1.1, 0.8, 0.0])
    t = trace_model(B.model_integrand_param_val, 3); t.set_integration(rel_error=1e-10)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['integrand_param_val'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 12. ... formed from the %val of a fitted parameter AND the abscissa (tests/fortran/fit_param_val_x.F90)
    x, y = B.param_val_x_data()
    start = np.array([2.5, 0.9, 0.3])
    t = trace_model(B.model_param_val_x, 3)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['param_val_x'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 13. ... the same with geodesic acceleration (STEP 3 at the parameters of the sweep after a trial chi2() elsewhere)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3)
    r = p.fit(lambda_=1.0, max_iter=6, accth=0.9)
    out['param_val_x_accel'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 14. ... under use_ad = .false.: the forward differences of fitfunction.F90:155-174 evaluate the model at p + step e_j, where the
    # real has moved with the parameter (tests/fortran/fit_param_val_x.F90 'fd': sets of columns, gfh_set_fd_column_sets)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3, use_ad=False)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['param_val_x_fd'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    # 15. a black-box eval(): every operation in plain real arithmetic on %val, the result assigned to the advar -- what use_ad = .false.
    # exists for (tests/fortran/fit_param_val_x.F90 'blackbox'); to finite differences it is the plain function
    t = trace_model(B.model_param_x_plain, 3)
    p = orc.OracleProblem(t, [x], [y], [np.ones_like(x)], [start], [0, 1, 2], [0] * 3, use_ad=False)
    r = p.fit(lambda_=1.0, max_iter=6)
    out['param_x_blackbox_fd'] = dict(start=start.tolist(), pars=p.pars[0].tolist(), iterations=r.iterations, chi2=r.chi2)
    json.dump(out, open(os.path.join(HERE, 'branching_goldens.json'), 'w'), indent=1)
    for k, v in out.items():
        print(k, v['iterations'], ' '.join('%.17g' % q for q in v['pars']), 'chi2 %.17g' % v['chi2'])


if __name__ == '__main__':
    main()
