#!/usr/bin/env python3
"""debug: iterated integral int_0^x w(t) int_0^t f(u) du dt on the device (sweep) -- which ingredient breaks it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import integrate, exp, trace_model
from oracle import binding as orc
truth = np.array([1.3, 1.2, 0.8, 0.1])
x = np.linspace(0.3, 4.0, 5)

def make(kind):
    def model(p, x):
        def inner(u, q):
            return q[0] * (1.0 + 0.5 * (u - q[1])) * exp(-(q[2] * u))
        def outer(t, q):
            if kind == 'upper = t':
                return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, t)
            if kind == 'upper = 2 q1':
                return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, 2.0 * q[1])
            if kind == 'upper = t, lower = -1':
                return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], -1.0, t)
            return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, t + 1.0)
        if kind.startswith('outer from'):
            a = float(kind.split()[-1])
            return integrate(lambda t, q: exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, t), [p[0], p[1], p[2]], a, x) + p[3]
        return integrate(outer, [p[0], p[1], p[2]], 0.0, x) + p[3]
    return model

for kind in ('upper = t', 'outer from 1e-3', 'outer from 1e-9', 'outer from 0.0'):
    t = trace_model(make(kind), 4); t.set_integration(rel_error=1e-5, rel_error_inner=1e-8, dbl=True)
    y = np.zeros_like(x); w = np.ones_like(x)
    p = orc.OracleProblem(t, [x], [y], [w], [truth], [0, 1, 2, 3], [0] * 4)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    for env in ({}, {'GADFIT_HIP_FAST_DIV': '0'}):
        os.environ.update(env)
        c = _lib.Context(0)
        for k in env: os.environ.pop(k)
        c.set_model(t); c.set_data(x, y, w, [0, x.size])
        try:
            jac, dim = c.jacobian_indices([0, 1, 2, 3], [0] * 4)
            try:
                chi = c.chi2([truth]); print('    chi2 ok', abs(chi - float(res0 @ res0)), end='')
            except Exception as e:
                print('    chi2 FAILED', end='')
            try:
                c.omega([truth], np.array([0.1, -0.1, 0.05, 0.02])); print('  omega ok', end='')
            except Exception as e:
                print('  omega FAILED', end='')
            print()
            c.sweep([truth], [0, 1, 2, 3], jac, dim)
            print('%-24s %-28s sweep ok; res %.1e J %.1e' % (kind, env, np.max(np.abs(c.residuals() - res0)), np.max(np.abs(c.jacobian(4) - JT0))), flush=True)
        except Exception as e:
            print('%-24s %-28s sweep FAILED %s' % (kind, env, str(e)[:40]), flush=True)
        c.close()
