#!/usr/bin/env python3
"""debug: kinked INNER integrand of a double integral, device against oracle pass by pass"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib, tape as T
from gadfit_amd.ad import integrate, exp
from oracle import binding as orc
from tests import branching as B
truth = B.KINKED_TRUTH
x = np.linspace(0.3, 4.0, 5)

def make(kind):
    def model(p, x):
        def inner(u, q):
            if kind == 'plain':
                return q[0] * (1.0 + 0.5 * (u - q[1])) + 0.0 * q[2]
            if kind == 'one-sided':
                if u > q[1] + 100.0:
                    return q[0] * exp(-((u - q[1]) / q[2]))
                return q[0] * (1.0 + 0.5 * (u - q[1])) + 0.0 * q[2]
            if u > q[1]:
                return q[0] * exp(-((u - q[1]) / q[2]))
            return q[0] * (1.0 + 0.5 * (u - q[1]))
        def outer(t, q):
            return exp(-(0.3 * t)) * integrate(inner, [q[0], q[1], q[2]], 0.0, t)
        return integrate(outer, [p[0], p[1], p[2]], 0.0, x) + p[3]
    return model

for kind in ('plain', 'one-sided', 'kink'):
    V = T.Variants(make(kind), 4, configure=lambda t: t.set_integration(rel_error=1e-5, rel_error_inner=1e-8, dbl=True))
    V.explore(x, truth)
    y = np.zeros_like(x); w = np.ones_like(x)
    p = orc.OracleProblem(V, [x], [y], [w], [truth], [0, 1, 2, 3], [0] * 4)
    JTJ0, JTr0, res0, JT0 = p.sweep(want_J=True)
    c = _lib.Context(0)
    c.set_model(V); c.set_data(x, y, w, [0, x.size])
    try:
        jac, dim = c.jacobian_indices([0, 1, 2, 3], [0] * 4)
        c.sweep([truth], [0, 1, 2, 3], jac, dim)
        print(kind, len(V), 'sweep ok; res', np.max(np.abs(c.residuals() - res0)), 'J', np.max(np.abs(c.jacobian(4) - JT0)))
    except Exception as e:
        print(kind, len(V), 'sweep FAILED', str(e)[:60], c.counters())
    c.close()
