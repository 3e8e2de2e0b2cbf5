"""The oracle's adaptive Gauss-Kronrod quadrature through AD (oracle/gadfit_oracle.c, restating numerical_integration.F90)
against CLOSED FORMS (tests/golden/quadrature_closed_forms.json, mpmath at 50 digits): every rule 15 ... 61
(gauss_kronrod_parameters.F90:74-617) and every kind of bound -- finite, (a, inf), (-inf, b), (-inf, inf), passive and
active -- for the value, the reverse-mode gradient (Leibniz terms AD:1637-1654) and the forward-mode (d, dd) (NI:425-437)."""
import json
import os

import numpy as np
import pytest

from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests import quadrature_cases as Q

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'quadrature_closed_forms.json')))
REL_ERROR = 1e-12          # requested of the adaptive rule
TOL = 1e-14                # asserted against the closed forms (observed: value 4e-16, gradient 6e-16, d 6e-16, dd 7e-16)


def tape_of(name, rule):
    t = trace_model(Q.CASES[name][0], len(Q.CASES[name][1]))
    t.set_integration(rel_error=REL_ERROR, rule=rule)
    return t


@pytest.mark.parametrize('rule', Q.RULES)
@pytest.mark.parametrize('name', sorted(Q.CASES))
def test_oracle_quadrature_vs_closed_form(name, rule):
    g = GOLD[name]
    n = len(g['values'])
    t = tape_of(name, rule)
    f, grad = orc.eval_reverse(t, 0.0, g['values'], [1] * n)
    assert abs(f - g['F']) <= TOL * abs(g['F'])
    scale = max(abs(v) for v in g['grad'])
    assert np.max(np.abs(np.asarray(grad[:n]) - g['grad'])) <= TOL * scale
    val, d, dd = orc.eval_forward(t, 0.0, g['values'], [1] * n, g['direction'], np.zeros(n))
    assert abs(val - g['F']) <= TOL * abs(g['F']) and abs(d - g['d']) <= TOL * max(abs(g['d']), scale)
    assert abs(dd - g['dd']) <= TOL * max(abs(g['dd']), scale)
    # passive parameters: value only, no derivative carried
    f0, grad0 = orc.eval_reverse(t, 0.0, g['values'], [0] * n)
    assert f0 == f or abs(f0 - f) <= 1e-15 * abs(f)
