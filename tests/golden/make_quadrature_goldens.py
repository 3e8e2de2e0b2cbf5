#!/usr/bin/env python3
"""Writes tests/golden/quadrature_closed_forms.json: value, gradient and second directional derivative of the closed forms
of tests/quadrature_cases.py at 50 digits (mpmath), rounded to double.  Run in the build container; the JSON is the fixture."""
import json
import os
import sys

import mpmath as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import quadrature_cases as Q

mp.mp.dps = 50
out = {}
for name, (_, values, direction) in Q.CASES.items():
    F = Q.closed_form(name, mp)
    # the decimal literals the tests use, to 50 digits
    p = [mp.mpf(repr(v)) for v in values]
    d = [mp.mpf(repr(v)) for v in direction]
    grad = [mp.diff(lambda t, k=k: F([p[j] + (t if j == k else 0) for j in range(len(p))]), 0) for k in range(len(p))]
    dd = mp.diff(lambda s: F([p[j] + s * d[j] for j in range(len(p))]), 0, 2)
    d1 = mp.diff(lambda s: F([p[j] + s * d[j] for j in range(len(p))]), 0, 1)
    out[name] = {'values': values, 'direction': direction, 'F': float(F(p)), 'grad': [float(g) for g in grad],
                 'd': float(d1), 'dd': float(dd)}
with open(os.path.join(ROOT, 'tests', 'golden', 'quadrature_closed_forms.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
