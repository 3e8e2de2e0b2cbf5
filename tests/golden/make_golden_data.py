#!/usr/bin/env python3
"""Extract the DATA (inputs) and GOLDEN VALUES (expected outputs) that the reference's own
known-answer tests hold, into small JSON fixtures.  Runs only in the build container (reads
/root/reference); the outputs under tests/golden/ are committed and travel to the GPU box.

Sources:
  fortran/tests/{1_gaussian,2_integral_single,3_integral_double,4_multiple_curves}_data.F90
      -> x/y(/weights) arrays
  fortran/tests/example_data1, example_data2 -> two-column text data
  c++/tests/fixtures.h                       -> fix_d and x/y arrays of the C++ LM-solver tests
Golden constants (fit results, AD values) are written by hand in tests/golden/goldens.py
with the reference file:line they come from.
"""
import json, re, sys, os

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def fortran_arrays(path):
    src = open(path).read()
    out = {}
    for m in re.finditer(r'real\(kp\), parameter :: (\w+)\(\*\) = \[(.*?)\]', src, re.S):
        vals = [float(v.replace('d', 'e').replace('_kp', ''))
                for v in re.findall(r'[-+]?\d+\.?\d*(?:[de][-+]?\d+)?(?:_kp)?', m.group(2).replace('&', ' '))]
        out[m.group(1)] = vals
    return out


def main():
    fx = {}
    for name in ['1_gaussian', '2_integral_single', '3_integral_double', '4_multiple_curves']:
        fx[name] = fortran_arrays('%s/fortran/tests/%s_data.F90' % (REF, name))
        print(name, {k: len(v) for k, v in fx[name].items()})
    for name in ['example_data1', 'example_data2']:
        rows = [list(map(float, l.split())) for l in open('%s/fortran/tests/%s' % (REF, name)) if l.strip()]
        fx[name] = {'x': [r[0] for r in rows], 'y': [r[1] for r in rows]}
        print(name, len(rows))
    # c++/tests/fixtures.h: fix_d and the two decay curves of the C++ LM-solver tests
    src = open('%s/c++/tests/fixtures.h' % REF).read()
    cx = {}
    for m in re.finditer(r'(?:constexpr std::array|const std::vector<double>) (\w+)\s*\{(.*?)\};', src, re.S):
        if m.group(1) in ('fix_d', 'x_data_1', 'y_data_1', 'x_data_2', 'y_data_2', 'x_data_single', 'y_data_single',
                          'x_data_double', 'y_data_double', 'weights_double'):
            cx[m.group(1)] = [float(v) for v in re.findall(r'[-+]?\d+\.?\d*(?:e[-+]?\d+)?', m.group(2))]
    fx['cxx_lm_solver'] = cx
    print('cxx_lm_solver', {k: len(v) for k, v in cx.items()})
    json.dump(fx, open(os.path.join(OUT, 'reference_test_data.json'), 'w'))


if __name__ == '__main__':
    main()
