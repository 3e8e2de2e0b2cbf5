#!/usr/bin/env python3
"""Does GADFIT_HIP_KEEP_WARM (the upload thread keeps the part busy while the host records eval()) buy anything on the first gadf_fit?
10 runs each way of tests/fortran/bench_headline at N = 1e7, interleaved; medians of the first gadf_fit and of its LM loop.
VERDICT r4 item 7: a product default needs a measured gain on file; default off below 5 %."""
import json, os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, 'tests', 'fortran', 'build', 'bench_headline')
n = sys.argv[1] if len(sys.argv) > 1 else '10000000'
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
res = {'0': {'first': [], 'loop': []}, '1': {'first': [], 'loop': []}}
for k in range(runs):
    for kw in ('0', '1'):
        p = subprocess.run([exe, n, '10'], capture_output=True, text=True, env=dict(os.environ, GADFIT_HIP_KEEP_WARM=kw, GADFIT_HIP_SETUP_TIMES='1'))
        out = p.stdout + p.stderr
        m1 = re.search(r'first gadf_fit\s*:\s*([0-9.]+) ms', out); m2 = re.search(r'of which the LM loop\s*([0-9.]+)\)', out)
        if m1 and m2:
            res[kw]['first'].append(float(m1.group(1))); res[kw]['loop'].append(float(m2.group(1)))
out = {'points': int(n), 'runs_each': runs}
for kw in ('0', '1'):
    out['keep_warm_' + kw] = {'first_gadf_fit_ms_median': statistics.median(res[kw]['first']), 'lm_loop_ms_median': statistics.median(res[kw]['loop']),
                              'first_gadf_fit_ms': res[kw]['first'], 'lm_loop_ms': res[kw]['loop']}
a, b = out['keep_warm_0'], out['keep_warm_1']
out['gain_of_keep_warm'] = {'first_gadf_fit': 1.0 - b['first_gadf_fit_ms_median'] / a['first_gadf_fit_ms_median'],
                            'lm_loop': 1.0 - b['lm_loop_ms_median'] / a['lm_loop_ms_median'],
                            'lm_loop_ms_saved': a['lm_loop_ms_median'] - b['lm_loop_ms_median']}
print(json.dumps(out))
