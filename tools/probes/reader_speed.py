#!/usr/bin/env python3
"""The text reader of gadf_add_dataset(path): libgadfit_hip's (reader.cpp, threads) against Fortran list-directed input the way the
reference reads (tests/fortran/list_directed_reader.F90, two passes) and numpy.loadtxt.   python tools/probes/reader_speed.py [lines]"""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from gadfit_amd import _lib
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
d = tempfile.mkdtemp()
path = os.path.join(d, 'data.txt')
x = np.linspace(0.0, 100.0, n); y = 5 * np.exp(-x / 20) + 1; s = 0.01 * (1 + np.abs(y))
t0 = time.perf_counter()
with open(path, 'w') as f:
    f.write('# x y sigma\n')
    blk = 100000
    for b in range(0, n, blk):
        f.write(''.join('%.17g %.17g %.17g\n' % t for t in zip(x[b:b + blk], y[b:b + blk], s[b:b + blk])))
print('wrote %d lines, %.0f MB in %.1f s' % (n, os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)
for threads in ('1', '4', '16', ''):
    if threads:
        os.environ['GADFIT_HIP_READ_THREADS'] = threads
    else:
        os.environ.pop('GADFIT_HIP_READ_THREADS', None)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); a, b, c = _lib.read_columns(path, 3); best = min(best, time.perf_counter() - t0)
    print('reader.cpp, threads %-8s %8.1f ms  (%.0f MB/s)' % (threads or 'default', 1e3 * best, os.path.getsize(path) / 1e6 / best), flush=True)
assert np.array_equal(a, x) and np.array_equal(b, y) and np.array_equal(c, s)
exe = os.path.join(d, 'ldr')
subprocess.run(['/opt/rocm/bin/amdflang', '-O2', os.path.join(ROOT, 'tests', 'fortran', 'list_directed_reader.F90'), '-o', exe], check=True, capture_output=True)
t0 = time.perf_counter(); subprocess.run([exe, path, '3'], stdout=subprocess.DEVNULL, check=True); dt = time.perf_counter() - t0
print('Fortran list-directed input, two passes (+ printing)   %8.1f ms' % (1e3 * dt), flush=True)
if n <= 2_000_000:
    t0 = time.perf_counter(); np.loadtxt(path); print('numpy.loadtxt %27.1f ms' % (1e3 * (time.perf_counter() - t0)))
