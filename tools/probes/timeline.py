#!/usr/bin/env python3
"""Print the device timeline (kernels and memory copies, rocprofv3 --kernel-trace --memory-copy-trace)
of the last LM iterations of a bench.py run: start offset, duration and the gap to the previous operation."""
import csv
import glob
import os
import sys

d = sys.argv[1]
ops = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:60]))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
ops.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# `bench.py --legs main`: one leg (look-ahead schedule); show a window from its timed region (the last launches)
sw = [i for i, o in enumerate(ops) if 'gfh_k_sweep_gram' in o[2]]
idx = sw[-12] if len(sw) > 12 else 0
print('---- look-ahead schedule, last iterations of the timed region')
t0 = ops[idx][0]; prev_end = t0
for s, e, name in ops[idx:idx + n]:
    print('%10.1f us  dur %8.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e
