#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run (gpurun_out/prof) into profiles/<tag>_*.{csv,md}."""
import collections
import csv
import glob
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_r03'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r03'
os.makedirs('profiles', exist_ok=True)
# gpurun merges every call's files into the same directory: take the newest run of each pass
stats = max(glob.glob(src + '/trace/*/*_kernel_stats.csv'), key=os.path.getmtime)
shutil.copy(stats, 'profiles/%s_kernel_stats.csv' % tag)
rows = list(csv.DictReader(open(stats)))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ['fetch', 'write', 'mfma']:
    for f in sorted(glob.glob(src + '/%s/*/*_counter_collection.csv' % name), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            cnt[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            cnt[r['Kernel_Name']]['VGPR'] = [float(r['VGPR_Count'])]
            cnt[r['Kernel_Name']]['LDS'] = [float(r['LDS_Block_Size'])]
with open('profiles/%s_summary.md' % tag, 'w') as o:
    o.write('# rocprofv3 summary (%s): `python3 bench.py` at N=1e7 points x 32 active parameters, 1 x MI355X\n\n' % tag)
    o.write('Source: tools/profile_bench.sh (separate passes: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; '
            '--pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).\n'
            'FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md §HBM FETCH_SIZE under-reports coalesced streaming '
            'reads by exactly 2x on gfx950, so HBM read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.\n\n')
    o.write('| kernel | calls | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (corrected) | MFMA_F64 insts '
            '| MFMA busy cycles | MFMA util % | FP64 matrix TFLOP/s | VGPR | LDS B |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n')
    for r in rows:
        k = r['Name']
        c = cnt.get(k, {})

        def m(n):
            return (sum(c[n]) / len(c[n])) if n in c and c[n] else float('nan')
        hbm = 2 * m('FETCH_SIZE') * 1024 + m('WRITE_SIZE') * 1024
        # MFMA utilisation = matrix-pipe busy cycles / (1024 SIMDs x shader cycles of the dispatch);
        # shader cycles from GRBM_GUI_ACTIVE (summed over the 8 XCDs) of the same pass
        cyc = m('GRBM_GUI_ACTIVE') / 8.0
        util = 100.0 * m('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * cyc) if cyc == cyc and cyc > 0 else float('nan')
        # flop per counted instruction: the fused kernel at 32 parameters issues, per 4 points, one v_mfma_f64_16x16x4 (2048 flop)
        # and five v_mfma_f64_4x4x4_4b (512 flop each): 768 flop on average; every other kernel issues 16x16x4 only
        fpi = 768.0 if r['Name'].startswith('gfh_k_sweep_gram') else 2048.0
        tfl = m('SQ_INSTS_VALU_MFMA_F64') * fpi / (float(r['AverageNs']) * 1e-9) / 1e12
        o.write('| `%s` | %s | %.1f | %.4g | %.4g | %.4g | %.4g | %.4g | %.1f | %.1f | %.0f | %.0f |\n' % (
            k[:60], r['Calls'], float(r['AverageNs']) / 1e3, m('FETCH_SIZE'), m('WRITE_SIZE'), hbm,
            m('SQ_INSTS_VALU_MFMA_F64'), m('SQ_VALU_MFMA_BUSY_CYCLES'), util, tfl, m('VGPR'), m('LDS')))
    # the trace pass runs `bench.py --legs main`: warm-up + pre-roll + exactly K timed iterations, in that order, so the
    # LAST K launches of the fused kernel are the timed region that bench.py's roofline line averages
    tr = glob.glob(src + '/trace/*/*_kernel_trace.csv')
    blog = [ln for ln in open(src + '/bench_trace.log') if ln.startswith('{')] if os.path.exists(src + '/bench_trace.log') else []
    if tr and blog:
        import json as _json
        bj = _json.loads(blog[-1])
        K = bj['steps'] * bj.get('repeats', {}).get('n', 1)          # the main leg repeats its K iterations (bench.py, MIN_TIMED_S)
        d = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']))
                   for r in csv.DictReader(open(tr[0])) if r['Kernel_Name'].startswith('gfh_k_sweep_gram') and 'nostore' not in r['Kernel_Name'])
        durs = [x[1] / 1e3 for x in d]
        if len(durs) >= K:
            o.write('\n`gfh_k_sweep_gram` launch by launch (kernel trace of `bench.py --legs main`): all %d launches average %.1f us; '
                    'launches 3-40 (power-management transient after the idle gap) %.1f us; the last %d launches = all timed repeats '
                    '**%.1f us** (bench.py reports `roofline.avg_ms` = %.4f for its median repeat).\n'
                    % (len(durs), sum(durs) / len(durs), sum(durs[2:40]) / max(1, len(durs[2:40])), K, sum(durs[-K:]) / K,
                       _json.loads(blog[-1])['roofline']['avg_ms']))
    for log in sorted(glob.glob(src + '/bench_*.log')):
        last = [ln for ln in open(log) if ln.startswith('{')]
        if last:
            o.write('\n`%s`:\n```\n%s```\n' % (os.path.basename(log), last[-1]))
import json
traffic = {}
for r in rows:
    c = cnt.get(r['Name'], {})
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        traffic[r['Name'].split('(')[0].replace('void ', '')] = (2 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) + sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])) * 1024
mfma = {}
for r in rows:
    c = cnt.get(r['Name'], {})
    if c.get('SQ_VALU_MFMA_BUSY_CYCLES') and sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) > 0:
        cyc = sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE']) / 8.0
        mfma[r['Name'].split('(')[0].replace('void ', '')] = {
            'busy_cycles': sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES']),
            'insts_f64': sum(c['SQ_INSTS_VALU_MFMA_F64']) / len(c['SQ_INSTS_VALU_MFMA_F64']),
            'util': sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024.0 * cyc)}
json.dump({'tag': tag, 'points': 10000000, 'hbm_bytes_per_launch': traffic, 'mfma': mfma}, open('profiles/traffic.json', 'w'), indent=1)
print(open('profiles/%s_summary.md' % tag).read()[:1800])
# BASELINE configs 2-4 (tools/profile_bench.sh: one `bench.py --legs configs --only-config N` kernel trace each): the dominant kernel's
# average from the trace beside the HIP-event figure of the same run's JSON line
cfg_rows = []
for cnum, kern in ((2, 'gfh_k_sweep_gram'), (3, 'gfh_k_sweep_gram'), (4, 'gfh_k_sweep')):
    st = sorted(glob.glob(src + '/cfg%d/*/*_kernel_stats.csv' % cnum), key=os.path.getmtime)
    lg = src + '/bench_cfg%d.log' % cnum
    if not st or not os.path.exists(lg):
        continue
    shutil.copy(st[-1], 'profiles/%s_cfg%d_kernel_stats.csv' % (tag, cnum))
    line = [ln for ln in open(lg) if ln.startswith('{')]
    ev = json.loads(line[-1])['configs']['cfg%d' % cnum] if line else {}
    for r in csv.DictReader(open(st[-1])):
        if r['Name'].split('(')[0] == kern:
            cfg_rows.append((cnum, kern, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3 if 'MinNs' in r else float('nan'), ev))
if cfg_rows:
    with open('profiles/%s_configs.md' % tag, 'w') as o:
        o.write('# BASELINE configs 2-4 on one MI355X (%s): rocprofv3 kernel trace of `python3 bench.py --legs configs --only-config N` beside the '
                'HIP-event figure of the same run\n\n(The trace average includes the untimed pre-roll and the launches of the LM-iteration leg, '
                'which run other store forms of the same kernel name; the shortest launch is the steadier comparison.)\n\n'
                '| config | kernel | calls in the trace | trace avg us | trace min us | bench line: HIP events ms (%s launches after pre-roll) | roofline of the line | LM iteration ms |\n|---|---|---|---|---|---|---|---|\n' % (tag, 'n'))
        for cnum, kern, calls, avg, mn, ev in cfg_rows:
            rf = ev.get('roofline', {})
            o.write('| %d: %s | `%s` | %s | %.1f | %.1f | %.4f | %s %.3f | %.4f |\n' % (cnum, ev.get('workload', ''), kern, calls, avg, mn, ev.get('kernel_ms', float('nan')),
                                                                                 rf.get('bound', ''), rf.get('frac') if rf.get('frac') is not None else float('nan'), ev.get('lm_iteration_ms', float('nan'))))
    print(open('profiles/%s_configs.md' % tag).read())
