# Device timeline of BASELINE config 3 fits: which kernels an LM iteration is made of and how long the GPU idles between them
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gaps3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps3/t -- python3 tools/probes/fit_cfg3.py > gpurun_out/gaps3/t.log 2>&1
tail -1 gpurun_out/gaps3/t.log
python3 - <<'PY'
import csv, glob, os, collections
f = sorted(glob.glob('gpurun_out/gaps3/t/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]                     # the timed half
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[a['Kernel_Name'][:28]].append(int(a['End_Timestamp']) - int(a['Start_Timestamp']))
    gap[(a['Kernel_Name'][:28], b['Kernel_Name'][:28])].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
print('kernel durations (us):')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])): print('  %-30s n=%4d mean=%8.2f' % (k, len(v), sum(v) / len(v) / 1e3))
print('idle between consecutive kernels (us):')
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])): print('  %-30s -> %-30s n=%4d mean=%8.2f' % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
PY
