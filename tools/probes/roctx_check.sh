# named ranges of the library under rocprofv3 --marker-trace (GADFIT_HIP_ROCTX=1)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/roctx; mkdir -p gpurun_out/roctx
export GADFIT_HIP_ROCTX=1
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d gpurun_out/roctx -- python3 tools/probes/fit_cfg2.py 1e6 > gpurun_out/roctx/run.log 2>&1
tail -1 gpurun_out/roctx/run.log
f=$(find gpurun_out/roctx -name "*marker_api_trace.csv" | head -1)
echo "marker file: $f"
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
cnt = collections.Counter(); dur = collections.defaultdict(float)
for r in rows:
    name = r.get('Function') or r.get('Name') or str(r)
    cnt[name] += 1; dur[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
for k, v in cnt.most_common():
    print('%6d x %-60s avg %9.1f us' % (v, k[:60], dur[k] / v))
PY
find gpurun_out/roctx -name "*.csv" -size +1M -delete
