# rocprofv3 passes over gfh_k_sweep_gram_nostore (tools/probes/nostore_probe.py): kernel trace, then counters in passes of their own.
# usage (on the GPU box): bash tools/pmc_nostore.sh <out-dir-under-gpurun_out> [N]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmc_nostore}
N=${2:-10000000}
mkdir -p $OUT
python3 tools/probes/nostore_probe.py $N 200 5 > $OUT/plain.json 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probes/nostore_probe.py $N 200 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 tools/probes/nostore_probe.py $N 40 1 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -- python3 tools/probes/nostore_probe.py $N 40 1 > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {}
for name in ['a', 'b']:
    for f in glob.glob('%s/%s/*/*_counter_collection.csv' % (out, name)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('gfh_k'):
                agg[(r['Kernel_Name'][:28], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            v = v[len(v) // 4:]                 # (the first launches run in the clock transient)
            res['%s %s' % k] = sum(v) / len(v)
            print(k, 'n=%d mean=%.6g' % (len(v), sum(v) / len(v)))
for f in glob.glob('%s/trace/*/*_kernel_stats.csv' % out):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith('gfh_k'):
            print('trace', r['Name'][:28], 'calls', r['Calls'], 'avg ns', r['AverageNs'])
            res['trace_avg_ns %s' % r['Name'][:28]] = float(r['AverageNs'])
json.dump(res, open('%s/summary.json' % out, 'w'), indent=1)
PY
cat $OUT/plain.json | tail -1
