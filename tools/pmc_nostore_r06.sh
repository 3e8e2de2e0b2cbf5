# Round 6: instruction counts of gfh_k_sweep_gram_nostore ON THE CURRENT SOURCES (headline model, N = 1e7), counter
# pass only.  -> gpurun_out/r06/nostore_pmc.json (copied to profiles/r06_nostore_pmc.json): SQ_INSTS_VALU (includes the matrix
# instructions), SQ_INSTS_VALU_MFMA_F64, per launch, with the sha1 of the generated source.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/r06/pmc_nostore
mkdir -p $D
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $D/a -- python3 tools/probes/nostore_probe.py 10000000 40 1 > $D/a.log 2>&1
python3 - $D <<'PY'
import csv, glob, collections, sys, json, os
D = sys.argv[1]
f = sorted(glob.glob('%s/a/*/*_counter_collection.csv' % D), key=os.path.getmtime)[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Kernel_Name'].startswith('gfh_k_sweep_gram_nostore'):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
rec = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in sorted(agg.items())}      # (the first launches run in the clock transient)
last = json.loads(next(l for l in reversed(open('%s/a.log' % D).read().splitlines()) if l.startswith('{"kernel"')))      # (the probe's own line)
out = {'what': 'rocprofv3 --pmc averages per launch of gfh_k_sweep_gram_nostore (tools/pmc_nostore_r06.sh, tools/probes/nostore_probe.py), headline model, N = 1e7',
       'points': last['points'], 'source_sha1': last['source_sha1'], 'ms_per_launch_under_counters': last['ms_per_launch'], 'counters': rec}
json.dump(out, open('gpurun_out/r06/nostore_pmc.json', 'w'), indent=1)
print(json.dumps(out))
PY
