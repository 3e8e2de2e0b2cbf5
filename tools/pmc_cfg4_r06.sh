# Round 6: instruction counts of BASELINE config 4's kernels ON THE CURRENT SOURCES, scratch form (default) and pool
# form (GADFIT_HIP_WS_FAST=0), one counter pass per kind of launch (tools/probes/cfg4_modes.py: 4 bisecting sweep, 2 chi2, 8 sweep
# replaying the meshes).  Counter passes only (--pmc, no trace domains).  -> gpurun_out/r06/cfg4_pmc.json (copied to profiles/r06_cfg4_pmc.json),
# keyed by the sha1 of the generated translation unit so that bench.py can tell a stale count.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/r06/pmc4
mkdir -p $D
for form in scratch pool; do
  for w in 4 2 8; do
    if [ $form = pool ]; then export GADFIT_HIP_WS_FAST=0; else unset GADFIT_HIP_WS_FAST; fi
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $D/${form}_m$w -- python3 tools/probes/cfg4_modes.py $w 20 > $D/${form}_m$w.log 2>&1
  done
done
unset GADFIT_HIP_WS_FAST
python3 - $D <<'PY'
import csv, glob, collections, os, sys, json
D = sys.argv[1]
names = {4: 'gfh_k_sweep', 2: 'gfh_k_chi2', 8: 'gfh_k_sweep'}
label = {4: 'sweep_bisecting', 2: 'chi2', 8: 'sweep_mesh_replay'}
out = {'what': 'rocprofv3 --pmc averages over the last 20 dispatches of the named kernel (tools/pmc_cfg4_r06.sh, tools/probes/cfg4_modes.py), N = 1e6, BASELINE config 4', 'forms': {}}
for form in ('scratch', 'pool'):
    for w in (4, 2, 8):
        fs = sorted(glob.glob('%s/%s_m%d/*/*_counter_collection.csv' % (D, form, w)), key=os.path.getmtime)
        log = open('%s/%s_m%d.log' % (D, form, w)).read().strip().splitlines()
        if not fs or not log:
            continue
        rows = [r for r in csv.DictReader(open(fs[-1])) if r['Kernel_Name'].startswith(names[w])]
        ids = sorted({int(r['Dispatch_Id']) for r in rows})[-20:]
        agg = collections.defaultdict(list)
        for r in rows:
            if int(r['Dispatch_Id']) in ids:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        last = next((l for l in reversed(log) if l.startswith('which ')), '').split()      # (the probe's own line: rocprofv3 writes its notes behind it)
        rec = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
        rec['avg_ms_under_counters'] = float(last[last.index('avg_ms') + 1]) if 'avg_ms' in last else None
        rec['source_sha1'] = last[last.index('source_sha1') + 1] if 'source_sha1' in last else None
        out['forms'].setdefault(form, {})[label[w]] = rec
        print(form, label[w], {k: ('%.5g' % v if isinstance(v, float) else v) for k, v in rec.items()})
json.dump(out, open('gpurun_out/r06/cfg4_pmc.json', 'w'), indent=1)
PY
