# one counter pass per kind of launch of BASELINE config 4 (tools/probes/cfg4_modes.py): VALU instructions, VALU-busy quad-cycles, active lanes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/pmc4_r03_modes
mkdir -p $D
for w in 4 2 3 8 9; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $D/m$w -- python3 tools/probes/cfg4_modes.py $w 20 > $D/m$w.log 2>&1
done
python3 - $D <<'PY'
import csv,glob,collections,os,sys
D=sys.argv[1]
names={4:'gfh_k_sweep',2:'gfh_k_chi2',3:'gfh_k_omega',8:'gfh_k_sweep',9:'gfh_k_omega'}
for w in (4,2,3,8,9):
    f=sorted(glob.glob('%s/m%d/*/*_counter_collection.csv'%(D,w)), key=os.path.getmtime)[-1]
    rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith(names[w])]
    # the last 20 dispatches of that kernel are the timed launches of the requested kind
    ids=sorted({int(r['Dispatch_Id']) for r in rows})[-20:]
    agg=collections.defaultdict(list)
    for r in rows:
        if int(r['Dispatch_Id']) in ids: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('mode', w, names[w], {k: '%.5g'%(sum(v)/len(v)) for k,v in sorted(agg.items())}, open('%s/m%d.log'%(D,w)).read().strip().splitlines()[-1][:60])
PY
