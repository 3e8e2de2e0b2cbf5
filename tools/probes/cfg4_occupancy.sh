# BASELINE config 4: how many waves does a SIMD hold while the bisecting sweep runs?  occupancy = SQ_WAVE_CYCLES (quad-cycles) x 4 /
# (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  usage: cfg4_occupancy.sh [N]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export CFG4_N=${1:-1e6}
D=gpurun_out/pmc4_r03_occ_$CFG4_N
mkdir -p $D
for w in 4; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $D/m$w -- python3 tools/probes/cfg4_modes.py $w 20 > $D/m$w.log 2>&1
done
python3 - $D <<'PY'
import csv,glob,collections,os,sys
D=sys.argv[1]
names={4:'gfh_k_sweep',2:'gfh_k_chi2'}
for w in (4,):
    fs=sorted(glob.glob('%s/m%d/*/*_counter_collection.csv'%(D,w)), key=os.path.getmtime)
    if not fs:
        print('mode', w, 'no counter file:', open('%s/m%d.log'%(D,w)).read()[-600:]); continue
    rows=[r for r in csv.DictReader(open(fs[-1])) if r['Kernel_Name'].startswith(names[w])]
    ids=sorted({int(r['Dispatch_Id']) for r in rows})[-20:]
    agg=collections.defaultdict(list)
    for r in rows:
        if int(r['Dispatch_Id']) in ids: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    a={k: sum(v)/len(v) for k,v in agg.items()}
    print('N', os.environ['CFG4_N'], 'mode', w, {k: '%.5g'%v for k,v in sorted(a.items())})
    if 'SQ_WAVE_CYCLES' in a and 'GRBM_GUI_ACTIVE' in a:
        print('  waves resident per SIMD on average: %.2f   VALU busy %.3f' % (4*a['SQ_WAVE_CYCLES']/(a['GRBM_GUI_ACTIVE']/8*1024), 4*a.get('SQ_ACTIVE_INST_VALU',0)/(a['GRBM_GUI_ACTIVE']/8*1024)))
PY
