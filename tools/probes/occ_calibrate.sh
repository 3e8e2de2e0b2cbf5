# calibration of "waves resident per SIMD" from SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE on a kernel whose occupancy is known from its own stamps
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/occ_cal
mkdir -p $D
rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $D/c -- tools/microbench/scratch_occupancy > $D/c.log 2>&1
python3 - $D <<'PY'
import csv,glob,collections,os,sys
fs=sorted(glob.glob(sys.argv[1]+'/c/*/*_counter_collection.csv'), key=os.path.getmtime)
rows=list(csv.DictReader(open(fs[-1])))
by=collections.defaultdict(dict)
for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value']); by[int(r['Dispatch_Id'])]['_']=(r['Kernel_Name'][:40], r['Grid_Size'], r['Scratch_Size'])
for d in sorted(by)[::3]:
    a=by[d]
    print(d, a['_'], 'waves/SIMD from counters: %.2f' % (4*a['SQ_WAVE_CYCLES']/(a['GRBM_GUI_ACTIVE']/8*1024)), {k:'%.4g'%v for k,v in a.items() if k!='_'})
PY
grep "grid 32" $D/c.log | head -3
