# Round 3: lane occupancy and traffic of the quadrature kernels (BASELINE config 4).  Separate counter passes (rocprofv3 --pmc, no
# trace domains mixed in) + one kernel trace of tools/bench_configs.py BENCH_CFG=4.  Output directory = argument 1 (default pmc4_r03).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BENCH_CFG=4 BENCH_REPS=5
D=gpurun_out/${1:-pmc4_r03}
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/t -- python3 tools/bench_configs.py > $D/t.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $D/a -- python3 tools/bench_configs.py > $D/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT --output-format csv -d $D/b -- python3 tools/bench_configs.py > $D/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/f -- python3 tools/bench_configs.py > $D/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/w -- python3 tools/bench_configs.py > $D/w.log 2>&1
python3 - $D <<'PY'
import csv,glob,collections,os,sys
D=sys.argv[1]
for name in ['a','b','f','w']:
    fs=sorted(glob.glob('%s/%s/*/*_counter_collection.csv'%(D,name)), key=os.path.getmtime)[-1:]
    for f in fs:
        agg=collections.defaultdict(list); meta={}
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('gfh_k'):
                agg[(r['Kernel_Name'][:24], r['Counter_Name'])].append(float(r['Counter_Value']))
                meta[r['Kernel_Name'][:24]]=(r.get('VGPR_Count'), r.get('Scratch_Size'), r.get('LDS_Block_Size'), r.get('Grid_Size'), r.get('Workgroup_Size'))
        for k,v in sorted(agg.items()):
            print(k, 'n=%d mean=%.6g'%(len(v), sum(v)/len(v)))
        for k,v in meta.items(): print('meta', k, 'vgpr/scratch/lds/grid/wg', v)
f=sorted(glob.glob('%s/t/*/*_kernel_stats.csv'%D), key=os.path.getmtime)[-1]
print(open(f).read())
print(open('%s/t.log'%D).read()[-1500:])
PY
