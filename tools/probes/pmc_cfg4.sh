# PMC detail of the quadrature kernels (BASELINE config 4): two counter passes + a kernel trace of tools/bench_configs.py BENCH_CFG=4
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BENCH_CFG=4 BENCH_REPS=5
mkdir -p gpurun_out/pmc4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc4/t -- python3 tools/bench_configs.py > gpurun_out/pmc4/t.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc4/a -- python3 tools/bench_configs.py > gpurun_out/pmc4/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc4/b -- python3 tools/bench_configs.py > gpurun_out/pmc4/b.log 2>&1
# (FETCH_SIZE and WRITE_SIZE in ONE pass abort rocprofv3 on this pool: they need separate passes, as tools/profile_bench.sh does)
python3 - <<'PY'
import csv,glob,collections,os
for name in ['a','b']:
    fs=sorted(glob.glob('gpurun_out/pmc4/%s/*/*_counter_collection.csv'%name), key=os.path.getmtime)[-1:]
    for f in fs:
        agg=collections.defaultdict(list); meta={}
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('gfh_k'):
                agg[(r['Kernel_Name'][:24], r['Counter_Name'])].append(float(r['Counter_Value']))
                meta[r['Kernel_Name'][:24]]=(r.get('VGPR_Count'), r.get('Scratch_Size'), r.get('LDS_Block_Size'), r.get('Grid_Size'), r.get('Workgroup_Size'))
        for k,v in sorted(agg.items()):
            print(k, 'n=%d mean=%.5g'%(len(v), sum(v)/len(v)))
        for k,v in meta.items(): print('meta', k, 'vgpr/scratch/lds/grid/wg', v)
f=sorted(glob.glob('gpurun_out/pmc4/t/*/*_kernel_stats.csv'), key=os.path.getmtime)[-1]
print(open(f).read())
PY
