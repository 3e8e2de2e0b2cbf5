export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r02
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r02/a -- python3 bench.py --legs main --steps 3 --warmup 1 --pre-roll 0 --cpu-sample 0 --min-timed 0 > gpurun_out/pmc_r02/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_r02/b -- python3 bench.py --legs main --steps 3 --warmup 1 --pre-roll 0 --cpu-sample 0 --min-timed 0 > gpurun_out/pmc_r02/b.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for name in ['a','b']:
    for f in glob.glob('gpurun_out/pmc_r02/%s/*/*_counter_collection.csv'%name):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('gfh_k'):
                agg[(r['Kernel_Name'][:20], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()):
            print(k, 'n=%d mean=%.5g'%(len(v), sum(v)/len(v)))
PY
tail -2 gpurun_out/pmc_r02/a.log | cut -c1-300
