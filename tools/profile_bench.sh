export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_r06
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06/trace -- python3 bench.py --legs main --cpu-sample 0 > gpurun_out/prof_r06/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r06/fetch -- python3 bench.py --legs main --steps 4 --warmup 1 --pre-roll 0 --cpu-sample 0 --min-timed 0 > gpurun_out/prof_r06/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_r06/write -- python3 bench.py --legs main --steps 4 --warmup 1 --pre-roll 0 --cpu-sample 0 --min-timed 0 > gpurun_out/prof_r06/bench_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_r06/mfma -- python3 bench.py --legs main --steps 4 --warmup 1 --pre-roll 0 --cpu-sample 0 --min-timed 0 > gpurun_out/prof_r06/bench_mfma.log 2>&1
find gpurun_out/prof_r06 -name "*.csv" | head -30
du -sh gpurun_out/prof_r06
# BASELINE configs 2-4: one kernel trace per configuration (the kernels of different models share their names)
for c in 2 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06/cfg$c -- python3 bench.py --legs configs --only-config $c > gpurun_out/prof_r06/bench_cfg$c.log 2>&1
done
