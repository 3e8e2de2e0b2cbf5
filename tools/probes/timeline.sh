# kernel + memory-copy timeline of a few LM iterations (no counters): where the time between kernels goes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/timeline; mkdir -p gpurun_out/timeline
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/timeline -- python3 bench.py --legs main --steps 30 --warmup 5 --cpu-sample 0 > gpurun_out/timeline/bench.log 2>&1
python3 tools/probes/timeline.py gpurun_out/timeline > gpurun_out/timeline/summary.txt 2>&1
find gpurun_out/timeline -name "*.csv" -size +2M -delete
tail -60 gpurun_out/timeline/summary.txt
