# kernel + memory-copy timeline of the last LM iterations of BASELINE config 3 (global fit, 64 datasets x 1e5 points, dim 259)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/timeline3; mkdir -p gpurun_out/timeline3
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/timeline3 -- python3 tools/probes/fit_cfg3.py > gpurun_out/timeline3/run.log 2>&1
python3 tools/probes/timeline.py gpurun_out/timeline3 40 > gpurun_out/timeline3/summary.txt 2>&1
find gpurun_out/timeline3 -name "*.csv" -size +2M -delete
tail -3 gpurun_out/timeline3/run.log
tail -42 gpurun_out/timeline3/summary.txt
