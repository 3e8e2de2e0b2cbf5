export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/timeline2; mkdir -p gpurun_out/timeline2
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/timeline2 -- python3 tools/probes/fit_cfg2.py > gpurun_out/timeline2/run.log 2>&1
python3 tools/probes/timeline.py gpurun_out/timeline2 30 > gpurun_out/timeline2/summary.txt 2>&1
find gpurun_out/timeline2 -name "*.csv" -size +2M -delete
tail -1 gpurun_out/timeline2/run.log
tail -31 gpurun_out/timeline2/summary.txt
