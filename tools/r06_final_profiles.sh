# Round 6: every counter / trace pass the bench line's rooflines are reproduced from, in one go (each rocprofv3 pass is either a kernel
# trace or a --pmc pass, never both).  Run on the GPU box from the repository root; copy the summaries into profiles/ afterwards
# (tools/summarize_profile.py gpurun_out/prof_r06 r06; the json files under gpurun_out/r06/).
set -x
bash tools/profile_bench.sh > gpurun_out/r06/profile_bench.log 2>&1
bash tools/pmc_configs.sh > gpurun_out/r06/pmc_configs.log 2>&1
bash tools/pmc_cfg4_r06.sh > gpurun_out/r06/pmc4.log 2>&1
bash tools/pmc_nostore_r06.sh > gpurun_out/r06/pmc_ns.log 2>&1
tail -3 gpurun_out/r06/pmc_configs.log | cut -c1-600
