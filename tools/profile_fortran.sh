# rocprofv3 kernel trace of the product path a Fortran user runs: tests/fortran/bench_headline.F90 (gadf_init / gadf_add_dataset /
# gadf_set / gadf_fit of the headline workload, N = 1e7, 32 active parameters).  The program itself stands after `--`.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_r04f
python3 gadfit_amd/fortran/build.py > gpurun_out/prof_r04f/build.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04f/trace -- tests/fortran/build/bench_headline 10000000 10 > gpurun_out/prof_r04f/headline.log 2>&1
find gpurun_out/prof_r04f -name "*kernel_stats.csv" | head -3
