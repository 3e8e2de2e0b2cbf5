# first gadf_fit of the headline program by number of recorder threads and keep-warm setting: is the bimodal record time the cgroup's CPU quota?
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for kw in 1 0; do for t in 8 12 14 16; do for i in 1 2 3; do
  echo -n "keep_warm=$kw threads=$t: "
  GADFIT_HIP_KEEP_WARM=$kw GADFIT_HIP_RECORD_THREADS=$t GADFIT_HIP_SETUP_TIMES=1 tests/fortran/build/bench_headline 10000000 10 2>&1 | grep "first gadf_fit" | tr -s ' '
done; done; done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
