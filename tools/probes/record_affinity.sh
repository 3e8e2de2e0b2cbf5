# first gadf_fit of the headline program: does binding the recorder threads to distinct cores (no SMT siblings) steady the capture?
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; lscpu | grep -E "Thread|Core|Socket|Model name" 
for rep in 1 2 3 4 5; do
  for mode in default close; do
    echo -n "$mode: "
    if [ $mode = close ]; then
      OMP_PLACES=cores OMP_PROC_BIND=close GADFIT_HIP_SETUP_TIMES=1 tests/fortran/build/bench_headline 10000000 10 2>&1 | grep "first gadf_fit" | tr -s ' '
    else
      GADFIT_HIP_SETUP_TIMES=1 tests/fortran/build/bench_headline 10000000 10 2>&1 | grep "first gadf_fit" | tr -s ' '
    fi
  done
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_throttled|throttled_usec"
