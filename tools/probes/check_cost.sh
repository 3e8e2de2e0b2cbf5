# cost of the every-abscissa capture on this host: the single-thread check of one eval() and the whole first gadf_fit of the headline program
set -e
amdflang -O2 -cpp -fopenmp -I gadfit_amd/fortran/build tools/probes/check_cost.F90 gadfit_amd/fortran/build/libgadfit_f.a -Lgadfit_amd/lib -lgadfit_hip -Wl,-rpath,$PWD/gadfit_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib -o /tmp/check_cost
for i in 1 2 3; do /tmp/check_cost | tail -1; done
nproc
for i in 1 2 3; do GADFIT_HIP_SETUP_TIMES=3 tests/fortran/build/bench_headline 10000000 10 2>&1 | grep "threaded check\|first gadf_fit\|gadf_init\|gadf_fit \[ms\]" | head -4; done
