#!/bin/bash
# AddressSanitizer + UBSan over what runs on the CPU (GPU sanitizers are not available on the pool):
#  (1) the oracle (gcc -fsanitize=address,undefined) under tests/test_oracle_goldens.py + tests/test_cpu_branching.py;
#  (2) the host side of libgadfit_hip.so (hipcc -fsanitize=address on the .cpp files: code generator, hiprtc cache, compile-only
#      contexts, the device group's threads/barrier/host sum, LM helpers) under tests/test_cpu_cabi.py + tests/test_cpu_api_mirror.py.
# The instrumented libraries replace the in-tree ones for the run and are put back afterwards.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD; T=$(mktemp -d)
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $T/oracle.so oracle/gadfit_oracle.c -lm
cp oracle/libgadfit_oracle.so $T/oracle.bak; cp $T/oracle.so oracle/libgadfit_oracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_oracle_goldens.py tests/test_cpu_branching.py -x -q || RC=1
cp $T/oracle.bak oracle/libgadfit_oracle.so
RT=$(dirname $(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1))
for f in codegen rtc context group lm reader; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -fsanitize=address -shared-libasan -fno-omit-frame-pointer -w -c gadfit_amd/csrc/$f.cpp -o $T/$f.o
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -w -c gadfit_amd/csrc/kernels.hip -o $T/kernels.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -shared-libasan -o $T/lib.so $T/*.o -L/opt/rocm/lib -lhiprtc -lrccl -pthread -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=gadfit_amd/csrc/exports.map
cp gadfit_amd/lib/libgadfit_hip.so $T/lib.bak; cp $T/lib.so gadfit_amd/lib/libgadfit_hip.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so LD_LIBRARY_PATH=$RT \
  python -m pytest tests/test_cpu_cabi.py tests/test_cpu_api_mirror.py tests/test_cpu_multirank_layout.py tests/test_cpu_branching.py tests/test_cpu_reader.py tests/test_cpu_bench_schema.py -x -q || RC=1
cp $T/lib.bak gadfit_amd/lib/libgadfit_hip.so
rm -rf $T
exit ${RC:-0}
