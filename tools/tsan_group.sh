#!/bin/bash
# ThreadSanitizer over the single-process device group (csrc/group.cpp): a C++ harness (tools/tsan_group_main.cpp) drives groups of
# 2, 5 and 8 compile-only members through 300 rounds of the barrier + ordered host sum, including members that fail on purpose.
# CPU only (the members bind no GPU).  Prints "tsan harness ok" and no ThreadSanitizer report when clean.
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
for f in codegen rtc context group lm; do /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=thread -w -c gadfit_amd/csrc/$f.cpp -o $T/$f.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -c gadfit_amd/csrc/kernels.hip -o $T/kernels.o
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude -c tools/tsan_group_main.cpp -o $T/main.o
/opt/rocm/bin/hipcc -fsanitize=thread --offload-arch=gfx950 -w $T/*.o -o $T/harness -L/opt/rocm/lib -lhiprtc -lrccl -pthread -Wl,-rpath,/opt/rocm/lib
TSAN_OPTIONS=halt_on_error=0 $T/harness
rm -rf $T
