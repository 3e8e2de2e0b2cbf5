#!/bin/bash
# Round 6: the short streaming kernels of BASELINE configs 2 and 3 against the number of gram workgroups and
# the waves per workgroup.  Prints one line per setting: config, GADFIT_HIP_GB_TARGET, GADFIT_HIP_FUSED_WAVES, kernel ms, fraction.
out=${1:-gpurun_out/r06/gb_scan.txt}
: > $out
for cfg in 2 3; do
  for fw in 8 4; do
    for t in 512 768 1024 1536 2048 4096; do
      GADFIT_HIP_GB_TARGET=$t GADFIT_HIP_FUSED_WAVES=$fw python bench.py --legs configs --only-config $cfg 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']['cfg$cfg']; print('cfg$cfg gb_target $t waves $fw kernel_ms %.4f frac %.3f lm_iter_ms %.4f' % (d['kernel_ms'], d['roofline']['frac'], d['lm_iteration_ms']))" >> $out
    done
  done
done
cat $out
