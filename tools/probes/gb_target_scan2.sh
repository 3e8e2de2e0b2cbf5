#!/bin/bash
# Round 6: fewer gram workgroups (one generation per CU?) for configs 2, 3 and the headline kernel.
out=${1:-gpurun_out/r06/gb_scan2.txt}
: > $out
for cfg in 2 3; do
  for t in 128 192 256 320 384 512; do
    GADFIT_HIP_GB_TARGET=$t python bench.py --legs configs --only-config $cfg 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']['cfg$cfg']; print('cfg$cfg gb_target $t kernel_ms %.4f frac %.3f lm_iter_ms %.4f' % (d['kernel_ms'], d['roofline']['frac'], d['lm_iteration_ms']))" >> $out
  done
done
for t in 256 512 384; do
  GADFIT_HIP_GB_TARGET=$t python bench.py --legs main --steps 100 --min-timed 1.0 --cpu-sample 0 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline gb_target $t ms_per_step %.4f kernel avg_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac']))" >> $out
done
cat $out
