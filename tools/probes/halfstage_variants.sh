# fused no-store kernel at the headline size: variants of the matrix phase, HIP-event average per launch and a digest of the sums
# (bitwise identity across variants that share a partition): bash tools/probes/halfstage_variants.sh [out.jsonl]
OUT=${1:-gpurun_out/r05_nostore_variants.jsonl}
run() { env "$@" python tools/probes/nostore_probe.py 1e7 200 3 2>&1 | grep '^{' >> $OUT; }
: > $OUT
run A=1
run GADFIT_HIP_FRAG_LATE=1
run GADFIT_HIP_FRAG_LATE=1 GADFIT_HIP_FRAG_AHEAD=2
run GADFIT_HIP_FUSED_WAVES=4
run GADFIT_HIP_FUSED_WAVES=4 GADFIT_HIP_FRAG_LATE=1
run GADFIT_HIP_FRAG_LATE=1 GADFIT_HIP_MATRIX_PRIO=0
cat $OUT
