# soaks of every kind of random Fortran case beyond the suite's seeds; one result block per kind (kind and use_ad per line, per-kind worst)
OUT=${1:-gpurun_out/r05_fortran_fuzz_soaks.txt}
LO=${2:-200}; HI=${3:-320}
: > $OUT
for spec in "300 straight" "400 branching" "60 integral" "60 integral_branching" "30 integral_nested" "300 layout" "300 layout_branching" "300 sessions"; do
  set -- $spec
  echo "== python tools/probes/soak_fortran_fuzz.py $LO $HI $1 $2" >> $OUT
  python tools/probes/soak_fortran_fuzz.py $LO $HI $1 $2 2>&1 | grep -v "^seed [0-9]* \[" >> $OUT
  tail -3 $OUT
done
