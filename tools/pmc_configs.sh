# HBM traffic of the dominant kernel of BASELINE configs 2 and 3 (the HBM-bound ones): separate --pmc passes of
# `bench.py --legs configs --only-config N` (FETCH_SIZE, WRITE_SIZE), per launch of gfh_k_sweep_gram (the stored form the line times);
# FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section).  -> gpurun_out/r06_configs_traffic.json (copied to profiles/)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_cfg
for c in 2 3; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_cfg/cfg${c}_$ctr -- python3 bench.py --legs configs --only-config $c > gpurun_out/pmc_cfg/cfg${c}_$ctr.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for c in (2, 3):
    v = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = max(glob.glob('gpurun_out/pmc_cfg/cfg%d_%s/*/*_counter_collection.csv' % (c, ctr)), key=lambda p: __import__('os').path.getmtime(p))
        vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Kernel_Name'] == 'gfh_k_sweep_gram' and r['Counter_Name'] == ctr]
        v[ctr] = sum(vals) / len(vals); v[ctr + '_launches'] = len(vals)
    out['cfg%d' % c] = dict(v, hbm_bytes_per_launch=2 * v['FETCH_SIZE'] * 1024 + v['WRITE_SIZE'] * 1024)
json.dump(out, open('gpurun_out/r06_configs_traffic.json', 'w'), indent=1)
print(json.dumps(out))
PY
