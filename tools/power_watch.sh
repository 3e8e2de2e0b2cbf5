# Samples rocm-smi (clocks, power, temperature) every 0.2 s while bench.py --legs main runs: is the launch-to-launch spread of the fused
# kernel (0.50 .. 0.59 ms in one run) a clock / power-cap effect?
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showtemp --showperflevel --json 2>/dev/null | tr -d '\n' | cut -c1-900; echo; sleep 0.2; done ) > gpurun_out/power_watch.log 2>&1 &
W=$!
python3 bench.py --legs main --cpu-sample 0 --steps 2000 > gpurun_out/power_watch_bench.json 2>/dev/null
kill $W 2>/dev/null
python3 - <<'PY'
import json,re
rows=[l for l in open('gpurun_out/power_watch.log') if l.strip().startswith('{')]
print(len(rows),'samples')
for l in rows[::3][:20]:
    try:
        d=json.loads(l)
    except Exception:
        print(l[:200]); continue
    c=d.get('card0', d)
    keys=[k for k in c if any(s in k.lower() for s in ('sclk','mclk','power','temperature (sensor junction)','performance'))]
    print({k:c[k] for k in keys})
b=json.loads(open('gpurun_out/power_watch_bench.json').read().strip().splitlines()[-1])
print('bench', b['ms_per_step'], b['roofline']['avg_ms'], b['roofline']['min_ms'], b['roofline']['max_ms'])
PY
