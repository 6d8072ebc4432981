#!/bin/bash
# the other BASELINE configs through bench.py on the final code: C3 (32 tracks), C4 (vpbd_acoustic + Silero network), C5 (30-min tracks)
set -o pipefail
mkdir -p gpurun_out
for c in "c3" "c4 --steps 4 --warmup 1" "c5 --steps 2 --warmup 1"; do
  n=$(echo $c | cut -d" " -f1)
  echo "== bench --config $c"
  timeout -k 10 500 python bench.py --config $c --cpu-baseline-seconds 0 > gpurun_out/r03x_bench_$n.json 2> gpurun_out/r03x_bench_$n.err || { tail -5 gpurun_out/r03x_bench_$n.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r03x_bench_$n.json').read().strip().splitlines()[-1])
p=d['parity_vs_single_gpu']
print({k:d[k] for k in ('value','ms_per_step','steps','parity_ok')}, 'checked', p['tracks_checked'], 'identical', p['tracks_identical'], 'mismatched', p['mismatched_seeds'], p.get('vs_cpu_oracle_fixture'))
PY
done
