#!/bin/bash
# one gpurun call: resample variants, TDF baseline, conv sanity, U-Net concurrency (separation gate) experiment
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03c.log
: > $L
for t in _base _sk2 _sk4; do echo "== resample lib$t" >> $L; AC_LIB=libaudiocut_hip$t.so timeout -k 10 150 python tools/resample_bench.py 32 >> $L 2>&1 || exit 1; done
echo "== tdf" >> $L; timeout -k 10 150 python tools/tdf_tile_bench.py 32 >> $L 2>&1 || exit 1
echo "== conv" >> $L; timeout -k 10 150 python tools/conv_pf_bench.py 32 >> $L 2>&1 || exit 1
echo "== unet tests" >> $L; timeout -k 10 400 python -m pytest tests/test_unet_gpu.py -m gpu -x -q >> $L 2>&1 || exit 1
for a in "" "--no-separation-gate" "--pipeline-depth 3 --no-separation-gate"; do
  echo "== bench $a" >> $L
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 $a > gpurun_out/r03c_bench.json 2>> $L || { cat gpurun_out/r03c_bench.json >> $L; exit 1; }
  python - >> $L <<'PY'
import json
d=json.loads(open('gpurun_out/r03c_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','parity_ok','single_stream_latency_ms')}, d['roofline']['avg_launch_ms'], d['phases_ms_per_step']['unet'], d['socket_under_load'])
PY
done
