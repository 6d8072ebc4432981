#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03h.log
: > $L
for t in "" _dual _prio1 _prio2 ""; do echo "== conv lib$t" >> $L; AC_LIB=libaudiocut_hip$t.so timeout -k 10 150 python tools/conv_pf_bench.py 32 >> $L 2>&1 || exit 1; done
echo "== unet tests on the dual-tile build" >> $L
AUDIOCUT_HIP_LIBNAME=libaudiocut_hip_dual.so timeout -k 10 600 python -m pytest tests/test_unet_gpu.py -m gpu -x -q >> $L 2>&1 || exit 1
echo "== bench on the dual-tile build" >> $L
AUDIOCUT_HIP_LIBNAME=libaudiocut_hip_dual.so timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 > gpurun_out/r03h_bench.json 2>> $L || { cat gpurun_out/r03h_bench.json >> $L; exit 1; }
python - >> $L <<'PY'
import json
d=json.loads(open('gpurun_out/r03h_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','parity_ok','single_stream_latency_ms')}, d['phases_ms_per_step'], d['roofline']['avg_launch_ms'], d['socket_under_load'])
PY
