#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03e.log
: > $L
for t in _stftold "" _stft2; do echo "== stft lib$t" >> $L; AC_LIB=libaudiocut_hip$t.so timeout -k 10 200 python tools/stft_bench.py >> $L 2>&1 || exit 1; done
echo "== resample_poly (one wave per output)" >> $L; timeout -k 10 200 python tools/resample_poly_bench.py >> $L 2>&1 || exit 1
echo "== tests" >> $L
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_export_loader.py tests/test_silero_vad.py -m gpu -x -q >> $L 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "c4_full or epsilon_plateau or c5_loader or feature_cache" >> $L 2>&1 || exit 1
echo "== bench" >> $L
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 > gpurun_out/r03e_bench.json 2>> $L || { cat gpurun_out/r03e_bench.json >> $L; exit 1; }
python - >> $L <<'PY'
import json
d=json.loads(open('gpurun_out/r03e_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','parity_ok','single_stream_latency_ms')}, d['single_stream_latency_note'], d['phases_ms_per_step'])
for r in d['framewise_rooflines']: print(r)
PY
