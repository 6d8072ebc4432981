#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03k.log
: > $L
echo "== tests (guard / tempogram loops with grouped LDS reads)" >> $L
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -m gpu -q -k "not c5_long_form_end_to_end and not c3_batch and not c5_loader" >> $L 2>&1 || exit 1
echo "== bench" >> $L
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trk
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r03k_bench.json 2>> $GRAFT_REPO_ROOT/$L || exit 1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats_from_db.py $(find gpurun_out/trk -name "*.db" | head -1) gpurun_out/r03k_kernel_stats.csv >> $L 2>&1
rm -rf gpurun_out/trk
python - >> $L <<'PY'
import json, csv
d=json.loads(open('gpurun_out/r03k_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','parity_ok','single_stream_latency_ms')}, d['phases_ms_per_step'], d['socket_under_load'])
for r in csv.DictReader(open('gpurun_out/r03k_kernel_stats.csv')):
    if any(s in r['Name'] for s in ('k_tempogram','k_quiet_guard_slow','k_pause','k_stft2048')): print(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e6)
PY
