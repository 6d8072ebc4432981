#!/bin/bash
# one gpurun call: new kernels' tests, conv with mixed-width tiles, resampler timing, pipelining schemes, kernel trace + idle gaps
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03d.log
: > $L
echo "== tests" >> $L
timeout -k 10 900 python -m pytest tests/test_unet_gpu.py tests/test_export_loader.py tests/test_silero_vad.py -m gpu -x -q >> $L 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "track_pipeline or c4_full or epsilon_plateau or c5_loader" >> $L 2>&1 || exit 1
echo "== resample_poly" >> $L; timeout -k 10 200 python tools/resample_poly_bench.py >> $L 2>&1 || exit 1
echo "== conv (mixed-width tiles on C = 144 / 240)" >> $L; timeout -k 10 150 python tools/conv_pf_bench.py 32 >> $L 2>&1 || exit 1
echo "== conv (AC_NO_MIX=1)" >> $L; AC_NO_MIX=1 timeout -k 10 150 python tools/conv_pf_bench.py 32 >> $L 2>&1 || exit 1
for a in "--shared-unet-stream 0" "" "--pipeline-depth 3"; do
  echo "== bench $a" >> $L
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 $a > gpurun_out/r03d_bench.json 2>> $L || { cat gpurun_out/r03d_bench.json >> $L; exit 1; }
  python - >> $L <<'PY'
import json
d=json.loads(open('gpurun_out/r03d_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','parity_ok','single_stream_latency_ms')}, d['roofline']['avg_launch_ms'], d['phases_ms_per_step'], d['socket_under_load'])
PY
done
echo "== kernel trace (default scheme)" >> $L
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03d_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03d_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r03d_bench_rocprof.json 2>> $GRAFT_REPO_ROOT/$L || exit 1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r03d_trace -name "*.db" | head -1)
echo "db: $DB" >> $L
python tools/kernel_stats_from_db.py $DB gpurun_out/r03d_kernel_stats.csv >> $L 2>&1
python tools/kernel_gaps.py $DB 20 30 >> $L 2>&1
rm -rf gpurun_out/r03d_trace
