#!/bin/bash
# ONE parametrised gpurun batch script (round 4 on; it replaces the per-experiment run_r03*.sh files).  Usage, from the repo root on the box:
#   gpurun --timeout N -- 'bash tools/run_gpu_batch.sh <tag> <step> [<step> ...]'
# Every step appends to gpurun_out/<tag>.log and stops the batch on its first failure (no GPU step is started after a failed one).
# Steps:
#   conv_levels[:lib]   tools/conv_pf_bench.py 32 (optionally on another build of the library, AC_LIB=libaudiocut_hip_<lib>.so)
#   conv_order | conv_tile   tools/conv_order_probe.py: time per band width (conv_tile: per tile width of the 48-channel tile) and level, then FETCH_SIZE / WRITE_SIZE per variant
#   conv_ablation:<tags>  tools/conv_ablation.py on the product and on each libaudiocut_hip_<tag>.so (comma-separated W9_PROBE builds)
#   tdf_order           tools/tdf_order_probe.py on the probe build: time, FETCH_SIZE / WRITE_SIZE per (column blocks, row tiles) super-group
#   calib               tools/probes/build/fetch_calib under the raw TCC request counters (FETCH_SIZE / WRITE_SIZE calibration)
#   tdf_levels[:lib]    tools/tdf_tile_bench.py 32
#   unet_tests[:lib]    tests/test_unet_gpu.py
#   gpu_tests           the whole -m gpu suite
#   bench[:args]        bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 (args: comma-separated extra flags) -> gpurun_out/<tag>_bench.json
#   bench_default       python bench.py (the driver's command) -> gpurun_out/<tag>_bench_default.json
#   trace               rocprofv3 --kernel-trace --stats of the bench -> <tag>_bench_kernel_stats.csv, <tag>_bench_kernel_gaps.txt
#   tail_trace          kernel + copy trace of bench.py --pipeline-depth 1 -> <tag>_track_tail.txt (what runs after a track's U-Net)
#   pmc                 the two HBM-traffic passes of the bench -> <tag>_pmc_summary.json
#   sq_probe            tools/kernel_pmc_probe.py under three SQ counter passes -> <tag>_sq_probe.txt / .json
#   soak:<cases>        tools/parity_soak.py; cases separated by ';' (e.g. soak:90,501,71,c2_song,4;120,502,72,vocal_like)
#   configs             bench.py --config c3 / c4 / c5 -> <tag>_bench_c3.json ...
#   pytest:<args>       python -m pytest <args, comma-separated> -m gpu -x -q
#   py:<script>[:args]  any tools/<script>.py (args comma-separated)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
TAG=$1; shift
L=$O/$TAG.log
mkdir -p $O
cd $R
export TMPDIR=/tmp
libenv() { [ -n "$1" ] && echo "libaudiocut_hip_$1.so"; }
for STEP in "$@"; do
  NAME=${STEP%%:*}; ARG=""; [ "$STEP" != "$NAME" ] && ARG=${STEP#*:}
  echo "== $STEP" >> $L
  case $NAME in
    conv_levels) AC_LIB=$(libenv $ARG) timeout -k 10 200 python tools/conv_pf_bench.py 32 >> $L 2>&1 || exit 1 ;;
    conv_ablation) for T in "" ${ARG//,/ }; do AC_LIB=$(libenv $T) timeout -k 10 120 python tools/conv_ablation.py 32 >> $L 2>&1 || exit 1; done ;;
    tdf_levels)  AC_LIB=$(libenv $ARG) timeout -k 10 200 python tools/tdf_tile_bench.py 32 >> $L 2>&1 || exit 1 ;;
    conv_order|conv_tile)
      export AC_PROBE_SET=${NAME#conv_} AC_LIB=libaudiocut_hip_probe.so
      timeout -k 10 400 python tools/conv_order_probe.py 32 >> $L 2>&1 || exit 1
      for CTR in FETCH_SIZE WRITE_SIZE; do
        rm -rf $O/pmc_tmp
        ( cd /tmp && AC_PROBE_PMC=1 timeout -k 10 300 rocprofv3 --pmc $CTR --output-format csv -d $O/pmc_tmp -- python3 $R/tools/conv_order_probe.py 32 > $O/pmc_tmp.log 2>&1 ) || { tail -5 $O/pmc_tmp.log; exit 1; }
        python3 tools/pmc_by_dispatch.py k_conv3x3_f16x3_w96 $(find $O/pmc_tmp -name "*counter_collection.csv") --seq $O/pmc_tmp.log >> $L 2>&1
      done
      rm -rf $O/pmc_tmp ;;
    tdf_order)
      export AC_LIB=libaudiocut_hip_probe.so
      timeout -k 10 400 python tools/tdf_order_probe.py 32 >> $L 2>&1 || exit 1
      for CTR in FETCH_SIZE WRITE_SIZE; do
        rm -rf $O/pmc_tmp
        ( cd /tmp && AC_PROBE_PMC=1 timeout -k 10 300 rocprofv3 --pmc $CTR --output-format csv -d $O/pmc_tmp -- python3 $R/tools/tdf_order_probe.py 32 > $O/pmc_tmp.log 2>&1 ) || { tail -5 $O/pmc_tmp.log; exit 1; }
        python3 tools/pmc_by_dispatch.py k_tdf_linear_f16x3 $(find $O/pmc_tmp -name "*counter_collection.csv") --seq $O/pmc_tmp.log >> $L 2>&1
      done
      rm -rf $O/pmc_tmp ;;
    calib)
      for CTRS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
        rm -rf $O/pmc_tmp
        ( cd /tmp && timeout -k 10 120 rocprofv3 --pmc $CTRS --output-format csv -d $O/pmc_tmp -- $R/tools/probes/build/fetch_calib > $O/pmc_tmp.log 2>&1 ) || { tail -5 $O/pmc_tmp.log; exit 1; }
        echo "-- $CTRS" >> $L
        python3 tools/pmc_by_dispatch.py k_ $(find $O/pmc_tmp -name "*counter_collection.csv") >> $L 2>&1
        python3 - $(find $O/pmc_tmp -name "*counter_collection.csv") >> $L <<'PY'
import csv, sys
names = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Kernel_Name"]: names[int(r["Dispatch_Id"])] = r["Kernel_Name"][:60]
print("dispatch order:", [names[k] for k in sorted(names)])
PY
      done
      cat $O/pmc_tmp.log >> $L; rm -rf $O/pmc_tmp ;;
    unet_tests) [ -n "$ARG" ] && export AUDIOCUT_HIP_LIBNAME=$(libenv $ARG); timeout -k 10 700 python -m pytest tests/test_unet_gpu.py -m gpu -x -q >> $L 2>&1 || { tail -30 $L; exit 1; } ;;
    gpu_tests)  timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=15 >> $L 2>&1 || { tail -40 $L; exit 1; }; tail -3 $L ;;
    bench)
      timeout -k 10 600 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 ${ARG//,/ } > $O/${TAG}_bench.json 2>> $L || { tail -5 $L; exit 1; }
      python3 - $O/${TAG}_bench.json >> $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "parity_ok", "single_stream_latency_ms")}, d.get("phases_ms_per_step"), d["roofline"], d.get("socket_under_load"))
PY
      tail -1 $L ;;
    bench_default) ( time timeout -k 10 900 python bench.py > $O/${TAG}_bench_default.json ) 2>> $L || { tail -5 $L; exit 1; } ;;
    trace)
      rm -rf $O/tr
      ( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/tr -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 > $O/${TAG}_bench_under_rocprof.json 2>> $L ) || exit 1
      DB=$(find $O/tr -name "*.db" | head -1)
      python3 tools/kernel_stats_from_db.py $DB $O/${TAG}_bench_kernel_stats.csv >> $L 2>&1
      python3 tools/kernel_gaps.py $DB 20 12 14 > $O/${TAG}_bench_kernel_gaps.txt 2>&1
      rm -rf $O/tr ;;
    tail_trace)
      rm -rf $O/tr
      ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr -- python3 $R/bench.py --steps 3 --warmup 1 --pipeline-depth 1 --cpu-baseline-seconds 0 > $O/${TAG}_bench_depth1.json 2>> $L ) || exit 1
      python3 tools/track_tail_timeline.py $(find $O/tr -name "*.db" | head -1) 2 > $O/${TAG}_track_tail.txt 2>&1
      tail -25 $O/${TAG}_track_tail.txt >> $L
      rm -rf $O/tr ;;
    pmc)
      rm -rf $O/pf $O/pw
      ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-seconds 0 > /dev/null 2>> $L ) || exit 1
      ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-seconds 0 > /dev/null 2>> $L ) || exit 1
      python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_summary.json >> $L 2>&1
      rm -rf $O/pf $O/pw ;;
    sq_probe)           # where the waves of the big U-Net kernels spend their cycles: three --pmc passes of 8 SQ counters each
      P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
      P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      P3="SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
      rm -rf $O/pmcp; mkdir -p $O/pmcp; i=0
      for P in "$P1" "$P2" "$P3"; do
        i=$((i+1))
        ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/pmcp/p$i -- python3 $R/tools/kernel_pmc_probe.py > $O/pmcp/p$i.log 2>&1 ) || { tail -5 $O/pmcp/p$i.log; exit 1; }
      done
      python3 tools/pmc_counters_summary.py $O/${TAG}_sq_probe.json $(find $O/pmcp -name "*counter_collection.csv") > $O/${TAG}_sq_probe.txt 2>&1
      rm -rf $O/pmcp ;;
    soak) timeout -k 10 1150 python tools/parity_soak.py ${ARG//;/ } >> $L 2>&1 || { tail -20 $L; exit 1; }; tail -2 $L ;;
    configs)
      for CFG in c3 c4 c5; do
        ST=8; [ $CFG = c3 ] && ST=32; [ $CFG = c5 ] && ST=3
        timeout -k 10 500 python bench.py --config $CFG --steps $ST --warmup 2 --cpu-baseline-seconds 0 > $O/${TAG}_bench_$CFG.json 2>> $L || { tail -5 $L; exit 1; }
      done ;;
    pytest) timeout -k 10 1100 python -m pytest ${ARG//,/ } -m gpu -x -q >> $L 2>&1 || { tail -40 $L; exit 1; }; tail -2 $L ;;
    py) S=${ARG%%:*}; A=""; [ "$ARG" != "$S" ] && A=${ARG#*:}; timeout -k 10 900 python tools/$S.py ${A//,/ } >> $L 2>&1 || { tail -20 $L; exit 1; } ;;
    *) echo "unknown step $STEP" | tee -a $L; exit 2 ;;
  esac
done
echo "batch $TAG done: $*" | tee -a $L
