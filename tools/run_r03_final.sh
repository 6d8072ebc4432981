#!/bin/bash
# Final round-3 evidence in one gpurun call: the whole GPU test suite, the default bench command (with the CPU baseline leg), a longer
# bench, its rocprofv3 kernel trace (per-kernel stats + idle gaps) and the two HBM-traffic PMC passes.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== conv per level" > $O/r03_final.log
timeout -k 10 200 python tools/conv_pf_bench.py 32 >> $O/r03_final.log 2>&1 || exit 1
echo "== pytest -m gpu" >> $O/r03_final.log
timeout -k 10 1700 python -m pytest tests -m gpu -q >> $O/r03_final.log 2>&1 || { tail -30 $O/r03_final.log; exit 1; }
tail -3 $O/r03_final.log
echo "== python bench.py (driver's default command)" >> $O/r03_final.log
( time timeout -k 10 900 python bench.py > $O/r03_bench_default.json ) 2>> $O/r03_final.log || { tail -5 $O/r03_final.log; exit 1; }
echo "== bench --steps 10 --warmup 2" >> $O/r03_final.log
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 > $O/r03_bench.json 2>> $O/r03_final.log || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr $O/pf $O/pw
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/tr -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 > $O/r03_bench_under_rocprof.json 2>> $O/r03_final.log || exit 1
DB=$(find $O/tr -name "*.db" | head -1)
python3 $R/tools/kernel_stats_from_db.py $DB $O/r03_bench_kernel_stats.csv >> $O/r03_final.log 2>&1
python3 $R/tools/kernel_gaps.py $DB 20 12 > $O/r03_bench_kernel_gaps.txt 2>&1
rm -rf $O/tr
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-seconds 0 > /dev/null 2>> $O/r03_final.log || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-seconds 0 > /dev/null 2>> $O/r03_final.log || exit 1
python3 $R/tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/r03_pmc_summary.json >> $O/r03_final.log 2>&1
rm -rf $O/pf $O/pw
echo done
