#!/bin/bash
# where do the waves of the big U-Net kernels spend their cycles?  Separate --pmc passes (8 SQ slots each), no tracing domains beside them.
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcp
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmcp/counters_list.txt 2>&1 || true
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmcp/p$i -- python3 $R/tools/kernel_pmc_probe.py > $R/gpurun_out/pmcp/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmcp/p$i.log; }
done
cd $R
python tools/pmc_counters_summary.py gpurun_out/r03f_pmc_probe.json $(find gpurun_out/pmcp -name "*counter_collection.csv") > gpurun_out/r03f_pmc_probe.txt 2>&1
grep -c . gpurun_out/pmcp/counters_list.txt
find gpurun_out/pmcp -name "*.csv" -size +20M -delete
