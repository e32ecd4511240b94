#!/bin/bash
# developer profiling: one rocprofv3 --pmc pass per counter group over a 1-step bench run (usage: tools/pmc.sh <outdir> [bench args])
# counters are collected WITHOUT any trace domain other than --kernel-trace (gpurun policy)
OUT=${1:-gpurun_out/pmc}; shift
ARGS=${@:---steps 1 --warmup 1 --no-cpu-baseline --serial-kernels}
mkdir -p $OUT; OUT=$(realpath $OUT); REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT -o p$i -- python $REPO/bench.py $ARGS > $OUT/p$i.log 2>&1
done <<'CNT'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_LDS_BANK_CONFLICT
CNT
cd $REPO && python tools/pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
