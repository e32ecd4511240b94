#!/bin/bash
# Round profile of bench.py's default workload on one MI355X (run on the GPU box, e.g. through gpurun):
#   pass 0   rocprofv3 --kernel-trace --stats over 3 serial-kernel steps          -> <out>/stats_kernel_stats.csv
#   pass 1-4 rocprofv3 --kernel-trace --pmc <group> over 1 serial-kernel step     -> <out>/pN_counter_collection.csv   (separate passes: FETCH_SIZE and
#            WRITE_SIZE do not fit one pass; counters are never combined with other trace domains)
#   summary  tools/profile_summary.py -> <out>/counters.json  (copy to profiles/rNN_counters.json; bench.py quotes it in roofline{})
# usage: tools/profile_round.sh <outdir>
OUT=${1:-gpurun_out/profile}; mkdir -p $OUT; OUT=$(realpath $OUT); REPO=$PWD
cd /tmp && export TMPDIR=/tmp
# PIPELINED=1: the product's composition (four batches, fused traversal launches, compacted pool) instead of serial-kernel steps: what k_trace_pair and k_shade<..., COMPACT> move
SERIAL=--serial-kernels; [ -n "$PIPELINED" ] && SERIAL=""
BENCH="python $REPO/bench.py --warmup 0 --no-cpu-baseline $SERIAL --skip-roofline-steps"
[ -n "$SKIP_STATS" ] || timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $BENCH --steps 3 > $OUT/stats.log 2>&1      # SKIP_STATS=1: the stats pass is already in <out>
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT -o p$i -- $BENCH --steps 1 > $OUT/p$i.log 2>&1
done <<'CNT'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
CNT
cd $REPO && python tools/profile_summary.py $OUT > $OUT/counters.json; cat $OUT/counters.json
