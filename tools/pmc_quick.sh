#!/bin/bash
# developer profiling: two SQ counter passes over a 1-step bench run (usage: tools/pmc_quick.sh <outdir>)
OUT=${1:-gpurun_out/pmcq}; mkdir -p $OUT; OUT=$(realpath $OUT); REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT -o p1 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial-kernels > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o p2 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial-kernels > $OUT/p2.log 2>&1
cd $REPO && python tools/pmc_summary.py $OUT | grep -A22 "k_extend<false>\|k_shade\|k_shadow<false, false>"
