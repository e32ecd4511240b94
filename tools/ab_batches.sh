#!/bin/bash
# developer A/B (GPU box): more than four pipelined batches (a library built with -DPT_PIPELINE_BATCHES=8; MI355PT_BATCHES picks the count, GPU_MAX_HW_QUEUES the hardware queues): rank 0 of 8 | full frame
for q in 4 8; do for b in 4 5 6 8; do
  echo "hw queues $q batches $b: $(GPU_MAX_HW_QUEUES=$q MI355PT_BATCHES=$b MI355PT_LIB=$PWD/gpurun_ab/lib_b8.so python tools/rank_profile.py 8 6 2>/dev/null | tail -1 | cut -c1-60) | $(GPU_MAX_HW_QUEUES=$q MI355PT_BATCHES=$b MI355PT_LIB=$PWD/gpurun_ab/lib_b8.so python tools/rank_profile.py 1 4 2>/dev/null | tail -1 | cut -c1-60)"
done; done
