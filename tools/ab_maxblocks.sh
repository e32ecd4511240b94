#!/bin/bash
# developer A/B (GPU box): the traversal grid bound of pipelined batches (MI355PT_MAX_BLOCKS) under fused launches, by call size: rank 0 of an N-way sharded C3 frame
for w in 8 4 2 1; do for mb in 0 672 896 1120 1344 1792 2688; do
  echo "world $w max blocks $mb: $(MI355PT_MAX_BLOCKS=$mb python tools/rank_profile.py $w 6 2>/dev/null | tail -1 | cut -c1-62)"
done; done
