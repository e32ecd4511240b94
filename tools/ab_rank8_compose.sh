#!/bin/bash
# developer A/B (GPU box): rank 0 of an 8-way sharded C3 frame — batches x grid bound x compaction, with fused launches and free-running passes
for cp in 1 0; do for b in 2 3 4; do for mb in 896 1344 1792; do
  echo "compact $cp batches $b max blocks $mb: $(MI355PT_COMPACT_POOL=$cp MI355PT_BATCHES=$b MI355PT_MAX_BLOCKS=$mb python tools/rank_profile.py 8 8 2>/dev/null | tail -1 | cut -c1-62)"
done; done; done
