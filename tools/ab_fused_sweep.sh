#!/bin/bash
# developer A/B (GPU box): fused traversal launches on rank 0 of an 8-way sharded C3 frame — batches x grid bound. usage: tools/ab_fused_sweep.sh
for b in 1 2 3 4; do for mb in 448 896 1344 1792; do
  echo "world 8 fused 1 batches $b maxBlocks $mb: $(MI355PT_FUSED_TRAVERSAL=1 MI355PT_BATCHES=$b MI355PT_MAX_BLOCKS=$mb python tools/rank_profile.py 8 8 2>/dev/null | tail -1)"
done; done
