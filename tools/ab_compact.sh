#!/bin/bash
# developer A/B (GPU box): the compacted path pool off / on (MI355PT_COMPACT_POOL): rank 0 of an 8- / 4- / 2-way sharded C3 frame and the full frame
for w in 1 8 4 2 1; do for m in 0 1; do
  echo "world $w compact $m: $(MI355PT_COMPACT_POOL=$m python tools/rank_profile.py $w 6 2>/dev/null | tail -1 | cut -c1-62)"
done; done
