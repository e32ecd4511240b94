#!/bin/bash
# developer A/B (GPU box): fused traversal launches off / on — rank 0 of an 8-, 4- and 2-way sharded C3 frame and the full frame. usage: tools/ab_fused.sh [frames]
N=${1:-8}
for w in 8 4 2 1; do
  for m in 0 1 0 1; do
    echo "world $w fused $m: $(MI355PT_FUSED_TRAVERSAL=$m python tools/rank_profile.py $w $N 2>/dev/null | tail -1)"
  done
done
