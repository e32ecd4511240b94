#!/bin/bash
# developer A/B (GPU box): the tail kernel's threshold and hand-back bound under fused traversal launches, rank 0 of an 8-way sharded C3 frame. usage: tools/ab_tail_sweep.sh
for tp in 0 8192 16384 32768 65536 131072 262144; do for td in 512 4096; do
  echo "world 8 tail paths $tp defer $td: $(MI355PT_FUSED_TRAVERSAL=1 MI355PT_TAIL_PATHS=$tp MI355PT_TAIL_DEFER=$td python tools/rank_profile.py 8 8 2>/dev/null | tail -1)"
done; done
