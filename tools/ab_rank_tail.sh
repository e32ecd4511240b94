#!/bin/bash
# developer A/B (GPU box): every variant library under gpurun_ab/ x tail-kernel threshold {0, 32768}: rank 0 of an 8-way sharded C3 frame, and the full frame. usage: tools/ab_rank_tail.sh
for lib in gpurun_ab/lib_*.so; do for tp in 0 32768; do
  r=$(MI355PT_TAIL_PATHS=$tp MI355PT_LIB=$PWD/$lib python tools/rank_profile.py 8 8 2>/dev/null | tail -1 | cut -c1-60)
  f=$(MI355PT_TAIL_PATHS=$tp MI355PT_LIB=$PWD/$lib python tools/rank_profile.py 1 4 2>/dev/null | tail -1 | cut -c1-60)
  echo "$(basename $lib) tail $tp | $r | $f"
done; done
