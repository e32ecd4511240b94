#!/bin/bash
# developer A/B: every variant library under gpurun_ab/ on (1) rank 0 of an 8-way sharded C3 frame and (2) the full frame (usage: tools/ab_rank.sh)
for lib in gpurun_ab/lib_*.so; do
  r=$(MI355PT_LIB=$PWD/$lib python tools/rank_profile.py ${AB_WORLD:-8} 6 2>/dev/null | tail -1)
  f=$(MI355PT_LIB=$PWD/$lib python bench.py --steps 4 --warmup 1 --no-cpu-baseline --skip-roofline-steps 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.1f Mrays/s %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "$(basename $lib) | $r | full frame: $f"
done
