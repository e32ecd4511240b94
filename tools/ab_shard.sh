#!/bin/bash
# developer A/B: every variant library under gpurun_ab/ on rank 0 of an 8-way sharded C3 frame and on the full frame (tools/shard_probe.py), tail-kernel thresholds from AB_TAILS (default 0)
for lib in gpurun_ab/lib_*.so; do
  echo "== $(basename $lib)"
  MI355PT_LIB=$PWD/$lib SHARD_PROBE_TAILS=${AB_TAILS:-0} SHARD_PROBE_RANKS=1 python tools/shard_probe.py ${AB_WORLDS:-8 1} 2>&1 | grep "^world"
done
