#!/bin/bash
# developer A/B (GPU box): small tail-kernel thresholds under fused + free-running passes: rank 0 of 8 and the full frame, C3 and C5 (animated, nested dielectrics quality 2), and C1 / C2
for tp in 0 256 1024 4096 16384; do
  export MI355PT_TAIL_PATHS=$tp
  a=$(SHARD_PROBE_RANKS=1 python tools/shard_probe.py 8 1 2>/dev/null | grep "per rank ms" | tr '\n' ' ')
  b=$(SHARD_PROBE_RANKS=1 SHARD_PROBE_ANIMATE=1 python tools/shard_probe.py 8 1 2>/dev/null | grep "per rank ms" | tr '\n' ' ')
  c=$(python tools/run_configs.py --only C1,C2 2>/dev/null | grep ms_per_frame | tr -d ' \n')
  echo "tail $tp | C3 rank of 8, full: $a | C5 rank of 8, full: $b | C1, C2: $c"
done
