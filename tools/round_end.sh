#!/bin/bash
# End-of-round measurement bundle on one MI355X (GPU box; run through gpurun): usage tools/round_end.sh <tag e.g. r06z> -> gpurun_out/<tag>/ with
#   <tag>_bench.json                 python bench.py (the contract line: timed steps, serial-kernel roofline steps, cpu_baseline, parity against the oracle and the reference text)
#   <tag>_counters.json, <tag>_serial_kernel_stats.csv      tools/profile_round.sh (rocprofv3 --kernel-trace --stats, then --pmc in separate passes)
#   <tag>_configs.json               tools/run_configs.py (C1 .. C5 on one GPU)
#   <tag>_shard4.txt / _shard16.txt / _shard_c5.txt          tools/shard_probe.py (per-rank frames of 1 / 2 / 4 / 8-way sharded frames on one GPU)
#   <tag>_rank8_kernel_breakdown.txt tools/rank_breakdown.sh 8
# The counters are collected FIRST so that bench.py, which quotes profiles/<newest>_counters.json by source digest, can be re-run after they have been copied to profiles/.
TAG=${1:-r06z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
tools/profile_round.sh $OUT/profile > $OUT/profile.log 2>&1
cp $OUT/profile/counters.json $OUT/${TAG}_counters.json; cp $(find $OUT/profile -name "stats_kernel_stats.csv" | head -1) $OUT/${TAG}_serial_kernel_stats.csv
mkdir -p profiles; cp $OUT/${TAG}_counters.json profiles/${TAG}_counters.json      # (in the box's copy of the tree: what the bench run below quotes)
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python tools/run_configs.py > $OUT/${TAG}_configs.json 2> $OUT/configs.err
SHARD_PROBE_RANKS=3 python tools/shard_probe.py 1 2 4 8 > $OUT/${TAG}_shard4.txt 2>&1
SHARD_PROBE_RANKS=3 SHARD_PROBE_SPP=16 python tools/shard_probe.py 1 2 4 8 > $OUT/${TAG}_shard16.txt 2>&1
SHARD_PROBE_RANKS=3 SHARD_PROBE_ANIMATE=1 python tools/shard_probe.py 1 2 4 8 > $OUT/${TAG}_shard_c5.txt 2>&1
tools/rank_breakdown.sh 8 $OUT/rank8 > /dev/null 2>&1; cp $OUT/rank8/breakdown.txt $OUT/${TAG}_rank8_kernel_breakdown.txt
rm -rf $OUT/profile/*/ $OUT/profile/*.csv 2>/dev/null; find $OUT -name "*.csv" -size +2M -delete
tail -c 1500 $OUT/${TAG}_bench.json; echo; grep "^world" $OUT/${TAG}_shard4.txt $OUT/${TAG}_shard16.txt $OUT/${TAG}_shard_c5.txt
