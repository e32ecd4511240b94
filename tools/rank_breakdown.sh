#!/bin/bash
# developer tool (GPU box): where one rank of an N-way sharded C3 frame spends its time — rocprofv3 kernel trace of tools/rank_profile.py (pipelined, product defaults), the last frame's
# kernels summed by name, how long 0 / 1 / 2 / ... kernels were resident at once, and the launch sequence. usage: tools/rank_breakdown.sh <world> <outdir>
W=${1:-8}; OUT=${2:-gpurun_out/rank_breakdown}; mkdir -p $OUT; OUT=$(realpath $OUT); REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $REPO/tools/rank_profile.py $W 3 > $OUT/run.log 2>&1
cd $REPO; f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/stream_gantt.py $f 3 > $OUT/gantt.txt
python - <<PY > $OUT/breakdown.txt
import re, collections
rows=[]
for l in open("$OUT/gantt.txt"):
    m=re.match(r'q(\S+)\s+([\d.]+) ms\s+([\d.]+) us\s+gap\s+(-?[\d.]+) us\s+grid\s+(\S+)\s+(.*)',l)
    if m: rows.append((m.group(1),float(m.group(2))*1e3,float(m.group(3)),m.group(6)))
end=max(r[1]+r[2] for r in rows)
print(open("$OUT/run.log").read().strip().splitlines()[-1])
print("last frame under the profiler: %.2f ms from the first kernel's start to the last kernel's end, %d launches" % (end/1e3, len(rows)))
by=collections.Counter(); cnt=collections.Counter()
for r in rows: k=re.sub(r'<.*','',r[3]); by[k]+=r[2]; cnt[k]+=1
for k,v in by.most_common(): print("  %-22s %4d launches %9.1f us" % (k,cnt[k],v))
pts=[]
for r in rows: pts.append((r[1],1)); pts.append((r[1]+r[2],-1))
pts.sort(); lvl=0; last=0; hist=collections.Counter()
for t,d in pts: hist[lvl]+=t-last; last=t; lvl+=d
print("time with n kernels resident (us):", {k:round(v) for k,v in sorted(hist.items())})
PY
rm -rf $OUT/trace; cat $OUT/breakdown.txt
