#!/bin/bash
# developer tool: one rocprofv3 --pmc pass (instruction counts) over one serial-kernel step of bench.py's workload; prints per-kernel sums.
# usage: tools/pmc_quick.sh <outdir> [counters...]
OUT=${1:-gpurun_out/pmcq}; shift; CNT=${@:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU}
mkdir -p $OUT; OUT=$(realpath $OUT); REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o q -- python $REPO/bench.py --warmup 0 --no-cpu-baseline --serial-kernels --skip-roofline-steps --steps 1 > $OUT/q.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/**/q_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    k = "extend" if k.startswith("void ptk::k_extend<") or k.startswith("ptk::k_extend<") else "shadow" if "k_shadow<" in k else "shade" if "k_shade<" in k else None
    if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print(k, {n: "%.4g" % x for n, x in sorted(v.items())})
PY
