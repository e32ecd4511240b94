#!/bin/bash
# developer tool (GPU box): VALU instructions per k_shade wave for every variant library under gpurun_ab/ (rocprofv3 --pmc on one serial-kernel step of the bench workload)
for lib in gpurun_ab/lib_*.so; do
  D=/tmp/si_$(basename $lib .so); rm -rf $D
  (cd /tmp && TMPDIR=/tmp MI355PT_LIB=$OLDPWD/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $D -o p -- python $OLDPWD/bench.py --warmup 0 --no-cpu-baseline --serial-kernels --skip-roofline-steps --steps 1 > /dev/null 2>&1)
  python - <<PY
import csv,glob,collections
c=collections.defaultdict(lambda: collections.defaultdict(float)); t=collections.defaultdict(float); seen=set()
for f in glob.glob("$D/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].replace("void ptk::","").split("<")[0].split("(")[0]
        c[n][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); t[n]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6
for n in ("k_shade","k_extend","k_shadow"):
    d=c[n]; print("$(basename $lib) %-9s %7.2f ms  VALU/wave %7.1f  SALU/wave %6.1f  VMEM_RD/wave %5.1f  waves %d" % (n, t[n], d["SQ_INSTS_VALU"]/max(d["SQ_WAVES"],1), d["SQ_INSTS_SALU"]/max(d["SQ_WAVES"],1), d["SQ_INSTS_VMEM_RD"]/max(d["SQ_WAVES"],1), d["SQ_WAVES"]))
PY
  rm -rf $D
done
