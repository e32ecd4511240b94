#!/bin/bash
# HBM traffic of the traversal kernels for bench.py's roofline.traffic: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (+ L2 hit/miss)
# over one default-workload step; writes <outdir>/traffic.json (copy to profiles/r01_traffic.json). usage: tools/pmc_traffic.sh <outdir>
OUT=${1:-gpurun_out/traffic}; mkdir -p $OUT; OUT=$(realpath $OUT); REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o t$i -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial-kernels > $OUT/t$i.log 2>&1
done
cd $REPO && python - $OUT <<'PY'
import csv, glob, json, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_extend" if "k_extend<false>" in k or "k_extendILb0" in k else "k_shadow" if "k_shadow<false, false>" in k or "k_shadowILb0ELb0" in k else "k_shade" if "k_shade" in k else None
        if not k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add((f, r["Dispatch_Id"]))
out = {}
for k in tot:
    launches = len({x[1] for x in n[k] if x[0].endswith("t1_counter_collection.csv")})
    fetch_kb, write_kb = tot[k].get("FETCH_SIZE", 0.0), tot[k].get("WRITE_SIZE", 0.0)
    out[k] = {"launches": launches, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
              "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0 / max(1, launches),
              "hbm_bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0,
              "l2_hit_rate": tot[k].get("TCC_HIT_sum", 0.0) / max(1.0, tot[k].get("TCC_REQ_sum", 0.0)),
              "note": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted as traffic"}
json.dump(out, open(d + "/traffic.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
