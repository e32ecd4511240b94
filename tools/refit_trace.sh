cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04n -o c5 -- python $GRAFT_REPO_ROOT/tools/c5_probe.py 2>&1 | grep "^frame"; cd $GRAFT_REPO_ROOT; t=$(find gpurun_out/r04n -name "*kernel_trace.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$t")))
ev=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("void ","").replace("ptk::","").split("(")[0]) for r in rows))
idx=[i for i,e in enumerate(ev) if e[2].startswith("k_init_bounds")]
i0=idx[-1]; t0=ev[i0][0]; out=[]
for s,e,n in ev[max(0,i0-3):]:
    if n.startswith("k_generate"): break
    out.append("%9.1f us  +%7.1f us  %s" % ((s-t0)*1e-3,(e-s)*1e-3,n[:70]))
open("gpurun_out/r04n_refit_sequence.txt","w").write("\n".join(out)+"\n"); print("\n".join(out))
PY
rm -rf gpurun_out/r04n
