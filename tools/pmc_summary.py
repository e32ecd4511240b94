"""Sum rocprofv3 counter_collection CSVs per kernel (developer tool): python tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set); dur = collections.defaultdict(float)
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        k = k.replace("void ptk::", "").replace("ptk::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); disp[k].add(key)
            if f.endswith("p1_counter_collection.csv"): dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for k in sorted(tot, key=lambda k: -dur[k]):
    if not k.startswith("k_"): continue
    print("== %s  (%.1f ms in pass 1)" % (k, dur[k]))
    for c, v in sorted(tot[k].items()): print("   %-40s %.6g" % (c, v))
