"""developer tool: concurrency / gap analysis of a rocprofv3 kernel trace (…_kernel_trace.csv): for the LAST frame in the trace prints the wall span,
the time with 0 / 1 / 2 / 3+ kernels resident, per-kernel-name totals and the per-stream busy time. usage: python tools/timeline.py <kernel_trace.csv> [frames]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("ptk::", "").split("(")[0], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows), key=lambda e: e[0])
# frames are delimited by k_generate launches: take everything from the first k_generate of the last frame to the last k_accumulate
gens = [i for i, e in enumerate(ev) if e[2].startswith("k_generate")]
accs = [i for i, e in enumerate(ev) if e[2].startswith("k_accumulate")]
per_frame = len(gens) // (frames + 1) if len(gens) >= frames + 1 else len(gens)
first = gens[-per_frame]
last_end = max(e[1] for e in ev[first:])
sel = [e for e in ev[first:]]
t0, t1 = sel[0][0], last_end
pts = []
for s, e, n, q in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
hist = collections.Counter(); cur = 0; prev = t0
for t, d in pts:
    hist[min(cur, 4)] += t - prev; prev = t; cur += d
print("frame span %.2f ms, %d kernels" % ((t1 - t0) * 1e-6, len(sel)))
for k in sorted(hist): print("  %s kernels resident: %.2f ms" % (("%d" % k) if k < 4 else "4+", hist[k] * 1e-6))
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in sel: tot[n][0] += 1; tot[n][1] += (e - s) * 1e-6
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]: print("  %-34s x%4d  %.2f ms  avg %.3f" % (n[:34], c, ms, ms / c))
byq = collections.defaultdict(float)
for s, e, n, q in sel: byq[q] += (e - s) * 1e-6
print("  per queue busy ms:", {k: round(v, 2) for k, v in byq.items()})
