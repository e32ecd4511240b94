"""developer tool: the kernel sequence of the LAST frame of a rocprofv3 kernel trace, one line per launch in start order: stream, start offset, duration, gap to the
previous kernel of the same stream, grid size, name. usage: python tools/stream_gantt.py <kernel_trace.csv> [frames in the trace after the warm-up frame]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("ptk::", "").split("(")[0], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows), key=lambda e: e[0])
gens = [i for i, e in enumerate(ev) if e[2].startswith("k_generate")]
per_frame = len(gens) // (frames + 1) if len(gens) >= frames + 1 else len(gens)
sel = ev[gens[-per_frame]:]
t0 = sel[0][0]; last = {}
for s, e, n, q, g in sel:
    gap = (s - last[q]) * 1e-3 if q in last else 0.0
    print("q%-3s %8.3f ms  %7.1f us  gap %6.1f us  grid %9s  %s" % (q, (s - t0) * 1e-6, (e - s) * 1e-3, gap, g, n[:60]))
    last[q] = e
