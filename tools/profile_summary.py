"""Per-kernel-group summary of tools/profile_round.sh's rocprofv3 output -> JSON on stdout (profiles/rNN_counters.json).
A "launch" of a traversal stage is the stage kernel plus its straggler task rounds and resolve pass (what bench.py brackets with HIP events).
Formulas (gfx94x definitions, the ones rocprofv3 falls back to on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  gpu_cycles       = GRBM_GUI_ACTIVE / 8        the counter comes back summed over the 8 XCDs (18.9 G "cycles" per second = 8 x 2.36 GHz)
  valu_busy        = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * gpu_cycles)      the VALUBusy formula. The counter charges every issued wave64 VALU instruction one quad-cycle
                     whatever its class, so this is "VALU instructions x 4 cycles": exact for the 4-cycle class, an over-statement for the two-operand ALU instructions that
                     issue every 2 cycles (profiles/r06a_valu_ceiling.txt, tools/valu_ceiling)
  valu_instr_per_simd_cycle = SQ_INSTS_VALU / (1024 * gpu_cycles)             measured ceilings on gfx950 (same file): 0.486 in total, 0.248 for the 4-cycle class
  valu_share_of_issue_ceiling = valu_instr_per_simd_cycle / 0.486
  lane_utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)           active lanes per issued VALU instruction
  hbm_bytes        = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated;
                     Infinity-Cache hits are counted as traffic
  l2_hit_rate      = TCC_HIT / (TCC_HIT + TCC_MISS)"""
import os, csv, glob, json, sys, collections

d = sys.argv[1]
GROUPS = (("extend", ("k_extend<false, false>", "k_extend_tasks", "k_resolve_extend")), ("shadow", ("k_shadow<false, false>", "k_shadow_tasks", "k_resolve_shadow")), ("shade", ("k_shade<false", "k_classify")),
          ("pair", ("k_trace_pair", "k_tasks_pair", "k_resolve_pair")),      # the fused traversal launches of a pipelined frame (PIPELINED=1 tools/profile_round.sh)
          ("generate", ("k_generate",)), ("accumulate", ("k_accumulate",)))
MAIN = {"pair": "k_trace_pair", "extend": "k_extend<false, false>", "shadow": "k_shadow<false, false>", "shade": "k_shade<false", "generate": "k_generate", "accumulate": "k_accumulate"}


def group_of(name):
    n = name.replace("void ptk::", "").replace("ptk::", "")
    for g, pre in GROUPS:
        if any(n.startswith(p) for p in pre):
            return g, n.startswith(MAIN[g])
    return None, False


cnt = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(int); dur = collections.defaultdict(float)
for f in sorted(glob.glob(d + "/**/p*_counter_collection.csv", recursive=True)):
    seen = set()
    first = "p1_" in f.split("/")[-1]
    for r in csv.DictReader(open(f)):
        g, main = group_of(r["Kernel_Name"])
        if not g:
            continue
        cnt[g][r["Counter_Name"]] += float(r["Counter_Value"])
        if first and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[g] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            if main:
                launches[g] += 1
stats = {}
for f in glob.glob(d + "/**/stats_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ptk::", "").replace("ptk::", "").split("(")[0]
        stats[n] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6, "total_ms": float(r["TotalDurationNs"]) * 1e-6, "percent": float(r["Percentage"])}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtxpt_amd
out = {"source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc (separate passes), 1 %s step of bench.py's default workload" % ("pipelined (product composition: fused traversal launches, compacted pool; rocprofv3 serialises the dispatches it counts)" if os.environ.get("PIPELINED") else "serial-kernel"),
       "kernel_source_sha256": rtxpt_amd.kernel_source_digest(),      # bench.py quotes these counters only for the kernels they were collected on
       "library_sha256": rtxpt_amd.library_digest(),                  # ... and the binary they ran in
       "issue_ceilings": {"total_instr_per_simd_cycle": 0.486, "four_cycle_class_instr_per_simd_cycle": 0.248, "source": "profiles/r06a_valu_ceiling.txt (tools/valu_ceiling on this GPU)"},
       "kernel_trace_stats": stats, "groups": {}}
for g, _ in GROUPS:
    c = cnt[g]
    if not c:
        continue
    n = max(1, launches[g])
    hbm = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
    e = {"launches_per_step": launches[g], "pmc_pass_ms_per_step": dur[g], "hbm_bytes_per_step": hbm, "hbm_bytes_per_launch": hbm / n,
         "hbm_counter_gbs": hbm / (dur[g] * 1e-3) / 1e9 if dur[g] else None,
         "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)),
         "valu_busy": c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0) if c.get("GRBM_GUI_ACTIVE") else None,
         "valu_instr_per_simd_cycle": c.get("SQ_INSTS_VALU", 0.0) / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0) if c.get("GRBM_GUI_ACTIVE") else None,
         "valu_share_of_issue_ceiling": c.get("SQ_INSTS_VALU", 0.0) / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0) / 0.486 if c.get("GRBM_GUI_ACTIVE") else None,
         "gpu_clock_ghz": c["GRBM_GUI_ACTIVE"] / 8.0 / (dur[g] * 1e-3) / 1e9 if c.get("GRBM_GUI_ACTIVE") and dur[g] else None,
         "lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else None,
         "wait_any_share_of_wave_cycles": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
         "valu_per_vmem_read": c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_VMEM_RD"] if c.get("SQ_INSTS_VMEM_RD") else None,
         "counters": dict(sorted(c.items()))}
    out["groups"][g] = e
print(json.dumps(out, indent=1))
