"""The committed bench line of the round (profiles/r03z_bench.json, produced by `python bench.py` on an MI355X) carries every field the bench contract asks for, and its
roofline figures are consistent with each other and with the committed rocprofv3 summaries. CPU only: nothing is run, the records are read."""
import csv, json, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = os.path.join(ROOT, "profiles", "r03z_bench.json")


def test_bench_line_has_the_contract_fields_and_consistent_numbers():
    d = json.loads(open(REC).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays per step / time per step
    assert abs(d["config"]["rays_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6 - d["value"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"):
        assert k in r, k
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["bound"] in ("hbm", "valu", "latency", "mfma")
    # achieved = algorithmic bytes per launch / average launch time
    bytes_per_launch = r["bytes_per_ray"] * d["config"]["extend_rays_per_step"] / (r["launches"] / 2)          # (the launches of the two serial-kernel steps)
    assert abs(bytes_per_launch / (r["avg_launch_ms"] * 1e-3) / 1e9 - r["achieved"]) < 0.02 * r["achieved"]
    # the counter file it quotes exists and says the same
    src = r["counters_source"].split(" ")[0]; c = json.load(open(os.path.join(ROOT, src)))["groups"]["extend"] if os.path.exists(os.path.join(ROOT, src)) else None
    c = c or json.load(open(os.path.join(ROOT, "profiles", "r03z_counters.json")))["groups"]["extend"]
    assert 0.5 < r["traffic"] / c["hbm_bytes_per_launch"] < 2.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"): assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    p = d["parity"]; assert p["differing_pixels"] == 0 and p["pixels"] == 1920 * 1080


def test_kernel_trace_summary_agrees_with_the_bench_line():
    d = json.loads(open(REC).read().strip().splitlines()[-1]); r = d["roofline"]
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r03z_serial_kernel_stats.csv"))))
    ext = [x for x in rows if x["Name"].startswith("void ptk::k_extend<false>")][0]
    avg_ms = float(ext["AverageNs"]) * 1e-6
    assert int(ext["Calls"]) == 27 and abs(avg_ms - r["avg_launch_ms"]) < 0.05 * r["avg_launch_ms"]          # rocprofv3's average k_extend launch vs bench.py's own HIP events


def test_final_tree_bench_line_and_its_realtime_leg():
    """profiles/r03zz_bench.json: the bench line of the round's final tree (same traversal / shading object code as r03z's) — the contract fields again, the parity block taken after the
    realtime leg ran on the same context, and that leg's figures consistent with the stand-alone probe (profiles/r03sq_stable_planes_probe.json: the single-batch fill pass)"""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03zz_bench.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "realtime_passes"):
        assert k in d, k
    assert abs(d["config"]["rays_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6 - d["value"]) < 1e-6 * d["value"]
    assert d["parity"]["differing_pixels"] == 0 and d["parity"]["pixels"] == 1920 * 1080
    old = json.loads(open(REC).read().strip().splitlines()[-1])
    assert abs(d["value"] / old["value"] - 1.0) < 0.02                                  # same kernels, same frame: within run-to-run noise of r03z
    rt = d["realtime_passes"]; assert "error" not in rt
    probe = json.load(open(os.path.join(ROOT, "profiles", "r03sq_stable_planes_probe.json")))
    assert rt["build_rays"] > 3840 * 2160 and abs(rt["build_rays"] / probe["build_pass"]["rays"] - 1.0) < 0.01      # (the build pass of another sample index: the camera jitter moves a few delta paths)
    assert abs(rt["fill_rays"] / (probe["fill_pass_one_subsample"]["extend_rays"] + probe["fill_pass_one_subsample"]["shadow_rays"]) - 1.0) < 0.01
    assert rt["fill_ms"] < probe["fill_pass_one_subsample"]["ms"]                       # pipelined batches against the probe's single batch
