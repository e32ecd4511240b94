"""The committed end-of-round bench bundle — the NEWEST profiles/rNNz_bench.json (the line `python bench.py` printed on an MI355X), profiles/rNNz_counters.json (tools/profile_round.sh:
rocprofv3 --pmc in separate passes) and profiles/rNNz_serial_kernel_stats.csv (rocprofv3 --kernel-trace --stats of serial-kernel steps) — carries every field the bench contract asks for,
and its roofline figures are consistent with each other and with the committed rocprofv3 summaries. CPU only: nothing is run, the records are read."""
import csv, glob, json, os, re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bundle():
    tags = sorted(re.match(r"(r\d\dz)_bench\.json", os.path.basename(f)).group(1) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]z_bench.json")))
    return tags[-1]


TAG = _newest_bundle()
REC = os.path.join(ROOT, "profiles", TAG + "_bench.json")


def _line(): return json.loads(open(REC).read().strip().splitlines()[-1])


def test_the_bundle_is_this_rounds_or_the_last_ones():
    assert int(TAG[1:3]) >= 5 and os.path.exists(os.path.join(ROOT, "profiles", TAG + "_counters.json")) and os.path.exists(os.path.join(ROOT, "profiles", TAG + "_serial_kernel_stats.csv"))


def test_bench_line_has_the_contract_fields_and_consistent_numbers():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["config"]["rays_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6 - d["value"]) < 1e-6 * d["value"]          # value = rays per step / time per step
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"):
        assert k in r, k
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["bound"] in ("hbm", "valu", "issue", "latency", "mfma")
    # achieved = algorithmic bytes per launch / average launch time (the launches of the two serial-kernel steps)
    bytes_per_launch = r["bytes_per_ray"] * d["config"]["extend_rays_per_step"] / (r["launches"] / 2)
    assert abs(bytes_per_launch / (r["avg_launch_ms"] * 1e-3) / 1e9 - r["achieved"]) < 0.02 * r["achieved"]
    # the counter file it quotes is the bundle's own, collected on the sources the line names, and says the same
    assert r["counters_source"].split(" ")[0] == "profiles/" + TAG + "_counters.json"
    cj = json.load(open(os.path.join(ROOT, "profiles", TAG + "_counters.json")))
    assert cj["kernel_source_sha256"] == d["config"]["kernel_source_sha256"]
    if "library_sha256" in cj: assert cj["library_sha256"] == d["config"]["library_sha256"]
    assert 0.5 < r["traffic"] / cj["groups"]["extend"]["hbm_bytes_per_launch"] < 2.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"): assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    p = d["parity"]; assert p["differing_pixels"] == 0 and p["pixels"] == 1920 * 1080
    rt = p["reference_text"]; assert rt["frame_sha256_equal"] is True and rt["ray_counts_equal"] is True and rt["differing_pixels_in_kept_rows"] == 0


def test_kernel_trace_summary_agrees_with_the_bench_line():
    r = _line()["roofline"]
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", TAG + "_serial_kernel_stats.csv"))))
    ext = [x for x in rows if x["Name"].startswith("void ptk::k_extend<false")][0]
    avg_ms = float(ext["AverageNs"]) * 1e-6
    assert int(ext["Calls"]) % (r["launches"] // 2) == 0 and abs(avg_ms - r["avg_launch_ms"]) < 0.06 * r["avg_launch_ms"]          # rocprofv3's average k_extend launch vs bench.py's own HIP events
