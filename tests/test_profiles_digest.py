"""profiles/r05z_counters.json (the rocprofv3 --pmc summary bench.py quotes in roofline{}) names the kernel sources it was collected on by their SHA-256
(rtxpt_amd.kernel_source_digest: every .h / .hip under rtxpt_amd/csrc). bench.py quotes the counters only on a match; this test says so ahead of a bench run:
it passes on a match and SKIPS — with the command to run — when a kernel source has changed since the last tools/profile_round.sh."""
import json
import os

import pytest

import rtxpt_amd as pt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counters_file_names_its_sources():
    import bench
    path = os.path.join(ROOT, "profiles", bench.COUNTERS_FILE)
    assert os.path.exists(path), "bench.py quotes %s" % path
    d = json.load(open(path))
    assert len(d.get("kernel_source_sha256", "")) == 64 and set(d["groups"]) >= {"extend", "shade", "shadow"}
    for g in ("extend", "shade", "shadow"):
        assert d["groups"][g]["hbm_bytes_per_launch"] > 0 and 0.0 < d["groups"][g]["lane_utilisation"] <= 1.0
    digest = pt.kernel_source_digest()
    assert len(digest) == 64 and digest == pt.kernel_source_digest()
    if d["kernel_source_sha256"] != digest:
        pytest.skip("kernel sources changed since the counters were taken: run `mkdir -p gpurun_out/p && tools/profile_round.sh gpurun_out/p` on the GPU box and copy "
                    "counters.json to profiles/%s (bench.py prints bound \"unknown\" until then)" % bench.COUNTERS_FILE)
