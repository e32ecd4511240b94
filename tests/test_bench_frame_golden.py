"""The frame bench.py times, as the REFERENCE'S integrator text renders it (tests/golden/bench_frame_golden.npz: SHA-256 of the 3840x2160 RGBA32F frame, every 120th row, ray counts;
made by tests/golden/make_bench_frame_golden.py). On the CPU the oracle is checked against the kept rows (the whole frame would take minutes here); bench.py and
tests/test_gpu_full_size.py compare the device's whole frame with the digest."""
import os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from rtxpt_amd import scenes
from oracle import ptref
import make_bench_frame_golden as gen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_frame_golden.npz")


def test_fixture_is_the_bench_workload():
    g = np.load(GOLD)
    assert tuple(int(v) for v in g["size"]) == (3840, 2160, 4) and int(g["row_step"][0]) == gen.ROW_STEP and g["rows"].shape == (18, 3840, 4) and g["sha256"].shape == (32,)
    assert np.isfinite(g["rows"]).all() and (g["rows"][..., :3] >= 0).all() and np.all(g["rows"][..., 3] == 1.0) and g["rows"][..., :3].max() > 0
    assert int(g["rays"][0]) > 3 * 3840 * 2160 * 4 // 2 and 0 < int(g["rays"][1]) < int(g["rays"][0])
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")).read()      # bench.py's defaults are the generator's workload
    for needle in ('sc["env_cube_dim"] = 2048', "useFp16Types=0 if args.fp32_lp_types else 1", "bench_frame_golden.npz"): assert needle in src, needle


def test_oracle_rows_equal_the_reference_text_frame():
    """six of the kept rows of the 4K bench frame (23 040 pixels x 4 samples) through the oracle, bit for bit against what the reference's text rendered"""
    g = np.load(GOLD)
    sc, cam, S = gen.bench_workload()
    o = ptref.Oracle(lp16=True); o.set_scene(sc); o.set_camera(scenes.bridge_camera(gen.W, gen.H, **cam)); o.set_settings(S); o.resize(gen.W, gen.H)
    for k in range(0, 18, 3):
        y = k * gen.ROW_STEP
        o.reset_accumulation(); o.render(0, gen.SPP, rect=(0, y, gen.W, y + 1))
        got = o.radiance()[y]
        bad = int((got.view(np.uint32) != g["rows"][k].view(np.uint32)).any(-1).sum())
        assert bad == 0, "row %d: %d pixels differ from the reference-text frame" % (y, bad)
    o.close()


CONFIG_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_frames_golden.npz")


def test_oracle_equals_the_reference_text_config_frames():
    """tests/golden/config_frames_golden.npz (BASELINE.json's C1, C2, C4, C5 at full size through the reference's integrator text): the oracle's whole C1 frame by digest, the kept
    rows of C2, two kept rows of C4 (16 samples) and of an animated C5 pose."""
    import make_config_frames_golden as cg
    import pin_scenes
    g = np.load(CONFIG_GOLD)
    for name, rows_checked in (("C1", None), ("C2", (0, 1, 2, 3)), ("C4", (1,)), ("C5_t2", (2,))):
        make, S, w, h, first, n, t = cg.configs()[name]
        sc, cam = make()
        o = ptref.Oracle(lp16=bool(int(S["useFp16Types"]))); o.set_scene(cg.posed(sc, t)); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h)
        if rows_checked is None:
            o.render(first, n); c = o.counters()
            assert np.array_equal(pin_scenes.frame_digest(o.radiance()), g[name + "_sha256"]) and (c["extendRays"], c["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])
        else:
            for k in rows_checked:
                y = cg.rows_of(h)[k]
                o.reset_accumulation(); o.render(first, n, rect=(0, y, w, y + 1))
                bad = int((o.radiance()[y].view(np.uint32) != g[name + "_rows"][k].view(np.uint32)).any(-1).sum())
                assert bad == 0, "%s row %d: %d pixels differ from the reference-text frame" % (name, y, bad)
        o.close()
    for name in cg.configs(): assert g[name + "_sha256"].shape == (32,) and np.isfinite(g[name + "_rows"]).all() and int(g[name + "_rays"][0]) > 0
