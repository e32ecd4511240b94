"""NEE-AT, the path tracer's side (first part) — oracle pinning on the CPU.

The reference's default light sampling (CommandLine.h:42 NEEType = 2) draws part of every NEE candidate set from a screen-tile local sampler, weighs the two
samplers against each other (inner MIS) and against the BSDF (outer MIS, both pdfs), and offers every visible light sample to its pixel's feedback reservoir
(LightSampler.hlsli:51-93,120-200,242-268,318-332,411-420; PathTracerNEE.hlsli:88-161,199-273; LightingTypes.hlsli:148-320). The oracle restates that; here it
is held against frames and feedback planes the reference's own text produced (tests/golden/neeat_golden.npz, made by tests/golden/make_neeat_golden.py) and,
where /root/reference exists, against the text compiled live. The tile tables are synthetic stand-ins for the baker's output."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from rtxpt_amd import scenes
import pin_scenes
import make_neeat_golden

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neeat_golden.npz")
CASES = pin_scenes.neeat_cases()


def _compare(name, got, want):
    for k in want:
        a, b = np.asarray(got[k]), np.asarray(want[k])
        assert a.shape == b.shape, k
        same = a.view(np.uint32) == b.view(np.uint32) if a.dtype == np.float32 else a == b
        assert same.all(), "%s: %d of %d values of %s differ from the reference text" % (name, int((~same).sum()), same.size, k)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_text_golden(name):
    g = np.load(GOLDEN)
    want = {k: g[k] for k in g.files if k == name or (k.startswith(name + "_") and k[len(name) + 1:].split("_")[0][:3] in ("ray", "lig", "fbw", "fbc"))}
    got = make_neeat_golden.frame(name, False)
    assert set(got) == set(want)
    _compare(name, got, want)
    assert want[name][..., :3].max() > 0
    if CASES[name][6]["feedback"]:
        w0, c0 = want[name + "_fbw0"], want[name + "_fbc0"]
        assert (w0 > 0).sum() > 100 and ((c0 != 0xFFFFFFFF) == (w0 > 0)).all()      # a slot has a candidate exactly when it received weight
        ssc = (c0 != 0xFFFFFFFF) & ((c0 >> 31) != 0)
        thr = CASES[name][6]["ssc_threshold"]
        assert ssc.any() == (thr > 0) and (thr < 1e8 or ssc.sum() == (w0 > 0).sum())


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_live_reference_text(name):
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine: the reference-text integrator cannot be built here")
    _compare(name, make_neeat_golden.frame(name, False), make_neeat_golden.frame(name, True))


def test_local_layer_changes_the_frame_and_only_where_it_should():
    """The table is not decoration: with it the default-settings frame differs from the global-sampler frame; with threshold 0 (no vertex coherent) the same table
    leaves the frame exactly as it is without one."""
    from oracle import ptref
    make, S, w, h, first, n, opts = CASES["bistro_like_neeat"]
    def run(table, thr):
        sc, cam = make(); o = ptref.Oracle()
        o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.L.ptref_prepare(o.h)
        t = pin_scenes.neeat_table(opts, len(o.lights()["lights"]), w, h) if table else None
        o.set_local_light_sampling(t, jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=thr, feedback=False); o.render(first, n)
        return o.radiance()
    plain, with_table, never = run(False, 0.3), run(True, 0.3), run(True, 0.0)
    assert (plain.view(np.uint32) != with_table.view(np.uint32)).any(-1).mean() > 0.3
    assert np.array_equal(plain.view(np.uint32), never.view(np.uint32))


def test_candidate_counts_and_packing():
    """ComputeCandidateSampleLocalCount (LightingTypes.hlsli:148-155): always one global candidate, then the ratio; the packing of a table entry (:172-175)."""
    f = lambda ratio, total: int(np.float32(np.float32(total - 1) * np.float32(ratio)) + np.float32(0.75))
    assert [f(0.65, k) for k in (1, 2, 5, 9)] == [0, 1, 3, 5] and f(0.0, 5) == 0 and f(0.95, 5) == 4
    t = scenes.synthetic_local_light_tables(1000, 64, 36, seed=3)
    assert t.shape == ((36 + 14) // 8, (64 + 14) // 8, 128)
    lights, counts = t >> 9, (t & 0x1FF) + 1
    assert (np.diff(lights.astype(np.int64), axis=-1) >= 0).all() and lights.max() < 1000
    for tile in t.reshape(-1, 128)[:8]:      # every entry of a light carries that light's number of proxies in the tile: the pdf is count / 128
        l, c = tile >> 9, (tile & 0x1FF) + 1
        for v in np.unique(l): assert (c[l == v] == (l == v).sum()).all()
