"""NEE-AT with the light baker in the loop — oracle pinning on the CPU.

Between two frames the reference's LightsBaker turns the path tracer's per-pixel feedback into the next frame's samplers: PreFilter, P0 (usage counts), the proxy
counts' feedback term, P1a / P1b (one candidate per pixel), P2 / P3 (tile tables), ClearFeedbackHistory (LightsBaker.hlsl:753-830, 880-948, 1062-1855; host side
LightsBaker.cpp:943-1420). oracle/ptref/neeat.h restates the passes; here whole runs — baker, path tracer, baker, ... — are held against what the reference's own
text produces frame by frame (tests/golden/neeat_loop_golden.npz; live where /root/reference exists, the passes executed thread by thread with their group-shared
memory and barriers: oracle/refpin/hlsl_lbfb_stubs.h)."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from rtxpt_amd import scenes
import pin_scenes
import make_neeat_loop_golden as loop

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neeat_loop_golden.npz")
CASES = pin_scenes.neeat_loop_cases()


def compare(name, got, want, keys=None):
    for k in (keys or want):
        a, b = np.asarray(got[k]), np.asarray(want[k])
        assert a.shape == b.shape, k
        same = a.view(np.uint32) == b.view(np.uint32) if a.dtype == np.float32 else a == b
        assert same.all(), "%s: %d of %d values of %s differ from the reference text" % (name, int((~same).sum()), same.size, k)


def golden(name):
    g = np.load(GOLDEN)
    return {k: g[k] for k in g.files if k == name or (k.startswith(name + "_") and k[len(name) + 1:].rstrip("0123456789") in ("table", "jitter", "counters", "fbw", "fbc"))}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_run_matches_reference_text_golden(name):
    want = golden(name)
    got = loop.run_oracle(name, False)
    assert set(got) == set(want)
    compare(name, got, want)
    frames = CASES[name][4]
    # the run does what it is for: tables start as draws from the global sampler and concentrate once feedback exists; the global table follows the usage counts
    distinct = [np.mean([len(np.unique(t >> 9)) for t in want["%s_table%d" % (name, f)].reshape(-1, 128)]) for f in range(frames)]
    assert distinct[-1] < distinct[0] or name.startswith("bistro_like_loop")
    assert not np.array_equal(want["%s_counters0" % name], want["%s_counters1" % name])
    for f in range(frames):
        t = want["%s_table%d" % (name, f)]; lights, counts = t >> 9, (t & 0x1FF) + 1
        assert (np.diff(lights.astype(np.int64), axis=-1) >= 0).all()
        tile = t.reshape(-1, 128)[len(t.reshape(-1, 128)) // 2]; l, c = tile >> 9, (tile & 0x1FF) + 1
        for v in np.unique(l): assert (c[l == v] == (l == v).sum()).all()


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_run_matches_live_reference_text(name):
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine: the reference-text passes cannot be built here")
    compare(name, loop.run_oracle(name, False), loop.run_oracle(name, True))


def test_jitter_sequence():
    """LightsBaker::UpdateLocalJitter (LightsBaker.cpp:943-962): the R2 sequence in tile pixels"""
    g, x, y, seq = 1.32471795724474602596, 0.0, 0.0, []
    for _ in range(4):
        x = float(np.float32(np.fmod(np.float32(x) + np.float32(1.0 / g), np.float32(1.0)))); y = float(np.float32(np.fmod(np.float32(y) + np.float32(1.0 / (g * g)), np.float32(1.0))))
        seq.append((min(int(x * 8), 7), min(int(y * 8), 7)))
    want = golden("bistro_like_loop")
    assert [tuple(int(v) for v in want["bistro_like_loop_jitter%d" % f]) for f in range(4)] == seq


def test_reset_and_disable():
    from oracle import ptref
    make, S, w, h, frames, opts = CASES["bistro_like_loop"]
    sc, cam = make(); o = ptref.Oracle()
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h)
    o.render(0, 1); plain = o.radiance().copy(); o.reset_accumulation()
    o.set_neeat(True, **opts); o.render(0, 2); first = o.radiance().copy(); t1 = o.neeat_tables()[0].copy(); o.reset_accumulation()
    o.neeat_reset(); o.render(0, 2)                                          # a reset run repeats the first one
    assert np.array_equal(first.view(np.uint32), o.radiance().view(np.uint32)) and np.array_equal(t1, o.neeat_tables()[0])
    o.reset_accumulation(); o.render(0, 2)                                   # without the reset the history carries on
    assert not np.array_equal(t1, o.neeat_tables()[0])
    o.set_neeat(False); o.reset_accumulation(); o.render(0, 1)               # off: the plain global sampler again
    assert np.array_equal(plain.view(np.uint32), o.radiance().view(np.uint32))


def _tiny_run(kind, w, h, frames=4):
    """kind: 'oracle' | 'reference' | 'device'. Cornell box C2 at frame sizes around and below one tile (8 x 8) and one low-resolution pixel (2 x 2)."""
    sc, cam = scenes.cornell_box("C2"); S = scenes.default_settings(NEEType=2); camd = scenes.bridge_camera(w, h, **cam)
    if kind == "device":
        import rtxpt_amd as pt
        t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(w, h); t.set_neeat(True)
    else:
        from oracle import ptref
        t = ptref.Oracle(reference_integrator=(kind == "reference"), settings=S); t.set_scene(sc); t.set_camera(camd); t.set_settings(S); t.resize(w, h); t.set_neeat(True)
    out = []
    for f in range(frames):
        t.render(f, 1); tab = t.neeat_tables(); fw, fc = t.light_feedback(0)
        out.append(dict(radiance=t.radiance().copy(), table=tab[0], jitter=np.array(tab[1], np.uint32), fbw=fw, fbc=fc, counters=t.lights()["proxyCounters"]))
    t.close()
    return out


TINY = [(1, 1), (3, 2), (8, 8), (9, 17), (33, 5)]


@pytest.mark.parametrize("w,h", TINY, ids=["%dx%d" % s for s in TINY])
def test_tiny_frames_oracle_matches_live_reference_text(w, h):
    """Mirror coordinates, clamped neighbourhoods, partial tiles and a jittered grid larger than the frame: the restatement against the text where the edge handling is everything"""
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine")
    a, b = _tiny_run("oracle", w, h), _tiny_run("reference", w, h)
    for f, (x, y) in enumerate(zip(a, b)): compare("%dx%d frame %d" % (w, h, f), x, y)
