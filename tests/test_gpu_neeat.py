"""NEE-AT, the path tracer's side, on the device (run with -m gpu): pt_set_local_light_sampling / pt_get_light_feedback.

  * against the REFERENCE TEXT's frames and feedback planes (tests/golden/neeat_golden.npz) with no oracle code in the loop;
  * against the oracle on the same inputs, ray counts included;
  * the deferred feedback: the reservoir update and the Russian-roulette outcome of the "visible" case are applied by k_shadow (pt_path.h ShadowRequest) —
    the frames above contain thousands of vertices whose two roulette outcomes differ, and a sample-by-sample run must equal one call tracing both samples;
  * what the API refuses."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rtxpt_amd import scenes
import pin_scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neeat_golden.npz")
CASES = pin_scenes.neeat_cases()


def _tracer(name, num_lights=None, serial=False):
    import rtxpt_amd as pt
    make, S, w, h, first, n, opts = CASES[name]
    sc, cam = make()
    t = pt.PathTracer(serial_kernels=serial); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
    baked = len(t.lights()["lights"])
    if num_lights is not None: assert baked == num_lights, "the device baked %d lights, the reference-text fixture was made with %d" % (baked, num_lights)
    t.set_local_light_sampling(pin_scenes.neeat_table(opts, baked, w, h), jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
    return t, first, n, opts


def _bits(a): return np.asarray(a).view(np.uint32)


@pytest.mark.parametrize("name", list(CASES))
def test_device_matches_reference_text(name):
    g = np.load(GOLDEN)
    t, first, n, opts = _tracer(name, int(g[name + "_lights"][0]))
    stats = t.render(first, n)
    got, want = t.radiance(), g[name]
    bad = (_bits(got) != _bits(want)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert (int(stats["extendRays"]), int(stats["shadowRays"])) == tuple(int(v) for v in g[name + "_rays"])
    if opts["feedback"]:
        for s in range(n):
            w, c = t.light_feedback(s)
            assert np.array_equal(_bits(w), _bits(g["%s_fbw%d" % (name, s)])), "%s: feedback weights of sample %d" % (name, s)
            assert np.array_equal(c, g["%s_fbc%d" % (name, s)]), "%s: feedback candidates of sample %d" % (name, s)
    t.close()


@pytest.mark.parametrize("name", ["bistro_like_neeat", "bistro_like_neeat_lp16", "c2_neeat_table_only_nee3"])
def test_device_matches_oracle(name):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_neeat_golden
    want = make_neeat_golden.frame(name, False)
    t, first, n, opts = _tracer(name, int(want[name + "_lights"][0]), serial=True)      # one batch, one stream: the other launch configuration
    t.render(first, n)
    assert np.array_equal(_bits(t.radiance()), _bits(want[name]))
    if opts["feedback"]:
        for s in range(n):
            w, c = t.light_feedback(s)
            assert np.array_equal(_bits(w), _bits(want["%s_fbw%d" % (name, s)])) and np.array_equal(c, want["%s_fbc%d" % (name, s)])
    t.close()


def test_one_call_equals_sample_by_sample():
    """pt_render(first, 2) keeps one feedback plane per sample; tracing the samples in two calls gives the same planes and the same accumulated frame."""
    t, first, n, opts = _tracer("bistro_like_neeat")
    t.render(first, 2); whole = t.radiance(); planes = [t.light_feedback(s) for s in range(2)]
    t.reset_accumulation()
    for s in range(2):
        t.render(first + s, 1)
        w, c = t.light_feedback(0)
        assert np.array_equal(_bits(w), _bits(planes[s][0])) and np.array_equal(c, planes[s][1])
        with pytest.raises(Exception): t.light_feedback(1)
    assert np.array_equal(_bits(t.radiance()), _bits(whole))
    t.close()


def test_removing_the_local_layer_restores_the_plain_frame():
    import rtxpt_amd as pt
    make, S, w, h, first, n, opts = CASES["bistro_like_neeat"]
    t, _, _, _ = _tracer("bistro_like_neeat")
    t.render(first, n); with_table = t.radiance()
    t.set_local_light_sampling(None, feedback=False); t.reset_accumulation(); t.render(first, n); removed = t.radiance()
    with pytest.raises(Exception): t.light_feedback(0)
    sc, cam = make()
    p = pt.PathTracer(); p.set_scene(sc); p.set_settings(S); p.set_camera(scenes.bridge_camera(w, h, **cam)); p.resize(w, h); p.render(first, n)
    assert np.array_equal(_bits(removed), _bits(p.radiance())) and not np.array_equal(_bits(with_table), _bits(removed))
    t.close(); p.close()


def test_api_refuses_bad_tables():
    import rtxpt_amd as pt
    make, S, w, h, first, n, opts = CASES["bistro_like_neeat"]
    t, _, _, _ = _tracer("bistro_like_neeat")
    good = pin_scenes.neeat_table(opts, len(t.lights()["lights"]), w, h)
    bad = good.copy(); bad[0, 0, 5], bad[0, 0, 6] = good[0, 0, 100], good[0, 0, 0]
    with pytest.raises(Exception, match="sorted"): t.set_local_light_sampling(bad)
    with pytest.raises(Exception, match="jitter"): t.set_local_light_sampling(good, jitter=(8, 0))
    with pytest.raises(Exception, match="0.95"): t.set_local_light_sampling(good, ratio=1.5)
    t.set_local_light_sampling(good[:2, :2].copy())
    with pytest.raises(Exception, match="smaller than the frame"): t.render(0, 1)
    far = good.copy(); far[-1, -1, -1] = (0x7FFFF0 << 9)
    t.set_local_light_sampling(far)
    with pytest.raises(Exception, match="beyond the baked light table"): t.render(0, 1)
    S3 = S.copy(); S3["NEEFullSamples"] = 3
    t.set_local_light_sampling(good, feedback=True); t.set_settings(S3)
    with pytest.raises(Exception, match="NEEFullSamples 1"): t.render(0, 1)
    t.set_local_light_sampling(good, feedback=False); t.render(0, 1)      # the local layer alone works with any NEEFullSamples
    t.close()
