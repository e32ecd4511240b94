"""Streaming frames (pt_set_stream_paths; rtxpt_amd/csrc/pt_api.hip pt_render) on the device (run with -m gpu): a batch generates its paths in slices and tops its extend
queue up after every bounce, so launches hold paths of different bounces. Launch composition must not change anything: the frame, the ray counts and the hit count
equal the frame that generates everything up front, bit for bit — with and without the tail kernel, for several in-flight sizes and batch counts, for a call that
continues an accumulation (first > 0), and for a NEE-AT frame whose feedback reservoirs are fed by the visibility rays."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _bits(a): return np.asarray(a).view(np.uint32)


def _tracer(scale=0.05, w=640, h=360, **settings):
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.bistro_like(scale=scale, tex_size=128)
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(scenes.default_settings(useFp16Types=1, **settings)); t.resize(w, h)
    return t


@pytest.mark.parametrize("tail", [0, 32768])
def test_streaming_frames_equal_the_up_front_frame(tail):
    """0.9 M paths: up front == 64 k / 128 k / 300 k in flight on 1, 2, 3 and 4 batches; every streaming frame needs more passes than the bounce bound alone allows."""
    t = _tracer(); spp = 4
    t.set_tail_paths(tail); t.set_stream_paths(0); t.reset_accumulation(); st = t.render(0, spp)
    ref = (t.radiance(), st["extendRays"], st["shadowRays"], st["hits"]); base_passes = st["iterations"]
    for batches, k in ((1, 65536), (2, 65536), (2, 131072), (3, 131072), (4, 65536), (1, 300000), (0, 100000)):
        t.set_stream_paths(k, batches); t.reset_accumulation(); st = t.render(0, spp)
        img = t.radiance()
        assert np.array_equal(_bits(img), _bits(ref[0])), "batches %d, %d in flight: %d pixels differ" % (batches, k, int((_bits(img) != _bits(ref[0])).any(-1).sum()))
        assert (st["extendRays"], st["shadowRays"], st["hits"]) == ref[1:], "batches %d, %d in flight: counts %s vs %s" % (batches, k, (st["extendRays"], st["shadowRays"], st["hits"]), ref[1:])
        assert st["iterations"] > base_passes, "batches %d, %d in flight: %d passes — the frame was not streamed" % (batches, k, st["iterations"])
    t.close()


def test_streaming_continues_an_accumulation():
    """Samples 0..1 up front, then samples 2..5 streamed == samples 0..5 up front: the accumulation weights and the sample indices of streamed slices."""
    t = _tracer(w=512, h=288)
    t.set_stream_paths(0); t.reset_accumulation(); t.render(0, 6); ref = t.radiance()
    t.reset_accumulation(); t.render(0, 2); t.set_stream_paths(70000, 2); t.render(2, 4)
    assert np.array_equal(_bits(t.radiance()), _bits(ref))
    t.close()


def test_streaming_nested_dielectrics_and_the_bounce_bound():
    """Nested dielectrics quality 2 (rejected hits re-trace without a bounce): the bounce bound of a streaming frame counts from the pass that carries the last fresh paths."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.bistro_like(scale=0.05, tex_size=128, animated=True)
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(640, 360, **cam)); t.set_settings(scenes.default_settings(useFp16Types=1, nestedDielectricsQuality=2)); t.resize(640, 360)
    t.set_tail_paths(0); t.set_stream_paths(0); st = t.render(0, 2); ref = (t.radiance(), st["extendRays"], st["shadowRays"])
    for tail in (0, 32768):
        t.set_tail_paths(tail); t.set_stream_paths(65536, 2); t.reset_accumulation(); st = t.render(0, 2)
        assert np.array_equal(_bits(t.radiance()), _bits(ref[0])) and (st["extendRays"], st["shadowRays"]) == ref[1:]
    t.close()


def test_streaming_neeat_feedback_frame():
    """NEE-AT with temporal feedback (the visibility rays feed per-pixel reservoirs through the shadow queue's fourth word group): reservoirs and frame unchanged by streaming."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    import pin_scenes
    CASES = pin_scenes.neeat_cases()
    name = "bistro_like_neeat_lp16" if "bistro_like_neeat_lp16" in CASES else sorted(CASES)[0]
    make, S, w, h, first, n, opts = CASES[name]
    if int(S["NEEFullSamples"]) > 1: pytest.skip("grouped NEE samples")
    sc, cam = make()
    out = []
    for k in (0, 4096):
        t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
        baked = len(t.lights()["lights"])
        t.set_local_light_sampling(pin_scenes.neeat_table(opts, baked, w, h), jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
        t.set_stream_paths(k, 1); st = t.render(first, n)
        fb = [t.light_feedback(s) for s in range(n)] if opts["feedback"] else []
        out.append((t.radiance(), st["extendRays"], st["shadowRays"], fb)); t.close()
    for o in out[1:]:
        assert np.array_equal(_bits(out[0][0]), _bits(o[0])) and out[0][1:3] == o[1:3]
        for (w0, c0), (w1, c1) in zip(out[0][3], o[3]): assert np.array_equal(_bits(w0), _bits(w1)) and np.array_equal(c0, c1)
