"""Fused traversal launches (pt_set_fused_traversal; rtxpt_amd/csrc/pt_wavefront.hip k_trace_pair, pt_api.hip pt_render) on the device (run with -m gpu): the visibility rays of
path vertex k are traced in the launch that traces the closest-hit rays of vertex k + 1, block by block, with shared straggler rounds and resolve passes. Launch composition must
not change anything: the frame, the ray counts and the hit count equal the frame of separate launches, bit for bit — on one, two and four pipelined batches, with and without the
tail kernel (which makes a batch trace its pending visibility rays first), when nearly every ray goes through the straggler rounds, for a call that continues an accumulation, with
nested dielectrics (rejected hits re-trace), on tile shards, and for a NEE-AT frame whose feedback reservoirs are fed by the visibility rays. Semantics preserved:
/root/reference/Rtxpt/Shaders/PathTracer/PathTracerNEE.hlsli:185-275 (the light sample of a vertex lands before the emission of the next one is added)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _bits(a): return np.asarray(a).view(np.uint32)


def _tracer(scale=0.05, w=640, h=360, animated=False, shard=(0, 1), **settings):
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.bistro_like(scale=scale, tex_size=128, animated=animated)
    t = pt.PathTracer(shard_rank=shard[0], shard_count=shard[1]); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(scenes.default_settings(useFp16Types=1, **settings)); t.resize(w, h)
    return t


def _frame(t, first, n):
    t.reset_accumulation(); st = t.render(first, n)
    return t.radiance(), (st["extendRays"], st["shadowRays"], st["hits"])


@pytest.mark.parametrize("tail", [0, 32768])
@pytest.mark.parametrize("size", [(320, 180, 2), (640, 360, 4), (1280, 720, 4)])      # 0.1 M paths: one batch; 0.9 M: one; 3.7 M: four pipelined batches
def test_fused_frames_equal_separate_launches(size, tail):
    w, h, spp = size
    t = _tracer(w=w, h=h)
    t.set_tail_paths(tail)
    t.set_fused_traversal(0); ref = _frame(t, 0, spp)
    t.set_fused_traversal(1); got = _frame(t, 0, spp)
    assert np.array_equal(_bits(got[0]), _bits(ref[0])), "%d pixels differ" % int((_bits(got[0]) != _bits(ref[0])).any(-1).sum())
    assert got[1] == ref[1]
    t.set_fused_traversal(2); auto = _frame(t, 0, spp)
    assert np.array_equal(_bits(auto[0]), _bits(ref[0])) and auto[1] == ref[1]
    t.close()


def test_fused_two_batches_and_a_continued_accumulation():
    """1.6 M paths per call (two pipelined batches); samples 0..1 separate, then samples 2..4 fused == samples 0..4 separate."""
    t = _tracer(w=1024, h=520)
    t.set_fused_traversal(0); t.reset_accumulation(); t.render(0, 5); ref = t.radiance()
    t.reset_accumulation(); t.render(0, 2); t.set_fused_traversal(1); t.render(2, 3)
    assert np.array_equal(_bits(t.radiance()), _bits(ref))
    t.close()


def test_fused_nested_dielectrics_animated_scene():
    """C5's scene, nested dielectrics quality 2 (rejected hits re-trace without a bounce: more passes, some without visibility rays), refit pose."""
    from rtxpt_amd import scenes
    t = _tracer(animated=True, nestedDielectricsQuality=2)
    for tail in (0, 32768):
        t.set_tail_paths(tail)
        t.set_fused_traversal(0); ref = _frame(t, 0, 2)
        t.set_fused_traversal(1); got = _frame(t, 0, 2)
        assert np.array_equal(_bits(got[0]), _bits(ref[0])) and got[1] == ref[1]
    t.close()


def test_fused_tile_shards():
    """ranks 0 and 2 of 3: the owned tiles equal the separate-launch frame's, nothing else is written"""
    for rank in (0, 2):
        t = _tracer(shard=(rank, 3))
        t.set_fused_traversal(0); ref = _frame(t, 0, 3)
        t.set_fused_traversal(1); got = _frame(t, 0, 3)
        assert np.array_equal(_bits(got[0]), _bits(ref[0])) and got[1] == ref[1]
        t.close()


def test_fused_cornell_and_no_nee():
    """C2 (environment + emissive NEE on a small closed scene) and a frame without NEE (no visibility rays at all: every launch is a plain closest-hit launch)."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C2"); S = scenes.config_settings("C2")
    for nee in (1, 0):
        S2 = S.copy(); S2["NEEEnabled"] = nee
        t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(480, 270, **cam)); t.set_settings(S2); t.resize(480, 270)
        t.set_fused_traversal(0); ref = _frame(t, 0, 4)
        t.set_fused_traversal(1); got = _frame(t, 0, 4)
        assert np.array_equal(_bits(got[0]), _bits(ref[0])) and got[1] == ref[1]
        if not nee: assert ref[1][1] == 0
        t.close()


def test_fused_neeat_feedback_frame():
    """NEE-AT with temporal feedback (the visibility rays feed per-pixel reservoirs through the shadow queue's fourth word group): reservoirs and frame unchanged by fusing."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    import pin_scenes
    CASES = pin_scenes.neeat_cases()
    name = "bistro_like_neeat_lp16" if "bistro_like_neeat_lp16" in CASES else sorted(CASES)[0]
    make, S, w, h, first, n, opts = CASES[name]
    if int(S["NEEFullSamples"]) > 1: pytest.skip("grouped NEE samples")
    sc, cam = make()
    out = []
    for mode in (0, 1):
        t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
        baked = len(t.lights()["lights"])
        t.set_local_light_sampling(pin_scenes.neeat_table(opts, baked, w, h), jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
        t.set_fused_traversal(mode); st = t.render(first, n)
        fb = [t.light_feedback(s) for s in range(n)] if opts["feedback"] else []
        out.append((t.radiance(), st["extendRays"], st["shadowRays"], fb)); t.close()
    assert np.array_equal(_bits(out[0][0]), _bits(out[1][0])) and out[0][1:3] == out[1][1:3]
    for (w0, c0), (w1, c1) in zip(out[0][3], out[1][3]): assert np.array_equal(_bits(w0), _bits(w1)) and np.array_equal(c0, c1)
