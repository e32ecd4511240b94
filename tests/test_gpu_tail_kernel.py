"""The tail kernel (rtxpt_amd/csrc/pt_tail.hip, pt_set_tail_paths) on the device (run with -m gpu): one launch in which every wave runs 32 paths to their end —
trace, shade, visibility, next bounce — instead of a chain of wavefront passes. The image must not depend on where the hand-over happens:

  * WHOLE frames through the tail kernel (threshold above the frame size) against the frames of the REFERENCE'S integrator text, both lp builds, ray counts included,
    and the NEE-AT frames (feedback reservoirs, the roulette fix-up of the visible case) against the reference-text NEE-AT fixtures — no oracle call;
  * the hand-back path: with MI355PT_TAIL_DEFER = 8 nearly every ray is "a straggler" — its path leaves the kernel (state of the bounce's start, or the shadow request
    after shading) and the wavefront kernels finish it; the frame stays the same;
  * frames that start as wavefront passes and end in the tail kernel equal the all-wavefront frame bit for bit for several thresholds.
(tests/conftest.py switches the tail kernel off for every other test so that small frames keep driving the wavefront kernels.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
WHOLE = 1 << 30

NAMES = ["c1", "c2", "c2_firefly", "c2_nee_off", "c2_nested2_norr_nold", "c2_nested0_uniform", "c2_sphere_lights", "c2_exclude_from_nee", "c2_env_rotated_mip2", "c2_mirrored_room",
         "bistro_like", "bistro_like_material_zoo", "bistro_like_c5", "c2_spec_gloss", "bistro_like_spec_gloss", "c2_sphere_light_proxy", "c2_sun_discs_bc6"]


def _bits(a): return np.asarray(a).view(np.uint32)


def _render(case, tail, serial=False):
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    make, S, w, h, first, n = case
    sc, cam = make()
    t = pt.PathTracer(serial_kernels=serial); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(S); t.resize(w, h)
    t.set_tail_paths(tail)
    st = t.render(first, n)
    img = t.radiance(); t.close()
    return img, st


@pytest.mark.parametrize("lp16", [False, True], ids=["fp32", "lp16"])
@pytest.mark.parametrize("name", NAMES)
def test_whole_frame_through_the_tail_kernel_matches_reference_text(name, lp16):
    import pin_scenes
    case = (pin_scenes.cases_lp16() if lp16 else pin_scenes.cases())[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden_lp16.npz" if lp16 else "reference_integrator_golden.npz"))
    got, st = _render(case, WHOLE)
    bad = (_bits(got) != _bits(g[name])).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert (st["extendRays"], st["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])
    assert st["tailLaunches"] >= 1 and st["extendLaunches"] == st["tailLaunches"], "every pass of this frame is a tail launch"


def test_grouped_nee_samples_stay_on_the_wavefront_path():
    """NEEFullSamples 3: the samples of a vertex are folded by k_resolve_nee; pt_render ignores the tail threshold there."""
    import pin_scenes
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden.npz"))
    got, st = _render(pin_scenes.cases()["c2_nee3"], WHOLE)
    assert np.array_equal(_bits(got), _bits(g["c2_nee3"])) and st["tailLaunches"] == 0


@pytest.mark.parametrize("name", ["c2", "bistro_like", "bistro_like_c5", "c2_nested2_norr_nold"])
def test_hand_back_of_stragglers(name, monkeypatch):
    """Eight iterations after a wave's first ray has finished, everything still in flight goes back to the host loop: extend rays as paths in the next pass's queue,
    visibility rays as shadow-queue entries whose paths wait for them."""
    import pin_scenes
    case = pin_scenes.cases_lp16()[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden_lp16.npz"))
    monkeypatch.setenv("MI355PT_TAIL_DEFER", "8")
    got, st = _render(case, WHOLE)
    bad = (_bits(got) != _bits(g[name])).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ" % (name, int(bad.sum()), bad.size)
    assert (st["extendRays"], st["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])
    monkeypatch.delenv("MI355PT_TAIL_DEFER")
    _, st_all = _render(case, WHOLE)
    assert st["tailLaunches"] > st_all["tailLaunches"], "with the short fuse the frame needs more launches: paths did come back"


@pytest.mark.parametrize("name", ["bistro_like_neeat", "bistro_like_neeat_lp16", "c2_neeat_table_only_nee3"])
def test_neeat_frames_through_the_tail_kernel(name):
    """NEE-AT: the local sampler, the feedback reservoirs and the roulette fix-up of the visible case, applied by the tail kernel on the path's own registers."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    import pin_scenes
    CASES = pin_scenes.neeat_cases()
    if name not in CASES: pytest.skip("no such NEE-AT case")
    g = np.load(os.path.join(ROOT, "tests", "golden", "neeat_golden.npz"))
    make, S, w, h, first, n, opts = CASES[name]
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
    baked = len(t.lights()["lights"]); assert baked == int(g[name + "_lights"][0])
    t.set_local_light_sampling(pin_scenes.neeat_table(opts, baked, w, h), jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
    t.set_tail_paths(WHOLE)
    st = t.render(first, n)
    grouped = int(S["NEEFullSamples"]) > 1
    assert (st["tailLaunches"] == 0) if grouped else (st["tailLaunches"] >= 1)
    bad = (_bits(t.radiance()) != _bits(g[name])).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert (int(st["extendRays"]), int(st["shadowRays"])) == tuple(int(v) for v in g[name + "_rays"])
    if opts["feedback"]:
        for s in range(n):
            fw, fc = t.light_feedback(s)
            assert np.array_equal(_bits(fw), _bits(g["%s_fbw%d" % (name, s)])) and np.array_equal(fc, g["%s_fbc%d" % (name, s)]), "feedback planes of sample %d" % s
    t.close()


def test_mixed_frames_do_not_depend_on_the_threshold():
    """A frame of 0.9 M paths in pipelined batches: all-wavefront == the product default (32768) == 65536 == an early hand-over (300 000) == a late one (2 000), ray counts included."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.bistro_like(scale=0.05, tex_size=128)
    w, h, spp = 640, 360, 4
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(scenes.default_settings(useFp16Types=1)); t.resize(w, h)
    frames = {}
    for tail in (0, 32768, 65536, 300000, 2000):
        t.set_tail_paths(tail); t.reset_accumulation(); st = t.render(0, spp)
        frames[tail] = (t.radiance(), st["extendRays"], st["shadowRays"], st["hits"], st["tailLaunches"])
    t.close()
    ref = frames[0]; assert ref[4] == 0
    for tail in (32768, 65536, 300000, 2000):
        f = frames[tail]
        assert f[4] >= 1, "threshold %d: no tail launch" % tail
        assert np.array_equal(_bits(f[0]), _bits(ref[0])), "threshold %d: %d pixels differ" % (tail, int((_bits(f[0]) != _bits(ref[0])).any(-1).sum()))
        assert f[1:4] == ref[1:4], "threshold %d: ray / hit counts %s vs %s" % (tail, f[1:4], ref[1:4])
