"""The ORACLE against a sample of the at-scale reference-text fixtures (the device is compared with all of them on the GPU box, tests/test_gpu_parity_hd.py and
tests/test_gpu_full_size.py): two integrator cases and one NEE-AT case at 1920x1080 x 8 samples, a stable-plane case with object motion at 1920x1080 — whole frames by SHA-256, ray counts."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from rtxpt_amd import scenes
from oracle import ptref
import make_pin_cases_hd_golden as gen
import make_stable_planes_hd_golden as sph


@pytest.mark.parametrize("key", ["lp16_bistro_like_material_zoo_firefly", "fp32_c2_sphere_light_proxy", "neeat_bistro_like_neeat_lp16"])
def test_oracle_frame_equals_the_reference_text_frame(key):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "pin_cases_hd_golden.npz"))
    make, S, first, opts = gen.case_setup(key)
    sc, cam = make()
    o = ptref.Oracle(lp16=bool(int(S["useFp16Types"]))); o.set_scene(sc); o.set_camera(scenes.bridge_camera(gen.W, gen.H, **cam)); o.set_settings(S); o.resize(gen.W, gen.H); o.L.ptref_prepare(o.h)
    nl = len(o.lights()["lights"]); assert nl == int(gold[key + "_lights"][0])
    if opts is not None:
        tab = None if opts["table_seed"] is None else scenes.synthetic_local_light_tables(nl, gen.W, gen.H, seed=opts["table_seed"], jitter=opts["jitter"])
        o.set_local_light_sampling(tab, jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
    o.render(first, gen.N); c = o.counters()
    assert np.array_equal(gen.digest(o.radiance()), gold[key]), "%s: the oracle's frame differs from the reference text's" % key
    assert (c["extendRays"], c["shadowRays"]) == tuple(int(v) for v in gold[key + "_rays"])
    if opts is not None and opts["feedback"]:
        wgt, cand = o.light_feedback(gen.N - 1)
        assert np.array_equal(gen.digest(np.concatenate([wgt.view(np.uint32).ravel(), cand.ravel()])), gold["%s_fb%d" % (key, gen.N - 1)])
    o.close()


@pytest.mark.parametrize("key", ["motion_zoo_object_motion_lp16"])
def test_oracle_stable_planes_equal_the_reference_text(key):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "stable_planes_hd_golden.npz"))
    sc, cam, S, prm, lp16, prev_pose = sph.setup(key)
    o = ptref.Oracle(lp16=lp16); o.set_scene(sc); o.set_camera(scenes.bridge_camera(sph.W, sph.H, **cam)); o.set_settings(S); o.resize(sph.W, sph.H)
    if prev_pose is not None: o.set_previous_pose(*prev_pose)
    frame = o.build_stable_planes(sph.SAMPLE, prm)
    for k, v in sph.digests(frame).items(): assert np.array_equal(v, gold["%s_build_%s" % (key, k)]), "%s, build pass: %s differs" % (key, k)
    for s in range(sph.SUBS): o.fill_stable_planes(sph.SAMPLE + s, prm, frame)
    for k, v in sph.digests(frame).items(): assert np.array_equal(v, gold["%s_fill_%s" % (key, k)]), "%s, fill passes: %s differs" % (key, k)
    o.close()
