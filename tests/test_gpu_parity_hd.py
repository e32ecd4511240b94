"""Every pin case at 1920x1080 x 8 samples against the REFERENCE'S integrator text (run with -m gpu): tests/golden/pin_cases_hd_golden.npz holds, per case of tests/pin_scenes.py — the
integrator cases in both lp builds and the NEE-AT cases with their synthetic tile tables —, the SHA-256 of the frame the reference's text rendered (16.6 M paths instead of the ~10 000
of the small fixtures), of each sample's reservoir planes, and the ray counts (tests/golden/make_pin_cases_hd_golden.py). The device's frames are digested the same way; no oracle in
the loop. Rounds 1-3 compared with the reference's text on frames of at most 96x54: one-in-a-million branches (seams, ties, clamps) were never met."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_pin_cases_hd_golden as gen

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "pin_cases_hd_golden.npz")


@pytest.mark.parametrize("key", list(gen.all_cases()))
def test_device_frame_equals_the_reference_text_frame(key):
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    gold = np.load(GOLD)
    if key not in gold.files: pytest.skip("not in the fixture")
    make, S, first, opts = gen.case_setup(key)
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(gen.W, gen.H, **cam)); t.set_settings(S); t.resize(gen.W, gen.H)
    nl = len(t.lights()["lights"]); assert nl == int(gold[key + "_lights"][0])
    if opts is not None:
        tab = None if opts["table_seed"] is None else scenes.synthetic_local_light_tables(nl, gen.W, gen.H, seed=opts["table_seed"], jitter=opts["jitter"])
        t.set_local_light_sampling(tab, jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
    st = t.render(first, gen.N)
    assert np.array_equal(gen.digest(t.radiance()), gold[key]), "%s: the frame's digest differs from the reference text's" % key
    assert (int(st["extendRays"]), int(st["shadowRays"])) == tuple(int(v) for v in gold[key + "_rays"])
    if opts is not None and opts["feedback"]:
        for s in range(gen.N):
            wgt, cand = t.light_feedback(s)
            assert np.array_equal(gen.digest(np.concatenate([wgt.view(np.uint32).ravel(), cand.ravel()])), gold["%s_fb%d" % (key, s)]), "%s: reservoirs of sample %d differ" % (key, s)
    t.close()


import make_stable_planes_hd_golden as sph
SP_GOLD = os.path.join(ROOT, "tests", "golden", "stable_planes_hd_golden.npz")


@pytest.mark.tail_once("tail_default")      # (no pt_render in this test: the tail kernel plays no part)
@pytest.mark.parametrize("key", sph.all_cases())
def test_device_stable_planes_equal_the_reference_text(key):
    """the stable-plane cases (incl. object motion and the edge cases) at 1920x1080: the build pass and two fill sub-samples against the REFERENCE'S text of those passes
    (tests/golden/stable_planes_hd_golden.npz: digests of every plane buffer and of all live plane records after each, ray counts)"""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    gold = np.load(SP_GOLD)
    sc, cam, S, prm, lp16, prev_pose = sph.setup(key)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(sph.W, sph.H, **cam)); t.resize(sph.W, sph.H)
    if prev_pose is not None: t.set_previous_pose(*prev_pose)
    built = t.build_stable_planes(sph.SAMPLE, prm)
    for k, v in sph.digests(built).items(): assert np.array_equal(v, gold["%s_build_%s" % (key, k)]), "%s, build pass: %s differs" % (key, k)
    assert int(built["stats"]["extendRays"]) == int(gold[key + "_build_rays"][0])
    filled = t.fill_stable_planes(sph.SAMPLE, prm, sub_samples=sph.SUBS)
    for k, v in sph.digests(filled).items(): assert np.array_equal(v, gold["%s_fill_%s" % (key, k)]), "%s, fill passes: %s differs" % (key, k)
    assert (int(filled["stats"]["extendRays"]), int(filled["stats"]["shadowRays"])) == tuple(int(v) for v in gold[key + "_fill_rays"])
    t.close()


import make_env_cube_2048_golden as cubes
CUBE_GOLD = os.path.join(ROOT, "tests", "golden", "env_cube_2048_golden.npz")


@pytest.mark.tail_once("tail_default")      # (no pt_render in this test: the tail kernel plays no part)
@pytest.mark.parametrize("name", list(cubes.cases()))
def test_device_env_cube_2048_equals_the_reference_text(name):
    """the 2048^2 environment cube with its mips (33.5 M texels) as the device bakes it against the bake of the REFERENCE'S EnvMapBaker text (SHA-256 of the whole cube): the bench's
    sky with "Fast" BC6U compression, a noisy HDR source uncompressed, sun discs with "Fast" and "Quality" compression"""
    import rtxpt_amd as pt
    gold = np.load(CUBE_GOLD)
    sc = cubes.cases()[name]()
    t = pt.PathTracer(); t.set_scene(sc)
    cube, dim, lv = t.env_cube()
    assert (dim, lv, cube.shape[0]) == tuple(int(v) for v in gold[name + "_dim"])
    assert np.array_equal(cubes.digest(cube), gold[name]), "%s: the cube's digest differs from the reference text's bake" % name
    t.close()


import make_neeat_loop_hd_golden as nlh
import pin_scenes as _pins
NL_GOLD = os.path.join(ROOT, "tests", "golden", "neeat_loop_hd_golden.npz")


@pytest.mark.parametrize("name", list(_pins.neeat_loop_cases()))
def test_device_neeat_runs_equal_the_reference_text(name):
    """NEE-AT with the light baker in the loop at 1920x1080 (pt_set_neeat; one pt_render per frame) for the option sets of pin_scenes.neeat_loop_cases(), against the REFERENCE'S text
    with LightsBaker.hlsl run thread by thread (tests/golden/neeat_loop_hd_golden.npz): every frame's tile tables, jitter, proxy counters and reservoirs, the accumulated frame, ray counts"""
    import rtxpt_amd as pt
    gold = np.load(NL_GOLD)
    def make_device(sc, camd, S, w, h):
        t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(w, h)
        return t, (lambda t: t.lights()["proxyCounters"])
    got = nlh.run(name, make_device)
    keys = list(got)
    assert keys and all(k in gold.files for k in keys)
    bad = [k for k in keys if not np.array_equal(np.asarray(got[k]), gold[k])]
    assert not bad, bad[:6]


import make_realtime_hd_golden as rth
import realtime_cases as _rc
RT_GOLD = os.path.join(ROOT, "tests", "golden", "realtime_hd_golden.npz")


@pytest.mark.tail_once("tail_default")      # (no pt_render in this test: the tail kernel plays no part)
@pytest.mark.parametrize("name", list(_rc.cases()))
def test_device_realtime_runs_equal_the_reference_text(name):
    """the coupled realtime runs of tests/realtime_cases.py at 1920x1080 (pt_realtime_frame per frame; the animated case through pt_set_motion_history + pt_animate) against the
    REFERENCE'S text with the baker run thread by thread (tests/golden/realtime_hd_golden.npz): digests of every output of every frame, ray counts"""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    gold = np.load(RT_GOLD)
    sc, cam, S, subs, frames = rth.frames_of(name)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(rth.W, rth.H, **cam)); t.resize(rth.W, rth.H); t.set_neeat(True)
    if frames[0][2] is not None: t.set_motion_history(True)
    rays = [0, 0]
    for f, (camd, prm, pose, prev) in enumerate(frames):
        if pose is not None: t.animate(pose[0], pose[1], vertex_ranges=scenes.animated_vertex_ranges(sc) if f % 2 else None)
        t.set_camera(camd)
        frame, bst, fst = t.realtime_frame(f * subs, prm)
        rays[0] += int(bst["extendRays"]) + int(fst["extendRays"]); rays[1] += int(fst["shadowRays"])
        tab, jit = t.neeat_tables(); fw, fc = t.light_feedback(0)
        got = rth.record(name, f, frame, tab, jit, t.lights()["proxyCounters"], fw, fc)
        bad = [k for k, v in got.items() if not np.array_equal(np.asarray(v), gold[k])]
        assert not bad, "frame %d: %s" % (f, bad)
    assert rays == [int(v) for v in gold[name + "_rays"]]
    t.close()


import fuzz_cases as _fz
FUZZ_GOLD = os.path.join(ROOT, "tests", "golden", "fuzz_hd_golden.npz")


@pytest.mark.parametrize("lp16", [False, True], ids=["fp32", "lp16"])
@pytest.mark.parametrize("seed", _fz.SEEDS)
def test_device_fuzz_frame_equals_the_reference_text_frame(seed, lp16):
    """40 seeded random scene / camera / settings cases at 1280x720 in both lp builds against frames of the REFERENCE'S integrator text (tests/golden/fuzz_hd_golden.npz); animated
    cases reach their pose through pt_animate (refit for even seeds, rebuild for odd ones) while the fixture's scene was built in that pose"""
    import rtxpt_amd as pt
    import make_pin_cases_hd_golden as gen
    gold = np.load(FUZZ_GOLD); key = "%d_%s" % (seed, "lp16" if lp16 else "fp32")
    if key not in gold.files: pytest.skip("not in the fixture")
    sc, camd, S, first, count, pose = _fz.case(seed, lp16)
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(camd); t.set_settings(S); t.resize(_fz.W, _fz.H)
    if pose is not None: t.animate(pose[0], pose[1], rebuild=bool(seed & 1))
    st = t.render(first, count)
    assert np.array_equal(gen.digest(t.radiance()), gold[key]), "seed %d: the frame's digest differs from the reference text's" % seed
    assert (int(st["extendRays"]), int(st["shadowRays"])) == tuple(int(v) for v in gold[key + "_rays"])
    t.close()


import make_fuzz_sp_hd_golden as fsp
FUZZ_SP_GOLD = os.path.join(ROOT, "tests", "golden", "fuzz_sp_hd_golden.npz")


@pytest.mark.tail_once("tail_default")      # (no pt_render in this test: the tail kernel plays no part)
@pytest.mark.parametrize("seed", _fz.SP_SEEDS)
def test_device_fuzz_stable_planes_equal_the_reference_text(seed):
    """30 seeded random stable-plane frames at 1280x720 (random viewpoints, plane counts, vertex depths, settings, previous poses, sub-sample counts) against the REFERENCE'S text of
    both passes (tests/golden/fuzz_sp_hd_golden.npz)"""
    import rtxpt_amd as pt
    gold = np.load(FUZZ_SP_GOLD)
    if "%d_build_rays" % seed not in gold.files: pytest.skip("not in the fixture")
    sc, camd, S, prm, lp16, prev_pose, sample, subs = _fz.stable_planes_case(seed)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(_fz.W, _fz.H)
    if prev_pose is not None: t.set_previous_pose(*prev_pose)
    built = t.build_stable_planes(sample, prm)
    for k, v in fsp.digests(built).items(): assert np.array_equal(v, gold["%d_build_%s" % (seed, k)]), "seed %d, build pass: %s differs" % (seed, k)
    assert int(built["stats"]["extendRays"]) == int(gold["%d_build_rays" % seed][0])
    filled = t.fill_stable_planes(sample, prm, sub_samples=subs)
    for k, v in fsp.digests(filled).items(): assert np.array_equal(v, gold["%d_fill_%s" % (seed, k)]), "seed %d, fill passes: %s differs" % (seed, k)
    assert (int(filled["stats"]["extendRays"]), int(filled["stats"]["shadowRays"])) == tuple(int(v) for v in gold["%d_fill_rays" % seed])
    t.close()
