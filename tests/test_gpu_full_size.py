"""GPU parity at BASELINE.json's FULL sizes (run with -m gpu): the configuration bench.py times — C3, 2.8 M triangles, 3840x2160, 4 spp, 8 bounces —
and C5 (the animated variant: per-frame refit + nested dielectrics) are compared with the oracle on complete pixel rows and on a pixel block, bit for
bit, plus the size-independent properties (accumulation associativity, finite non-negative radiance, ray budget). These are the sizes at which 32-bit
node / triangle offsets, the 2 GB sparse table of the BVH builder, the 4-batch pipeline and the multi-round straggler splitting are exercised; the
small-scale tests in test_gpu_parity.py never reach them."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, SPP = 3840, 2160, 4


def _imports():
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    from oracle import ptref
    return pt, scenes, ptref


def _oracle_rects(o, rects, first, n):
    out = []
    for r in rects:
        o.reset_accumulation(); o.render(first, n, rect=r)
        out.append(o.radiance()[r[1]:r[3], r[0]:r[2], :3].copy())
    return out


def test_c3_full_size_4k_rows_and_block_bit_exact():
    """BASELINE configs[2] = bench.py's workload at full size: 9 complete rows of the 4K frame and a 256x128 block around the frame centre are
    bit-identical to the oracle; ray counts of a sub-frame equal the oracle's."""
    pt, scenes, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
    S = scenes.default_settings()
    camd = scenes.bridge_camera(W, H, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
    st = g.render(0, SPP); a = g.radiance()
    assert np.isfinite(a).all() and (a >= 0).all() and np.all(a[..., 3] == 1.0)
    assert st["pathsTraced"] == W * H * SPP
    assert st["extendRays"] <= 10 * st["pathsTraced"] and st["shadowRays"] <= st["extendRays"]
    # accumulation is associative over pt_render calls (bit-exact) at this size too
    g.reset_accumulation(); g.render(0, 1); g.render(1, 3)
    assert np.array_equal(a, g.radiance())
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(W, H)
    rows = list(range(7, H, 256))
    rects = [(0, y, W, y + 1) for y in rows] + [(W // 2 - 128, H // 2 - 64, W // 2 + 128, H // 2 + 64)]
    want = _oracle_rects(o, rects, 0, SPP)
    for r, wnt in zip(rects, want):
        got = a[r[1]:r[3], r[0]:r[2], :3]
        bad = int((got.view(np.uint32) != wnt.view(np.uint32)).any(-1).sum())
        assert bad == 0, "rect %s: %d pixels differ" % (r, bad)
    # ray counts: the same frame at 1/6 resolution per axis is small enough for a whole-frame oracle render
    w2, h2 = W // 6, H // 6
    camd2 = scenes.bridge_camera(w2, h2, **cam)
    g.set_camera(camd2); g.resize(w2, h2); st2 = g.render(0, SPP)
    o.set_camera(camd2); o.resize(w2, h2); o.reset_accumulation(); o.render(0, SPP)      # (reset also clears the oracle's ray counters)
    c = o.counters()
    assert np.array_equal(g.radiance(), o.radiance())
    assert (st2["extendRays"], st2["shadowRays"], st2["hits"]) == (c["extendRays"], c["shadowRays"], c["hits"])


def test_c5_full_size_4k_animated_refit_rows_bit_exact():
    """BASELINE configs[4] on one GPU at full size: two animated frames (rigid clutter groups + deforming banner; refit only, no rebuild) with
    nestedDielectricsQuality 2; per frame 5 complete rows equal an oracle rebuilt from scratch for that frame."""
    pt, scenes, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=True)
    S = scenes.default_settings(nestedDielectricsQuality=2, useFp16Types=1)      # the reference's default build of the lp types (C3 above runs the fp32 build)
    camd = scenes.bridge_camera(W, H, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
    for frame, t in enumerate((0.0, 0.6)):
        inst, pos = scenes.animate_instances(sc, t), scenes.animate_positions(sc, t)
        g.animate(instances=inst, positions=pos, rebuild=False)
        g.reset_accumulation(); st = g.render(frame * SPP, SPP); a = g.radiance()
        assert np.isfinite(a).all() and (a >= 0).all()
        sc_t = dict(sc); sc_t["positions"] = pos
        o = ptref.Oracle(lp16=True); o.set_scene(sc_t); o.set_instances(inst); o.set_camera(camd); o.set_settings(S); o.resize(W, H)
        rects = [(0, y, W, y + 1) for y in range(11 + 37 * frame, H, 512)]
        for r, wnt in zip(rects, _oracle_rects(o, rects, frame * SPP, SPP)):
            got = a[r[1]:r[3], r[0]:r[2], :3]
            bad = int((got.view(np.uint32) != wnt.view(np.uint32)).any(-1).sum())
            assert bad == 0, "frame %d rect %s: %d pixels differ" % (frame, r, bad)
        o.close()
    assert g.build_stats()["refitMs"] > 0
    # a rebuild between animated frames takes the fast builder (PLOC on the device), whatever made the first tree; the frame does not depend on it
    inst, pos = scenes.animate_instances(sc, 0.6), scenes.animate_positions(sc, 0.6)
    g.animate(instances=inst, positions=pos, rebuild=True)
    assert g.build_stats()["buildMs"] < 100.0
    g.reset_accumulation(); g.render(SPP, SPP)
    assert np.array_equal(a, g.radiance())


@pytest.mark.tail_once("tail_default")      # (the stable-plane passes and the display path have their own loops: the tail kernel is pt_render's)
def test_realtime_frames_4k_animated_match_oracle():
    """The realtime mode's coupled frame (pt_realtime_frame: baker UpdateBegin, build pass, UpdateEnd on the frame's depth + motion vectors, fill pass feeding the reservoirs) at
    3840x2160 on C5's scene, two frames with the camera and the scene moving in between (pt_set_motion_history, pt_animate_ranges: refit, light re-bake, object motion in the motion
    vectors): header, depth, motion vectors, specular hit distances, stable radiance, throughput, every live plane record, NEE-AT's tile tables and the reservoirs the fill pass
    left — all of both frames bit for bit against the oracle. At this size the passes run as four pipelined batches and the fill pass's first launch uses the narrowed ray interval."""
    pt, scenes, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=True)
    S = scenes.default_settings(NEEType=2, nestedDielectricsQuality=2, useFp16Types=1)      # the reference's default build
    step, dt, frames = (0.35, 0.02, -0.2), 0.45, 2
    def camera(f): c = dict(cam); c["pos"] = tuple(np.asarray(cam["pos"], np.float64) + np.asarray(step) * f); return c
    poses = [(scenes.animate_instances(sc, dt * f), scenes.animate_positions(sc, dt * f)) for f in range(frames)]
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(S); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.resize(W, H); g.set_neeat(True); g.set_motion_history(True)
    o = ptref.Oracle(lp16=bool(int(S["useFp16Types"]))); o.set_scene(sc); o.set_settings(S); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.resize(W, H); o.set_neeat(True)
    same = lambda a, b: np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    for f in range(frames):
        cur, prev = camera(f), camera(max(f - 1, 0))
        prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cur), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=1)
        camd = scenes.bridge_camera(W, H, **cur)
        g.animate(poses[f][0], poses[f][1], vertex_ranges=scenes.animated_vertex_ranges(sc) if f else None); g.set_camera(camd)
        got, bst, fst = g.realtime_frame(f, prm)
        posed = dict(sc); posed["instances"], posed["positions"] = poses[f]
        o.set_scene(posed); o.set_previous_pose(*(poses[f - 1] if f else (sc["instances"], sc["positions"]))); o.set_camera(camd)
        o.neeat_update_begin(); want = o.build_stable_planes(f, prm); o.neeat_update_end(want["depth"], want["motion_vectors"]); o.fill_stable_planes(f, prm, want)
        for k in ("header", "depth", "motion_vectors", "spec_hit_t", "stable_radiance", "throughput"):
            assert same(got[k], want[k]), "frame %d: %s differs" % (f, k)
        hd = want["header"]; P, Q = got["planes"].reshape(-1, 20), want["planes"].reshape(-1, 20)
        for pl in range(3):      # the records of the planes that exist (sampled: every 7th live pixel of each plane)
            ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF); ys, xs = ys[::7], xs[::7]
            idx = np.array([scenes.stable_planes_address(int(x), int(y), pl, W, H) for x, y in zip(xs, ys)], np.int64)
            assert same(P[idx], Q[idx]), "frame %d: plane %d records differ" % (f, pl)
        tab, jit = g.neeat_tables(); wtab, wjit, _ = o.neeat_tables()
        assert same(tab, wtab) and tuple(int(v) for v in jit) == tuple(int(v) for v in wjit), "frame %d: tile tables" % f
        fw, fc = g.light_feedback(0); wfw, wfc = o.neeat_feedback()
        assert same(fw, wfw) and same(fc, wfc), "frame %d: reservoirs" % f
        mv = np.asarray(want["motion_vectors"]).reshape(H, W, -1)
        assert (mv[..., :2] != 0).any()
    assert int(bst["extendRays"]) > W * H and int(fst["shadowRays"]) > 0
    g.close(); o.close()


def test_bench_frame_equals_the_reference_text_frame():
    """bench.py's default workload at full size against the frame the REFERENCE'S integrator text rendered of it (tests/golden/bench_frame_golden.npz): the SHA-256 of the whole
    3840x2160 RGBA32F frame, the kept rows pixel for pixel, the ray counts — no oracle in the loop."""
    import hashlib, sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_bench_frame_golden as gen
    gold = np.load(os.path.join(root, "tests", "golden", "bench_frame_golden.npz"))
    sc, cam, S = gen.bench_workload()
    g = pt.PathTracer(); g.set_tail_paths(32768); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(S); g.resize(W, H)
    st = g.render(0, SPP); frame = g.radiance()
    rows = frame[::int(gold["row_step"][0])]
    bad = int((rows.view(np.uint32) != gold["rows"].view(np.uint32)).any(-1).sum())
    assert bad == 0, "%d pixels of the kept rows differ" % bad
    assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(frame, np.float32).tobytes()).digest(), np.uint8), gold["sha256"]), "the frame's digest differs"
    assert (int(st["extendRays"]), int(st["shadowRays"])) == tuple(int(v) for v in gold["rays"])
    g.close()


@pytest.mark.parametrize("name", ["C1", "C2", "C4", "C5_t1", "C5_t2"])
def test_config_frames_equal_the_reference_text_frames(name):
    """BASELINE.json's other configurations at their full sizes (tools/run_configs.py's definitions) against the frames the REFERENCE'S integrator text rendered of them
    (tests/golden/config_frames_golden.npz): SHA-256 of the whole frame, four rows pixel for pixel, ray counts. C5 reaches its pose the way the product does: pt_animate on the
    scene as uploaded (refit, light re-bake), the fixture's frame came from a scene built in that pose. No oracle in the loop."""
    import hashlib, sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_config_frames_golden as gen
    gold = np.load(os.path.join(root, "tests", "golden", "config_frames_golden.npz"))
    make, S, w, h, first, n, t = gen.configs()[name]
    sc, cam = make()
    g = pt.PathTracer(); g.set_tail_paths(32768); g.set_scene(sc); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(S); g.resize(w, h)
    if t is not None:
        g.render(first, 1)                                             # (a frame on the rest pose first: the tree the refit starts from is the rest pose's)
        g.animate(scenes.animate_instances(sc, t), scenes.animate_positions(sc, t), vertex_ranges=scenes.animated_vertex_ranges(sc)); g.reset_accumulation()
    st = g.render(first, n); frame = g.radiance()
    rows = frame[gen.rows_of(h)]
    bad = int((rows.view(np.uint32) != gold[name + "_rows"].view(np.uint32)).any(-1).sum())
    assert bad == 0, "%s: %d pixels of the kept rows differ" % (name, bad)
    assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(frame, np.float32).tobytes()).digest(), np.uint8), gold[name + "_sha256"]), "%s: the frame's digest differs" % name
    assert (int(st["extendRays"]), int(st["shadowRays"])) == tuple(int(v) for v in gold[name + "_rays"])
    g.close()


@pytest.mark.tail_once("tail_default")      # (the stable-plane passes and the display path have their own loops: the tail kernel is pt_render's)
def test_realtime_passes_4k_equal_the_reference_text():
    """The stable-plane build pass and one fill sub-sample at 3840x2160 on C5's scene in an animated pose (previous pose = the rest pose: object motion; the camera moved as well)
    against what the REFERENCE'S text of those passes produced (tests/golden/realtime_4k_golden.npz: SHA-256 digests of header, depth, motion vectors, stable radiance, throughput,
    hit distances and of all live plane records after each pass, ray counts). The device reaches the pose through pt_set_motion_history + pt_animate_ranges. No oracle in the loop."""
    import sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_realtime_4k_golden as gen
    gold = np.load(os.path.join(root, "tests", "golden", "realtime_4k_golden.npz"))
    sc, cam, S, prm, pose = gen.workload()
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(S); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.resize(W, H); g.set_motion_history(True)
    g.animate(pose[0], pose[1], vertex_ranges=scenes.animated_vertex_ranges(sc))
    built = g.build_stable_planes(gen.SAMPLE, prm)
    for k, v in gen.digests(built).items(): assert np.array_equal(v, gold["build_" + k]), "build pass: %s differs" % k
    filled = g.fill_stable_planes(gen.SAMPLE, prm)
    for k, v in gen.digests(filled).items(): assert np.array_equal(v, gold["fill_" + k]), "fill pass: %s differs" % k
    assert int(built["stats"]["extendRays"]) == int(gold["build_rays"][0])
    assert (int(filled["stats"]["extendRays"]), int(filled["stats"]["shadowRays"])) == tuple(int(v) for v in gold["fill_rays"])
    g.close()


def test_neeat_4k_equals_the_reference_text():
    """NEE-AT's path-tracer side at 3840x2160 (NEEType 2, the reference's default sampler; tile tables handed in as a host's baker would) against the REFERENCE'S integrator text
    (tests/golden/neeat_4k_golden.npz): SHA-256 of the frame and of both samples' reservoir planes, ray counts. No oracle in the loop."""
    import sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_neeat_4k_golden as gen
    gold = np.load(os.path.join(root, "tests", "golden", "neeat_4k_golden.npz"))
    sc, cam, S = gen.workload()
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(S); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.resize(W, H)
    n_lights = len(g.lights()["lights"]); assert n_lights == int(gold["lights"][0])
    g.set_local_light_sampling(gen.table(n_lights), jitter=gen.OPTS["jitter"], ratio=gen.OPTS["ratio"], ssc_threshold=gen.OPTS["ssc_threshold"], feedback=gen.OPTS["feedback"])
    st = g.render(gen.FIRST, gen.N)
    assert np.array_equal(gen.digest(g.radiance()), gold["frame"]), "the frame's digest differs"
    for s in range(gen.N):
        wgt, cand = g.light_feedback(s)
        assert np.array_equal(gen.digest(wgt), gold["fbw%d" % s]) and np.array_equal(gen.digest(cand), gold["fbc%d" % s]), "reservoirs of sample %d differ" % s
    assert (int(st["extendRays"]), int(st["shadowRays"])) == tuple(int(v) for v in gold["rays"])
    g.close()


@pytest.mark.tail_once("tail_default")      # (the stable-plane passes and the display path have their own loops: the tail kernel is pt_render's)
def test_coupled_realtime_frames_4k_equal_the_reference_text():
    """Two coupled realtime frames at 3840x2160 — baker UpdateBegin, build pass, UpdateEnd on the frame's depth and motion vectors, fill pass feeding the reservoirs — against the
    REFERENCE'S text with LightsBaker.hlsl run thread by thread (tests/golden/realtime_coupled_4k_golden.npz): per frame the digests of the tile tables, proxy counters, reservoirs
    and of every plane buffer, the tile jitter, and the ray counts of the run. C5's scene; camera and scene move between the frames (pt_set_motion_history, pt_animate_ranges). No
    oracle in the loop."""
    import sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_realtime_coupled_4k_golden as gen
    import make_realtime_4k_golden as rt
    gold = np.load(os.path.join(root, "tests", "golden", "realtime_coupled_4k_golden.npz"))
    sc, cam, S, poses = gen.workload()
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(S); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.resize(W, H); g.set_neeat(True); g.set_motion_history(True)
    rays = [0, 0]
    for f in range(gen.FRAMES):
        prm, camd = gen.params(cam, f)
        g.animate(poses[f][0], poses[f][1], vertex_ranges=scenes.animated_vertex_ranges(sc) if f else None); g.set_camera(camd)
        frame, bst, fst = g.realtime_frame(f, prm)
        rays[0] += int(bst["extendRays"]) + int(fst["extendRays"]); rays[1] += int(fst["shadowRays"])
        tab, jit = g.neeat_tables(); fw, fc = g.light_feedback(0)
        assert np.array_equal(rt.digest(tab), gold["table%d" % f]), "frame %d: tile tables" % f
        assert tuple(int(v) for v in jit) == tuple(int(v) for v in gold["jitter%d" % f]), "frame %d: jitter" % f
        assert np.array_equal(rt.digest(g.lights()["proxyCounters"]), gold["counters%d" % f]), "frame %d: proxy counters" % f
        assert np.array_equal(rt.digest(fw), gold["fbw%d" % f]) and np.array_equal(rt.digest(fc), gold["fbc%d" % f]), "frame %d: reservoirs" % f
        for k, v in rt.digests(frame).items(): assert np.array_equal(v, gold["%s%d" % (k, f)]), "frame %d: %s differs" % (f, k)
    assert rays == [int(v) for v in gold["rays"]]
    g.close()


def test_neeat_loop_4k_equals_the_reference_text():
    """Reference mode with the reference's default sampler at 3840x2160: three accumulated frames with NEE-AT's baker in the loop (pt_set_neeat, pt_set_view_projection,
    pt_set_light_importance_boost; one pt_render per frame) against the REFERENCE'S text with LightsBaker.hlsl run thread by thread (tests/golden/neeat_loop_4k_golden.npz): per frame
    the digests of tile tables, proxy counters, reservoirs and the tile jitter; the accumulated frame's digest; the ray counts. No oracle in the loop."""
    import sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_neeat_loop_4k_golden as gen
    import make_realtime_4k_golden as rt
    gold = np.load(os.path.join(root, "tests", "golden", "neeat_loop_4k_golden.npz"))
    sc, cam, S = gen.workload()
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(S); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.resize(W, H); gen.setup(g, cam)
    rays = [0, 0]
    for f in range(gen.FRAMES):
        st = g.render(f, 1); rays[0] += int(st["extendRays"]); rays[1] += int(st["shadowRays"])
        tab, jit = g.neeat_tables(); fw, fc = g.light_feedback(0)
        assert np.array_equal(rt.digest(tab), gold["table%d" % f]) and tuple(int(v) for v in jit) == tuple(int(v) for v in gold["jitter%d" % f]), "frame %d: tile tables / jitter" % f
        assert np.array_equal(rt.digest(g.lights()["proxyCounters"]), gold["counters%d" % f]), "frame %d: proxy counters" % f
        assert np.array_equal(rt.digest(fw), gold["fbw%d" % f]) and np.array_equal(rt.digest(fc), gold["fbc%d" % f]), "frame %d: reservoirs" % f
    assert np.array_equal(rt.digest(g.radiance()), gold["frame"]), "the accumulated frame's digest differs"
    assert rays == [int(v) for v in gold["rays"]]
    g.close()


@pytest.mark.tail_once("tail_default")      # (the stable-plane passes and the display path have their own loops: the tail kernel is pt_render's)
def test_display_path_4k_equals_the_reference_text():
    """pt_tonemap of the device's bench frame for the six operators x three exposure compensations and auto exposure against tests/golden/display_4k_golden.npz: the SRGBA8 image of
    the frame the reference's integrator text rendered, through the reference's ToneMapping.ps.hlsli text (== its restatement on all 8.3 M pixels, asserted when the fixture was made)"""
    import sys
    pt, scenes, ptref = _imports()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_bench_frame_golden as gen
    import make_display_4k_golden as disp
    gold = np.load(os.path.join(root, "tests", "golden", "display_4k_golden.npz"))
    sc, cam, S = gen.bench_workload()
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(S); g.resize(W, H); g.render(0, SPP)
    for name, kw in disp.parameter_sets().items():
        assert np.array_equal(disp.digest(g.tonemap(pt.default_tonemap(**kw))), gold[name]), "%s: the LDR image differs" % name
    g.close()
