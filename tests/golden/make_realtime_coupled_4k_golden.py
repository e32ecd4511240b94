"""Generates tests/golden/realtime_coupled_4k_golden.npz: TWO COUPLED REALTIME FRAMES at full size through the REFERENCE'S text — per frame LightsBaker::UpdateBegin, the stable-plane
build pass, LightsBaker::UpdateEnd on that frame's depth and motion vectors, one fill sub-sample feeding the reservoirs (Rtxpt/Sample.cpp:2438-2516) — with LightsBaker.hlsl's passes
run thread by thread (groups as coroutines, group-shared memory and barriers as written) and PathTracer.hlsli & co. in both PATH_TRACER_MODEs: C5's scene (2.86 M triangles, 7 400
lights), 3840x2160, the reference's default lp16 build, NEEType 2, nested dielectrics quality 2; between the frames the camera and the scene move (frame 1 reprojects frame 0's
reservoirs through motion vectors that carry object motion). Per frame: SHA-256 digests of the tile tables, the global proxy counters, the reservoirs after the fill pass, and of
every plane buffer (header, depth, motion vectors, stable radiance, throughput, hit distances, live plane records); the tile jitter; the ray counts.
tests/test_gpu_full_size.py runs pt_realtime_frame twice and compares. Run in the build container only (about a quarter of an hour of CPU time):
    python tests/golden/make_realtime_coupled_4k_golden.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from rtxpt_amd import scenes
from oracle import ptref
import make_realtime_4k_golden as rt

W, H, FRAMES, DT, STEP = rt.W, rt.H, 2, 0.45, (0.35, 0.02, -0.2)


def workload():
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=True); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    S = scenes.default_settings(useFp16Types=1, NEEType=2, nestedDielectricsQuality=2)
    poses = [(scenes.animate_instances(sc, DT * f), scenes.animate_positions(sc, DT * f)) for f in range(FRAMES)]
    return sc, cam, S, poses


def camera(cam, f): c = dict(cam); c["pos"] = tuple(np.asarray(cam["pos"], np.float64) + np.asarray(STEP) * f); return c


def params(cam, f):
    cur, prev = camera(cam, f), camera(cam, max(f - 1, 0))
    return scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cur), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=1), scenes.bridge_camera(W, H, **cur)


if __name__ == "__main__":
    sc, cam, S, poses = workload()
    def mk(mode):
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=True, mode=mode); o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H); return o
    filler, builder = mk(2), mk(1); filler.set_neeat(True)
    out = {}
    for f in range(FRAMES):
        t0 = time.time()
        prm, camd = params(cam, f)
        posed = dict(sc); posed["instances"], posed["positions"] = poses[f]
        for o in (filler, builder): o.set_scene(posed); o.set_previous_pose(*(poses[f - 1] if f else (sc["instances"], sc["positions"]))); o.set_camera(camd)
        filler.neeat_update_begin(); t1 = time.time()
        frame = builder.build_stable_planes(f, prm); t2 = time.time()
        filler.neeat_update_end(frame["depth"], frame["motion_vectors"]); t3 = time.time()
        filler.fill_stable_planes(f, prm, frame); t4 = time.time()
        tab, jit, cnt = filler.neeat_tables(); fw, fc = filler.neeat_feedback()
        out["table%d" % f] = rt.digest(tab); out["jitter%d" % f] = np.array(jit, np.uint32); out["counters%d" % f] = rt.digest(cnt); out["fbw%d" % f] = rt.digest(fw); out["fbc%d" % f] = rt.digest(fc)
        for k, v in rt.digests(frame).items(): out["%s%d" % (k, f)] = v
        print("frame %d: begin %.0f s, build %.0f s, end %.0f s, fill %.0f s; reservoirs filled %d" % (f, t1 - t0, t2 - t1, t3 - t2, t4 - t3, int((fw > 0).sum())), flush=True)
    out["rays"] = np.array([filler.counters()["extendRays"] + builder.counters()["extendRays"], filler.counters()["shadowRays"]], np.uint64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "realtime_coupled_4k_golden.npz"), **out)
    print("rays", out["rays"].tolist())
