"""Generates tests/golden/realtime_4k_golden.npz: the realtime mode's two path-tracing passes at FULL size through the REFERENCE'S text (PATH_TRACER_MODE_BUILD_STABLE_PLANES and
PATH_TRACER_MODE_FILL_STABLE_PLANES of PathTracer.hlsli & co., compiled by oracle/refpin/hlsl_tu.py --integrator) — C5's scene (2.86 M triangles) in an animated pose with the rest pose
as the previous frame (object motion in the motion vectors) and a camera that moved, 3840x2160, the reference's default lp16 build, nested dielectrics quality 2, the global light
sampler (the coupled frame with the baker's text in the loop: make_realtime_coupled_4k_golden.py): SHA-256 digests of
the header, depth, motion vectors, stable radiance, throughput, specular hit distances and of the live plane records (noisy radiance included), plus the ray counts of both passes.
tests/test_gpu_full_size.py compares the device with it. Run in the build container only (a few minutes of CPU time):  python tests/golden/make_realtime_4k_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref

W, H, SAMPLE, T_POSE = 3840, 2160, 3, 0.45
KEYS = ("header", "depth", "motion_vectors", "stable_radiance", "throughput", "spec_hit_t")


def workload():
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=True); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    S = scenes.default_settings(useFp16Types=1, nestedDielectricsQuality=2)
    prev = dict(cam); prev["pos"] = tuple(np.asarray(cam["pos"], np.float64) - np.array([0.35, 0.02, -0.2]))
    prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=1)
    pose = (scenes.animate_instances(sc, T_POSE), scenes.animate_positions(sc, T_POSE))
    return sc, cam, S, prm, pose


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def live_planes_digest(frame):
    """the records of the planes that exist, in (plane, address) order"""
    hd = frame["header"]; P = frame["planes"].reshape(-1, 20); h = hashlib.sha256()
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        idx = np.sort(scenes.stable_planes_address(xs.astype(np.int64), ys.astype(np.int64), pl, W, H))
        h.update(np.ascontiguousarray(P[idx]).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def digests(frame):
    out = {k: digest(frame[k]) for k in KEYS}; out["live_planes"] = live_planes_digest(frame); return out


if __name__ == "__main__":
    sc, cam, S, prm, pose = workload()
    posed = dict(sc); posed["instances"], posed["positions"] = pose
    out = {}
    t0 = time.time()
    b = ptref.Oracle(reference_integrator=True, settings=S, lp16=True, mode=1)
    b.set_scene(posed); b.set_previous_pose(sc["instances"], sc["positions"]); b.set_camera(scenes.bridge_camera(W, H, **cam)); b.set_settings(S); b.resize(W, H)
    frame = b.build_stable_planes(SAMPLE, prm); out["build_rays"] = np.array([b.counters()["extendRays"]], np.uint64)
    for k, v in digests(frame).items(): out["build_" + k] = v
    print("build pass: rays %d, %.0f s" % (int(out["build_rays"][0]), time.time() - t0), flush=True)
    t0 = time.time()
    f = ptref.Oracle(reference_integrator=True, settings=S, lp16=True, mode=2)
    f.set_scene(posed); f.set_camera(scenes.bridge_camera(W, H, **cam)); f.set_settings(S); f.resize(W, H)
    f.fill_stable_planes(SAMPLE, prm, frame); c = f.counters(); out["fill_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
    for k, v in digests(frame).items(): out["fill_" + k] = v
    print("fill pass: rays %s, %.0f s" % (out["fill_rays"].tolist(), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "realtime_4k_golden.npz"), **out)
