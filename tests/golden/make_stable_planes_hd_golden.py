"""Generates tests/golden/stable_planes_hd_golden.npz: every stable-plane case of tests/stable_planes_cases.py (cases(), motion_cases(), edge_cases()) at 1920x1080 through the
REFERENCE'S text of the two passes (PATH_TRACER_MODE_BUILD_STABLE_PLANES, then two FILL sub-samples): SHA-256 digests of every plane buffer and of the live plane records after the
build pass and after the fill passes, ray counts. tests/test_gpu_parity_hd.py compares the device with it. Run in the build container only (about ten minutes of CPU time):
    python tests/golden/make_stable_planes_hd_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import stable_planes_cases as spc

W, H, SAMPLE, SUBS = 1920, 1080, 5, 2
KEYS = ("header", "depth", "motion_vectors", "stable_radiance", "throughput", "spec_hit_t")


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def digests(frame):
    out = {k: digest(frame[k]) for k in KEYS}
    hd = frame["header"]; P = frame["planes"].reshape(-1, 20); h = hashlib.sha256()
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        h.update(np.ascontiguousarray(P[np.sort(scenes.stable_planes_address(xs.astype(np.int64), ys.astype(np.int64), pl, W, H))]).tobytes())
    out["live_planes"] = np.frombuffer(h.digest(), np.uint8).copy()
    return out


def all_cases(): return ["case_" + n for n in spc.cases()] + ["motion_" + n for n in spc.motion_cases()] + ["edge_" + n for n in spc.edge_cases()]


def setup(key):
    """(scene, camera kwargs, settings, params, lp16, previous pose or None) of a case at W x H (the cases' own frame sizes are replaced)"""
    kind, name = key.split("_", 1)
    prev_pose = None
    if kind == "edge":
        make, over, kw, _, _ = spc.edge_cases()[name]; lp16 = False
        sc, cam = make(); S = scenes.config_settings("C2")
        for k, v in over.items(): S[k] = v
        prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), sub_samples=SUBS, **kw)
    else:
        base = spc.motion_cases()[name] if kind == "motion" else name
        lp16, over, kw = spc.cases()[base][:3]
        sc, cam = scenes.stable_planes_zoo(*spc.cases()[base][3:]); S = scenes.config_settings("C2")
        for k, v in over.items(): S[k] = v
        if lp16: S["useFp16Types"] = 1
        prev = dict(cam); prev["pos"] = tuple(np.asarray(cam["pos"]) + np.array([0.03, 0.01, 0.02]))
        prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=SUBS, **kw)
        if kind == "motion": prev_pose = scenes.previous_pose(sc)
    return sc, cam, S, prm, lp16, prev_pose


if __name__ == "__main__":
    out = {}
    for key in all_cases():
        sc, cam, S, prm, lp16, prev_pose = setup(key)
        t0 = time.time()
        b = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=1); b.set_scene(sc); b.set_camera(scenes.bridge_camera(W, H, **cam)); b.set_settings(S); b.resize(W, H)
        if prev_pose is not None: b.set_previous_pose(*prev_pose)
        frame = b.build_stable_planes(SAMPLE, prm); out[key + "_build_rays"] = np.array([b.counters()["extendRays"]], np.uint64)
        for k, v in digests(frame).items(): out["%s_build_%s" % (key, k)] = v
        f = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=2); f.set_scene(sc); f.set_camera(scenes.bridge_camera(W, H, **cam)); f.set_settings(S); f.resize(W, H)
        for s in range(SUBS): f.fill_stable_planes(SAMPLE + s, prm, frame)
        c = f.counters(); out[key + "_fill_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
        for k, v in digests(frame).items(): out["%s_fill_%s" % (key, k)] = v
        print("%-32s build rays %d, fill rays %s, %.0f s" % (key, int(out[key + "_build_rays"][0]), out[key + "_fill_rays"].tolist(), time.time() - t0), flush=True)
        b.close(); f.close()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stable_planes_hd_golden.npz"), **out)
