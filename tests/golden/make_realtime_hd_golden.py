"""Generates tests/golden/realtime_hd_golden.npz: the coupled realtime runs of tests/realtime_cases.py (baker UpdateBegin, build pass, UpdateEnd on the frame's depth + motion vectors,
fill sub-samples feeding the reservoirs; moving camera, one case with an animated scene, one with two sub-samples in the lp16 build, the delta-tree zoo) at 1920x1080 through the
REFERENCE'S text (LightsBaker.hlsl thread by thread, PathTracer.hlsli & co. in both PATH_TRACER_MODEs): per frame SHA-256 digests of every output tests/realtime_cases.KEYS names;
the run's ray counts. tests/test_gpu_parity_hd.py compares the device with it. Run in the build container only (about a quarter of an hour of CPU time):
    python tests/golden/make_realtime_hd_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from rtxpt_amd import scenes
import realtime_cases as rc

W, H = 1920, 1080


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def live_noisy(frame):
    hd = frame["header"]; P = frame["planes"].reshape(-1, 20); rows = []
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        rows.append(P[np.sort(scenes.stable_planes_address(xs.astype(np.int64), ys.astype(np.int64), pl, W, H)), 16:18])
    return np.concatenate(rows)


def frames_of(name):
    """[(camera struct, params, pose or None, previous pose or None)] per frame of a case at W x H"""
    make, _, _, _, frames, subs, step, kw = rc.cases()[name]
    sc, cam = make()
    P = rc.poses(sc, kw, frames); kw = {k: v for k, v in kw.items() if k != "anim_dt"}
    out = []
    for f in range(frames):
        cur, prev = rc.camera(cam, step, f), rc.camera(cam, step, max(f - 1, 0))
        prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cur), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=subs, **kw)
        out.append((scenes.bridge_camera(W, H, **cur), prm, None if P is None else P[f], None if P is None else (P[f - 1] if f else (sc["instances"], sc["positions"]))))
    return sc, cam, rc.settings_for(name), subs, out


def record(name, f, frame, tab, jit, cnt, fw, fc):
    return {"%s_table%d" % (name, f): digest(tab), "%s_jitter%d" % (name, f): np.array(jit, np.uint32), "%s_counters%d" % (name, f): digest(cnt),
            "%s_fb%d" % (name, f): digest(np.concatenate([fw.view(np.uint32).ravel(), fc.ravel()])), "%s_noisy%d" % (name, f): digest(live_noisy(frame)),
            **{"%s_%s%d" % (name, k, f): digest(frame[k]) for k in ("spec_hit_t", "depth", "motion_vectors", "header", "stable_radiance", "throughput")}}


if __name__ == "__main__":
    from oracle import ptref
    out = {}
    for name in rc.cases():
        t0 = time.time()
        sc, cam, S, subs, frames = frames_of(name); lp16 = bool(int(S["useFp16Types"]))
        def mk(mode):
            o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=mode); o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H); return o
        filler, builder = mk(2), mk(1); filler.set_neeat(True)
        for f, (camd, prm, pose, prev) in enumerate(frames):
            if pose is not None:
                posed = dict(sc); posed["instances"], posed["positions"] = pose
                for o in (filler, builder): o.set_scene(posed); o.set_previous_pose(*prev)
            for o in (filler, builder): o.set_camera(camd)
            filler.neeat_update_begin(); frame = builder.build_stable_planes(f * subs, prm); filler.neeat_update_end(frame["depth"], frame["motion_vectors"])
            for s in range(subs): filler.fill_stable_planes(f * subs + s, prm, frame)
            tab, jit, cnt = filler.neeat_tables(); fw, fc = filler.neeat_feedback()
            out.update(record(name, f, frame, tab, jit, cnt, fw, fc))
        out[name + "_rays"] = np.array([filler.counters()["extendRays"] + builder.counters()["extendRays"], filler.counters()["shadowRays"]], np.uint64)
        print("%-36s frames %d  rays %s  %.0f s" % (name, len(frames), out[name + "_rays"].tolist(), time.time() - t0), flush=True)
        filler.close(); builder.close()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "realtime_hd_golden.npz"), **out)
