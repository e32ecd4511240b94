"""Generates tests/golden/realtime_golden.npz: runs of the realtime mode's coupled frame (tests/realtime_cases.py) as the REFERENCE TEXT produces them — LightsBaker.hlsl's passes thread by thread,
PathTracer.hlsli & co. compiled with PATH_TRACER_MODE_BUILD_STABLE_PLANES for the build pass and with PATH_TRACER_MODE_FILL_STABLE_PLANES for the fill passes (two pin libraries; the baker's state lives
with the fill library, the build pass's buffers travel as arrays). Per frame: the tile tables and jitter the fill passes sampled, the global proxy counters, the feedback reservoirs the fill passes left,
the planes' noisy radiance, the specular hit distance, and the build pass's depth / motion vectors / header.
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_realtime_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
import realtime_cases as rc


def run_oracle(name, reference):
    from oracle import ptref
    make, _, w, h, frames, subs, step, kw = rc.cases()[name]
    S = rc.settings_for(name); lp16 = bool(int(S["useFp16Types"]))
    sc, cam = make()
    def mk(mode):
        o = ptref.Oracle(reference_integrator=reference, settings=S, lp16=lp16, mode=mode) if reference else ptref.Oracle(lp16=lp16)
        o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); return o
    filler = mk(2); builder = mk(1) if reference else filler
    filler.set_neeat(True)
    def set_camera(c):
        filler.set_camera(c)
        if builder is not filler: builder.set_camera(c)
    def read(frame):
        t, j, pc = filler.neeat_tables(); fw, fc = filler.neeat_feedback()
        return dict(table=t, jitter=np.array(j, np.uint32), counters=pc, fbw=fw, fbc=fc, noisy=rc.live_noisy(frame, w, h), spec_hit_t=frame["spec_hit_t"], depth=frame["depth"],
                    motion_vectors=frame["motion_vectors"], header=frame["header"])
    def pose(cur, prev):      # an animated case: the frame's pose into both contexts (a fresh upload: the oracle has no refit), the previous one as the scene's motion history
        posed = dict(sc); posed["instances"], posed["positions"] = cur
        for o in ([filler] if builder is filler else [filler, builder]): o.set_scene(posed); o.set_previous_pose(*prev)
    out = rc.run(name, filler.neeat_update_begin, lambda s, prm: builder.build_stable_planes(s, prm), lambda fr: filler.neeat_update_end(fr["depth"], fr["motion_vectors"]),
                 lambda s, prm, fr: filler.fill_stable_planes(s, prm, fr), read, set_camera, pose)
    out[name + "_rays"] = np.array([filler.counters()["extendRays"] + (0 if builder is filler else builder.counters()["extendRays"]), filler.counters()["shadowRays"]], np.uint64)
    filler.close()
    if builder is not filler: builder.close()
    return out


if __name__ == "__main__":
    out = {}
    for name in rc.cases():
        r = run_oracle(name, True); out.update(r)
        frames = rc.cases()[name][4]
        print(name, "frames", frames, "reservoirs filled", [int((r["%s_fbw%d" % (name, f)] > 0).sum()) for f in range(frames)],
              "pixels with motion", [int((r["%s_motion_vectors%d" % (name, f)][..., :2] != 0).any(-1).sum()) for f in range(frames)], "rays", r[name + "_rays"].tolist())
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "realtime_golden.npz"), **out)
