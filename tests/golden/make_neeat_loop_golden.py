"""Generates tests/golden/neeat_loop_golden.npz: NEE-AT runs with the light baker in the loop, as the REFERENCE TEXT produces them — LightsBaker.hlsl's
ProcessFeedbackHistoryPreFilter / P0 / P1a / P1b / P2 / P3 and ClearFeedbackHistory executed thread by thread (oracle/refpin/hlsl_lbfb_stubs.h) between frames of the
reference's path tracer text, for tests/pin_scenes.neeat_loop_cases(). Per case: the accumulated frame and, for every frame, the tile tables and jitter it was traced with,
the global proxy counters, and the feedback reservoirs it left.
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_neeat_loop_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
import pin_scenes


def _collect(name, frames, step):
    out = {}
    for f in range(frames):
        rad, table, jitter, counters, fbw, fbc = step(f)
        out["%s_table%d" % (name, f)] = table; out["%s_jitter%d" % (name, f)] = np.array(jitter, np.uint32); out["%s_fbw%d" % (name, f)] = fbw; out["%s_fbc%d" % (name, f)] = fbc
        if counters is not None: out["%s_counters%d" % (name, f)] = counters
    out[name] = rad
    return out


def _extras(t, opts, w, h, cam):
    o = dict(opts); vp = o.pop("view_projection", False); boost = o.pop("importance_boost", False)
    M = scenes.view_projection(w, h, **cam)
    if vp: t.set_view_projection(M)
    if boost: t.set_light_importance_boost(M)
    t.set_neeat(True, **o)


def run_oracle(name, reference):
    from oracle import ptref
    make, S, w, h, frames, opts = pin_scenes.neeat_loop_cases()[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=reference, settings=S, lp16=bool(S["useFp16Types"]))
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); _extras(o, opts, w, h, cam)
    def step(f):
        o.render(f, 1); t, j, pc = o.neeat_tables(); fw, fc = o.light_feedback(0)
        return o.radiance(), t, j, pc, fw, fc
    return _collect(name, frames, step)


def run_device(name, one_call=False):
    import rtxpt_amd as pt
    make, S, w, h, frames, opts = pin_scenes.neeat_loop_cases()[name]
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h); _extras(t, opts, w, h, cam)
    if one_call:
        t.render(0, frames); tab, j = t.neeat_tables(); fw, fc = t.light_feedback(0)
        out = {name: t.radiance(), "%s_table%d" % (name, frames - 1): tab, "%s_jitter%d" % (name, frames - 1): np.array(j, np.uint32), "%s_fbw%d" % (name, frames - 1): fw, "%s_fbc%d" % (name, frames - 1): fc}
        t.close(); return out
    def step(f):
        t.render(f, 1); tab, j = t.neeat_tables(); fw, fc = t.light_feedback(0)
        return t.radiance(), tab, j, t.lights()["proxyCounters"], fw, fc
    out = _collect(name, frames, step); t.close()
    return out


if __name__ == "__main__":
    out = {}
    for name in pin_scenes.neeat_loop_cases():
        r = run_oracle(name, True); out.update(r)
        frames = pin_scenes.neeat_loop_cases()[name][4]
        print(name, r[name].shape, "frames", frames, "feedback slots filled", [int((r["%s_fbw%d" % (name, f)] > 0).sum()) for f in range(frames)],
              "lights per tile", ["%.1f" % np.mean([len(np.unique(x >> 9)) for x in r["%s_table%d" % (name, f)].reshape(-1, 128)]) for f in range(frames)])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neeat_loop_golden.npz"), **out)
