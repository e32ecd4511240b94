"""Generates tests/golden/neeat_golden.npz: frames AND feedback reservoirs of the REFERENCE'S integrator text (LightSampler.hlsli's local sampler, its pdfs and
MIS, PathTracerNEE.hlsli's candidate loop and feedback insert, LightingTypes.hlsli's LightFeedbackReservoir — compiled from /root/reference by
oracle/refpin/hlsl_tu.py --integrator) for tests/pin_scenes.neeat_cases(). Per case: the frame, the ray counts, the number of baked lights (the synthetic tile
tables are regenerated from it and the case's seed) and one (total weight, candidate) plane pair per sample.
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_neeat_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes


def frame(name, reference, L=None):
    make, S, w, h, first, n, opts = pin_scenes.neeat_cases()[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=reference, settings=S, lp16=bool(S["useFp16Types"]))
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h)
    o.L.ptref_prepare(o.h)
    num_lights = len(o.lights()["lights"])
    o.set_local_light_sampling(pin_scenes.neeat_table(opts, num_lights, w, h), jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
    o.render(first, n)
    c = o.counters()
    out = {name: o.radiance(), name + "_rays": np.array([c["extendRays"], c["shadowRays"]], np.uint64), name + "_lights": np.array([num_lights], np.uint32)}
    if opts["feedback"]:
        for s in range(n):
            wgt, cand = o.light_feedback(s)
            out["%s_fbw%d" % (name, s)] = wgt; out["%s_fbc%d" % (name, s)] = cand
    return out


if __name__ == "__main__":
    out = {}
    for name in pin_scenes.neeat_cases():
        r = frame(name, True); out.update(r)
        print(name, r[name].shape, r[name + "_rays"], "lights", int(r[name + "_lights"][0]), "feedback slots filled", [int((r[k] > 0).sum()) for k in r if "_fbw" in k])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neeat_golden.npz"), **out)
