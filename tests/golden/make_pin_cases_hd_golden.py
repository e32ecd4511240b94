"""Generates tests/golden/pin_cases_hd_golden.npz: every case of tests/pin_scenes.py (cases() in the fp32 build, cases_lp16() in the reference's default build; NEE-AT's
neeat_cases() with their synthetic tile tables) at 1920x1080 x 8 samples through the REFERENCE'S integrator text — 16.6 M paths per case instead of the ~10 000 of the small fixtures,
so that one-in-a-million branches (seams, ties, clamps) are met — kept as SHA-256 of the frame (and of the reservoir planes) plus the ray counts. tests/test_gpu_parity_hd.py compares
the device with it. Run in the build container only (about twenty minutes of CPU time):   python tests/golden/make_pin_cases_hd_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

W, H, N = 1920, 1080, 8


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def all_cases():
    """key -> (kind, case name, lp16)"""
    out = {}
    for name in pin_scenes.cases(): out["fp32_" + name] = ("pt", name, False)
    for name in pin_scenes.cases_lp16(): out["lp16_" + name] = ("pt", name, True)
    for name in pin_scenes.neeat_cases(): out["neeat_" + name] = ("neeat", name, None)
    return out


def case_setup(key):
    kind, name, lp16 = all_cases()[key]
    if kind == "pt":
        make, S, w, h, first, n = (pin_scenes.cases_lp16() if lp16 else pin_scenes.cases())[name]; opts = None
    else:
        make, S, w, h, first, n, opts = pin_scenes.neeat_cases()[name]
    return make, S, first, opts


if __name__ == "__main__":
    out = {}
    only = sys.argv[1:]
    for key in all_cases():
        if only and not any(o in key for o in only): continue
        make, S, first, opts = case_setup(key)
        sc, cam = make()
        t0 = time.time()
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=bool(int(S["useFp16Types"])))
        o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H); o.L.ptref_prepare(o.h)
        nl = len(o.lights()["lights"])
        if opts is not None:
            tab = None if opts["table_seed"] is None else scenes.synthetic_local_light_tables(nl, W, H, seed=opts["table_seed"], jitter=opts["jitter"])
            o.set_local_light_sampling(tab, jitter=opts["jitter"], ratio=opts["ratio"], ssc_threshold=opts["ssc_threshold"], feedback=opts["feedback"])
        o.render(first, N); c = o.counters()
        out[key] = digest(o.radiance()); out[key + "_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64); out[key + "_lights"] = np.array([nl], np.uint32)
        if opts is not None and opts["feedback"]:
            for s in range(N):
                wgt, cand = o.light_feedback(s); out["%s_fb%d" % (key, s)] = digest(np.concatenate([wgt.view(np.uint32).ravel(), cand.ravel()]))
        print("%-55s rays %s  %.0f s" % (key, out[key + "_rays"].tolist(), time.time() - t0), flush=True)
        o.close()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pin_cases_hd_golden.npz")
    if only and os.path.exists(path): old = dict(np.load(path)); old.update(out); out = old
    np.savez_compressed(path, **out)
