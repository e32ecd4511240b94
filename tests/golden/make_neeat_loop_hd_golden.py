"""Generates tests/golden/neeat_loop_hd_golden.npz: the NEE-AT runs of tests/pin_scenes.neeat_loop_cases() — light baker in the loop, one sample per frame — at 1920x1080 (1917x1083
for the case that asks for partial tiles) through the REFERENCE'S text (LightsBaker.hlsl thread by thread + PathTracer.hlsli & co.): the options the 4K run leaves at their defaults
(no pre-filter, other feedback weights and ratios, analytic lights with the firefly filter, the C5 scene with three candidates, depth export + frustum boost). Per frame SHA-256
digests of the tile tables, proxy counters and reservoirs and the tile jitter; the accumulated frame's digest; ray counts. tests/test_gpu_parity_hd.py compares the device with it.
Run in the build container only (about a quarter of an hour of CPU time):   python tests/golden/make_neeat_loop_hd_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from rtxpt_amd import scenes
import pin_scenes
import make_neeat_loop_golden as small


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def size_of(name):
    _, _, w, h, _, _ = pin_scenes.neeat_loop_cases()[name]
    return (1917, 1083) if (w % 8 or h % 8) else (1920, 1080)


def run(name, make_tracer):
    """drives one case at its HD size through a tracer made by make_tracer(sc, camd, S, w, h) (oracle / reference text or device): {key: digest or small array}"""
    make, S, _, _, frames, opts = pin_scenes.neeat_loop_cases()[name]
    w, h = size_of(name)
    sc, cam = make()
    t, read_counters = make_tracer(sc, scenes.bridge_camera(w, h, **cam), S, w, h)
    small._extras(t, opts, w, h, cam)
    out = {}; rays = [0, 0]
    for f in range(frames):
        st = t.render(f, 1)
        tab, jit = t.neeat_tables()[:2]; fw, fc = t.light_feedback(0)
        out["%s_table%d" % (name, f)] = digest(tab); out["%s_jitter%d" % (name, f)] = np.array(jit, np.uint32); out["%s_counters%d" % (name, f)] = digest(read_counters(t))
        out["%s_fb%d" % (name, f)] = digest(np.concatenate([fw.view(np.uint32).ravel(), fc.ravel()]))
        if st is not None: rays[0] += int(st["extendRays"]); rays[1] += int(st["shadowRays"])
    out[name + "_frame"] = digest(t.radiance())
    if st is None: c = t.counters(); rays = [c["extendRays"], c["shadowRays"]]
    out[name + "_rays"] = np.array(rays, np.uint64)
    t.close()
    return out


if __name__ == "__main__":
    from oracle import ptref
    def make_reference(sc, camd, S, w, h):
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=bool(int(S["useFp16Types"]))); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h)
        return o, (lambda t: t.neeat_tables()[2])
    out = {}
    for name in pin_scenes.neeat_loop_cases():
        t0 = time.time(); out.update(run(name, make_reference)); print("%-36s %s  rays %s  %.0f s" % (name, size_of(name), out[name + "_rays"].tolist(), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neeat_loop_hd_golden.npz"), **out)
