"""Generates tests/golden/config_frames_golden.npz: BASELINE.json's other configurations at their full sizes (C1 Cornell 256^2 x 1, C2 Cornell 1920x1080 x 4, C4 the bench scene
3840x2160 x 16, C5 the animated scene 3840x2160 x 4 at two poses, nested dielectrics quality 2; the definitions of tools/run_configs.py) rendered by the REFERENCE'S integrator text
(oracle/refpin/hlsl_tu.py --integrator over the oracle's scene services): per frame the SHA-256 of the RGBA32F frame, four rows and the ray counts. C3 is bench_frame_golden.npz.
tests/test_gpu_full_size.py compares the device's frames with it. Run in the build container only (about a quarter of an hour of CPU time):
    python tests/golden/make_config_frames_golden.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes


def configs():
    """name -> (scene maker, settings, width, height, first sample, samples, animation time or None)"""
    def bench_scene(animated):
        def make():
            sc, cam = scenes.bistro_like(animated=animated); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1; return sc, cam
        return make
    d = scenes.default_settings
    return {"C1": (lambda: scenes.cornell_box("C1"), scenes.config_settings("C1"), 256, 256, 0, 1, None),
            "C2": (lambda: scenes.cornell_box("C2"), scenes.config_settings("C2"), 1920, 1080, 0, 4, None),
            "C4": (bench_scene(False), d(useFp16Types=1), 3840, 2160, 0, 16, None),
            "C5_t1": (bench_scene(True), d(useFp16Types=1, nestedDielectricsQuality=2), 3840, 2160, 0, 4, 0.1),
            "C5_t2": (bench_scene(True), d(useFp16Types=1, nestedDielectricsQuality=2), 3840, 2160, 0, 4, 0.2)}


def posed(sc, t):
    if t is None: return sc
    out = dict(sc); out["instances"] = scenes.animate_instances(sc, t); out["positions"] = scenes.animate_positions(sc, t); return out


def rows_of(h): return [h // 7, h // 3, h // 2, (6 * h) // 7]


if __name__ == "__main__":
    out = {}
    for name, (make, S, w, h, first, n, t) in configs().items():
        sc, cam = make()
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=bool(int(S["useFp16Types"])))
        o.set_scene(posed(sc, t)); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h)
        t0 = time.time(); o.render(first, n); dt = time.time() - t0
        rad = o.radiance(); c = o.counters()
        out[name + "_sha256"] = pin_scenes.frame_digest(rad); out[name + "_rows"] = rad[rows_of(h)].copy(); out[name + "_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
        print("%s: %dx%d x %d, rays %s, %.0f s" % (name, w, h, n, out[name + "_rays"].tolist(), dt), flush=True)
        o.close()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_frames_golden.npz"), **out)
