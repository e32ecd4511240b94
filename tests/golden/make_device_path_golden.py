"""Generates tests/golden/device_path_golden.npz: what the REFERENCE TEXT answers on the two gathers the HIP path restructures in round 3 —
  * Bridge::loadSurface (PathTracerBridgeDonut.hlsli:612-853 and everything RTXPT-side below it, compiled by oracle/refpin/hlsl_tu.py --integrator) on random hits
    of four pin scenes, both builds of the lp types: the device reads the same hits through its flat 128-byte ShadeTri record (pt_scene.h);
  * Bridge::AlphaTest / AlphaTestVisibilityRay (:929-989) on random candidates: the device tests them against its per-texture alpha planes (byte opacities).
The GPU tests compare the device with these outputs directly, no oracle code in the loop (tests/test_gpu_reference_goldens.py).
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_device_path_golden.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ptref
import pin_scenes

out = {}
SURFACE = ["c2", "bistro_like", "bistro_like_material_zoo", "c2_spec_gloss"]
for lp16, cases in ((False, pin_scenes.cases()), (True, pin_scenes.cases_lp16())):
    for name in SURFACE:
        make, S, w, h, first, n = cases[name]
        sc, cam = make()
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16); o.set_scene(sc); o.set_settings(S); o.resize(8, 8)
        nt = o.num_tris(); rng = np.random.default_rng(0xD0 + len(name) + (1 if lp16 else 0)); k = 2500
        prims = rng.integers(0, nt, k).astype(np.uint32); u = rng.uniform(0, 1, k); v = rng.uniform(0, 1, k) * (1 - u)
        d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        rows = np.column_stack([u, v, d, rng.uniform(0, 0.5, k), rng.uniform(0, 0.01, k)]).astype(np.float32)
        R, Q = ptref.surface_probe(o, prims, rows)
        assert np.array_equal(R, Q)
        tag = "surface_%s_%s" % (name, "lp16" if lp16 else "fp32")
        out[tag + "_prims"], out[tag + "_rows"], out[tag + "_out"] = prims, rows, R
        print(tag, R.shape); o.close()
for name in ("bistro_like", "c2_exclude_from_nee"):
    make, S, w, h, first, n = pin_scenes.cases()[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S); o.set_scene(sc); o.set_settings(S); o.resize(8, 8)
    nt = o.num_tris(); rng = np.random.default_rng(0xA1FB); k = 20000
    prims = rng.integers(0, nt, k).astype(np.uint32); u = rng.uniform(0, 1, k); v = rng.uniform(0, 1, k) * (1 - u)
    uv = np.ascontiguousarray(np.column_stack([u, v]), np.float32); res = np.zeros((k, 4), np.uint32)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    o.L.refpt_alpha_probe(o.h, ctypes.c_uint32(k), vp(prims), vp(uv), vp(res))
    out["alpha_%s_prims" % name], out["alpha_%s_uv" % name], out["alpha_%s_out" % name] = prims, uv, res[:, :2].copy()      # (columns 0, 1: the reference text's scatter / visibility answers)
    print("alpha", name, int(res[:, 0].sum()), int(res[:, 1].sum())); o.close()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "device_path_golden.npz"), **out)
