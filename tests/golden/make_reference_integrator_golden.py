"""Generates tests/golden/reference_integrator_golden.npz: linear-radiance frames rendered by the REFERENCE'S integrator text (PathTracer.hlsli,
PathTracerNEE.hlsli, PathTracerNestedDielectrics.hlsli, LightSampler.hlsli, PolymorphicLight.hlsli, EnvMap.hlsli, PathState.hlsli, BxDF.hlsli ...
compiled from /root/reference by oracle/refpin/hlsl_tu.py --integrator) over the oracle's scene services, for the cases of tests/pin_scenes.py.
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_reference_integrator_golden.py [--wide-only | --xl-only]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

for lp16, cases, fname in (() if ("--wide-only" in sys.argv or "--xl-only" in sys.argv) else ((False, pin_scenes.cases(), "reference_integrator_golden.npz"), (True, pin_scenes.cases_lp16(), "reference_integrator_golden_lp16.npz"))):
    out = {}            # lp16: the reference text compiled with RTXPT_LP_TYPES_USE_16BIT_PRECISION=1 (its default build) over hlsl_shim.h's binary16 type
    for name, (make, S, w, h, first, n) in cases.items():
        sc, cam = make()
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16)
        o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
        out[name] = o.radiance(); c = o.counters()
        out[name + "_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
        print("lp16" if lp16 else "fp32", name, out[name].shape, out[name + "_rays"])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), fname), **out)

# one notch wider (pin_scenes.wide_cases): 256 x 144 x 4 samples per pin family, both lp builds in one file
out = {}
for name, (make, S, w, h, first, n) in ({} if "--xl-only" in sys.argv else pin_scenes.wide_cases()).items():
    lp16 = bool(int(S["useFp16Types"]))
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
    out[name] = o.radiance(); c = o.counters()
    out[name + "_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
    print("wide", name, out[name].shape, out[name + "_rays"])
if out: np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_integrator_golden_wide.npz"), **out)

# and one more notch (pin_scenes.xl_cases): 1280 x 720 x 4 samples; every sixteenth row + a digest of the whole frame
out = {}
for name, (make, S, w, h, first, n) in pin_scenes.xl_cases().items():
    lp16 = bool(int(S["useFp16Types"]))
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
    rad = o.radiance(); c = o.counters()
    out[name + "_rows"] = rad[::pin_scenes.XL_ROW_STEP].copy(); out[name + "_sha256"] = pin_scenes.frame_digest(rad)
    out[name + "_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
    print("xl", name, rad.shape, out[name + "_rays"], "triangles", len(sc["indices"]) // 3)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_integrator_golden_xl.npz"), **out)
