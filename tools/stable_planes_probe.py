"""Times the realtime mode's two passes on one MI355X: the stable-plane build pass and one sub-sample of the fill pass on the bench scene (bistro-like, 2.8 M triangles), next to the
reference-mode frame of the same sample count -> JSON on stdout. GPU only, nothing under tools/ touches oracle/.   usage: python tools/stable_planes_probe.py [--width 3840 --height 2160]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtxpt_amd as pt
from rtxpt_amd import scenes


def main():
    ap = argparse.ArgumentParser(description=__doc__); ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160); ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--fill-only", type=int, default=0, help="for rocprofv3 kernel traces: one build pass, then this many fill passes as ONE batch (launches never overlap), nothing else")
    a = ap.parse_args(); w, h = a.width, a.height
    sc, cam = scenes.bistro_like(); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1      # as bench.py
    S = scenes.default_settings(useFp16Types=1)
    g = pt.PathTracer(device=0); g.set_scene(sc); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(S); g.resize(w, h)
    prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cam), sub_samples=1)
    ptr = lambda x: x.ctypes.data_as(__import__("ctypes").c_void_p)
    import ctypes
    fb = g.L.pt_build_stable_planes; fb.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]; fb.restype = ctypes.c_int32
    ff = g.L.pt_fill_stable_planes; ff.argtypes = fb.argtypes; ff.restype = ctypes.c_int32
    p = np.ascontiguousarray(prm); build, fill = [], []
    if a.fill_only:
        g._chk(fb(g.h, 0, ptr(p), None), "pt_build_stable_planes"); g.set_serial_kernels(True)
        for f in range(a.fill_only):
            sf = pt.PtFrameStats(); g._chk(ff(g.h, f, ptr(p), ctypes.byref(sf)), "pt_fill_stable_planes"); fill.append(sf.as_dict())
        print(json.dumps({"fill_pass_single_batch_ms": [r["gpuMilliseconds"] for r in fill]})); return
    for f in range(a.frames + 1):
        sb, sf = pt.PtFrameStats(), pt.PtFrameStats()
        g._chk(fb(g.h, f, ptr(p), ctypes.byref(sb)), "pt_build_stable_planes"); g._chk(ff(g.h, f, ptr(p), ctypes.byref(sf)), "pt_fill_stable_planes")
        if f: build.append(sb.as_dict()); fill.append(sf.as_dict())
    g.reset_accumulation(); g.render(0, 1); g.reset_accumulation(); ref = g.render(0, 1)
    med = lambda rows, k: float(np.median([r[k] for r in rows]))
    out = {"width": w, "height": h, "scene": "bistro_like (bench.py's)", "build_pass": {"ms": med(build, "gpuMilliseconds"), "rays": int(build[-1]["extendRays"]), "passes": int(build[-1]["iterations"])},
           "fill_pass_one_subsample": {"ms": med(fill, "gpuMilliseconds"), "extend_rays": int(fill[-1]["extendRays"]), "shadow_rays": int(fill[-1]["shadowRays"]), "passes": int(fill[-1]["iterations"])},
           "reference_mode_one_sample": {"ms": float(ref["gpuMilliseconds"]), "extend_rays": int(ref["extendRays"]), "shadow_rays": int(ref["shadowRays"])}}
    # the coupled frame with the baker in the loop (pt_realtime_frame, pt_set_neeat): wall time of the call = baker (UpdateBegin + UpdateEnd) + build pass + fill pass
    import time
    g.set_neeat(True)
    for f in range(3): g.realtime_frame(f, prm)
    walls, b_ms, f_ms = [], [], []
    for f in range(3, 3 + a.frames):
        t0 = time.perf_counter(); _, bs, fs = g.realtime_frame(f, prm); walls.append((time.perf_counter() - t0) * 1e3); b_ms.append(bs["gpuMilliseconds"]); f_ms.append(fs["gpuMilliseconds"])
    out["realtime_frame_with_neeat"] = {"wall_ms_incl_read_back": float(np.median(walls)), "build_ms": float(np.median(b_ms)), "fill_ms": float(np.median(f_ms)), "fill_extend_rays": int(fs["extendRays"]), "fill_shadow_rays": int(fs["shadowRays"])}
    for k in ("build_pass",): out[k]["mrays_per_s"] = out[k]["rays"] / out[k]["ms"] / 1e3
    out["fill_pass_one_subsample"]["mrays_per_s"] = (out["fill_pass_one_subsample"]["extend_rays"] + out["fill_pass_one_subsample"]["shadow_rays"]) / out["fill_pass_one_subsample"]["ms"] / 1e3
    out["reference_mode_one_sample"]["mrays_per_s"] = (out["reference_mode_one_sample"]["extend_rays"] + out["reference_mode_one_sample"]["shadow_rays"]) / out["reference_mode_one_sample"]["ms"] / 1e3
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
