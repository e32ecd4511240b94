"""Measure BASELINE.json configs C1..C5 on one MI355X -> JSON on stdout (GPU only: the CPU oracle is timed by bench.py's cpu_baseline leg
and by the tests, nothing under tools/ touches oracle/).  usage: python tools/run_configs.py      (developer tool; bench.py is the contract benchmark: C3)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtxpt_amd as pt
from rtxpt_amd import scenes


def gpu_run(sc, cam, S, w, h, spp, frames=5, warmup=2, animate=None):
    g = pt.PathTracer(device=0); g.set_scene(sc); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(S); g.resize(w, h)
    ms, rays, refit = [], 0, []
    for f in range(warmup + frames):
        if animate is not None:
            g.animate(instances=scenes.animate_instances(sc, 0.1 * f), positions=scenes.animate_positions(sc, 0.1 * f), rebuild=False)
            refit.append(g.build_stats()["refitMs"])
        g.reset_accumulation(); st = g.render(0, spp)
        if f >= warmup:
            ms.append(st["gpuMilliseconds"] + (refit[-1] if animate is not None else 0.0)); rays = st["extendRays"] + st["shadowRays"]
    m = float(np.median(ms))
    out = {"ms_per_frame": m, "rays_per_frame": int(rays), "mrays_per_s": rays / m / 1e3, "extend_rays": int(st["extendRays"]), "shadow_rays": int(st["shadowRays"]),
           "build_ms": g.build_stats()["buildMs"], "bvh": g.bvh_info()}
    if animate is not None:
        out["refit_ms"] = float(np.median(refit[warmup:]))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__); ap.add_argument("--only", default="", help="comma-separated subset, e.g. C1,C2"); args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    res = {}
    sc, cam = scenes.cornell_box("C1")
    res["C1"] = {"gpu": gpu_run(sc, cam, scenes.config_settings("C1"), 256, 256, 1)}
    sc, cam = scenes.cornell_box("C2")
    res["C2"] = {"gpu": gpu_run(sc, cam, scenes.config_settings("C2"), 1920, 1080, 4)}
    if only and not (only - {"C1", "C2"}): print(json.dumps({k: v for k, v in res.items() if k in only}, indent=1)); return
    sc, cam = scenes.bistro_like(); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1      # as bench.py
    res["C3"] = {"gpu": gpu_run(sc, cam, scenes.default_settings(useFp16Types=1), 3840, 2160, 4)}
    res["C4"] = {"gpu": gpu_run(sc, cam, scenes.default_settings(useFp16Types=1), 3840, 2160, 16, frames=2, warmup=1)}
    sc, cam = scenes.bistro_like(animated=True); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    res["C5"] = {"gpu": gpu_run(sc, cam, scenes.default_settings(useFp16Types=1, nestedDielectricsQuality=2), 3840, 2160, 4, animate=True)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
