"""Measure BASELINE.json configs C1..C5 on one MI355X (+ the CPU oracle on a bounded sample) -> JSON on stdout.
usage: python tools/run_configs.py [--cpu]      (developer tool; bench.py is the contract benchmark: C3)"""
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
           "build_ms": g.build_stats()["buildMs"]}
    if animate is not None:
        out["refit_ms"] = float(np.median(refit[warmup:]))
    return out


def cpu_run(sc, cam, S, w, h, spp, rect=None):
    from oracle import ptref
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h)
    o.L.ptref_prepare(o.h)
    t0 = time.perf_counter()
    if rect is None: o.render(0, spp)
    else: o.render(0, spp, rect=rect)
    dt = time.perf_counter() - t0
    c = o.counters(); rays = c["extendRays"] + c["shadowRays"]
    return {"seconds": dt, "rays": int(rays), "mrays_per_s": rays / dt / 1e6, "threads": ptref.num_threads(), "sample": "full frame" if rect is None else "rect %s" % (rect,)}


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cpu", action="store_true"); a = ap.parse_args()
    res = {}
    sc, cam = scenes.cornell_box("C1")
    res["C1"] = {"gpu": gpu_run(sc, cam, scenes.config_settings("C1"), 256, 256, 1)}
    if a.cpu: res["C1"]["cpu"] = cpu_run(sc, cam, scenes.config_settings("C1"), 256, 256, 1)
    sc, cam = scenes.cornell_box("C2")
    res["C2"] = {"gpu": gpu_run(sc, cam, scenes.config_settings("C2"), 1920, 1080, 4)}
    if a.cpu: res["C2"]["cpu"] = cpu_run(sc, cam, scenes.config_settings("C2"), 1920, 1080, 4, rect=(720, 405, 1200, 675))
    sc, cam = scenes.bistro_like()
    res["C3"] = {"gpu": gpu_run(sc, cam, scenes.default_settings(), 3840, 2160, 4)}
    if a.cpu: res["C3"]["cpu"] = cpu_run(sc, cam, scenes.default_settings(), 3840, 2160, 4, rect=(1440, 810, 2400, 1350))
    res["C4"] = {"gpu": gpu_run(sc, cam, scenes.default_settings(), 3840, 2160, 16, frames=2, warmup=1)}
    sc, cam = scenes.bistro_like(animated=True)
    res["C5"] = {"gpu": gpu_run(sc, cam, scenes.default_settings(), 3840, 2160, 4, animate=True)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
