#!/usr/bin/env python3
"""Developer probe (needs the GPU): what NEE-AT costs and what it buys on the bench workload (bistro_like, 2.8 M triangles, 1920 x 1080 by default).
  * cost: ms per frame of the plain global sampler (NEEType 1, one sample per call) against NEE-AT with the baker in the loop (pt_set_neeat), and the baker's share
    (the passes run between frames: PreFilter, P0, proxy rebuild, P1a, P1b, tile fill, sort, Clear);
  * benefit: relative RMSE of 8 x 8 block means against a converged frame at equal sample counts — NEE-AT is a variance-reduction technique, the expectation of both estimators is the same image.
usage: python tools/neeat_probe.py [--width W --height H --scale S --frames N --ref-spp R]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rtxpt_amd as pt
from rtxpt_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080); ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--tex", type=int, default=1024)
ap.add_argument("--frames", type=int, default=16); ap.add_argument("--ref-spp", type=int, default=512)
a = ap.parse_args()
sc, cam = scenes.bistro_like(scale=a.scale, tex_size=a.tex)
camd = scenes.bridge_camera(a.width, a.height, **cam)


def tracer(nee_type):
    S = scenes.default_settings(useFp16Types=1, NEEType=nee_type)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(a.width, a.height)
    return g


def lum(img): return img[..., 0] * 0.2126 + img[..., 1] * 0.7152 + img[..., 2] * 0.0722


g = tracer(1)
g.render(0, 4); g.reset_accumulation()
t0 = time.time(); g.render(1000, a.ref_spp); ref = g.radiance()[..., :3].astype(np.float64); t_ref = time.time() - t0; g.reset_accumulation()
print("reference: %d spp of the global sampler in %.2f s; %d lights" % (a.ref_spp, t_ref, len(g.lights()["lights"])))
def blocks(img):      # 8 x 8 block means of the luminance: linear in the samples (no bias at low sample counts), and what a viewer's eye or a denoiser integrates over
    l = lum(img[..., :3].astype(np.float64)); h, w = l.shape[0] // 8 * 8, l.shape[1] // 8 * 8
    return l[:h, :w].reshape(h // 8, 8, w // 8, 8).mean((1, 3))


ref_b = blocks(ref); keep = ref_b < np.percentile(ref_b, 99.5)      # the reference's own unconverged fireflies should not decide the comparison


def rmse(img): return float(np.sqrt(np.mean((blocks(img)[keep] - ref_b[keep]) ** 2)) / ref_b[keep].mean())


rows = []
for name, neeat in (("global sampler (NEEType 1)", False), ("NEE-AT, baker in the loop", True)):
    g = tracer(2 if neeat else 1)
    if neeat: g.set_neeat(True)
    g.render(0, 1); g.reset_accumulation()
    if neeat: g.neeat_reset()
    ms, gpu_ms, errs = [], [], {}
    for f in range(a.frames):
        t0 = time.time(); st = g.render(f, 1); ms.append((time.time() - t0) * 1e3); gpu_ms.append(st["gpuMilliseconds"])
        if f + 1 in (1, 2, 4, 8, 16, 32, 64): errs[f + 1] = rmse(g.radiance())
    rows.append((name, ms, gpu_ms, errs))
    print("%-28s wall ms/frame: first %.2f, median of the rest %.2f; path tracing (GPU events) median %.2f ms -> baker + host %.2f ms" % (
        name, ms[0], float(np.median(ms[1:])), float(np.median(gpu_ms[1:])), float(np.median(ms[1:]) - np.median(gpu_ms[1:]))))
    print("%-28s relative RMSE of the accumulated luminance (8 x 8 block means) vs the reference: %s" % ("", "  ".join("%d spp: %.4f" % kv for kv in sorted(errs.items()))))
    g.close()
e0, e1 = rows[0][3], rows[1][3]
print("RMSE ratio NEE-AT / global at equal samples: " + "  ".join("%d spp: %.3f" % (k, e1[k] / e0[k]) for k in sorted(e0)))
print("equal-time: NEE-AT frames cost %.3f x a global-sampler frame (median wall)" % (np.median(rows[1][1][1:]) / np.median(rows[0][1][1:])))
