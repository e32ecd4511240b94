"""developer probe: C5 (animated bistro-like, full size) per-frame costs on one GPU: refit, light re-bake (device-side weights / proxy table), frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=True)
g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(nestedDielectricsQuality=2, useFp16Types=1)); g.resize(W, H)
g.render(0, SPP); print("first frame build stats", g.build_stats())
for f, t in enumerate((0.2, 0.4, 0.6, 0.8, 1.0)):
    inst, pos = scenes.animate_instances(sc, t), scenes.animate_positions(sc, t)
    t0 = time.perf_counter(); g.animate(instances=inst, positions=pos, rebuild=False, vertex_ranges=None if os.environ.get("C5_PROBE_FULL") else scenes.animated_vertex_ranges(sc)); t1 = time.perf_counter()
    g.reset_accumulation(); st = g.render(f * SPP, SPP); t2 = time.perf_counter()
    b = g.build_stats()
    print("frame %d: animate call %.2f ms (refit %.2f ms, light re-bake %.2f ms), render %.2f ms, %.1f Mrays/s" % (f, (t1 - t0) * 1e3, b["refitMs"], b["lightBakeMs"], (t2 - t1) * 1e3, (st["extendRays"] + st["shadowRays"]) / (t2 - t1) / 1e6))
