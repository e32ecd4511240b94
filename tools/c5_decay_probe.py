"""developer probe: C5 (animated bistro-like, full size) over a sequence of frames with REFIT ONLY (the fast-trace topology of frame 0 is kept, as the reference keeps its BLAS topology and
calls PerformUpdate, Sample.cpp:1065, 1170-1198): how the frame time and the rays per second develop while the animated parts move away from where the tree was built, then a rebuild
(prefer-fast-build PLOC, what pt_animate(rebuild=1) runs) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=True)
g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(nestedDielectricsQuality=2, useFp16Types=1)); g.resize(W, H)
g.render(0, SPP); g.reset_accumulation(); st = g.render(0, SPP)
print("frame  0 (t = 0.0, tree as built: %s, %.1f ms): render %.2f ms, %.1f Mrays/s" % (g.bvh_info()["builderName"], g.build_stats()["buildMs"], st["gpuMilliseconds"], (st["extendRays"] + st["shadowRays"]) / st["gpuMilliseconds"] / 1e3), flush=True)
for f in range(1, 25):
    t = 0.1 * f
    g.animate(instances=scenes.animate_instances(sc, t), positions=scenes.animate_positions(sc, t), rebuild=False)
    b = g.build_stats(); g.reset_accumulation(); st = g.render(0, SPP)
    if f in (1, 2, 4, 8, 12, 16, 20, 24):
        print("frame %2d (t = %.1f): refit %.2f ms + light re-bake %.2f ms, render %.2f ms, %.1f Mrays/s" % (f, t, b["refitMs"], b["lightBakeMs"], st["gpuMilliseconds"], (st["extendRays"] + st["shadowRays"]) / st["gpuMilliseconds"] / 1e3), flush=True)
g.animate(instances=scenes.animate_instances(sc, 2.4), positions=scenes.animate_positions(sc, 2.4), rebuild=True)
b = g.build_stats(); g.reset_accumulation(); st = g.render(0, SPP)
print("rebuild at t = 2.4 (%s): build %.2f ms, render %.2f ms, %.1f Mrays/s" % (g.bvh_info()["builderName"], b["buildMs"], st["gpuMilliseconds"], (st["extendRays"] + st["shadowRays"]) / st["gpuMilliseconds"] / 1e3), flush=True)
