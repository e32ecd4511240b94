import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
def run(sc, cam, label):
    sc = dict(sc); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H)
    g.set_serial_kernels(True); g.reset_accumulation(); g.render(0, SPP); g.reset_accumulation(); s = g.render(0, SPP)
    print("%-44s frame %.2f extend %.2f shade %.2f shadow %.2f | hits %d shadow rays %d extend rays %d" % (label, s["gpuMilliseconds"], s["extendKernelMs"], s["shadeKernelMs"], s["shadowKernelMs"], s["hits"], s["shadowRays"], s["extendRays"]), flush=True)
    g.close()
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
run(sc, cam, "as generated (64 materials)")
m = sc["materials"].copy()
print("material dtype", m.dtype if hasattr(m, "dtype") else type(m), getattr(m, "shape", None))
sc2 = dict(sc); m2 = m.copy(); m2[:] = m[0]; sc2["materials"] = m2
run(sc2, cam, "every material = material 0")
sc3 = dict(sc); m3 = m.copy(); m3[:] = m[5]; sc3["materials"] = m3
run(sc3, cam, "every material = material 5")
