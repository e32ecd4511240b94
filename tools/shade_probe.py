"""developer tool: where does k_shade's time go? Serial-kernel timings of the bench workload with parts of the shading work made cheap:
tiny textures (every texel fetch hits a cache), one NEE candidate instead of five, NEE off. Not a parity run — the frames differ by design."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtxpt_amd as pt
from rtxpt_amd import scenes

W, H, SPP = 3840, 2160, 4
for tex in (1024, 16):
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=tex); sc["env_cube_dim"] = 2048
    camd = scenes.bridge_camera(W, H, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.resize(W, H); g.set_serial_kernels(True)
    for name, kw in (("default", {}), ("1 NEE candidate", dict(NEECandidateSamples=1)), ("NEE off", dict(NEEEnabled=0))):
        S = scenes.default_settings(useFp16Types=1, **kw); g.set_settings(S)
        g.reset_accumulation(); g.render(0, SPP); g.reset_accumulation(); st = g.render(0, SPP)
        print("tex %4d  %-16s  shade %.2f ms  extend %.2f  shadow %.2f  hits %d  shadow rays %d" % (tex, name, st["shadeKernelMs"], st["extendKernelMs"], st["shadowKernelMs"], st["hits"], st["shadowRays"]), flush=True)
    g.close()
