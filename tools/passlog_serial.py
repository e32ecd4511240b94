import os, sys
sys.path.insert(0, os.getcwd())
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H); g.set_serial_kernels(True)
g.reset_accumulation(); g.render(0, SPP); g.reset_accumulation(); st = g.render(0, SPP)
