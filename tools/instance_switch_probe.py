"""developer probe (a library built with -DT8_PROBE_INSTANCE_SWITCHES, MI355PT_LIB): per closest-hit ray of the C3 frame, how many of the visited leaves belong to another instance than the
leaf visited before — a lower bound of the instance entries a two-level (TLAS / BLAS) traversal would pay for — next to node and leaf visits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H); g.set_serial_kernels(True)
g.set_counters(True); g.reset_accumulation(); st = g.render(0, SPP)
n = st["extendRays"]
print("instances %d, triangles %d; per closest-hit ray: node visits %.2f, leaf visits %.2f, instance switches %.2f (wave iterations per ray %.3f)" % (
    len(sc["instances"]), g.scene_info()["triangles"], st["nodeVisitsExtend"] / n, st["leafVisitsExtend"] / n, st["extendEvents"][4] / n, st["waveItersExtend"] / n))
