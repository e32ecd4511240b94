"""developer probe: the pass log (MI355PT_PASS_LOG) of one rank of an N-way sharded C3 frame for a given tail threshold. usage: tail_log_probe.py <world> <tail paths>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
world, tail = int(sys.argv[1]), int(sys.argv[2])
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
g = pt.PathTracer(device=0, shard_rank=0, shard_count=world)
g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H); g.set_tail_paths(tail)
g.render(0, SPP); g.reset_accumulation()
os.environ["MI355PT_PASS_LOG"] = "1"
st = g.render(0, SPP)
print("frame %.2f ms, %d passes, %d tail launches" % (st["gpuMilliseconds"], st["iterations"], st["tailLaunches"]))
