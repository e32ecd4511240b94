"""developer probe: renders rank 0 of an N-way tile-sharded C3 frame a few times (run it under rocprofv3 --kernel-trace --stats to see where a
small per-rank frame spends its time). usage: python tools/rank_profile.py [world] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, int(os.environ.get("SHARD_PROBE_SPP", "4"))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
sc["env_cube_dim"] = 2048; sc["env_compression"] = 1      # as bench.py: EnvMapBaker's cube for an image source, BC6H on (the reference's D3D12 default)
g = pt.PathTracer(device=0, shard_rank=0, shard_count=world)
g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H)
if os.environ.get("RANK_SERIAL"): g.set_serial_kernels(True)
g.reset_accumulation(); g.render(0, SPP)
t0 = time.perf_counter()
for _ in range(frames):
    g.reset_accumulation(); st = g.render(0, SPP)
torch.cuda.synchronize()
print("world %d rank 0: %.2f ms/frame, gpu %.2f ms, passes %d, spans ext %.2f shade %.2f shadow %.2f ms" % (
    world, (time.perf_counter() - t0) / frames * 1e3, st["gpuMilliseconds"], st["iterations"], st["extendKernelMs"], st["shadeKernelMs"], st["shadowKernelMs"]))
