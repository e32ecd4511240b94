import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
for world in (8, 1):
    g = pt.PathTracer(device=0, shard_rank=0, shard_count=world); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H)
    for _ in range(2): g.reset_accumulation(); g.render(0, SPP)
    tr = tn = tg = 0.0; N = 8
    for _ in range(N):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.reset_accumulation(); torch.cuda.synchronize(); t1 = time.perf_counter(); st = g.render(0, SPP); t2 = time.perf_counter()
        tr += t1 - t0; tn += t2 - t1; tg += st["gpuMilliseconds"]
    print("world %d: reset (+ sync) %.3f ms, render call %.3f ms, of which GPU events %.3f ms -> host / un-timed %.3f ms" % (world, tr / N * 1e3, tn / N * 1e3, tg / N, (tn / N * 1e3) - tg / N))
    del g
