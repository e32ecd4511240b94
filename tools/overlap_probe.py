"""developer probe: does running two half-frame renders concurrently (two contexts = two HIP streams, tile shards 0/1 of 2) beat one full-frame
render? (kernel-level overlap of the VALU-bound traversal of one half with the latency-bound shading of the other)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtxpt_amd as pt
from rtxpt_amd import scenes

W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
camd = scenes.bridge_camera(W, H, **cam)
S = scenes.default_settings()


def make(rank, world):
    g = pt.PathTracer(device=0, shard_rank=rank, shard_count=world)
    g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
    g.reset_accumulation(); g.render(0, SPP)
    return g


full = make(0, 1)
t0 = time.perf_counter()
for _ in range(3):
    full.reset_accumulation(); st = full.render(0, SPP)
torch.cuda.synchronize(); t_full = (time.perf_counter() - t0) / 3
print("one context, full frame: %.1f ms  (rays %.1fM)" % (t_full * 1e3, (st["extendRays"] + st["shadowRays"]) / 1e6))
for world in (2, 3, 4):
    gs = [make(r, world) for r in range(world)]
    def work(g):
        for _ in range(3):
            g.reset_accumulation(); g.render(0, SPP)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(g,)) for g in gs]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t_c = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for g in gs: work(g)
    torch.cuda.synchronize(); t_s = (time.perf_counter() - t0) / 3
    print("%d contexts (tile shards): concurrent %.1f ms, back-to-back %.1f ms" % (world, t_c * 1e3, t_s * 1e3))
    del gs
