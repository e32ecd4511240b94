"""developer probe: streaming frames (pt_set_stream_paths) on one rank of an N-way tile-sharded C3 frame and on the full frame — ms per frame for every
(batches, paths in flight) of a sweep, with the frame's SHA-256 and ray counts checked against the non-streaming frame of the same context.
usage: tools/stream_probe.py [world ...]      environment: STREAM_PROBE_SWEEP="b:k,b:k,..." (k in units of 1024 paths; 0:0 = off), SHARD_PROBE_SPP, SHARD_PROBE_ANIMATE"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, int(os.environ.get("SHARD_PROBE_SPP", "4"))
ANIMATE = os.environ.get("SHARD_PROBE_ANIMATE") is not None
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=ANIMATE)
SETTINGS = scenes.default_settings(useFp16Types=1, nestedDielectricsQuality=2) if ANIMATE else scenes.default_settings(useFp16Types=1)
sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
camd = scenes.bridge_camera(W, H, **cam)
worlds = [int(x) for x in sys.argv[1:]] or [8]
SWEEP = [tuple(int(v) for v in (x.split(":") + ["0"])[:3]) for x in os.environ.get("STREAM_PROBE_SWEEP", "0:0,1:1024,1:2048,2:512,2:1024,2:1536,3:512,3:1024,4:512,4:768,0:0").split(",")]      # batches : kilo-paths in flight [: overlapped visibility rays]
REPS = int(os.environ.get("STREAM_PROBE_REPS", "4"))
print("stream probe: %dx%d, %d spp; sweep (batches:kilo-paths in flight) %s" % (W, H, SPP, SWEEP))
for world in worlds:
    g = pt.PathTracer(device=0, shard_rank=0, shard_count=world)
    g.set_scene(sc); g.set_camera(camd); g.set_settings(SETTINGS); g.resize(W, H)
    g.reset_accumulation(); g.render(0, SPP)
    ref = None
    for b, k, ov in SWEEP:
        g.set_stream_paths(k * 1024, b); g.reset_accumulation(); st = g.render(0, SPP)
        digest = hashlib.sha256(g.radiance().tobytes()).hexdigest()[:16]; rays = (st["extendRays"], st["shadowRays"])
        if ref is None: ref = (digest, rays)
        best = 1e9; tot = 0.0
        for _ in range(REPS):
            g.reset_accumulation(); torch.cuda.synchronize(); t0 = time.perf_counter(); st = g.render(0, SPP); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
            best = min(best, ms); tot += ms
        print("world %d rank 0  batches %d  in flight %5dk  overlap %d: %6.2f ms mean %6.2f ms best (gpu %6.2f), %2d passes, %d tail launches, %.1f Mrays  %s" % (
            world, b, k, ov, tot / REPS, best, st["gpuMilliseconds"], st["iterations"], st["tailLaunches"], sum(rays) / 1e6, "== off" if (digest, rays) == ref else "DIFFERS from the first entry: %s %s vs %s %s" % (digest, rays, ref[0], ref[1])))
        sys.stdout.flush()
    del g
