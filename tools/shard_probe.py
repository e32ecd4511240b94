"""developer probe: per-rank frame time of an N-way tile-sharded C3 frame (what each GPU does in the multi-GPU bench): max and mean over ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, int(os.environ.get("SHARD_PROBE_SPP", "4"))
ANIMATE = os.environ.get("SHARD_PROBE_ANIMATE") is not None      # C5: the animated scene, nested dielectrics quality 2, pt_animate (refit + light re-bake) inside every timed frame
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024, animated=ANIMATE)
SETTINGS = scenes.default_settings(useFp16Types=1, nestedDielectricsQuality=2) if ANIMATE else scenes.default_settings(useFp16Types=1)
RANGES = (None if os.environ.get("SHARD_PROBE_ANIMATE") == "full" else scenes.animated_vertex_ranges(sc)) if ANIMATE else None      # SHARD_PROBE_ANIMATE=full: pt_animate with the whole vertex array
POSES = [(scenes.animate_instances(sc, 0.1 * f), scenes.animate_positions(sc, 0.1 * f)) for f in range(1, 4)] if ANIMATE else []
sc["env_cube_dim"] = 2048; sc["env_compression"] = 1      # as bench.py: EnvMapBaker's cube for an image source, BC6H on (the reference's D3D12 default)
camd = scenes.bridge_camera(W, H, **cam)
FRAMES = int(os.environ.get("SHARD_PROBE_FRAMES", "6"))      # timed frames per rank
MAXR = int(os.environ.get("SHARD_PROBE_RANKS", "0"))          # > 0: time only this many evenly spaced ranks per world size (every rank costs a scene build)
worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
TAILS = [int(x) for x in os.environ.get("SHARD_PROBE_TAILS", "").split(",") if x]      # tail-kernel thresholds to A/B on one context per rank (pt_set_tail_paths); empty: the product default only
print("shard probe: %dx%d, %d spp%s" % (W, H, SPP, ", animated (C5): pt_animate inside every timed frame" if ANIMATE else ""))
base = None
for world in worlds if not TAILS else []:
    times, rays = [], []
    ranks = list(range(world)) if not MAXR or world <= MAXR else sorted({int(round(i * (world - 1) / (MAXR - 1))) for i in range(MAXR)}) if MAXR > 1 else [0]
    for rank in ranks:
        g = pt.PathTracer(device=0, shard_rank=rank, shard_count=world)
        g.set_scene(sc); g.set_camera(camd); g.set_settings(SETTINGS); g.resize(W, H)
        for _ in range(2): g.reset_accumulation(); g.render(0, SPP)      # (first calls allocate; round 6: two warm-up frames and FRAMES timed ones instead of 1 + 2 — a 12 ms frame needs more than two samples)
        if ANIMATE:
            for inst, pos in POSES: g.animate(instances=inst, positions=pos, rebuild=False, vertex_ranges=RANGES)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in range(FRAMES):
            if ANIMATE: g.animate(instances=POSES[f % len(POSES)][0], positions=POSES[f % len(POSES)][1], rebuild=False, vertex_ranges=RANGES)
            g.reset_accumulation(); st = g.render(0, SPP)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / FRAMES); rays.append(st["extendRays"] + st["shadowRays"])
        del g
    print("   ranks timed:", ranks); print("   per rank ms:", " ".join("%.1f" % (t * 1e3) for t in times)); print("   per rank Mrays:", " ".join("%.1f" % (r / 1e6) for r in rays))
    base = base or max(times)
    print("world %d: frame time per rank max %.1f mean %.1f min %.1f ms; rays per rank max %.1fM min %.1fM; speed-up vs 1 GPU %.2f (efficiency %.2f)" % (
        world, max(times) * 1e3, sum(times) / len(times) * 1e3, min(times) * 1e3, max(rays) / 1e6, min(rays) / 1e6, base / max(times), base / max(times) / world))

for world in worlds if TAILS else []:
    ranks = list(range(world)) if not MAXR or world <= MAXR else sorted({int(round(i * (world - 1) / (MAXR - 1))) for i in range(MAXR)}) if MAXR > 1 else [0]
    for rank in ranks:
        g = pt.PathTracer(device=0, shard_rank=rank, shard_count=world)
        g.set_scene(sc); g.set_camera(camd); g.set_settings(SETTINGS); g.resize(W, H)
        g.reset_accumulation(); g.render(0, SPP)
        for rep in range(2):
            for tail in TAILS:
                g.set_tail_paths(tail); g.reset_accumulation(); g.render(0, SPP)
                t0 = time.perf_counter()
                for _ in range(3):
                    g.reset_accumulation(); st = g.render(0, SPP)
                torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
                print("world %d rank %d tail %7d: %.2f ms per frame (gpu %.2f), %d tail launches, %d passes, %.1f Mrays" % (world, rank, tail, ms, st["gpuMilliseconds"], st["tailLaunches"], st["iterations"], (st["extendRays"] + st["shadowRays"]) / 1e6))
        del g
