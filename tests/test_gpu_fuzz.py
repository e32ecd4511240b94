"""Randomised GPU-vs-oracle parity (run with -m gpu): seeded scenes, cameras and settings outside the hand-picked configurations.
Every case must be bit-identical (radiance words and ray counts); the seeds are fixed so that a failure is reproducible."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
import os
FUZZ_SEEDS = [int(x) for x in os.environ.get("PT_FUZZ_SEEDS", "11,12,13,14,15,16,17,18").split(",")]


@pytest.mark.parametrize("lp16", [False, True], ids=["fp32", "lp16"])
@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_random_scene_camera_settings(seed, lp16):
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    from oracle import ptref
    rng = np.random.default_rng(0xF00D + seed)
    animated = bool(rng.integers(0, 2))
    sc, cam = scenes.bistro_like(scale=float(rng.uniform(0.004, 0.012)), seed=scenes.SEED_BASE + 100 + seed, tex_size=int(rng.choice([32, 64, 128])), animated=animated)
    yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-0.5, 0.6)
    cam = dict(cam, pos=(float(rng.uniform(5, 110)), float(rng.uniform(0.5, 18.0)), float(rng.uniform(10.0, 30.0))),
               direction=(math.cos(yaw) * math.cos(pitch), math.sin(pitch), math.sin(yaw) * math.cos(pitch)), fov_y=float(rng.uniform(0.5, 1.4)),
               aperture_radius=float(rng.choice([0.0, 0.02])), focal_distance=float(rng.uniform(3.0, 30.0)))
    S = scenes.default_settings(bounceCount=int(rng.integers(1, 9)), diffuseBounceCount=int(rng.integers(1, 9)), NEEType=int(rng.integers(0, 2)),
                                NEECandidateSamples=int(rng.integers(1, 8)), enableRussianRoulette=int(rng.integers(0, 2)),
                                nestedDielectricsQuality=int(rng.integers(0, 3)), fireflyFilterThreshold=float(rng.choice([0.0, 0.5])),
                                texLODBias=float(rng.uniform(-2.0, 1.0)), enableLDSamplerForBSDF=int(rng.integers(0, 2)), diffuseBrdf=int(rng.choice([0, 2])))
    if lp16:        # the reference's default build of the lp types; the firefly filter (all-half arithmetic in that build) always on
        S["useFp16Types"] = 1; S["fireflyFilterThreshold"] = float(rng.choice([0.3, 1.0, 4.0]))
    w, h = int(rng.integers(40, 260)), int(rng.integers(30, 150))
    first, count = int(rng.integers(0, 50)), int(rng.integers(1, 4))
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h)
    o = ptref.Oracle(lp16=lp16); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h)
    if animated:
        t = float(rng.uniform(0, 3))
        inst, pos = scenes.animate_instances(sc, t), scenes.animate_positions(sc, t)
        g.animate(instances=inst, positions=pos, rebuild=bool(rng.integers(0, 2)))
        sc_t = dict(sc); sc_t["positions"] = pos
        o = ptref.Oracle(lp16=lp16); o.set_scene(sc_t); o.set_instances(inst); o.set_camera(camd); o.set_settings(S); o.resize(w, h)
    st = g.render(first, count); o.render(first, count)
    a, b = g.radiance(), o.radiance()
    assert not np.isnan(a).any()
    c = o.counters()
    assert (st["extendRays"], st["shadowRays"]) == (c["extendRays"], c["shadowRays"])
    bad = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
    assert bad == 0, "%d of %d pixels differ (seed %d)" % (bad, w * h, seed)
