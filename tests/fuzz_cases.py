"""Seeded random scene / camera / settings cases at 1280x720 (the recipe of tests/test_gpu_fuzz.py at a size where rare branches are met): shared by
tests/golden/make_fuzz_hd_golden.py (the REFERENCE'S integrator text renders them) and tests/test_gpu_parity_hd.py (the device)."""
import math
import numpy as np
from rtxpt_amd import scenes

W, H = 1280, 720
SEEDS = list(range(101, 141))


def case(seed, lp16):
    """(scene in its pose, camera struct, settings, first sample, sample count, (instances, positions) to animate to or None)"""
    rng = np.random.default_rng(0xBEEF + seed)
    animated = bool(rng.integers(0, 2))
    sc, cam = scenes.bistro_like(scale=float(rng.uniform(0.004, 0.03)), seed=scenes.SEED_BASE + 300 + seed, tex_size=int(rng.choice([32, 64, 128])), animated=animated)
    yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-0.5, 0.6)
    cam = dict(cam, pos=(float(rng.uniform(5, 110)), float(rng.uniform(0.5, 18.0)), float(rng.uniform(10.0, 30.0))),
               direction=(math.cos(yaw) * math.cos(pitch), math.sin(pitch), math.sin(yaw) * math.cos(pitch)), fov_y=float(rng.uniform(0.5, 1.4)),
               aperture_radius=float(rng.choice([0.0, 0.02])), focal_distance=float(rng.uniform(3.0, 30.0)))
    S = scenes.default_settings(bounceCount=int(rng.integers(1, 9)), diffuseBounceCount=int(rng.integers(1, 9)), NEEType=int(rng.integers(0, 2)),
                                NEECandidateSamples=int(rng.integers(1, 8)), NEEFullSamples=int(rng.choice([1, 1, 1, 2, 4])), enableRussianRoulette=int(rng.integers(0, 2)),
                                nestedDielectricsQuality=int(rng.integers(0, 3)), fireflyFilterThreshold=float(rng.choice([0.0, 0.5])),
                                texLODBias=float(rng.uniform(-2.0, 1.0)), enableLDSamplerForBSDF=int(rng.integers(0, 2)), diffuseBrdf=int(rng.choice([0, 2])),
                                envMapDiffuseSampleMIPLevel=float(rng.choice([0.0, 2.0])), perPixelJitterAAScale=float(rng.choice([0.0, 1.0])))
    if lp16: S["useFp16Types"] = 1; S["fireflyFilterThreshold"] = float(rng.choice([0.3, 1.0, 4.0]))
    first, count = int(rng.integers(0, 50)), int(rng.integers(2, 5))
    pose = None
    if animated:
        t = float(rng.uniform(0, 3)); pose = (scenes.animate_instances(sc, t), scenes.animate_positions(sc, t))
    return sc, scenes.bridge_camera(W, H, **cam), S, first, count, pose


SP_SEEDS = list(range(201, 231))


def stable_planes_case(seed):
    """a random stable-plane frame at W x H: (scene, camera struct, settings, params, lp16, previous pose or None, sample index, fill sub-samples)"""
    rng = np.random.default_rng(0xFACE + seed)
    zoo = bool(rng.integers(0, 3))                                   # two of three on the delta-tree zoo, the others on the small street scene with nested-dielectric props
    if zoo:
        sc, cam = scenes.stable_planes_zoo()
        cam = dict(cam, pos=tuple(np.asarray(cam["pos"], np.float64) + rng.uniform(-0.35, 0.35, 3)), fov_y=float(rng.uniform(0.6, 1.3)))
        d = np.asarray(cam["direction"], np.float64) + rng.uniform(-0.25, 0.25, 3); cam["direction"] = tuple(d / np.linalg.norm(d))
    else:
        sc, cam = scenes.bistro_like(scale=float(rng.uniform(0.006, 0.02)), seed=scenes.SEED_BASE + 500 + seed, tex_size=64, animated=True)
    lp16 = bool(rng.integers(0, 2))
    S = scenes.config_settings("C2")
    for k, v in dict(bounceCount=int(rng.integers(0, 9)), diffuseBounceCount=int(rng.integers(0, 5)), nestedDielectricsQuality=int(rng.integers(0, 3)), enableRussianRoulette=int(rng.integers(0, 2)),
                     fireflyFilterThreshold=float(rng.choice([0.0, 0.7, 3.0])), NEEEnabled=int(rng.integers(0, 4) > 0), NEECandidateSamples=int(rng.integers(1, 7)), NEEType=int(rng.integers(0, 2)),
                     texLODBias=float(rng.uniform(-1.0, 1.0)), useFp16Types=int(lp16)).items(): S[k] = v
    prev = dict(cam); prev["pos"] = tuple(np.asarray(cam["pos"], np.float64) + rng.uniform(-0.05, 0.05, 3))
    subs = int(rng.integers(1, 4))
    prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=subs,
                                      active_planes=int(rng.integers(1, 4)), max_vertex_depth=int(rng.integers(0, 10)), allow_psr=bool(rng.integers(0, 2)))
    prev_pose = scenes.previous_pose(sc, seed) if rng.integers(0, 2) else None
    return sc, scenes.bridge_camera(W, H, **cam), S, prm, lp16, prev_pose, int(rng.integers(0, 40)), subs
