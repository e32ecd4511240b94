"""Generates tests/golden/env_cube_golden.npz: environment cubes baked by the REFERENCE'S EnvMapBaker.hlsl text (BaseLayerCS, MIPReduceCS, GenerateTexel,
SampleSource, ComputeLightContribution, CubemapGetDirectionFor, CubemapTexelSolidAngle4: compiled from /root/reference by oracle/refpin/hlsl_tu.py
--integrator over the stand-in bindings of oracle/refpin/hlsl_envbake_stubs.h) for the cases of tests/pin_scenes.env_cube_cases().
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_env_cube_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

out = {}
S = scenes.default_settings()
for name, sc in pin_scenes.env_cube_cases().items():
    o = ptref.Oracle(reference_integrator=True, settings=S); o.set_scene(sc)
    cube, dim, levels = o.env_cube(reference=True)
    out[name] = cube; out[name + "_dim"] = np.array([dim, levels], np.uint32)
    out[name + "_importance64"] = o.env_importance(64, reference=True)        # BuildMIPDescentImportanceMapCS of EnvMapImportanceSamplingBaker.hlsl over that cube (RGBA16F store)
    print(name, dim, levels, cube.shape)
# BC6UCompress.hlsl's EncodeP1 on blocks of several kinds (smooth HDR, flat, black, one varying channel, wide range, ramps, outliers, large values)
rng = np.random.default_rng(3); blocks = []
for k in range(1600):
    kind = k % 8; base = rng.uniform(0, 1, 3) * 10 ** rng.uniform(-3, 3)
    if kind == 0: t = base[None, :] * rng.uniform(0.5, 1.5, (16, 3))
    elif kind == 1: t = np.repeat(base[None, :], 16, 0)
    elif kind == 2: t = np.zeros((16, 3))
    elif kind == 3: t = np.repeat(base[None, :], 16, 0); t[:, 1] = rng.uniform(0, 5, 16)
    elif kind == 4: t = 10 ** rng.uniform(-4, 4.5, (16, 3))
    elif kind == 5: t = base[None, :] * (1 + np.linspace(0, 1, 16)[:, None] * rng.uniform(0, 3))
    elif kind == 6: t = np.repeat(base[None, :], 16, 0); t[rng.integers(0, 16)] *= rng.uniform(1.5, 100)
    else: t = rng.uniform(0, 60000, (16, 3))
    blocks.append(np.clip(t, 0, 65504))
T = np.array(blocks, np.float32).astype(np.float16).astype(np.float32)
out["bc6_texels"] = T; out["bc6_blocks"] = ptref.bc6_encode(T, reference=True)
out["bc6_blocks_quality"] = ptref.bc6_encode(T, reference=True, quality=True)       # CSMain with QUALITY 1: EncodeP1, EvaluateP2Pattern over the 32 partitions, EncodeP2Pattern
print("bc6 blocks", out["bc6_blocks"].shape)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_cube_golden.npz"), **out)
