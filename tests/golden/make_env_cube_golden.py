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
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_cube_golden.npz"), **out)
