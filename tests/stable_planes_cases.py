"""The cases of the stable-plane build pass shared by the fixture generator (tests/golden/make_stable_planes_golden.py), the CPU tests and the GPU tests."""
import numpy as np
from rtxpt_amd import scenes

W, H, SAMPLE = 64, 48, 5
SUBSAMPLES = 2      # noisy (fill) passes per frame: sample indices SAMPLE, SAMPLE + 1 (the build pass uses the first one's camera ray)
KEYS = ("header", "planes", "stable_radiance", "depth", "spec_hit_t", "motion_vectors", "throughput")


def cases():
    """name -> (lp16, settings overrides, stable_planes_params keywords)"""
    return {"zoo_fp32": (False, {}, {}), "zoo_lp16": (True, {}, {}), "zoo_two_planes_no_psr": (False, {}, dict(active_planes=2, allow_psr=False)),
            "zoo_one_plane_depth4": (False, {}, dict(active_planes=1, max_vertex_depth=4))}


def setup(name):
    """(scene, bridge camera, settings, params) of a case: the camera moved a little since the last frame, so the motion vectors are not zero"""
    lp16, over, kw = cases()[name]
    sc, cam = scenes.stable_planes_zoo()
    S = scenes.config_settings("C2")
    for k, v in over.items(): S[k] = v
    if lp16: S["useFp16Types"] = 1
    camd = scenes.bridge_camera(W, H, **cam)
    prev = dict(cam); prev["pos"] = tuple(np.asarray(cam["pos"]) + np.array([0.03, 0.01, 0.02]))
    prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=SUBSAMPLES, **kw)
    return sc, camd, S, prm, lp16


def live_planes(out):
    """the records of the planes that exist (the others hold whatever the buffer held before: nothing writes them)"""
    hd = out["header"]; P = out["planes"].reshape(-1, 20)
    rows = []
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        for x, y in zip(xs.tolist(), ys.tolist()): rows.append(P[scenes.stable_planes_address(x, y, pl, W, H)])
    return np.array(rows, np.uint32)
