"""The cases of the stable-plane build pass shared by the fixture generator (tests/golden/make_stable_planes_golden.py), the CPU tests and the GPU tests."""
import numpy as np
from rtxpt_amd import scenes

W, H, SAMPLE = 64, 48, 5
SUBSAMPLES = 2      # noisy (fill) passes per frame: sample indices SAMPLE, SAMPLE + 1 (the build pass uses the first one's camera ray)
KEYS = ("header", "planes", "stable_radiance", "depth", "spec_hit_t", "motion_vectors", "throughput")


def cases():
    """name -> (lp16, settings overrides, stable_planes_params keywords[, scenes.stable_planes_zoo's auto_mv])"""
    return {"zoo_fp32": (False, {}, {}), "zoo_lp16": (True, {}, {}), "zoo_two_planes_no_psr": (False, {}, dict(active_planes=2, allow_psr=False)),
            "zoo_one_plane_depth4": (False, {}, dict(active_planes=1, max_vertex_depth=4)),
            # the automatic motion-vector block types of Bridge::loadSurface (AutoLow / AutoHigh, PathTracerBridgeDonut.hlsli:92-149, 704-716): curved surfaces whose pixel curvature
            # straddles the per-(pixel, vertex, sample) jittered thresholds
            "zoo_auto_mv": (False, {}, {}, "auto"), "zoo_auto_mv_lp16": (True, {}, {}, "auto")}


def setup(name):
    """(scene, bridge camera, settings, params) of a case: the camera moved a little since the last frame, so the motion vectors are not zero"""
    lp16, over, kw = cases()[name][:3]
    sc, cam = scenes.stable_planes_zoo(*cases()[name][3:])
    S = scenes.config_settings("C2")
    for k, v in over.items(): S[k] = v
    if lp16: S["useFp16Types"] = 1
    camd = scenes.bridge_camera(W, H, **cam)
    prev = dict(cam); prev["pos"] = tuple(np.asarray(cam["pos"]) + np.array([0.03, 0.01, 0.02]))
    prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), prev_world_to_clip=scenes.view_projection(W, H, **prev), sub_samples=SUBSAMPLES, **kw)
    return sc, camd, S, prm, lp16


def motion_cases():
    """Object motion (Bridge::loadSurface's prevPosW; pt_set_previous_pose / pt_set_motion_history): name -> the case of cases() it adds scenes.previous_pose() to. The camera moves as well."""
    return {"zoo_object_motion": "zoo_fp32", "zoo_object_motion_lp16": "zoo_lp16"}


def live_planes(out):
    """the records of the planes that exist (the others hold whatever the buffer held before: nothing writes them)"""
    hd = out["header"]; P = out["planes"].reshape(-1, 20)
    rows = []
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        for x, y in zip(xs.tolist(), ys.tolist()): rows.append(P[scenes.stable_planes_address(x, y, pl, W, H)])
    return np.array(rows, np.uint32)


# ---- edge cases (small frames whose sizes are no multiples of the 8 x 8 addressing tiles): (scene maker, settings overrides, params keywords, width, height)
def _empty_scene():
    b = scenes.SceneBuilder(); b.set_environment(scenes.sky_equirect(64, 32), color_multiplier=(1, 1, 1)); return b.finish(), scenes.stable_planes_zoo()[1]


def _inside_glass():
    sc, cam = scenes.stable_planes_zoo(); cam = dict(cam); cam["pos"] = (-0.35, 0.25, 1.1); cam["direction"] = (0.3, 0.05, -1.0); return sc, cam


def edge_cases():
    return {"empty_scene": (_empty_scene, {}, {}, 37, 21),                                        # every pixel is a miss on plane 0
            "no_env_cornell": (lambda: scenes.cornell_box("C1"), {}, {}, 37, 21),                # no environment, no delta lobes anywhere
            "bounce0": (scenes.stable_planes_zoo, dict(bounceCount=0), {}, 40, 24),              # the path stops at its first vertex
            "bounce1_depth1": (scenes.stable_planes_zoo, dict(bounceCount=1), dict(max_vertex_depth=1), 40, 24),
            "depth0": (scenes.stable_planes_zoo, {}, dict(max_vertex_depth=0), 40, 24),          # no delta exploration at all: the primary surface is the plane
            "planes0_clamped": (scenes.stable_planes_zoo, {}, dict(active_planes=0), 33, 17),    # an out-of-range plane count is clamped to 1
            "inside_glass": (_inside_glass, dict(nestedDielectricsQuality=1), {}, 40, 24),       # the camera sits in the nested glass cubes: rejected false hits from the first vertex on
            "diffuse_bounce0": (scenes.stable_planes_zoo, dict(diffuseBounceCount=0), {}, 40, 24)}


def edge_setup(name, sub_samples=2):
    make, over, kw, w, h = edge_cases()[name]
    sc, cam = make(); S = scenes.config_settings("C2")
    for k, v in over.items(): S[k] = v
    camd = scenes.bridge_camera(w, h, **cam)
    prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cam), sub_samples=sub_samples, **kw)
    return sc, camd, S, prm, w, h
