"""The realtime mode's frame with everything coupled (Rtxpt/Sample.cpp:2438-2516): LightsBaker::UpdateBegin -> the stable-plane build pass -> LightsBaker::UpdateEnd on THAT frame's depth
and screen-space motion vectors -> the fill passes, which sample the tile tables the baker just made and fill the feedback reservoirs the next frame's baker reads. Shared by the fixture
generator (tests/golden/make_realtime_golden.py), the CPU tests (oracle == the reference's text, live and committed) and the GPU tests (device == the committed reference-text run).
The camera moves between the frames, so the motion vectors are not zero and pixels do get disoccluded."""
import numpy as np
from rtxpt_amd import scenes

KEYS = ("table", "jitter", "counters", "fbw", "fbc", "noisy", "spec_hit_t", "depth", "motion_vectors", "header")


def _zoo_settings():
    S = scenes.config_settings("C2"); S["NEEType"] = 2; return S


def cases():
    """name -> (scene maker, settings, w, h, frames, sub-samples per frame, camera step per frame, stable-plane keywords)"""
    bl = lambda: scenes.bistro_like(scale=0.02, tex_size=128)
    d = scenes.default_settings
    return {
        "bistro_like_realtime": (bl, d(NEEType=2), 96, 54, 4, 1, (0.35, 0.02, -0.2), {}),                                  # the reference's defaults, one sub-sample
        "bistro_like_realtime_lp16_2sub": (bl, d(NEEType=2, useFp16Types=1), 64, 36, 3, 2, (0.6, 0.0, 0.25), {}),          # binary16 lp types, two sub-samples feeding one reservoir plane
        "zoo_realtime": (scenes.stable_planes_zoo, _zoo_settings(), 64, 48, 3, 1, (0.03, 0.01, 0.02), {}),      # delta trees (mirror, glass): the dominant plane's depth and motion
        # the automatic motion-vector block types (AutoLow / AutoHigh, PathTracerBridgeDonut.hlsli:704-716) on curved mirrors and panes: where the planes stop following the delta path
        "zoo_auto_mv_realtime": (lambda: scenes.stable_planes_zoo("auto"), _zoo_settings(), 64, 48, 3, 1, (0.03, 0.01, 0.02), {}),
        # the scene moves as well (C5's: rigid props, a deforming mesh, nested-dielectric props, emissive triangles that are re-baked every frame): object motion in the motion vectors
        # (Bridge::loadSurface's prevPosW), so the baker's Reproject follows the objects; "anim_dt" is the scene time per frame (popped before the keywords reach stable_planes_params)
        "bistro_like_c5_realtime_animated": (lambda: scenes.bistro_like(scale=0.01, tex_size=64, animated=True), d(NEEType=2, nestedDielectricsQuality=2), 96, 54, 3, 1, (0.2, 0.0, -0.1), {"anim_dt": 0.45}),
    }


def poses(sc, kw, frames):
    """[(instances, positions)] per frame for an animated case, else None; frame f's previous pose is frame f - 1's, frame 0's the scene as uploaded"""
    if "anim_dt" not in kw: return None
    return [(scenes.animate_instances(sc, kw["anim_dt"] * f), scenes.animate_positions(sc, kw["anim_dt"] * f)) for f in range(frames)]


def camera(cam, step, f):
    c = dict(cam); c["pos"] = tuple(np.asarray(cam["pos"], np.float64) + np.asarray(step, np.float64) * f); return c


def live_noisy(frame, w, h):
    """the noisy radiance | specular average words of the planes that exist, in (plane, y, x) order"""
    hd = frame["header"]; P = frame["planes"].reshape(-1, 20); rows = []
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        for x, y in zip(xs.tolist(), ys.tolist()): rows.append(P[scenes.stable_planes_address(x, y, pl, w, h), 16:18])
    return np.array(rows, np.uint32).reshape(-1, 2)


def settings_for(name): return cases()[name][1]


def run(name, begin, build, end, fill, read, set_camera, pose=None):
    """Drives one case through callbacks (oracle, reference text or device): returns {"<name>_<key><frame>": array}. pose(current, previous): called before every frame of an
    animated case with the (instances, positions) pairs of the frame and of the one before."""
    make, _, w, h, frames, subs, step, kw = cases()[name]
    sc, cam = make()
    P = poses(sc, kw, frames); kw = {k: v for k, v in kw.items() if k != "anim_dt"}
    out = {}
    for f in range(frames):
        if P is not None: pose(P[f], P[f - 1] if f else (sc["instances"], sc["positions"]))
        cur, prev = camera(cam, step, f), camera(cam, step, max(f - 1, 0))
        prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cur), prev_world_to_clip=scenes.view_projection(w, h, **prev), sub_samples=subs, **kw)
        set_camera(scenes.bridge_camera(w, h, **cur))
        begin()
        frame = build(f * subs, prm)
        end(frame)
        for s in range(subs): fill(f * subs + s, prm, frame)
        rec = read(frame)
        for k in KEYS:
            if k in rec and rec[k] is not None: out["%s_%s%d" % (name, k, f)] = np.asarray(rec[k]).copy()
    return out
