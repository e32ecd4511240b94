"""pt_convert_light (the host side of analytic lights) against the reference's own LightsBaker::ConvertLight with its helpers packLightColor, the host
NDirToOctUnorm32 and the truncating fp32ToFp16 (Rtxpt/Lighting/LightsBaker.cpp:384-556 compiled as they stand over Donut light stand-ins,
oracle/refpin/light_stubs.inc), and the records it produces through pt_set_lights on the GPU."""
import ctypes

import numpy as np
import pytest

import rtxpt_amd as pt


def _random_light(rng):
    kind = ["point", "spot"][int(rng.integers(0, 2))]
    d = rng.normal(size=3) * rng.choice([0.01, 1.0, 37.0])
    args = dict(position=rng.uniform(-50, 50, 3), color=rng.uniform(0, 1, 3) * rng.choice([0.0, 1.0, 1.0, 1.0]), intensity=float(np.exp(rng.uniform(np.log(1e-3), np.log(1e5)))),
                radius=float(rng.choice([0.0, 0.01, 0.05, 0.3, 2.5])) if kind == "point" else float(rng.choice([0.01, 0.05, 0.3, 2.5])), direction=d)
    if kind == "spot":
        outer = float(rng.uniform(1, 89)) * float(rng.choice([1.0, 1.0, -1.0]))
        args.update(inner_angle=float(rng.uniform(0, abs(outer) * 1.2)), outer_angle=outer if rng.random() > 0.05 else 0.0)
    return kind, args


def _desc_words(kind, a):
    f = lambda x: np.asarray(x, np.float32).reshape(-1)
    return np.concatenate([np.array([{"point": 0, "spot": 1}[kind]], np.uint32), f(a["position"]).view(np.uint32), f(a["direction"]).view(np.uint32), f(a["color"]).view(np.uint32),
                           f([a["intensity"], a["radius"], a.get("inner_angle", 0.0), a.get("outer_angle", 0.0)]).view(np.uint32)])


@pytest.mark.parametrize("seed", range(8))
def test_convert_light_matches_reference_text(seed):
    from oracle import ptref
    rng = np.random.default_rng(0x11A + seed)
    checked = 0
    for _ in range(500):
        kind, a = _random_light(rng)
        ref = ptref.reference_convert_light(_desc_words(kind, a))
        if ref is None:
            pytest.skip("librefpin_mat.so not available (no /root/reference on this machine)")
        if ref == "assert":
            continue
        base, ex = pt.convert_light(kind, **a)
        assert np.array_equal(base, ref[0]) and np.array_equal(ex, ref[1]), (kind, a, base, ref[0], ex, ref[1])
        checked += 1
    assert checked > 400


def test_convert_light_known_values():
    base, ex = pt.convert_light("point", (1.0, 2.0, 3.0), (1.0, 1.0, 1.0), intensity=np.pi * 0.25, radius=0.5)      # radiance = intensity / (pi r^2) = 1
    assert (base[3] >> 24) & 0xF == 0 and base[3] & 0xFFFFFF == 0xFFFFFF and base[6] == 0x3800 and not base[3] & (1 << 28)
    assert np.array_equal(base[:3].view(np.float32), np.float32([1, 2, 3])) and not ex.any()
    base, ex = pt.convert_light("spot", (0, 0, 0), (1, 0, 0), 10.0, 0.1, direction=(0, -2, 0), inner_angle=10.0, outer_angle=-30.0)
    assert base[3] & (1 << 28) and base[3] & (1 << 30) and ex[1] != 0 and (ex[2] & 0xFFFF) == 0x3AED       # cos 30 deg, truncated to fp16
    d = pt.PtAnalyticLightDesc(); d.type = 7                       # unknown light type
    base = np.zeros(8, np.uint32); ex = np.zeros(4, np.uint32)
    assert pt.load_library().pt_convert_light(ctypes.byref(d), base.ctypes.data_as(ctypes.c_void_p), ex.ctypes.data_as(ctypes.c_void_p)) == 1      # PT_ERROR_INVALID_ARGUMENT


@pytest.mark.gpu
def test_converted_lights_render_like_the_oracle():
    """Records from pt_convert_light through pt_set_lights: HIP frame == oracle frame on a Cornell box lit by two converted spheres and a converted spot."""
    from rtxpt_amd import scenes
    from oracle import ptref
    sc, cam = scenes.cornell_box("C2")
    recs = [pt.convert_light("point", (0.15, 0.45, 0.2), (1.0, 0.8, 0.4), 3.0, 0.02), pt.convert_light("point", (0.42, 0.3, 0.1), (0.3, 0.5, 1.0), 1.5, 0.03),
            pt.convert_light("spot", (0.28, 0.5, 0.3), (1.0, 1.0, 1.0), 8.0, 0.015, direction=(0.1, -1.0, 0.05), inner_angle=15.0, outer_angle=40.0)]
    sc = dict(sc); sc["lights"] = (np.stack([r[0] for r in recs]), np.stack([r[1] for r in recs]))
    S = scenes.default_settings(); w, h = 80, 45
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h); g.render(0, 2)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(0, 2)
    assert np.array_equal(g.radiance().view(np.uint32), o.radiance().view(np.uint32))
    plain = ptref.Oracle(); plain.set_scene(scenes.cornell_box("C2")[0]); plain.set_camera(camd); plain.set_settings(S); plain.resize(w, h); plain.render(0, 2)
    assert o.radiance()[..., :3].mean() > 1.02 * plain.radiance()[..., :3].mean()
