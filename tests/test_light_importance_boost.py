"""The frustum term (and, for NEE-AT runs, the intensity-delta term) of LightsBaker's ImportanceBooster — SURVEY.md §8 row N3's "frustum boost", on by default in the reference
for every NEEType (LightsBaker.h:245-249).

  * the planes: LightsBaker::UpdateFrustumConsts' C++ text (LightsBaker.cpp:886-909, compiled over Donut vector stand-ins) against the restatement, on random view-projection matrices;
  * the boost: LightsBaker.hlsl's ImportanceBooster + DistanceFromFrustum text against the restatement on the lights of a baked scene, with and without last frame's weights;
  * committed values of both for machines without /root/reference; and what the boost does to the proxy table."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rtxpt_amd import scenes
from oracle import ptref

HAVE_REF = os.path.isdir("/root/reference/Rtxpt/Shaders")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "light_importance_boost_golden.npz")


def matrices(n=64, seed=7):
    rng = np.random.default_rng(seed); out = []
    for k in range(n):
        pos = rng.normal(size=3) * 10; d = rng.normal(size=3); up = (0.0, 1.0, 0.0) if k % 3 else tuple(rng.normal(size=3))
        out.append(scenes.view_projection(int(rng.integers(16, 4000)), int(rng.integers(16, 3000)), pos, d, up, float(rng.uniform(0.2, 2.5)), near_z=float(10 ** rng.uniform(-3, 0))))
    return np.stack(out)


def scene_lights():
    sc, cam = scenes.bistro_like(scale=0.02, tex_size=64)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(scenes.bridge_camera(96, 54, **cam)); o.set_settings(scenes.default_settings()); o.resize(96, 54); o.L.ptref_prepare(o.h)
    L = o.lights(); o.close()
    return sc, cam, np.concatenate([L["lights"], L["lightsEx"]], 1)


def test_planes_match_reference_cpp_text():
    if not HAVE_REF: pytest.skip("no /root/reference on this machine")
    for m in matrices():
        a, b = ptref.frustum_planes(m), ptref.frustum_planes(m, reference=True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (m, a, b)


def test_boost_matches_reference_hlsl_text():
    if not HAVE_REF: pytest.skip("no /root/reference on this machine")
    sc, cam, lights = scene_lights()
    rng = np.random.default_rng(3); w = rng.uniform(0, 5, len(lights)).astype(np.float32); w[::7] = 0; hist = (w * rng.uniform(0.5, 1.5, len(w))).astype(np.float32)
    for k, m in enumerate([scenes.view_projection(96, 54, **cam)] + list(matrices(6, seed=11))):
        pl = ptref.frustum_planes(m)
        for h, dm in ((None, 0.0), (hist, 64.0)):
            a, b = ptref.importance_boost(lights, pl, 8.0, 5.0, w, h, dm), ptref.importance_boost(lights, pl, 8.0, 5.0, w, h, dm, reference=True)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "matrix %d: %d of %d weights differ" % (k, int((a.view(np.uint32) != b.view(np.uint32)).sum()), len(a))
    assert np.array_equal(ptref.importance_boost(lights, pl, 0.0, 5.0, w), w)                 # multiplier 0: off


def test_golden_values():
    """planes of 64 matrices and the boosted weights of the first 512 lights of the scene under its own camera, as the reference text gave them (made where it exists)"""
    sc, cam, lights = scene_lights(); M = matrices(); m0 = scenes.view_projection(96, 54, **cam)
    rng = np.random.default_rng(5); w = rng.uniform(0, 5, 512).astype(np.float32)
    if HAVE_REF and not os.path.exists(GOLDEN):
        np.savez_compressed(GOLDEN, planes=np.stack([ptref.frustum_planes(m, reference=True) for m in M]), boosted=ptref.importance_boost(lights[:512], ptref.frustum_planes(m0, reference=True), 8.0, 5.0, w, reference=True))
    g = np.load(GOLDEN)
    assert np.array_equal(np.stack([ptref.frustum_planes(m) for m in M]).view(np.uint32), g["planes"].view(np.uint32))
    got = ptref.importance_boost(lights[:512], ptref.frustum_planes(m0), 8.0, 5.0, w)
    assert np.array_equal(got.view(np.uint32), g["boosted"].view(np.uint32))
    r = got[w > 0] / w[w > 0]
    assert np.isclose(r, 5.0).any() and (r >= 1).all() and r.max() <= 9.0 + 1e-5      # environment quads: 1 + 8 * 0.5; local lights between 1 (far outside) and 9 (inside)


def test_boost_moves_proxies_towards_the_view():
    sc, cam, lights = scene_lights()
    def counters(boost):
        o = ptref.Oracle(); o.set_scene(sc); o.set_camera(scenes.bridge_camera(96, 54, **cam)); o.set_settings(scenes.default_settings()); o.resize(96, 54)
        if boost: o.set_light_importance_boost(scenes.view_projection(96, 54, **cam))
        o.L.ptref_prepare(o.h); c = o.lights()["proxyCounters"].copy(); o.close(); return c
    plain, boosted = counters(False), counters(True)
    pl = ptref.frustum_planes(scenes.view_projection(96, 54, **cam))
    centre = lights[:, 0:3].view(np.float32)
    inside = ((centre @ pl[:, :3].T - pl[:, 3]) > 0).all(1) & (((lights[:, 3] >> 24) & 0xF) == 1)      # emissive triangles inside the frustum (PolymorphicLightType kTriangle = 1)
    assert inside.sum() > 10 and boosted[inside].sum() > plain[inside].sum() and not np.array_equal(plain, boosted)


def test_proxy_counts_match_reference_hlsl_text():
    """ComputeProxyCounts (LightsBaker.hlsl:880-948): the budget formula and, for NEE-AT, the lerp towards last frame's usage counts — the text dispatched in groups of 128
    threads with its barrier, against build_light_proxies"""
    if not HAVE_REF: pytest.skip("no /root/reference on this machine")
    rng = np.random.default_rng(21)
    for n in (1, 127, 128, 129, 5000):
        w = (rng.uniform(0, 1, n) ** 6 * 50).astype(np.float32); w[rng.random(n) < 0.2] = 0
        if not (w > 0).any(): w[0] = 1.0
        for typ in (0, 1, 2):
            a, b = ptref.proxy_counts(w, sampling_type=typ), ptref.proxy_counts(w, sampling_type=typ, reference=True)
            assert np.array_equal(a[0], b[0]) and a[1] == b[1], (n, typ)
        pixels = 96 * 54; usage = np.zeros(n + 1, np.uint32)
        picks = rng.choice(n, size=pixels // 2, p=(w + 1e-3) / (w + 1e-3).sum()); np.add.at(usage, picks, 1); usage[n] = 64 * 12 * 7 - usage[:n].sum()
        for g in (0.0, 0.3, 0.75, 0.95):
            a, b = ptref.proxy_counts(w, usage, 64 * 12 * 7, g, 2), ptref.proxy_counts(w, usage, 64 * 12 * 7, g, 2, reference=True)
            assert np.array_equal(a[0], b[0]) and a[1] == b[1], (n, g)
        assert not np.array_equal(ptref.proxy_counts(w, usage, 64 * 12 * 7, 0.75, 2)[0], ptref.proxy_counts(w, sampling_type=2)[0]) or n == 1
