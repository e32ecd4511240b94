"""Oracle pinning, third layer: the oracle's INTEGRATOR against the reference's integrator source text.

oracle/refpin/hlsl_tu.py --integrator compiles PathTracer.hlsli and its whole include closure (HandleHit, HandleMiss, GenerateScatterRay, nested
dielectrics, NEE with its reservoir and MIS, Russian roulette, the packed PathState, LightSampler, PolymorphicLight, EnvMap, BxDF ...) from
/root/reference; oracle/refpin/hlsl_pt_wrappers.inc serves its `Bridge` from the oracle's scene services and runs the raygen loop. The frame that
comes out must equal the oracle's own frame bit for bit, with the same ray counts.

  * test_oracle_matches_reference_integrator_golden: committed frames (tests/golden/reference_integrator_golden.npz) — runs everywhere.
  * test_oracle_matches_live_reference_integrator: the libraries built here, where /root/reference exists (one per shader-macro combination)."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_integrator_golden.npz")
GOLDEN_LP16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_integrator_golden_lp16.npz")
CASES = pin_scenes.cases()
CASES_LP16 = pin_scenes.cases_lp16()


def _oracle_frame(name, reference=False, lp16=False):
    make, S, w, h, first, n = (CASES_LP16 if lp16 else CASES)[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16) if reference else ptref.Oracle(lp16=lp16)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
    c = o.counters()
    return o.radiance(), (c["extendRays"], c["shadowRays"])


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_integrator_golden(name):
    g = np.load(GOLDEN)
    got, rays = _oracle_frame(name)
    want = g[name]
    assert got.shape == want.shape
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert tuple(int(v) for v in g[name + "_rays"]) == rays
    assert want[..., :3].max() > 0


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_live_reference_integrator(name):
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine: the reference-text integrator cannot be built here")
    want, rays_ref = _oracle_frame(name, reference=True)
    got, rays = _oracle_frame(name)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and rays == rays_ref


@pytest.mark.parametrize("name", list(CASES_LP16))
def test_oracle_lp16_matches_reference_integrator_golden(name):
    """The reference's DEFAULT build — lp types in binary16 (SampleUI.h:182, Sample.cpp:1035, Utils.hlsli:28-48) — restated by libptref_lp16.so, against frames
    of the reference's integrator text compiled with RTXPT_LP_TYPES_USE_16BIT_PRECISION=1 over hlsl_shim.h's binary16 type (one rounding per operation,
    HLSL's scalar typing rules: half op half -> half, literals adapt, a float operand promotes)."""
    g = np.load(GOLDEN_LP16)
    got, rays = _oracle_frame(name, lp16=True)
    want = g[name]
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s (lp16): %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert tuple(int(v) for v in g[name + "_rays"]) == rays
    if name in CASES:       # and the two builds do differ: the fixture is not the fp32 frame under another name
        assert not np.array_equal(want, np.load(GOLDEN)[name]) or name == "c1"


@pytest.mark.parametrize("name", list(CASES_LP16))
def test_oracle_lp16_matches_live_reference_integrator(name):
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine: the reference-text integrator cannot be built here")
    want, rays_ref = _oracle_frame(name, reference=True, lp16=True)
    got, rays = _oracle_frame(name, lp16=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and rays == rays_ref


WIDE = pin_scenes.wide_cases()
GOLDEN_WIDE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_integrator_golden_wide.npz")


def _wide_frame(name, reference):
    make, S, w, h, first, n = WIDE[name]
    lp16 = bool(int(S["useFp16Types"]))
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16) if reference else ptref.Oracle(lp16=lp16)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
    c = o.counters()
    return o.radiance(), (c["extendRays"], c["shadowRays"])


@pytest.mark.parametrize("name", list(WIDE))
def test_oracle_matches_wide_reference_integrator_golden(name):
    """256 x 144 x 4 samples per pin family and lp build (pin_scenes.wide_cases): 147 456 paths of the reference's integrator text per frame, so the late bounces,
    the Russian-roulette survivors and the deep nested-dielectric stacks are compared thousands of times, not a handful."""
    g = np.load(GOLDEN_WIDE)
    got, rays = _wide_frame(name, reference=False)
    want = g[name]
    assert got.shape == want.shape == (144, 256, 4)
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert tuple(int(v) for v in g[name + "_rays"]) == rays
    assert rays[0] > 2 * 256 * 144 * 4 and np.isfinite(want).all()       # deeper than two bounces on average


@pytest.mark.parametrize("name", list(WIDE))
def test_wide_golden_is_the_live_reference_integrator(name):
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine: the reference-text integrator cannot be built here")
    g = np.load(GOLDEN_WIDE)
    want, rays = _wide_frame(name, reference=True)
    assert np.array_equal(g[name].view(np.uint32), want.view(np.uint32)) and tuple(int(v) for v in g[name + "_rays"]) == rays
    if name.endswith("_lp16"): assert not np.array_equal(g[name], g[name[:-5]])       # the two builds differ


XL = pin_scenes.xl_cases()
GOLDEN_XL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_integrator_golden_xl.npz")


@pytest.mark.parametrize("name", list(XL))
def test_oracle_matches_xl_reference_integrator_golden(name):
    """1280 x 720 x 4 samples (3.7 M paths, the bench configuration's settings) of the reference's integrator text: the oracle's whole frame by SHA-256, every sixteenth row pixel for
    pixel, the ray counts."""
    g = np.load(GOLDEN_XL)
    make, S, w, h, first, n = XL[name]; lp16 = bool(int(S["useFp16Types"]))
    sc, cam = make()
    o = ptref.Oracle(lp16=lp16); o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
    rad = o.radiance(); c = o.counters()
    rows = rad[::pin_scenes.XL_ROW_STEP]
    bad = (rows.view(np.uint32) != g[name + "_rows"].view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels of the kept rows differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert np.array_equal(pin_scenes.frame_digest(rad), g[name + "_sha256"]), "%s: the frame's digest differs (a pixel outside the kept rows)" % name
    assert (c["extendRays"], c["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])


@pytest.mark.parametrize("lp16", [False, True], ids=["fp32", "lp16"])
@pytest.mark.parametrize("name", ["c2", "c2_mirrored_room", "bistro_like", "bistro_like_material_zoo", "bistro_like_c5", "c2_spec_gloss", "bistro_like_spec_gloss", "c2_sphere_light_proxy"])
def test_load_surface_matches_reference_text(name, lp16):
    """Bridge::loadSurface and everything RTXPT-side below it (getGeometryFromHit, sampleGeometryMaterialRTXPT, EvaluateSceneMaterialRTXPT,
    ApplyNormalMapRTXPT, createTextureSampler + ray-cone LOD, computeTangentSpace / adjustShadingNormal, emissive light index) compiled from
    PathTracerBridgeDonut.hlsli, against the oracle's loadSurface: ShadingData + StandardBSDFData + interior IoR + light index, 45 words per hit."""
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine")
    make, S, w, h, first, n = (CASES_LP16 if lp16 else CASES)[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16); o.set_scene(sc); o.set_settings(S); o.resize(8, 8)
    o.L.ptref_num_tris.restype = __import__("ctypes").c_uint32
    nt = o.L.ptref_num_tris(o.h)
    rng = np.random.default_rng(0x5F + len(name)); k = 20000
    prims = rng.integers(0, nt, k); u = rng.uniform(0, 1, k); v = rng.uniform(0, 1, k) * (1 - u)
    d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    R, Q = ptref.surface_probe(o, prims, np.column_stack([u, v, d, rng.uniform(0, 0.5, k), rng.uniform(0, 0.01, k)]))
    bad = (R != Q).any(1)
    assert not bad.any(), "%d of %d surfaces differ (first: hit %d, words %s)" % (int(bad.sum()), k, int(np.flatnonzero(bad)[0]), np.flatnonzero(R[bad][0] != Q[bad][0]))
    assert len(np.unique(R[:, 23])) > 1 or name == "c2"       # several materials were hit


@pytest.mark.parametrize("name", ["bistro_like", "c2_exclude_from_nee"])
def test_alpha_tests_match_reference_text(name):
    """AlphaTestImpl + Bridge::AlphaTest / AlphaTestVisibilityRay (PathTracerBridgeDonut.hlsli:929-989) against the candidate filter of the oracle's traversal:
    alpha-tested foliage (bistro-like) and ExcludeFromNEE geometry (Cornell boxes)."""
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine")
    import ctypes
    make, S, w, h, first, n = CASES[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S); o.set_scene(sc); o.set_settings(S); o.resize(8, 8)
    o.L.ptref_num_tris.restype = ctypes.c_uint32
    nt = o.L.ptref_num_tris(o.h)
    rng = np.random.default_rng(0xA1FA); k = 60000
    prims = rng.integers(0, nt, k).astype(np.uint32); u = rng.uniform(0, 1, k); v = rng.uniform(0, 1, k) * (1 - u)
    uv = np.ascontiguousarray(np.column_stack([u, v]), np.float32); out = np.zeros((k, 4), np.uint32)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    o.L.refpt_alpha_probe(o.h, ctypes.c_uint32(k), vp(prims), vp(uv), vp(out))
    assert np.array_equal(out[:, 0], out[:, 2]) and np.array_equal(out[:, 1], out[:, 3])
    assert 0 < out[:, 1].sum() < k and (out[:, 1] <= out[:, 0]).all()        # some candidates are rejected; visibility rays reject at least as many
    if name == "c2_exclude_from_nee": assert out[:, 0].all() and not out[:, 1].all()      # excluded boxes: solid for scatter rays, transparent for shadow rays


@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26, 27, 28, 29, 30])
def test_random_scene_camera_settings_against_reference_text(seed):
    """The fuzz generator of tests/test_gpu_fuzz.py (random bistro-like scene, camera incl. depth of field, bounce limits, NEE type and candidate count,
    RR, nested-dielectrics quality, firefly filter, LOD bias, LD sampler, diffuse model), oracle integrator vs the reference's integrator text."""
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine")
    import math
    rng = np.random.default_rng(0xF00D + seed)
    sc, cam = scenes.bistro_like(scale=float(rng.uniform(0.004, 0.012)), seed=scenes.SEED_BASE + 100 + seed, tex_size=int(rng.choice([32, 64, 128])), animated=bool(rng.integers(0, 2)))
    yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-0.5, 0.6)
    cam = dict(cam, pos=(float(rng.uniform(5, 110)), float(rng.uniform(0.5, 18.0)), float(rng.uniform(10.0, 30.0))),
               direction=(math.cos(yaw) * math.cos(pitch), math.sin(pitch), math.sin(yaw) * math.cos(pitch)), fov_y=float(rng.uniform(0.5, 1.4)),
               aperture_radius=float(rng.choice([0.0, 0.02])), focal_distance=float(rng.uniform(3.0, 30.0)))
    S = scenes.default_settings(bounceCount=int(rng.integers(1, 9)), diffuseBounceCount=int(rng.integers(1, 9)), NEEType=int(rng.integers(0, 2)),
                                NEECandidateSamples=int(rng.integers(1, 8)), NEEFullSamples=int(rng.choice([1, 1, 2])), enableRussianRoulette=int(rng.integers(0, 2)),
                                nestedDielectricsQuality=int(rng.integers(0, 3)), fireflyFilterThreshold=float(rng.choice([0.0, 0.5])),
                                texLODBias=float(rng.uniform(-2.0, 1.0)), enableLDSamplerForBSDF=int(rng.integers(0, 2)), diffuseBrdf=int(rng.choice([0, 2])))
    lp16 = bool(seed % 2)                                 # odd seeds: the reference's default build of the lp types (binary16), with the firefly filter on
    if lp16: S["useFp16Types"] = 1; S["fireflyFilterThreshold"] = float(rng.choice([0.3, 1.0, 4.0]))
    w, h = int(rng.integers(40, 120)), int(rng.integers(30, 70))
    first, count = int(rng.integers(0, 50)), int(rng.integers(1, 3))
    camd = scenes.bridge_camera(w, h, **cam)
    frames = []
    for reference in (False, True):
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16) if reference else ptref.Oracle(lp16=lp16)
        o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(first, count)
        c = o.counters(); frames.append((o.radiance(), c["extendRays"], c["shadowRays"]))
    (a, ea, sa), (b, eb, sb) = frames
    bad = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
    assert bad == 0 and (ea, sa) == (eb, sb), "%d of %d pixels differ, rays %s vs %s (seed %d, variant %s)" % (bad, w * h, (ea, sa), (eb, sb), seed, ptref.pt_variant(S))


def test_lp16_reference_build_deviation():
    """How far the reference's DEFAULT build (lp types in 16 bits, RTXPT_LP_TYPES_USE_16BIT_PRECISION 1 / UseFp16Types) is from its fp32 build (both are
    restated by the oracle and the HIP path, PtSettings.useFp16Types): the reference's integrator text compiled both ways (hlsl_shim.h float16_t: a float rounded to binary16 after every
    operation), same scene services, same samples. Recorded in DESIGN.md 6; this test keeps the 16-bit build compiling and the numbers honest."""
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine")
    out = {}
    for name, spp in (("c2", 8), ("bistro_like", 8)):
        make, S, w, h, first, n = CASES[name]
        sc, cam = make(); camd = scenes.bridge_camera(w, h, **cam)
        frames = []
        for lp16 in (False, True):
            o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16)
            o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(0, spp); frames.append(o.radiance()[..., :3])
        a, b = frames
        out[name] = (float(np.linalg.norm(a - b) / np.linalg.norm(a)), float(b.mean() / a.mean()))
    assert 0 < out["c2"][0] < 5e-3 and abs(out["c2"][1] - 1) < 1e-3, out            # Cornell: 2e-3 relative L2 at 8 spp, mean equal to 1e-4
    # bistro-like: 1.4e-2 relative L2 at 8 spp, mean radiance equal to 2e-4. (Round 1 reported 5e-2 and a 1.4 % darker image: that was the shim resolving
    # `half op float` to a half operation — e.g. F0 = ((ior - 1.f) / (ior + 1.f))^2 computed in binary16 — where HLSL promotes to float.)
    assert 1e-3 < out["bistro_like"][0] < 0.05 and abs(out["bistro_like"][1] - 1) < 2e-3, out
