"""The HIP path's leaf functions against the REFERENCE TEXT directly (run with -m gpu): tests/golden/refpin_hlsl_golden.npz holds inputs and outputs of
functions compiled verbatim from the reference's .hlsli files (oracle/refpin/hlsl_tu.py, made by tests/golden/make_refpin_hlsl_golden.py in the build
container). The device evaluates the product's own implementation of each through pt_probe and must reproduce the reference's output bit for bit.
No oracle code runs here: the product's headers and the oracle's are textual twins, so "GPU == oracle" alone cannot catch a typo they share — this can."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refpin_hlsl_golden.npz")
PIN_NAMES = ["evalFresnelSchlick", "evalFresnelSchlick3", "evalFresnelDielectric", "evalNdfGGX", "evalPdfGGX_BVNDF", "sampleGGX_BVNDF", "evalLambdaGGX", "evalMaskingSmithGGXCorrelated",
             "ndir_to_oct_equal_area_unorm", "oct_to_ndir_equal_area_unorm", "sample_disk", "sample_disk_concentric", "sample_cosine_hemisphere_concentric", "perp_stark", "ComputeRayOrigin",
             "FastSqrt", "FastACos", "ComputeRayConeSpreadAngleExpansionByScatterPDF", "ComputeNewScatterFireflyFilterK", "FireflyFilter", "FireflyFilterShort", "ComputeLowGrazingAngleFalloff"]


def _same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.fixture(scope="module")
def tracer():
    import rtxpt_amd as pt
    return pt.PathTracer(test_hooks=True)      # pt_probe: the tests' build of the library (include/mi355pt_testhooks.h)


@pytest.mark.parametrize("fn", range(len(PIN_NAMES)), ids=PIN_NAMES)
def test_device_leaf_function_matches_reference_text(tracer, fn):
    g = np.load(GOLDEN)
    a, want = g["in_" + PIN_NAMES[fn]], g["out_" + PIN_NAMES[fn]]
    rows = np.zeros((a.shape[0], 9), np.float32); rows[:, 0] = fn; rows[:, 1:1 + a.shape[1]] = a
    got = tracer.probe(5, rows, (a.shape[0], 4))[:, :want.shape[1]]
    ok = _same(got, want)
    assert ok.all(), "%s: %d of %d rows differ from the reference text; first: in=%s device=%s reference=%s" % (
        PIN_NAMES[fn], int((~ok).any(1).sum()), len(a), a[(~ok).any(1)][0], got[(~ok).any(1)][0], want[(~ok).any(1)][0])


def test_device_whole_bsdf_matches_reference_text(tracer):
    """FalcorBSDF eval / evalPdf / getLobes / sample of BxDF.hlsli:55-970 (both DiffuseBrdf settings) as compiled from the reference text, against pt_bsdf.h on the device."""
    g = np.load(GOLDEN)
    rows, want = g["bsdf_in"], g["bsdf_out"]
    P = np.zeros((rows.shape[0], 24), np.float32); P[:, :23] = rows
    got = tracer.probe(3, P, (rows.shape[0], 10))
    ok = _same(got, want).all(1)
    assert ok.all(), "%d of %d BSDF cases differ from the reference text; first: case=%s device=%s reference=%s" % (int((~ok).sum()), len(rows), rows[~ok][0], got[~ok][0], want[~ok][0])


def test_device_sample_streams_match_reference_text(tracer):
    """SampleGeneratorVertexBase + the hash-Owen-Sobol / uniform sequence generators + sampleNext1D (NoiseAndSequences / StatelessSampleGenerators) on the device."""
    g = np.load(GOLDEN)
    cases, want = g["stream_in"], g["stream_out"]
    got = tracer.probe(2, cases, (cases.shape[0], 8))
    for i in range(cases.shape[0]):
        k = min(int(cases[i, 5]), 8)
        assert np.array_equal(got[i, :k].view(np.uint32), want[i, :k].view(np.uint32)), (cases[i], got[i], want[i])


@pytest.mark.parametrize("kind", range(5), ids=["PackLightColor", "TriangleLight_Store", "CalcSample_GetPower", "TriangleLight_PdfForMIS", "NDirToOctUnorm32"])
def test_device_polymorphic_lights_match_reference_text(tracer, kind):
    """PolymorphicLight.hlsli / LightShaping.hlsli (triangle, sphere with spot shaping, environment quad) compiled from the reference text, against pt_lights.h on the device."""
    g = np.load(GOLDEN)
    a, want = g["light%d_in" % kind], g["light%d_out" % kind]
    rows = np.zeros((a.shape[0], 19), np.uint32); rows[:, 0] = kind; rows[:, 1:1 + a.shape[1]] = a
    got = tracer.probe(6, rows, (a.shape[0], 12), out_dtype=np.uint32)[:, :want.shape[1]]
    wf, gf = want.view(np.float32), got.view(np.float32)
    ok = (got == want) | (np.isnan(wf) & np.isnan(gf))
    assert ok.all(), "light probe %d: %d of %d rows differ from the reference text; first: in=%s device=%s reference=%s" % (kind, int((~ok).any(1).sum()), len(a), a[(~ok).any(1)][0], got[(~ok).any(1)][0], want[(~ok).any(1)][0])


def test_device_half_operators_are_ieee_binary16(tracer):
    """LPOps<true> on the device (the half-typed operators of the reference's default build: a float rounded to binary16 after every operation, conversions by
    v_cvt_f16_f32) against numpy's float16 arithmetic on 60 000 operand sets per operator, including the binary16 denormal range, values around the overflow
    boundary and exact ties. numpy evaluates a half operation in float and rounds once, which is the correctly rounded half result (24 >= 2 * 11 + 2)."""
    rng = np.random.default_rng(0x16F10A7)
    n = 60000
    def operands():
        mag = np.where(rng.random(n) < 0.35, 10.0 ** rng.uniform(-9, -3, n), 10.0 ** rng.uniform(-3, 5, n))
        return (mag * np.where(rng.random(n) < 0.3, -1.0, 1.0)).astype(np.float32)
    h = lambda x: x.astype(np.float16)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        # lpfloat(<float expression>): the fp32 result is rounded to fp32 first and then to binary16 — two roundings, exactly what HLSL's float -> float16_t cast
        # of a float expression does. (A mixed-precision multiply-add that rounds the exact product straight to binary16 differs once in ~8000 products.)
        for op, name in ((7, "lpfloat(a*b)"), (8, "lpfloat(half*b)"), (9, "lpfloat(a+b)"), (10, "lpfloat(a/b)")):
            a, b = operands(), operands()
            fa = h(a).astype(np.float32) if op == 8 else a
            want = [fa * b, fa * b, a + b, a / b][op - 7].astype(np.float16).astype(np.float32)
            rows = np.stack([np.full(n, op, np.float32), a, b, b], 1)
            got = tracer.probe(7, rows, (n,))
            ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            assert ok.all(), "%s: %d of %d differ; first: a=%r b=%r device=%r numpy=%r" % (name, int((~ok).sum()), n, a[~ok][0], b[~ok][0], got[~ok][0], want[~ok][0])
        for op, name in enumerate(("round", "add", "sub", "mul", "div", "lerp", "average3")):
            a, b, c = operands(), operands(), operands()
            if op == 5: c = rng.random(n).astype(np.float32)
            if op == 4: a[: n // 3] = 1.0                 # 1 / x: the pattern a compiler likes to turn into a reciprocal instruction (one ulp, not correctly rounded)
            ha, hb, hc = h(a), h(b), h(c)
            want = [ha, ha + hb, ha - hb, ha * hb, ha / hb, ha + (hb - ha) * hc, ((ha + hb) + hc) / np.float16(3.0)][op].astype(np.float32)
            rows = np.stack([np.full(n, op, np.float32), a, b, c], 1)
            got = tracer.probe(7, rows, (n,))
            ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            assert ok.all(), "%s: %d of %d differ; first: a=%r b=%r c=%r device=%r numpy=%r" % (name, int((~ok).sum()), n, a[~ok][0], b[~ok][0], c[~ok][0], got[~ok][0], want[~ok][0])


DEVICE_PATH_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "device_path_golden.npz")


def _pin_case(name, lp16):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pin_scenes
    return (pin_scenes.cases_lp16() if lp16 else pin_scenes.cases())[name]


@pytest.mark.parametrize("lp16", [False, True], ids=["fp32", "lp16"])
@pytest.mark.parametrize("name", ["c2", "bistro_like", "bistro_like_material_zoo", "c2_spec_gloss"])
def test_device_load_surface_matches_reference_text(name, lp16):
    """Bridge::loadSurface on the device — since round 3 through the flat 128-byte ShadeTri record of the hit primitive (pt_scene.h, k_shade_tris) — against
    the outputs of PathTracerBridgeDonut.hlsli:612-853 compiled from the reference (tests/golden/make_device_path_golden.py): 2 500 random hits per scene,
    45 words each, both builds of the lp types. No oracle code runs."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    g = np.load(DEVICE_PATH_GOLDEN); tag = "surface_%s_%s" % (name, "lp16" if lp16 else "fp32")
    prims, rows, want = g[tag + "_prims"], g[tag + "_rows"], g[tag + "_out"]
    make, S, w, h, first, n = _pin_case(name, lp16)
    sc, cam = make()
    t = pt.PathTracer(test_hooks=True); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
    rows8 = np.zeros((len(prims), 8), np.float32); rows8[:, 0] = prims.view(np.float32); rows8[:, 1:] = rows
    got = t.probe(8, rows8, (len(prims), 45), out_dtype=np.uint32)
    bad = (got != want).any(1)
    assert not bad.any(), "%d of %d surfaces differ from the reference text; first: hit %d, words %s" % (int(bad.sum()), len(prims), int(np.flatnonzero(bad)[0]), np.flatnonzero(got[bad][0] != want[bad][0]))
    t.close()


@pytest.mark.parametrize("name", ["bistro_like", "c2_exclude_from_nee"])
def test_device_alpha_test_matches_reference_text(name):
    """The traversal's alpha test on the device — since round 3 against per-texture alpha planes (byte opacities) through a self-contained AlphaRec — against
    Bridge::AlphaTest / AlphaTestVisibilityRay (PathTracerBridgeDonut.hlsli:929-989) compiled from the reference: 20 000 random candidates per scene, the answer for
    scatter rays and for visibility rays (ExcludeFromNEE geometry lets those through)."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    g = np.load(DEVICE_PATH_GOLDEN)
    prims, uv, want = g["alpha_%s_prims" % name], g["alpha_%s_uv" % name], g["alpha_%s_out" % name]
    make, S, w, h, first, n = _pin_case(name, False)
    sc, cam = make()
    t = pt.PathTracer(test_hooks=True); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
    rows = np.zeros((len(prims), 3), np.uint32); rows[:, 0] = prims; rows[:, 1:] = uv.view(np.uint32)
    got = t.probe(10, rows, (len(prims), 2), out_dtype=np.uint32)
    bad = (got != want).any(1)
    assert not bad.any(), "%d of %d alpha tests differ from the reference text; first: primitive %d uv %s device %s reference %s" % (int(bad.sum()), len(prims), int(prims[bad][0]), uv[bad][0], got[bad][0], want[bad][0])
    assert 0 < int(want[:, 1].sum()) < len(prims)
    t.close()
