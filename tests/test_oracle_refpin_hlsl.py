"""Oracle pinning, second layer: functions of the reference's .hlsli files compiled VERBATIM (oracle/refpin/hlsl_tu.py streams them from
/root/reference through the HLSL-vocabulary shim hlsl_shim.h) against the oracle's restatement of the same functions, bit for bit.

  * test_restatement_matches_reference_golden: the committed fixture (tests/golden/refpin_hlsl_golden.npz, made by make_refpin_hlsl_golden.py
    from the reference text) — runs everywhere, also where /root/reference does not exist.
  * test_restatement_matches_live_reference: 20 000 fresh rows per function against the live library, where it can be built.

The shim maps what HLSL leaves to the implementation (transcendentals, dot/normalize summation order, mad, pow(x,5), fp16 conversion) to the
oracle's arithmetic contract, so a mismatch here is a restatement error: wrong operation order, constant, branch or clamp."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ptref
import pin_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refpin_hlsl_golden.npz")


def _same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))


@pytest.mark.parametrize("fn", range(len(ptref.PIN_NAMES)), ids=ptref.PIN_NAMES)
def test_restatement_matches_reference_golden(fn):
    name = ptref.PIN_NAMES[fn]
    g = np.load(GOLDEN)
    assert ("in_" + name) in g.files, "regenerate tests/golden/refpin_hlsl_golden.npz (make_refpin_hlsl_golden.py)"
    a, want = g["in_" + name], g["out_" + name]
    assert np.array_equal(a, pin_inputs.rows(name, 256, 0x5EED0100 + fn)), "input generator drifted from the fixture"
    got = ptref.pin_call(fn, a)
    ok = _same(got, want)
    assert ok.all(), "%s: %d of %d rows differ, first: in=%s oracle=%s reference=%s" % (name, int((~ok).any(1).sum()), len(a), a[(~ok).any(1)][0], got[(~ok).any(1)][0], want[(~ok).any(1)][0])


@pytest.mark.parametrize("fn", range(len(ptref.PIN_NAMES)), ids=ptref.PIN_NAMES)
def test_restatement_matches_live_reference(fn):
    if ptref.refpin_hlsl() is None:
        pytest.skip("librefpin_hlsl.so not available (no /root/reference on this machine)")
    name = ptref.PIN_NAMES[fn]
    a = pin_inputs.rows(name, 20000, 0xA11CE + fn)
    got, want = ptref.pin_call(fn, a), ptref.pin_call(fn, a, reference=True)
    ok = _same(got, want)
    assert ok.all(), "%s: %d of %d rows differ, first: in=%s oracle=%s reference=%s" % (name, int((~ok).any(1).sum()), len(a), a[(~ok).any(1)][0], got[(~ok).any(1)][0], want[(~ok).any(1)][0])


def _bsdf_report(rows, got, want):
    ok = _same(got, want).all(1)
    bad = np.flatnonzero(~ok)
    return ok, "whole BSDF: %d of %d cases differ; first: case=%s oracle=%s reference=%s" % (len(bad), len(rows), rows[bad[0]] if len(bad) else None, got[bad[0]] if len(bad) else None, want[bad[0]] if len(bad) else None)


def test_whole_bsdf_matches_reference_golden():
    """FalcorBSDF eval / evalPdf / getLobes / sample of BxDF.hlsli:55-970 (both DiffuseBrdf settings), compiled from the reference text, vs the oracle's StandardBSDF."""
    g = np.load(GOLDEN)
    rows, want = g["bsdf_in"], g["bsdf_out"]
    assert np.array_equal(rows, pin_inputs.bsdf_cases(3000, 0x5EED0200)), "input generator drifted from the fixture"
    ok, msg = _bsdf_report(rows, ptref.bsdf_probe(rows), want)
    assert ok.all(), msg
    assert (want[rows[:, 22] == 1][:, 9] == 1).sum() > 500 and (want[rows[:, 22] == 0][:, 4] > 0).sum() > 300      # the fixture exercises valid samples and non-zero pdfs


def test_whole_bsdf_matches_live_reference():
    if ptref.refpin_hlsl() is None:
        pytest.skip("librefpin_hlsl.so not available (no /root/reference on this machine)")
    rows = pin_inputs.bsdf_cases(40000, 0xB5DF)
    ok, msg = _bsdf_report(rows, ptref.bsdf_probe(rows), ptref.bsdf_probe(rows, reference=True))
    assert ok.all(), msg


def test_sample_generators_match_reference_golden():
    """SampleGeneratorVertexBase / SampleSequenceGenerator (Burley hash-Owen-Sobol) / UniformSampleSequenceGenerator / sampleNext1D of
    StatelessSampleGenerators.hlsli + SampleGenerators.hlsli, compiled from the reference text: integer streams, bit for bit."""
    g = np.load(GOLDEN)
    cases, want = g["stream_in"], g["stream_out"]
    assert np.array_equal(cases, pin_inputs.stream_cases(4000, 0x5EED0300)), "input generator drifted from the fixture"
    got = ptref.sample_streams(cases)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%d streams differ" % int((got != want).any(1).sum())
    assert ((want >= 0) & (want < 1)).all() and len(np.unique(want[:, 0])) > 3900


def test_sample_generators_match_live_reference():
    if ptref.refpin_hlsl() is None:
        pytest.skip("librefpin_hlsl.so not available (no /root/reference on this machine)")
    cases = pin_inputs.stream_cases(60000, 0x57EA)
    got, want = ptref.sample_streams(cases), ptref.sample_streams(cases, reference=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%d streams differ" % int((got != want).any(1).sum())


LIGHT_KINDS = {0: "PackColor/UnpackColor", 1: "TriangleLight::Store", 2: "PolymorphicLight::CalcSample+GetPower (triangle, sphere+shaping, environment quad)",
               3: "TriangleLight::CalcSolidAnglePdfForMIS", 4: "NDirToOctUnorm32/OctToNDirUnorm32"}


def _words_equal(got, want):
    g, w = got.view(np.float32), want.view(np.float32)
    return (got == want) | (np.isnan(g) & np.isnan(w))


@pytest.mark.parametrize("kind", sorted(LIGHT_KINDS), ids=lambda k: "kind%d" % k)
def test_lights_match_reference_golden(kind):
    """PolymorphicLight.hlsli:93-259, 399-520, 562-640, 643-676, 762-792, LightShaping.hlsli:16-99, Utils.hlsli:127-152, Packing.hlsli:17-51 compiled from the reference text."""
    g = np.load(GOLDEN)
    words, want = g["light%d_in" % kind], g["light%d_out" % kind]
    got = ptref.light_probe(kind, words)
    ok = _words_equal(got, want)
    bad = np.flatnonzero(~ok.all(1))
    assert ok.all(), "%s: %d of %d rows differ; first row %d: oracle=%s reference=%s" % (LIGHT_KINDS[kind], len(bad), len(words), bad[0], got[bad[0]], want[bad[0]])
    if kind == 2:
        pdf = want[:, 9].view(np.float32)
        assert (pdf[:512] > 0).sum() > 100 and (pdf[512:1024] > 0).sum() > 400 and (pdf[1024:] > 0).all()      # triangles facing the viewer, spheres, environment quads


def test_lights_match_live_reference():
    if ptref.refpin_hlsl() is None:
        pytest.skip("librefpin_hlsl.so not available (no /root/reference on this machine)")
    inputs = pin_inputs.light_inputs(20000, 0x11647, lambda kind, w: ptref.light_probe(kind, w))       # records built with the oracle's own packing ...
    for kind, words in inputs.items():
        got, want = ptref.light_probe(kind, words), ptref.light_probe(kind, words, reference=True)    # ... must read back identically through the reference text
        ok = _words_equal(got, want)
        bad = np.flatnonzero(~ok.all(1))
        assert ok.all(), "%s: %d of %d rows differ; first row %d: in=%s oracle=%s reference=%s" % (LIGHT_KINDS[kind], len(bad), len(words), bad[0], words[bad[0]], got[bad[0]], want[bad[0]])


def test_tonemapping_matches_reference_golden():
    """applyToneMapping of ToneMapper/ToneMapping.ps.hlsli (six operators, CPU auto exposure, colour transform, clamp) compiled from the reference text,
    against the oracle's tm_apply — the floats that go into the SRGBA8 store."""
    g = np.load(GOLDEN)
    cases = pin_inputs.tonemap_cases(0x5EED0500, ptref.TONEMAP_DTYPE)
    for k, (p, rgba) in enumerate(cases):
        got = ptref.tonemap_linear(rgba, p)
        ok = _same(got, g["tonemap_out"][k])
        assert ok.all(), "operator %d variant %d: %d of %d pixels differ" % (k // 4, k % 4, int((~ok).any(1).sum()), len(rgba))


def test_tonemapping_matches_live_reference():
    if ptref.refpin_hlsl() is None:
        pytest.skip("librefpin_hlsl.so not available (no /root/reference on this machine)")
    for k, (p, rgba) in enumerate(pin_inputs.tonemap_cases(0x70E3, ptref.TONEMAP_DTYPE)):
        ok = _same(ptref.tonemap_linear(rgba, p), ptref.tonemap_linear(rgba, p, reference=True))
        assert ok.all(), "operator %d variant %d: %d of %d pixels differ" % (k // 4, k % 4, int((~ok).any(1).sum()), len(rgba))


def test_lightbake_functions_match_reference_golden():
    """Lighting/LightsBaker.hlsl per-light / per-node functions compiled from the reference text: ComputeWeight (flux^0.8 + threshold), environment quad-tree
    node weight (PACK_20F_12UI) and node radiance / weight. (The passes around them use group-shared memory and float atomics whose order the GPU does
    not fix, so the baked tables themselves cannot be pinned bit for bit.)"""
    g = np.load(GOLDEN)
    k0, pyr, k1 = pin_inputs.lightbake_inputs(0x5EED0600, g["light2_in"][:, :12])
    assert np.array_equal(ptref.lightbake_probe(0, k0), g["lightbake0_out"])
    got = ptref.lightbake_probe(1, k1, pyr, (0.7, 1.3, 0.9), 0.0002)
    assert np.array_equal(got, g["lightbake1_out"]), "%d node rows differ" % int((got != g["lightbake1_out"]).any(1).sum())
    assert (g["lightbake0_out"].view(np.float32) > 0).sum() > 1000 and len(np.unique(g["lightbake1_out"][:, 0] >> 12)) > 100


def test_lightbake_functions_match_live_reference():
    if ptref.refpin_hlsl() is None:
        pytest.skip("librefpin_hlsl.so not available (no /root/reference on this machine)")
    recs = pin_inputs.light_inputs(8000, 0xBA4E, lambda kind, w: ptref.light_probe(kind, w))[2][:, :12]
    k0, pyr, k1 = pin_inputs.lightbake_inputs(0xBA4F, recs)
    assert np.array_equal(ptref.lightbake_probe(0, k0), ptref.lightbake_probe(0, k0, reference=True))
    assert np.array_equal(ptref.lightbake_probe(1, k1, pyr, (1.1, 0.4, 2.0), 0.0002), ptref.lightbake_probe(1, k1, pyr, (1.1, 0.4, 2.0), 0.0002, reference=True))
