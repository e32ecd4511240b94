"""pt_tonemap_color_transform (white balance + exposure, the host half of the display path) against the reference's own text:
ColorUtils.h (calculateWhiteBalanceTransformRGB_Rec709, colorTemperatureToXYZ) and ToneMappingPass::UpdateWhiteBalanceTransform / UpdateColorTransform
(ToneMappingPasses.cpp:392-441) compiled as they stand over Donut math stand-ins — committed golden vectors plus a live comparison where the pin is built.
Bit-exact: the transform is fp32 products in a fixed order over a double-precision chromaticity fit."""
import os
import sys
import numpy as np
import pytest
import rtxpt_amd as pt
from oracle import ptref
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_color_transform_golden import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_transform_golden.npz")


def product(row):
    t = np.zeros((), dtype=pt.TONEMAP_DTYPE); t["autoExposure"] = int(row[2])
    pt.tonemap_color_transform(t, bool(row[0]), row[1], row[3], row[4], row[5], row[6])
    return t["colorTransform"].copy()


def test_color_transform_matches_reference_golden():
    g = np.load(GOLD)
    assert np.array_equal(g["cases"], cases())
    for row, want in zip(g["cases"], g["transform"]):
        got = product(row)
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (row, got, want)


def test_color_transform_matches_reference_text_live():
    rng = np.random.default_rng(7)
    if ptref.reference_color_transform(0, 6500.0, 0, 0.0, 100.0, 1.0, 1.0) is None:
        pytest.skip("oracle/_ref/librefpin_mat.so not built (no /root/reference here)")
    for _ in range(2000):
        row = (int(rng.integers(0, 2)), float(np.float32(rng.uniform(1667.0, 25000.0))), int(rng.integers(0, 4) == 0), float(np.float32(rng.uniform(-6, 6))),
               float(np.float32(rng.uniform(25, 6400))), float(np.float32(rng.uniform(0.001, 1000.0))), float(np.float32(rng.uniform(0.5, 32.0))))
        want = ptref.reference_color_transform(*row)
        got = product(row)
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (row, got, want)


def test_color_transform_properties():
    # white balance off == pt_default_tonemap's transform; 6500 K is the identity to rounding; out-of-range temperatures give the reference's non-finite matrix
    d = pt.default_tonemap(0.75, 200.0, 0.5, 2.8)
    t = d.copy(); t["colorTransform"] = 0
    pt.tonemap_color_transform(t, False, 6500.0, 0.75, 200.0, 0.5, 2.8)
    assert np.array_equal(t["colorTransform"], d["colorTransform"])
    pt.tonemap_color_transform(t, True, 6500.0, 0.0, 100.0, 1.0, 1.0)
    assert np.allclose(t["colorTransform"].reshape(3, 3), np.eye(3), atol=2e-7)
    pt.tonemap_color_transform(t, True, 1000.0, 0.0, 100.0, 1.0, 1.0)
    assert not np.isfinite(t["colorTransform"]).all()
    # warmer target white point (lower T) boosts blue relative to red on a grey input, cooler does the opposite — through the constant-buffer convention
    grey = np.ones(3, np.float32)
    pt.tonemap_color_transform(t, True, 4000.0, 0.0, 100.0, 1.0, 1.0); warm = grey @ t["colorTransform"].reshape(3, 3)
    pt.tonemap_color_transform(t, True, 9000.0, 0.0, 100.0, 1.0, 1.0); cool = grey @ t["colorTransform"].reshape(3, 3)
    assert warm[2] / warm[0] > 1.0 > cool[2] / cool[0]
    # auto exposure: the manual factor drops out (ToneMappingPasses.cpp:434), shutter / fNumber are not even validated
    t["autoExposure"] = 1
    pt.tonemap_color_transform(t, False, 6500.0, 1.0, 100.0, 0.0, 0.0)
    assert np.array_equal(t["colorTransform"].reshape(3, 3), 2.0 * np.eye(3, dtype=np.float32))
    t["autoExposure"] = 0
    with pytest.raises(pt.PtError):
        pt.tonemap_color_transform(t, False, 6500.0, 1.0, 100.0, 0.0, 1.0)


# ---- the whole UI block: ToneMappingPass::PreRender + the constant fill of ::Render
def _ui_words(u):
    return np.frombuffer(u.tobytes(), np.uint32)


def _product_constants(u, avg, enabled):
    t = pt.tonemap_from_parameters(u, avg, enabled)
    scal = np.array([t["whiteScale"], t["whiteMaxLuminance"]], np.float32).view(np.uint32).tolist() + [int(t["toneMapOperator"]), int(t["clamped"]), int(t["autoExposure"])] + \
        np.array([t["avgLuminance"], t["autoExposureLumValueMin"], t["autoExposureLumValueMax"]], np.float32).view(np.uint32).tolist()
    M = t["colorTransform"].reshape(3, 3)
    rows = np.concatenate([M, np.zeros((3, 1), np.float32)], axis=1).reshape(-1)            # constant-buffer rows are float4(col(i), 0)
    return scal + rows.view(np.uint32).tolist() + [int(t["enabled"])]


def _random_ui(rng):
    return pt.default_tone_mapping_parameters(
        exposureMode=int(rng.integers(0, 2)), toneMapOperator=int(rng.integers(0, 6)), autoExposure=int(rng.integers(0, 3) == 0),
        exposureCompensation=np.float32(rng.uniform(-4, 4)), exposureValue=np.float32(rng.uniform(-30, 40)), filmSpeed=np.float32(rng.uniform(25, 3200)),
        fNumber=np.float32(rng.uniform(0.7, 22)), shutter=np.float32(rng.uniform(0.001, 1000)), whiteBalance=int(rng.integers(0, 2)),
        whitePoint=np.float32(rng.uniform(1667, 25000)), whiteMaxLuminance=np.float32(rng.uniform(0.5, 8)), whiteScale=np.float32(rng.uniform(1, 12)),
        clamped=int(rng.integers(0, 2)), exposureValueMin=np.float32(rng.uniform(-20, 0)), exposureValueMax=np.float32(rng.uniform(0, 20)))


def test_tone_mapping_parameters_match_reference_text_live():
    d = ptref.reference_tonemap_defaults()
    if d is None:
        pytest.skip("oracle/_ref/librefpin_mat.so not built (no /root/reference here)")
    assert _ui_words(pt.default_tone_mapping_parameters()).tolist() == d.tolist()
    rng = np.random.default_rng(11)
    for _ in range(1500):
        u = _random_ui(rng); avg = float(np.float32(rng.uniform(0.01, 4))); en = int(rng.integers(0, 2))
        want = ptref.reference_tonemap_constants(_ui_words(u), avg, en).tolist()
        got = _product_constants(u, avg, en)
        assert got == want, (u, got, want)


def test_tone_mapping_parameters_golden_and_semantics():
    g = np.load(GOLD)
    assert _ui_words(pt.default_tone_mapping_parameters()).tolist() == g["ui_defaults"].tolist()
    rng = np.random.default_rng(20260925)
    for want in g["ui_constants"]:
        u = _random_ui(rng); avg = float(np.float32(rng.uniform(0.01, 4))); en = int(rng.integers(0, 2))
        assert _product_constants(u, avg, en) == want.tolist(), u
    # defaults: EV 0, f/1 -> shutter 1: the same transform as pt_default_tonemap(0, 100, 1, 1)
    assert np.array_equal(pt.tonemap_from_parameters(pt.default_tone_mapping_parameters())["colorTransform"], pt.default_tonemap()["colorTransform"])
    # aperture priority ignores the UI's shutter; shutter priority ignores its fNumber
    a = pt.tonemap_from_parameters(pt.default_tone_mapping_parameters(exposureValue=3.0, shutter=1.0))["colorTransform"]
    b = pt.tonemap_from_parameters(pt.default_tone_mapping_parameters(exposureValue=3.0, shutter=77.0))["colorTransform"]
    assert np.array_equal(a, b) and np.allclose(a.reshape(3, 3), np.eye(3) / 8.0)
    a = pt.tonemap_from_parameters(pt.default_tone_mapping_parameters(exposureMode=1, exposureValue=2.0, shutter=1.0, fNumber=1.0))["colorTransform"]
    b = pt.tonemap_from_parameters(pt.default_tone_mapping_parameters(exposureMode=1, exposureValue=2.0, shutter=1.0, fNumber=9.0))["colorTransform"]
    assert np.array_equal(a, b) and np.allclose(a.reshape(3, 3), np.eye(3) / 4.0, rtol=1e-6)
    # auto exposure: manual factor 1, luminance limits from the EV range
    t = pt.tonemap_from_parameters(pt.default_tone_mapping_parameters(autoExposure=1, exposureValueMin=-3.0, exposureValueMax=5.0, exposureCompensation=1.0), 0.25)
    assert t["autoExposureLumValueMin"] == 0.125 and t["autoExposureLumValueMax"] == 32.0 and t["avgLuminance"] == 0.25 and np.array_equal(t["colorTransform"].reshape(3, 3), 2 * np.eye(3, dtype=np.float32))
    with pytest.raises(pt.PtError):
        pt.tonemap_from_parameters(pt.default_tone_mapping_parameters(toneMapOperator=9))
