"""Display path: tone mapping + sRGB8 store + screenshot writers.

CPU: the oracle (oracle/ptref/tonemap.h, restating Rtxpt/ToneMapper/ToneMapping.ps.hlsli:31-174) against an independent float64 numpy model
of the same formulas (tolerance 1 LSB of the 8-bit output); the product's host-only PNG/BMP writers round-trip.
GPU: k_tonemap through the C-ABI (pt_tonemap) == oracle, byte for byte, for every operator.
"""
import os
import struct
import zlib

import numpy as np
import pytest

import rtxpt_amd as pt
from oracle import ptref

OPS = ["linear", "reinhard", "reinhard_modified", "heji_hable_alu", "hable_uc2", "aces"]


def _model(rgba, t):
    """float64 restatement of applyToneMapping + SRGBA8 store, written independently of tonemap.h."""
    c = rgba[..., :3].astype(np.float64)
    if t["autoExposure"]:
        c = c * np.clip(0.042 / float(t["avgLuminance"]), float(t["autoExposureLumValueMin"]), float(t["autoExposureLumValueMax"]))
    if t["enabled"]:
        M = np.asarray(t["colorTransform"], np.float64).reshape(3, 3)
        c = c @ M
        op = int(t["toneMapOperator"])
        lum = c @ np.array([0.299, 0.587, 0.114])
        with np.errstate(all="ignore"):
            if op == 1:
                c = c * ((lum / (lum + 1)) / lum)[..., None]
            elif op == 2:
                w = float(t["whiteMaxLuminance"])
                c = c * ((lum * (1 + lum / (w * w)) * (1 + lum)) / lum)[..., None]
            elif op == 3:
                x = np.maximum(0, c - 0.004)
                c = ((x * (6.2 * x + 0.5)) / (x * (6.2 * x + 1.7) + 0.06)) ** 2.2
            elif op == 4:
                def uc2(v):
                    A, B, C, D, E, F = 0.22, 0.3, 0.1, 0.2, 0.01, 0.3
                    return ((v * (A * v + C * B) + D * E) / (v * (A * v + B) + D * F)) - E / F
                c = uc2(2.0 * c) * (1.0 / uc2(np.float64(t["whiteScale"])))
            elif op == 5:
                x = c * 0.6
                c = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1)
        if t["clamped"]:
            c = np.clip(c, 0, 1)
    c = np.nan_to_num(c, nan=0.0)
    v = np.clip(c, 0, 1)
    s = np.where(v <= 0.0031308, v * 12.92, 1.055 * np.power(v, 1 / 2.4) - 0.055)
    a = np.clip(rgba[..., 3:4].astype(np.float64), 0, 1)
    return np.concatenate([s * 255.0, a * 255.0], axis=-1)


def _radiance(seed=7, n=4096):
    rng = np.random.default_rng(seed)
    x = np.exp(rng.uniform(np.log(1e-4), np.log(50.0), size=(n, 4))).astype(np.float32)
    x[:64, :3] = 0.0                       # black
    x[64:96] = np.float32(1e6)            # blown out
    x[:, 3] = 1.0
    return x


@pytest.mark.parametrize("op", OPS)
def test_oracle_tonemap_matches_float64_model(op):
    t = pt.default_tonemap(exposure_compensation=-1.0, toneMapOperator=op)
    x = _radiance()
    if op in ("reinhard", "reinhard_modified"):
        x = x[96:]                        # luminance 0 is 0/0 in the reference's formula; not part of the numeric check
        if op == "reinhard_modified":
            x = x[(x[:, :3].max(axis=1) < 4.0)]          # the operator as written grows without bound; keep the unclamped range in the check
    got = ptref.tonemap(x, t).astype(np.float64)
    want = _model(x, t)
    assert np.abs(got - np.rint(want)).max() <= 1.0
    assert (np.abs(got - want) <= 1.001).all()


def test_tonemap_defaults_follow_the_reference():
    t = pt.default_tonemap()
    assert int(t["toneMapOperator"]) == 5 and int(t["clamped"]) == 1 and int(t["enabled"]) == 1 and int(t["autoExposure"]) == 0
    assert float(t["whiteScale"]) == np.float32(5.1) and float(t["whiteMaxLuminance"]) == 1.0
    # UpdateColorTransform: identity * 2^EC * (filmSpeed/100) / (shutter * fNumber^2)
    t2 = pt.default_tonemap(exposure_compensation=2.0, film_speed=200.0, shutter=0.5, f_number=2.0)
    assert np.allclose(np.asarray(t2["colorTransform"]).reshape(3, 3), np.eye(3) * (4.0 * 2.0 / (0.5 * 4.0)))
    # disabled tone mapping is a pass-through into the sRGB target
    t3 = pt.default_tonemap(enabled=0)
    x = np.array([[0.0, 0.5, 1.0, 1.0], [0.0021, 0.2, 2.0, 0.5]], np.float32)
    out = ptref.tonemap(x, t3)
    assert out.tolist() == [[0, 188, 255, 255], [7, 124, 255, 128]]


def _decode_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])
        assert crc == (zlib.crc32(typ + data) & 0xFFFFFFFF)
        if typ == b"IHDR":
            w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, comp, flt, inter) == (8, 6, 0, 0, 0)
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


def test_screenshot_writers_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    p = str(tmp_path / "shot.png")
    pt.write_image(p, img)
    assert (_decode_png(p) == img).all()
    q = str(tmp_path / "shot.bmp")
    pt.write_image(q, img)
    b = open(q, "rb").read()
    assert b[:2] == b"BM" and struct.unpack("<I", b[2:6])[0] == len(b) == 54 + img.size
    w, h, planes, bpp = struct.unpack("<iiHH", b[18:30])
    assert (w, h, planes, bpp) == (53, -37, 1, 32)
    px = np.frombuffer(b[54:], np.uint8).reshape(37, 53, 4)
    assert (px[..., [2, 1, 0, 3]] == img).all()
    with pytest.raises(pt.PtError):
        pt.write_image(str(tmp_path / "no_such_dir" / "x.png"), img)


@pytest.mark.gpu
def test_gpu_tonemap_equals_oracle_for_every_operator(tmp_path):
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C2")
    W, H = 96, 64
    g = pt.PathTracer(device=0)
    g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.config_settings("C2")); g.resize(W, H)
    g.render(0, 4)
    rad = g.radiance()
    assert rad[..., :3].max() > 1.0 and (rad[..., :3] > 0).mean() > 0.5
    for op in OPS:
        for ec in (0.0, -2.0):
            t = pt.default_tonemap(exposure_compensation=ec, toneMapOperator=op)
            got = g.tonemap(t)
            want = ptref.tonemap(rad, t)
            assert got.shape == (H, W, 4) and (got == want).all(), (op, ec, int((got != want).sum()))
    t = pt.default_tonemap(autoExposure=1, avgLuminance=0.3)
    assert (g.tonemap(t) == ptref.tonemap(rad, t)).all()
    shot = str(tmp_path / "c2.png")
    pt.write_image(shot, g.tonemap())
    assert (_decode_png(shot) == ptref.tonemap(rad, pt.default_tonemap())).all()


def test_oracle_average_luminance_properties():
    """The oracle of the luminance capture on images whose answer is known: a constant image, the 1e-4 floor, a power-of-two image (no resampling: the
    geometric mean of the luminances), and a non-power-of-two one against a direct per-texel evaluation."""
    assert abs(ptref.average_luminance(np.full((37, 91, 4), 0.5, np.float32)) - 0.5) < 1e-12
    assert abs(ptref.average_luminance(np.zeros((16, 16, 4), np.float32)) - 1e-4) < 1e-15
    rng = np.random.default_rng(3)
    img = np.exp(rng.uniform(-6, 3, size=(32, 64, 4))).astype(np.float32)
    lum = img[..., :3].astype(np.float64) @ np.array([0.299, 0.587, 0.114])
    assert abs(ptref.average_luminance(img) / np.exp(np.mean(np.log(lum))) - 1) < 1e-12
    img = np.exp(rng.uniform(-6, 3, size=(5, 7, 4))).astype(np.float32)       # 7x5 -> 4x4 target
    acc = 0.0
    for y in range(4):
        for x in range(4):
            sx, sy = (x + 0.5) / 4 * 7 - 0.5, (y + 0.5) / 4 * 5 - 0.5
            x0, y0 = int(np.floor(sx)), int(np.floor(sy)); fx, fy = sx - x0, sy - y0
            px = lambda yy, xx: img[min(max(yy, 0), 4), min(max(xx, 0), 6), :3].astype(np.float64)
            c = (px(y0, x0) * (1 - fx) + px(y0, x0 + 1) * fx) * (1 - fy) + (px(y0 + 1, x0) * (1 - fx) + px(y0 + 1, x0 + 1) * fx) * fy
            acc += np.log2(max(1e-4, c @ np.array([0.299, 0.587, 0.114])))
    assert abs(ptref.average_luminance(img) / 2.0 ** (acc / 16) - 1) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(96, 64), (100, 70), (64, 33), (1, 1), (257, 3)])
def test_gpu_average_luminance_matches_oracle(size):
    """pt_average_luminance (k_log_luminance + k_luminance_mip) against the float64 oracle. Tolerance 2e-5 relative: the kernel works in fp32 (bilinear
    weights, dm_log2, pairwise averages), and the reference's own texture-unit filter is fixed-point, so this quantity is defined to tolerance only."""
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C2")
    W, H = size
    g = pt.PathTracer(device=0)
    g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.config_settings("C2")); g.resize(W, H)
    g.render(0, 2)
    rad = g.radiance()
    got, want = g.average_luminance(), ptref.average_luminance(rad)
    assert np.isfinite(got) and got > 0 and abs(got / want - 1) < 2e-5, (size, got, want)
    # and it feeds the tone mapper the way TONEMAPPING_AUTOEXPOSURE_CPU does
    t = pt.default_tonemap(autoExposure=1, avgLuminance=got)
    assert (g.tonemap(t) == ptref.tonemap(rad, t)).all()
