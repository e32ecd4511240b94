"""pt_image_read_jpeg (rtxpt_amd/csrc/pt_jpeg.cpp) — the JPEG images of glTF files — against Pillow (libjpeg-turbo, the IJG reference decoder's default path, which the
reader restates: islow IDCT, fancy up-sampling, fixed-point YCbCr): baseline and progressive streams, 4:4:4 / 4:2:2 / 4:2:0, greyscale, odd sizes down to
1 x 1, restart intervals, optimised Huffman tables; damaged streams never crash. CPU only."""
import io
import numpy as np
import pytest

import rtxpt_amd as pt

pytest.importorskip("PIL.Image")
from PIL import Image


def _picture(w, h, seed):
    rng = np.random.default_rng(seed); y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0), 127 + 100 * np.sin((x + y) / 11.0), 40 + 2.0 * ((x * 3 + y * 5) % 100)], -1)
    img += rng.normal(0, 12, img.shape); img[h // 3: h // 3 + 3] = 255; img[:, w // 2: w // 2 + 2] = 0          # smooth + noise + hard edges
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")


def _check(data, exact=True):
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = pt.read_jpeg(data)
    assert got.shape == want.shape[:2] + (4,) and (got[..., 3] == 255).all()
    d = np.abs(got[..., :3].astype(int) - want.astype(int))
    assert d.max() == 0 if exact else d.max() <= 1, "max difference %d, %d samples differ" % (d.max(), int((d > 0).sum()))


@pytest.mark.parametrize("progressive", [False, True], ids=["baseline", "progressive"])
@pytest.mark.parametrize("subsampling", ["4:4:4", "4:2:2", "4:2:0"])
@pytest.mark.parametrize("size", [(64, 48), (37, 29), (3, 2), (1, 1), (17, 131)])
def test_equals_the_ijg_decoder(size, subsampling, progressive):
    for quality, optimize in ((90, False), (35, True)):
        buf = io.BytesIO(); _picture(size[0], size[1], quality).save(buf, "JPEG", quality=quality, subsampling=subsampling, progressive=progressive, optimize=optimize)
        _check(buf.getvalue())


def test_greyscale_restart_intervals_and_large_image():
    buf = io.BytesIO(); _picture(70, 50, 1).convert("L").save(buf, "JPEG", quality=80); _check(buf.getvalue())
    buf = io.BytesIO(); _picture(70, 50, 2).convert("L").save(buf, "JPEG", quality=80, progressive=True); _check(buf.getvalue())
    for sub in ("4:4:4", "4:2:0"):
        for prog in (False, True):
            buf = io.BytesIO(); _picture(150, 90, 3).save(buf, "JPEG", quality=75, subsampling=sub, progressive=prog, restart_marker_blocks=5); data = buf.getvalue()
            assert b"\xff\xdd" in data and b"\xff\xd0" in data
            _check(data)
    buf = io.BytesIO(); _picture(1024, 768, 4).save(buf, "JPEG", quality=85, subsampling="4:2:0"); _check(buf.getvalue())


def test_damaged_streams_fail_cleanly():
    buf = io.BytesIO(); _picture(48, 40, 5).save(buf, "JPEG", quality=70, progressive=True); good = buf.getvalue()
    rng = np.random.default_rng(6)
    for k in range(400):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        b = bytes(b[: int(rng.integers(4, len(b) + 1))])
        try: img = pt.read_jpeg(b); assert img.ndim == 3
        except pt.PtError: pass
    big = bytearray(good); i = good.index(b"\xff\xc2"); big[i + 5:i + 9] = bytes([0x7F, 0xFF, 0x7F, 0xFF])          # the same stream claiming 32767 x 32767 pixels: refused before any allocation
    import time; t0 = time.time()
    with pytest.raises(pt.PtError): pt.read_jpeg(bytes(big))
    assert time.time() - t0 < 1.0
    for junk in (b"", b"\xff", b"\xff\xd8", b"\xff\xd8\xff\xd9", b"not a jpeg at all", good[:200]):
        with pytest.raises(pt.PtError): pt.read_jpeg(junk)


def test_gltf_image_jpeg_is_imported(tmp_path):
    """A glTF whose base-colour image is a .jpg: the importer decodes it (it used to count as "texture not loaded")."""
    import json, os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gltf_writer import write_gltf
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C2")
    write_gltf(sc, str(tmp_path / "c.gltf"))
    _picture(40, 24, 9).save(tmp_path / "base.jpg", "JPEG", quality=88, subsampling="4:2:0")
    doc = json.loads((tmp_path / "c.gltf").read_text())
    doc["images"] = [{"uri": "base.jpg"}]; doc["textures"] = [{"source": 0}]; doc["materials"][0].setdefault("pbrMetallicRoughness", {})["baseColorTexture"] = {"index": 0}
    (tmp_path / "c.gltf").write_text(json.dumps(doc)); (tmp_path / "c.scene.json").write_text(json.dumps({"models": ["c.gltf"], "graph": [{"model": 0}]}))
    imp = pt.SceneImport(tmp_path / "c.scene.json")
    assert imp.info["numTextures"] == 1 and imp.info["texturesNotLoaded"] == 0
    px, fmt = imp.texture(0)
    assert fmt == pt.PT_TEX_RGBA8_SRGB and np.array_equal(px[..., :3], np.asarray(Image.open(tmp_path / "base.jpg").convert("RGB")))
    assert imp.materials[0]["BaseOrDiffuseTextureIndex"] == scenes.pack_texture_word(0, 40, 24)
