"""The PNG reader behind glTF images and `.material.json` textures (pt_gltf.cpp decode_png), CPU only, against Pillow: every colour type and bit depth of the specification's
table 11.1 (grey 1 / 2 / 4 / 8 / 16, RGB 8 / 16, palette 1 / 2 / 4 / 8, grey + alpha 8 / 16, RGBA 8 / 16), all five filters, Adam7 interlacing (written here: seven reduced
images), tRNS colour keys and palette alpha. 16-bit samples come out as their high byte, as stb_image gives them to an 8-bit texture request."""
import json, os, struct, sys, zlib
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rtxpt_amd as pt
from test_scene_json import make_folder

PIL = pytest.importorskip("PIL.Image")
import io

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def _chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _pack_rows(samples, depth, rng):
    """samples: uint16 [h, w, ch] -> filtered scanlines (a random filter type per row, encoded properly)"""
    h, w, ch = samples.shape; out = bytearray()
    if depth == 16: rows = samples.astype(">u2").tobytes()
    elif depth == 8: rows = samples.astype(np.uint8).tobytes()
    else:
        bits = np.unpackbits(samples.astype(np.uint8)[..., None], axis=-1)[..., 8 - depth:].reshape(h, -1)
        pad = (-bits.shape[1]) % 8; bits = np.pad(bits, ((0, 0), (0, pad))); rows = np.packbits(bits, axis=1).tobytes()
    stride = len(rows) // h; bpp = max(1, ch * depth // 8); prev = bytearray(stride)
    for y in range(h):
        cur = bytearray(rows[y * stride:(y + 1) * stride]); ft = int(rng.integers(0, 5)); enc = bytearray(stride)
        for x in range(stride):
            a = cur[x - bpp] if x >= bpp else 0; b = prev[x]; c = prev[x - bpp] if x >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) // 2
            else:
                pp = a + b - c; pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c); p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            enc[x] = (cur[x] - p) & 255
        out += bytes([ft]) + enc; prev = cur
    return bytes(out)


def _png(samples, ctype, depth, rng, interlace=False, plte=None, trns=None):
    h, w, ch = samples.shape
    if interlace:
        data = b"".join(_pack_rows(samples[ys::yst, xs::xst], depth, rng) for xs, ys, xst, yst in ADAM7 if samples[ys::yst, xs::xst].size)
    else: data = _pack_rows(samples, depth, rng)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None: out += _chunk(b"PLTE", plte)
    if trns is not None: out += _chunk(b"tRNS", trns)
    z = zlib.compress(data, 6); half = len(z) // 2
    return out + _chunk(b"IDAT", z[:half]) + _chunk(b"IDAT", z[half:]) + _chunk(b"IEND", b"")


CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [False, True], ids=["plain", "adam7"])
@pytest.mark.parametrize("ctype,depth", CASES)
def test_every_colour_type_and_depth_equals_pillow(tmp_path, ctype, depth, interlace):
    rng = np.random.default_rng(100 * ctype + depth + (7 if interlace else 0))
    for k, (w, h) in enumerate(((13, 9), (1, 1), (8, 8), (5, 17))):
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        s = rng.integers(0, 1 << depth, (h, w, ch)).astype(np.uint16)
        plte = rng.integers(0, 256, 3 << depth, dtype=np.uint8).tobytes() if ctype == 3 else None
        trns = None
        if ctype == 3 and k % 2 == 0: trns = rng.integers(0, 256, max(1, (1 << depth) // 2), dtype=np.uint8).tobytes()
        if ctype == 0 and k % 2 == 0: trns = struct.pack(">H", int(s[0, 0, 0]))
        if ctype == 2 and k % 2 == 0: trns = struct.pack(">HHH", *[int(v) for v in s[h // 2, w // 2]])
        data = _png(s, ctype, depth, rng, interlace, plte, trns)
        media, sc, _ = make_folder(tmp_path / ("c%d" % k), [{"model": 0}], {"red.material.json": {"BaseTexture": {"path": "Textures/t.png", "sRGB": False}}})
        os.makedirs(media / "Textures", exist_ok=True); (media / "Textures" / "t.png").write_bytes(data)
        imp = pt.SceneImport(media / "test.scene.json")
        assert imp.info["numTextures"] == 1 and imp.info["texturesNotLoaded"] == 0, (ctype, depth, interlace, w, h)
        got, fmt = imp.texture(0)
        im = PIL.open(io.BytesIO(data)); im.load()
        if depth == 16:                       # Pillow keeps 16 bits for grey and rescales otherwise: compare with the high byte of the samples directly
            hi = (s >> 8).astype(np.uint8)
            want = {0: np.concatenate([hi.repeat(3, 2), np.full((h, w, 1), 255, np.uint8)], 2), 2: np.concatenate([hi, np.full((h, w, 1), 255, np.uint8)], 2),
                    4: np.concatenate([hi[..., :1].repeat(3, 2), hi[..., 1:]], 2), 6: hi}[ctype].copy()
            if trns is not None and ctype == 0: want[..., 3] = np.where(s[..., 0] == s[0, 0, 0], 0, 255)
            if trns is not None and ctype == 2: want[..., 3] = np.where((s == s[h // 2, w // 2]).all(-1), 0, 255)
        else:
            want = np.asarray(im.convert("RGBA")).copy()
            # colour keys compare the samples as stored (PNG specification 11.3.2.1; stb_image scales the key like the samples): Pillow misses keys of sub-byte grey images
            if trns is not None and ctype == 0: want[..., 3] = np.where(s[..., 0] == s[0, 0, 0], 0, 255)
            if trns is not None and ctype == 2: want[..., 3] = np.where((s == s[h // 2, w // 2]).all(-1), 0, 255)
        assert got.shape == (h, w, 4) and np.array_equal(got, want), (ctype, depth, interlace, w, h)


def test_invalid_depths_and_damaged_files_are_textures_not_loaded(tmp_path):
    rng = np.random.default_rng(9); s = rng.integers(0, 256, (6, 6, 3)).astype(np.uint16)
    good = _png(s, 2, 8, rng)
    bad_depth = bytearray(good); bad_depth[24] = 4; bad_depth[29:33] = struct.pack(">I", zlib.crc32(bytes(bad_depth[12:29])) & 0xFFFFFFFF)      # RGB with 4 bits per sample: not a PNG
    for k, data in enumerate((bytes(bad_depth), good[:40], good[:-20], good.replace(b"IDAT", b"IDAX"))):
        media, sc, _ = make_folder(tmp_path / ("d%d" % k), [{"model": 0}], {"red.material.json": {"BaseTexture": {"path": "Textures/t.png", "sRGB": False}}})
        os.makedirs(media / "Textures", exist_ok=True); (media / "Textures" / "t.png").write_bytes(data)
        imp = pt.SceneImport(media / "test.scene.json")
        assert imp.info["texturesNotLoaded"] == 1 and imp.info["numTextures"] == 0


def test_screenshot_writers_are_read_back_by_an_independent_reader(tmp_path):
    """pt_write_png / pt_write_bmp (the reference's screenshot formats, CaptureScriptManager.cpp:29-60): Pillow reads back exactly what was written."""
    rng = np.random.default_rng(17)
    for (w, h) in ((1, 1), (7, 5), (64, 33)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8); img[..., 3] = 255
        pt.write_image(str(tmp_path / "s.png"), img); pt.write_image(str(tmp_path / "s.bmp"), img)
        assert np.array_equal(np.asarray(PIL.open(tmp_path / "s.png").convert("RGBA")), img)
        assert np.array_equal(np.asarray(PIL.open(tmp_path / "s.bmp").convert("RGB")), img[..., :3])
