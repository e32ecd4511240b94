"""pt_image_read_dds (rtxpt_amd/csrc/pt_dds.cpp) — the .dds files the reference's texture pipeline prefers next to a .png (MaterialsBaker.cpp:178-191; BC7 from its
compression script) — CPU only, against Pillow's independent DDS reader: BC7 on random blocks of every mode (any 128 bits are a valid block), BC1 / BC2 / BC3 / BC5
on Pillow-encoded and on random blocks, uncompressed and float files, non-multiple-of-4 sizes, a mip chain behind the top level, damaged files."""
import io, os, struct
import numpy as np
import pytest

import rtxpt_amd as pt

PIL = pytest.importorskip("PIL.Image")
from PIL import Image


def _dds(payload, w, h, dxgi=None, fourcc=None, masks=None, mips=1):
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | (0x20000 if mips > 1 else 0)
    if masks: pf = struct.pack("<2I4s5I", 32, 0x41, b"\0\0\0\0", 32, *masks)
    else: pf = struct.pack("<2I4s5I", 32, 0x4, b"DX10" if dxgi is not None else fourcc, 0, 0, 0, 0, 0)
    hdr = b"DDS " + struct.pack("<7I", 124, flags, h, w, 0, 0, mips) + b"\0" * 44 + pf + struct.pack("<5I", 0x1000, 0, 0, 0, 0)
    if dxgi is not None: hdr += struct.pack("<5I", dxgi, 3, 0, 1, 0)
    return hdr + payload


def _pil(data):
    im = Image.open(io.BytesIO(data)); im.load()
    return im


def _mine(tmp_path, data, name="t.dds"):
    p = tmp_path / name; p.write_bytes(data)
    return pt.read_dds(p)


def test_bc7_random_blocks_equal_pillow(tmp_path):
    rng = np.random.default_rng(7); w, h = 64, 64
    modes = np.zeros(8, int)
    for rep in range(6):
        blocks = rng.integers(0, 256, (w // 4) * (h // 4) * 16, dtype=np.uint8).reshape(-1, 16)
        if rep >= 2:                                 # random bytes mostly land in the low modes: force every mode (mode m = m zero bits, then a one)
            m = rng.integers(0, 8, len(blocks)); blocks[:, 0] = (blocks[:, 0] & ~((1 << (m + 1)) - 1).astype(np.uint8)) | (1 << m).astype(np.uint8)
        if rep == 5: blocks[::7, 0] = 0              # the reserved mode: transparent black
        for b in blocks[:, 0]:
            k = 0
            while k < 8 and not (b >> k) & 1: k += 1
            if k < 8: modes[k] += 1
        data = _dds(blocks.tobytes(), w, h, dxgi=98 if rep % 2 == 0 else 99)
        got, fmt = _mine(tmp_path, data)
        want = np.asarray(_pil(data).convert("RGBA"))
        assert fmt == (pt.PT_TEX_RGBA8_UNORM if rep % 2 == 0 else pt.PT_TEX_RGBA8_SRGB) and got.shape == (h, w, 4)
        reserved = np.repeat(np.repeat((blocks[:, 0] == 0).reshape(h // 4, w // 4), 4, 0), 4, 1)          # mode bits all zero: the reserved mode decodes to all-zero texels (D3D11 BC7
        assert (got[reserved] == 0).all()                                                                  # format: "an all-0 block is returned"); Pillow makes those opaque
        assert np.array_equal(got[~reserved], want[~reserved]), "rep %d: %d pixels differ" % (rep, int((got != want).any(-1)[~reserved].sum()))
    assert (modes > 100).all()          # every mode was exercised


@pytest.mark.parametrize("kind", ["DXT1", "DXT3", "DXT5", "BC5"])
def test_s3tc_and_rgtc_equal_pillow(tmp_path, kind):
    rng = np.random.default_rng(11); w, h = 52, 36
    img = Image.fromarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8), "RGBA")
    buf = io.BytesIO(); (img.convert("RGB") if kind == "BC5" else img).save(buf, "DDS", pixel_format=kind); data = buf.getvalue()
    got, fmt = _mine(tmp_path, data)
    want = np.asarray(_pil(data).convert("RGBA")) if kind != "BC5" else None
    if kind == "BC5":
        p = np.asarray(_pil(data).convert("RGB")); assert np.abs(got[..., :2].astype(int) - p[..., :2].astype(int)).max() <= 1 and (got[..., 3] == 255).all()
    else:                                                # colour ramps: the third-points are round((2 a + b) / 3) here (the ideal value to the nearest integer); Pillow truncates
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got >= want).all() and (kind == "DXT5" or np.array_equal(got[..., 3], want[..., 3]))      # (DXT5's alpha ramp likewise)
    # random payloads behind the same header (both endpoint orders, the three-colour BC1 mode, both BC3 alpha modes)
    nblk = ((w + 3) // 4) * ((h + 3) // 4); bs = 8 if kind == "DXT1" else 16
    head = data[:len(data) - nblk * bs]; rnd = head + rng.integers(0, 256, nblk * bs, dtype=np.uint8).tobytes()
    got, _ = _mine(tmp_path, rnd)
    if kind == "BC5": assert np.abs(got[..., :2].astype(int) - np.asarray(_pil(rnd).convert("RGB"))[..., :2].astype(int)).max() <= 1
    else:
        want = np.asarray(_pil(rnd).convert("RGBA"))
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (kind == "DXT5" or np.array_equal(got[..., 3], want[..., 3]))


BC6_MODES = [(2, 0, 10), (2, 1, 7), (5, 2, 11), (5, 6, 11), (5, 10, 11), (5, 14, 9), (5, 18, 8), (5, 22, 8), (5, 26, 8), (5, 30, 6), (5, 3, 10), (5, 7, 11), (5, 11, 12), (5, 15, 16)]      # (mode bits, mode value, endpoint precision)


def _bc6_blocks(rng, n, mode_bits, mode_value):
    blk = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    blk[:, 0] = (blk[:, 0] & (0xFF ^ ((1 << mode_bits) - 1))) | mode_value
    return blk


def _texels(a, n):
    return np.asarray(a, np.float64).reshape(128, 128, -1)[..., :3].reshape(32, 4, 32, 4, 3).transpose(0, 2, 1, 3, 4).reshape(n, 16, 3)


@pytest.mark.parametrize("mode_bits,mode_value,prec", BC6_MODES)
def test_bc6h_unsigned_random_blocks_of_every_mode_against_pillow(tmp_path, mode_bits, mode_value, prec):
    """BC6H_UF16: 1024 random blocks per mode (every partition, every index pattern, deltas that wrap) against Pillow's decoder at the 8 bits that one outputs
    (clamp to [0, 1], x 255, truncated). The format's interpolation rounds —
    (a (64 - w) + b w + 32) >> 6, which is also what the reference's own encoder assumes (BC6UCompress.hlsl FinishUnquantize) — and Pillow's truncates: at most one 8-bit step on a fraction of a percent of the texels
    with endpoints finer than 10 bits, and next to never below that (there the unquantised endpoints are all 32 mod 64, where the two rules coincide)."""
    rng = np.random.default_rng(100 + mode_value); n = 1024
    data = _dds(_bc6_blocks(rng, n, mode_bits, mode_value).tobytes(), 128, 128, dxgi=95)
    px, fmt = _mine(tmp_path, data)
    assert fmt == pt.PT_TEX_RGBA32F and px.shape == (128, 128, 4) and np.all(px[..., 3] == 1.0) and np.isfinite(px).all() and (px >= 0).all()
    mine = np.floor(np.clip(_texels(px, n), 0, 1) * 255).astype(int); pil = _texels(np.asarray(_pil(data).convert("RGB")), n).astype(int)
    d = np.abs(mine - pil)
    assert d.max() <= 1 and (d > 0).mean() < (0.0005 if prec <= 10 else 0.005)      # (<= 10 bits: only where an endpoint is 0 or all ones, the two values that are not 32 mod 64)


def test_bc6h_signed_untransformed_modes_against_pillow_and_reserved_modes(tmp_path):
    """BC6H_SF16: the two modes without the delta transform (6666 and 10:10) against Pillow wherever the texel is not negative (negative texels: Pillow clamps to 0). The
    transformed signed modes follow the format's rule (sign-extended base, wrap within its width, sign extension again) and are not checked here: Pillow's output for them
    disagrees with that rule. Reserved mode values decode to zero."""
    rng = np.random.default_rng(7); n = 1024
    for mode_bits, mode_value in ((5, 30), (5, 3)):
        data = _dds(_bc6_blocks(rng, n, mode_bits, mode_value).tobytes(), 128, 128, dxgi=96)
        px, fmt = _mine(tmp_path, data); t = _texels(px, n)
        assert (t < 0).mean() > 0.3                                                      # signed: about half of the random texels are negative
        mine = np.floor(np.clip(t, 0, 1) * 255).astype(int); pil = _texels(np.asarray(_pil(data).convert("RGB")), n).astype(int)
        assert np.abs(mine - pil)[t >= 0].max() <= 1 and (np.abs(mine - pil)[t >= 0] > 0).mean() < 0.001 and np.all(pil[t < 0] == 0)
    for reserved in (0x13, 0x17, 0x1B, 0x1F):
        px, _ = _mine(tmp_path, _dds(_bc6_blocks(rng, n, 5, reserved).tobytes(), 128, 128, dxgi=95))
        assert not px[..., :3].any() and np.all(px[..., 3] == 1.0)


def test_bc6h_blocks_of_the_reference_encoder_decode_to_the_same_half_floats(tmp_path):
    """Blocks written by the cube compressor (BC6UCompress.hlsl as the oracle restates it, pinned to the reference text: modes 11, 7.6 and 9.5) through the .dds reader: the float
    texels are exactly the half floats the oracle's own decode of those blocks gives — two independent implementations of the BC6H_UF16 rule, all 16 bits."""
    from oracle import ptref
    rng = np.random.default_rng(11); n = 1024
    T = np.zeros((n, 16, 3), np.float32)
    for k in range(n):
        c0, c1 = rng.uniform(0, 1, 3), rng.uniform(0, 1, 3); a = rng.uniform(0, 2 * np.pi); off = rng.uniform(-1.5, 1.5)
        side = ((np.arange(16) % 4 - 1.5) * np.cos(a) + (np.arange(16) // 4 - 1.5) * np.sin(a)) > off
        T[k] = np.where(side[:, None], c0, c1) * rng.uniform(0.9, 1.1, (16, 3)) * 10 ** rng.uniform(-2, 3)
    T = T.astype(np.float16).astype(np.float32)
    blk = ptref.bc6_encode(T, quality=True)
    assert ((blk[:, 0] & 31) == 3).sum() > 50 and ((blk[:, 0] & 3) == 1).sum() > 50 and ((blk[:, 0] & 31) == 0xE).sum() > 50
    px, _ = _mine(tmp_path, _dds(blk.astype("<u4").tobytes(), 128, 128, dxgi=95))
    want = ptref.bc6_decode(blk).astype(np.uint16).view(np.float16).astype(np.float32).reshape(n, 16, 3)
    assert np.array_equal(_texels(px, n).astype(np.float32), want)


def test_uncompressed_and_float_files(tmp_path):
    rng = np.random.default_rng(3); w, h = 19, 7
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    got, fmt = _mine(tmp_path, _dds(px.tobytes(), w, h, dxgi=28)); assert fmt == pt.PT_TEX_RGBA8_UNORM and np.array_equal(got, px)
    got, fmt = _mine(tmp_path, _dds(px.tobytes(), w, h, dxgi=29)); assert fmt == pt.PT_TEX_RGBA8_SRGB
    got, fmt = _mine(tmp_path, _dds(px[..., [2, 1, 0, 3]].tobytes(), w, h, masks=(0xFF0000, 0xFF00, 0xFF, 0xFF000000))); assert np.array_equal(got, px)
    got, fmt = _mine(tmp_path, _dds(px.tobytes(), w, h, masks=(0xFF, 0xFF00, 0xFF0000, 0xFF000000))); assert np.array_equal(got, px)
    f = rng.normal(size=(h, w, 4)).astype(np.float32) * 50
    got, fmt = _mine(tmp_path, _dds(f.tobytes(), w, h, dxgi=2)); assert fmt == pt.PT_TEX_RGBA32F and np.array_equal(got, f)
    got, fmt = _mine(tmp_path, _dds(f.tobytes(), w, h, fourcc=struct.pack("<I", 116))); assert np.array_equal(got, f)
    hf = f.astype(np.float16); hf[0, 0] = [np.float16(6e-8), np.float16(-0.0), np.float16(65504), np.float16(1e-5)]      # a denormal, -0, the largest half
    got, fmt = _mine(tmp_path, _dds(hf.tobytes(), w, h, dxgi=10)); assert fmt == pt.PT_TEX_RGBA32F and np.array_equal(got, hf.astype(np.float32))
    # ... and through the environment reader (RGB, alpha dropped); an 8-bit file is refused there
    (tmp_path / "e.dds").write_bytes(_dds(hf.tobytes(), w, h, fourcc=struct.pack("<I", 113)))
    assert np.array_equal(pt.read_float_image(tmp_path / "e.dds"), hf.astype(np.float32)[..., :3])
    (tmp_path / "e8.dds").write_bytes(_dds(px.tobytes(), w, h, dxgi=28))
    with pytest.raises(pt.PtError): pt.read_float_image(tmp_path / "e8.dds")


def test_top_level_of_a_mip_chain_and_odd_sizes(tmp_path):
    rng = np.random.default_rng(5); w, h = 10, 6                                     # 3 x 2 blocks, the last column / row partly outside
    blocks = rng.integers(0, 256, 3 * 2 * 16, dtype=np.uint8).tobytes(); tail = rng.integers(0, 256, 2 * 1 * 16 + 16 + 16, dtype=np.uint8).tobytes()
    data = _dds(blocks + tail, w, h, dxgi=98, mips=4)
    got, _ = _mine(tmp_path, data)
    full = np.asarray(_pil(_dds(blocks, 12, 8, dxgi=98)).convert("RGBA"))
    assert got.shape == (h, w, 4) and np.array_equal(got, full[:h, :w])


def test_damaged_and_unsupported_files(tmp_path):
    rng = np.random.default_rng(9); good = _dds(rng.integers(0, 256, 16 * 16, dtype=np.uint8).tobytes(), 16, 16, dxgi=98)
    for cut in (0, 3, 64, 127, 140, 148, 148 + 255):
        (tmp_path / "c.dds").write_bytes(good[:cut])
        with pytest.raises(pt.PtError): pt.read_dds(tmp_path / "c.dds")
    for dx in (81, 84, 61):                                                          # signed RGTC, R8_UNORM
        (tmp_path / "u.dds").write_bytes(_dds(b"\0" * 4096, 16, 16, dxgi=dx))
        with pytest.raises(pt.PtError) as e: pt.read_dds(tmp_path / "u.dds")
        assert e.value.code == pt.PT_ERROR_UNSUPPORTED
    bad = bytearray(good); bad[12:16] = struct.pack("<I", 0)                         # height 0
    (tmp_path / "z.dds").write_bytes(bytes(bad))
    with pytest.raises(pt.PtError): pt.read_dds(tmp_path / "z.dds")
    cube = bytearray(good); cube[112:116] = struct.pack("<I", 0x200 | 0xFC00)
    (tmp_path / "q.dds").write_bytes(bytes(cube))
    with pytest.raises(pt.PtError): pt.read_dds(tmp_path / "q.dds")
    for k in range(300):                                                              # random damage never crashes
        b = bytearray(good); i = int(rng.integers(0, 148)); b[i] = int(rng.integers(0, 256)); (tmp_path / "r.dds").write_bytes(bytes(b[: int(rng.integers(100, len(b) + 1))]))
        try: pt.read_dds(tmp_path / "r.dds")
        except pt.PtError: pass
    with pytest.raises(pt.PtError): pt.read_dds(tmp_path / "missing.dds")


def test_committed_bc7_tables_are_what_the_generator_produces(tmp_path):
    """rtxpt_amd/csrc/pt_bcn_tables.h is generated (tools/gen_bcn_tables.py reads the BC7 partition / anchor tables off Pillow's decoder): regenerating gives the committed file."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(os.path.join(root, "rtxpt_amd", "csrc", "pt_bcn_tables.h")).read()
    src = open(os.path.join(root, "tools", "gen_bcn_tables.py")).read().replace('os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rtxpt_amd", "csrc", "pt_bcn_tables.h")', repr(str(tmp_path / "t.h")))
    (tmp_path / "gen.py").write_text(src)
    subprocess.run([sys.executable, str(tmp_path / "gen.py")], check=True, capture_output=True)
    assert (tmp_path / "t.h").read_text() == committed


def test_gltf_msft_texture_dds(tmp_path):
    """MSFT_texture_dds (what Bistro's glTF carries; Donut's importer prefers the extension's image): the .dds image wins over `source`, also without a `source`; BC7 payload."""
    import json, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gltf_writer import write_gltf
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C2")
    write_gltf(sc, str(tmp_path / "c.gltf"))
    rng = np.random.default_rng(8); blocks = rng.integers(0, 256, 4 * 2 * 16, dtype=np.uint8); blocks[0::16] |= 1
    (tmp_path / "base.dds").write_bytes(_dds(blocks.tobytes(), 16, 8, dxgi=98))
    Image.fromarray(np.full((4, 4, 3), 200, np.uint8), "RGB").save(tmp_path / "base.png")
    doc = json.loads((tmp_path / "c.gltf").read_text())
    doc["images"] = [{"uri": "base.png"}, {"uri": "base.dds"}]
    doc["textures"] = [{"source": 0, "extensions": {"MSFT_texture_dds": {"source": 1}}}, {"extensions": {"MSFT_texture_dds": {"source": 1}}}]
    doc["materials"][0].setdefault("pbrMetallicRoughness", {})["baseColorTexture"] = {"index": 0}; doc["materials"][1]["emissiveTexture"] = {"index": 1}
    (tmp_path / "c.gltf").write_text(json.dumps(doc)); (tmp_path / "c.scene.json").write_text(json.dumps({"models": ["c.gltf"], "graph": [{"model": 0}]}))
    imp = pt.SceneImport(tmp_path / "c.scene.json")
    assert imp.info["numTextures"] == 1 and imp.info["texturesNotLoaded"] == 0            # one image, one sRGB flag: one texture (both slots are sRGB slots)
    px, fmt = imp.texture(0); want, _ = pt.read_dds(tmp_path / "base.dds")
    assert px.shape == (8, 16, 4) and np.array_equal(px, want) and fmt == pt.PT_TEX_RGBA8_SRGB
    word = scenes.pack_texture_word(0, 16, 8)
    assert imp.materials[0]["BaseOrDiffuseTextureIndex"] == word and imp.materials[1]["EmissiveTextureIndex"] == word
