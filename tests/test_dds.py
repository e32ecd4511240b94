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
    for dx in (95, 96, 81, 84, 61):                                                  # BC6H, signed RGTC, R8_UNORM
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
