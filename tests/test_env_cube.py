"""The environment cube: EnvMapBaker's lat-long -> RGBA16F cube bake, restated by the oracle (oracle/ptref/envcube.h,
ptref_api.cpp bake_env_cube), against the reference's own EnvMapBaker.hlsl text.

  * test_oracle_cube_matches_reference_text_golden: committed cubes (tests/golden/env_cube_golden.npz) baked by BaseLayerCS / MIPReduceCS of the reference
    text — runs everywhere, bit for bit.
  * test_oracle_cube_matches_live_reference_text: the same comparison on further inputs where /root/reference exists.
  * properties: alpha, fp16 range clamp, a constant source stays constant through the mips, the energy of a baked disc, cube addressing at texel centres.
The cube FETCH (TextureCube.SampleLevel: face selection, bilinear taps clamped to the face, trilinear between mips) is hardware behaviour the reference
has no text for; it is restated once in envcube.h and shared by the oracle and the reference-text integrator (hlsl_pt_wrappers.inc env_fetch)."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "env_cube_golden.npz")
CASES = pin_scenes.env_cube_cases()
HAVE_REF = os.path.isdir("/root/reference/Rtxpt/Shaders")


def _cube(sc, reference=False):
    o = ptref.Oracle(reference_integrator=True, settings=scenes.default_settings()) if reference else ptref.Oracle()
    o.set_scene(sc)
    return o.env_cube(reference=reference), o


def _half(cube):
    return cube.view(np.float16).astype(np.float32).reshape(-1, 4)


def _mip_slices(dim, levels):
    out, off = [], 0
    for l in range(levels):
        d = dim >> l; out.append((off, off + 6 * d * d, d)); off += 6 * d * d
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_cube_matches_reference_text_golden(name):
    g = np.load(GOLDEN)
    (cube, dim, levels), _ = _cube(CASES[name])
    assert (dim, levels) == tuple(int(v) for v in g[name + "_dim"])
    bad = (cube != g[name]).any(-1)
    assert not bad.any(), "%s: %d of %d texels differ from the reference-text bake" % (name, int(bad.sum()), bad.size)
    h = _half(cube)
    assert np.all(h[:, 3] == 1.0) and np.isfinite(h).all() and (h >= 0).all() and h[:, :3].max() > 0


@pytest.mark.skipif(not HAVE_REF, reason="no /root/reference on this machine: the reference text cannot be compiled here")
@pytest.mark.parametrize("dim,nlights,seed", [(16, 0, 1), (32, 3, 2), (128, 2, 3)])
def test_oracle_cube_matches_live_reference_text(dim, nlights, seed):
    rng = np.random.default_rng(seed)
    sc = dict(CASES["sky_32_discs"]); rgb, tw, cm = sc["env"]
    src = (rng.random((96, 192, 3), np.float32) ** 4 * 40.0).astype(np.float32)          # noisy HDR source: every bilinear tap matters
    d = rng.normal(size=(3, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    lights = np.concatenate([rng.random((3, 3)), rng.random((3, 1)) * 3, d, np.array([[0.02], [0.3], [1.2]])], axis=1).astype(np.float32)
    sc["env"] = (src, tw, cm); sc["env_cube_dim"] = dim; sc["env_directional_lights"] = lights[:nlights] if nlights else None
    (a, da, la), o = _cube(sc, reference=True)
    (b, _, _) = o.env_cube(reference=False)
    assert np.array_equal(a, b) and da == dim and la == {16: 2, 32: 3, 128: 5}[dim]


@pytest.mark.skipif(not HAVE_REF, reason="no /root/reference on this machine: the reference text cannot be compiled here")
@pytest.mark.parametrize("srcDim,dim,nlights,seed", [(8, 16, 0, 1), (33, 32, 3, 2), (128, 64, 1, 3)])
def test_oracle_cube_from_a_cube_map_source_matches_live_reference_text(srcDim, dim, nlights, seed):
    """BackgroundSourceType 2 (EnvMapBaker.hlsl:105-106): the image is a cube map (pt_set_environment_cube / scene key "env_cube_source")."""
    rng = np.random.default_rng(seed)
    sc = dict(CASES["cubesrc_32_discs"]); _, tw, cm = sc["env_cube_source"]
    faces = np.concatenate([(rng.random((6, srcDim, srcDim, 3), np.float32) ** 4 * 900.0).astype(np.float32), rng.random((6, srcDim, srcDim, 1), np.float32)], axis=-1)
    d = rng.normal(size=(3, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    lights = np.concatenate([rng.random((3, 3)), rng.random((3, 1)) * 3, d, np.array([[0.02], [0.3], [1.2]])], axis=1).astype(np.float32)
    sc["env_cube_source"] = (faces, tw, cm); sc["env_cube_dim"] = dim; sc["env_directional_lights"] = lights[:nlights] if nlights else None
    (a, da, _), o = _cube(sc, reference=True)
    (b, _, _) = o.env_cube(reference=False)
    assert np.array_equal(a, b) and da == dim and a.any()


def _dds_header(d, mips=1, dx10=None, caps2=0xFE00, fourcc=113):
    import struct
    flags = 0x1007 | (0x20000 if mips > 1 else 0)
    cc = b"DX10" if dx10 is not None else struct.pack("<I", fourcc)
    h = b"DDS " + struct.pack("<7I", 124, flags, d, d, 0, 0, mips) + b"\0" * 44 + struct.pack("<2I4s5I", 32, 4, cc, 0, 0, 0, 0, 0) + struct.pack("<5I", 0x1008, caps2, 0, 0, 0)
    if dx10 is not None: h += struct.pack("<5I", dx10[0], 3, dx10[1], dx10[2], 0)
    return h


def test_dds_cube_reader(tmp_path):
    """pt_image_read_dds_cube: the top level of all six faces — legacy FourCC 113 (RGBA16F) with the cube caps, a DX10 RGBA32F cube with a mip chain per face,
    a BC6H_UF16 cube (blocks from the library's own encoder, decoded like the 2D reader does); 2D files, partial cubes and cube arrays are refused."""
    import rtxpt_amd as pt
    rng = np.random.default_rng(9); d = 12
    faces = (rng.random((6, d, d, 4), np.float32) * 30.0).astype(np.float16)
    (tmp_path / "a.dds").write_bytes(_dds_header(d) + faces.tobytes())
    assert np.array_equal(pt.read_dds_cube(tmp_path / "a.dds"), faces.astype(np.float32))
    f32 = (rng.random((6, d, d, 4), np.float32) * 1e5).astype(np.float32); body = b""
    for f in range(6):                                    # every face carries its own mip chain (12, 6, 3, 1): the reader takes the first level and steps over the rest
        body += f32[f].tobytes() + b"".join(np.full((max(d >> l, 1), max(d >> l, 1), 4), 7.0 + l, np.float32).tobytes() for l in range(1, 4))
    (tmp_path / "b.dds").write_bytes(_dds_header(d, mips=4, dx10=(2, 0x4, 1)) + body)
    assert np.array_equal(pt.read_dds_cube(tmp_path / "b.dds"), f32)
    from oracle import ptref
    T = np.clip(rng.random((6 * 9, 16, 3), np.float32) * 20.0, 0, 65504).astype(np.float16).astype(np.float32)      # 3 x 3 blocks per face
    blocks = ptref.bc6_encode(T)
    (tmp_path / "c.dds").write_bytes(_dds_header(d, dx10=(95, 0x4, 1)) + blocks.astype(np.uint32).tobytes())
    got = pt.read_dds_cube(tmp_path / "c.dds")
    (tmp_path / "c2d.dds").write_bytes(_dds_header(d, dx10=(95, 0, 1), caps2=0)[:12] + np.array([d * 6], np.uint32).tobytes() + _dds_header(d, dx10=(95, 0, 1), caps2=0)[16:] + blocks.astype(np.uint32).tobytes())
    flat, fmt = pt.read_dds(tmp_path / "c2d.dds")         # the same blocks as one 12 x 72 image: the 2D reader's decode
    assert fmt == 2 and np.array_equal(got.reshape(6 * d, d, 4), flat) and np.all(got[..., 3] == 1.0)
    for name, data in (("flat.dds", _dds_header(d, caps2=0) + faces[0].tobytes()), ("partial.dds", _dds_header(d, caps2=0x0600) + faces[:1].tobytes()),
                       ("array.dds", _dds_header(d, dx10=(10, 0x4, 2)) + faces.tobytes() * 2), ("rgba8.dds", _dds_header(d, dx10=(28, 0x4, 1)) + bytes(6 * d * d * 4))):
        (tmp_path / name).write_bytes(data)
        with pytest.raises(pt.PtError) as e: pt.read_dds_cube(tmp_path / name)
        assert e.value.code == pt.PT_ERROR_UNSUPPORTED, name
    (tmp_path / "short.dds").write_bytes(_dds_header(d) + faces.tobytes()[:-8])
    with pytest.raises(pt.PtError) as e: pt.read_dds_cube(tmp_path / "short.dds")
    assert e.value.code == pt.PT_ERROR_IO


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_importance_map_matches_reference_text_golden(name):
    """Level 0 of the radiance / importance map the environment quad-tree lights are made from: BuildMIPDescentImportanceMapCS of the reference's
    EnvMapImportanceSamplingBaker.hlsl (4 x 4 cube fetches per texel through the equal-area octahedral map, the RGBA16_FLOAT store) == the oracle's, bit for bit."""
    g = np.load(GOLDEN)
    (_, _, _), o = _cube(CASES[name])
    got = o.env_importance(64)
    assert np.array_equal(got.view(np.uint32), g[name + "_importance64"].view(np.uint32))
    assert np.array_equal(got, got.astype(np.float16).astype(np.float32)) and got[..., 3].min() > 0          # every value is a binary16 value


@pytest.mark.skipif(not HAVE_REF, reason="no /root/reference on this machine: the reference text cannot be compiled here")
def test_oracle_importance_map_matches_live_reference_text():
    (_, _, _), o = _cube(CASES["sky_64_hdr_sun"], reference=True)
    for dim in (32, 256):
        assert np.array_equal(o.env_importance(dim).view(np.uint32), o.env_importance(dim, reference=True).view(np.uint32))


def test_cube_clamps_to_fp16_range_and_scales_by_quarter():
    """GenerateTexel: radiance x c_envMapRadianceScale (1/4, Sample.cpp:88) clamped to [0, HLF_MAX] (EnvMapBaker.hlsl:236-242)."""
    sc = dict(CASES["sky_16"]); rgb, tw, cm = sc["env"]
    src = np.full((32, 64, 3), 8.0, np.float32); src[:16] = 1e6
    sc["env"] = (src, tw, cm); sc["env_cube_dim"] = 32
    (cube, dim, levels), _ = _cube(sc)
    h = _half(cube); sl = _mip_slices(dim, levels)
    top = h[sl[0][0]:sl[0][1]].reshape(6, dim, dim, 4)
    assert np.all(top[3, :, :, :3] == 2.0)                       # -Y face: 8 x 1/4
    assert np.all(top[2, :, :, :3] == 65504.0)                   # +Y face: clamped
    assert h.max() == 65504.0 and np.isfinite(h).all()


def test_constant_source_stays_constant_through_the_mips():
    sc = dict(CASES["sky_16"]); rgb, tw, cm = sc["env"]
    sc["env"] = (np.full((16, 32, 3), 1.7, np.float32) * np.array([1.0, 2.0, 3.0], np.float32), tw, cm); sc["env_cube_dim"] = 64
    (cube, dim, levels), _ = _cube(sc)
    h = _half(cube); want = np.array([1.7, 3.4, 5.1], np.float32) * 0.25
    for a, b, d in _mip_slices(dim, levels):
        assert np.abs(h[a:b, :3] / want - 1.0).max() < 2e-3, d         # fp16 rounding of the weighted mean of equal values: 1 ulp at most
    assert levels == 4 and (dim >> (levels - 1)) == 8                  # mips stop at 8x8 (EnvMapBaker.cpp:318-320)


def test_baked_disc_carries_the_lights_energy():
    """ComputeLightContribution draws radiance = colour x intensity / solidAngle(disc): integrated over the cube's texels (x4 to undo the radiance scale)
    the disc returns colour x intensity, within the coverage approximation the reference documents as 'not physically correct' (EnvMapBaker.hlsl:164-165)."""
    sc = dict(CASES["sky_16"]); rgb, tw, cm = sc["env"]
    d = np.array([0.3, -0.7, 0.2]); d /= np.linalg.norm(d)
    base = dict(sc); base["env"] = (np.zeros((16, 32, 3), np.float32), tw, cm); base["env_cube_dim"] = 256
    lit = dict(base); lit["env_directional_lights"] = np.array([[1.0, 0.5, 0.25, 2.0, d[0], d[1], d[2], 0.25]], np.float32)
    (cube, dim, levels), _ = _cube(lit)
    h = _half(cube)[:6 * dim * dim, :3].reshape(6, dim, dim, 3).astype(np.float64)
    c = (np.arange(dim) + 0.5) * 2.0 / dim - 1.0
    area = lambda x, y: np.arctan2(x * y, np.sqrt(x * x + y * y + 1.0))
    x0, x1 = c - 1.0 / dim, c + 1.0 / dim
    sa = area(x0[None, :], x0[:, None]) - area(x0[None, :], x1[:, None]) - area(x1[None, :], x0[:, None]) + area(x1[None, :], x1[:, None])
    assert abs(6 * sa.sum() - 4 * np.pi) < 1e-9
    e = (h * sa[None, :, :, None]).sum(axis=(0, 1, 2)) * 4.0
    assert np.allclose(e, np.array([1.0, 0.5, 0.25]) * 2.0, rtol=0.06), e
    # and the disc sits where the light comes from (-Direction)
    f, y, x = np.unravel_index(np.argmax(h[..., 0]), h.shape[:3])
    assert f == 2 and -d[1] > max(abs(d[0]), abs(d[2]))              # +Y face


def test_cube_fetch_at_texel_centres_and_integer_lods_returns_the_texel():
    """TextureCube.SampleLevel restated (envcube.h): a direction through a texel centre at an integer lod reads exactly that texel; lod clamps to the chain."""
    sc = CASES["sky_32_discs"]
    (cube, dim, levels), o = _cube(sc)
    h = _half(cube); cm4 = np.asarray(sc["env"][2], np.float32) * np.float32(4.0)
    rng = np.random.default_rng(5)
    rows, want = [], []
    for a, b, d in _mip_slices(dim, levels):
        lod = int(np.log2(dim // d))
        for _ in range(64):
            f, y, x = int(rng.integers(6)), int(rng.integers(d)), int(rng.integers(d))
            cx, cy = (x + 0.5) * 2.0 / d - 1.0, 1.0 - (y + 0.5) * 2.0 / d
            dirv = [(1, cy, -cx), (-1, cy, cx), (cx, 1, -cy), (cx, -1, cy), (cx, cy, 1), (-cx, cy, -1)][f]
            rows.append([dirv[0] * 3.0, dirv[1] * 3.0, dirv[2] * 3.0, float(lod)])       # unnormalised: the fetch only uses ratios
            want.append(h[a + (f * d + y) * d + x, :3] * cm4)
    got = o.env_eval(np.array(rows, np.float32))
    assert np.allclose(got, np.array(want, np.float32), rtol=2e-6, atol=0)
    far = o.env_eval(np.array([[0.2, 0.9, 0.1, 50.0], [0.2, 0.9, 0.1, float(levels - 1)], [0.2, 0.9, 0.1, -3.0], [0.2, 0.9, 0.1, 0.0]], np.float32))
    assert np.array_equal(far[0], far[1]) and np.array_equal(far[2], far[3])


def test_cube_fetch_is_continuous_across_face_edges():
    """Bilinear taps are clamped to the face the direction selects (no cross-face filtering in this restatement); across an edge the two faces' border
    texels are neighbours in the source image, so the fetch may step by at most their difference — and must never read outside the cube."""
    sc = CASES["sky_32_discs"]
    _, o = _cube(sc)
    t = np.linspace(-1.0, 1.0, 257, dtype=np.float32)
    eps = np.float32(1e-3)
    a = o.env_eval(np.stack([np.ones_like(t), t, np.full_like(t, 1.0 - eps), np.zeros_like(t)], 1))      # just on the +X side of the +X/+Z edge
    b = o.env_eval(np.stack([np.full_like(t, 1.0 - eps), t, np.ones_like(t), np.zeros_like(t)], 1))      # just on the +Z side
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.abs(a - b).max() <= 0.35 * max(a.max(), b.max())


# ---- EnvMapBaker's BC6U compression of the cube (EnvMapBaker.cpp:593-633; BC6UCompress.hlsl "Fast" = EncodeP1, one-region mode 11) --------------------------------------

def test_bc6_encoder_matches_reference_text_golden():
    """The oracle's EncodeP1 against blocks encoded by the reference's BC6UCompress.hlsl text (committed; 1 600 blocks of eight kinds incl. flat and black ones)."""
    g = np.load(GOLDEN)
    got = ptref.bc6_encode(g["bc6_texels"])
    bad = (got != g["bc6_blocks"]).any(1)
    assert not bad.any(), "%d of %d blocks differ" % (int(bad.sum()), len(bad))
    assert ((got[:, 0] & 31) == 3).all()                                    # mode 11


def test_bc6_quality_encoder_matches_reference_text_golden():
    """QUALITY 1 (the reference's "Quality" setting): EncodeP1, then the best-scoring of the 32 two-region partitions encoded in modes 7.6 / 9.5 where its estimate is lower —
    against blocks from the reference's text (committed). A fair share of the blocks takes a two-region mode, in both of them."""
    g = np.load(GOLDEN)
    got = ptref.bc6_encode(g["bc6_texels"], quality=True)
    bad = (got != g["bc6_blocks_quality"]).any(1)
    assert not bad.any(), "%d of %d blocks differ" % (int(bad.sum()), len(bad))
    m76, m95, m11 = ((got[:, 0] & 3) == 1), ((got[:, 0] & 31) == 0xE), ((got[:, 0] & 31) == 3)
    assert (m76 | m95 | m11).all() and m76.sum() > 100 and m95.sum() > 50 and m11.sum() > 500
    assert np.array_equal(got[m11], g["bc6_blocks"][m11])                        # where one region wins, the block is the "Fast" one


@pytest.mark.skipif(not HAVE_REF, reason="no /root/reference on this machine: the reference text cannot be compiled here")
def test_bc6_quality_encoder_matches_live_reference_text():
    rng = np.random.default_rng(45); n = 2000
    T = np.zeros((n, 16, 3), np.float32)
    for k in range(n):                                                            # two colour regions behind a random straight edge, a little noise, HDR scale
        c0, c1 = rng.uniform(0, 1, 3), rng.uniform(0, 1, 3); a = rng.uniform(0, 2 * np.pi); off = rng.uniform(-1.5, 1.5)
        side = ((np.arange(16) % 4 - 1.5) * np.cos(a) + (np.arange(16) // 4 - 1.5) * np.sin(a)) > off
        T[k] = np.where(side[:, None], c0, c1) * rng.uniform(0.9, 1.1, (16, 3)) * 10 ** rng.uniform(-2, 3)
    T = T.astype(np.float16).astype(np.float32); T[::19] = T[::19, :1]
    a, b = ptref.bc6_encode(T, quality=True), ptref.bc6_encode(T, reference=True, quality=True)
    assert np.array_equal(a, b) and ((a[:, 0] & 31) != 3).mean() > 0.3


def test_bc6_two_region_decode_against_an_independent_decoder():
    """The decode of the modes QUALITY 1 adds (7.6: 2-bit mode field 01; 9.5: 01110 — delta endpoints, partition table, 3-bit indices, two anchor texels) against Pillow's
    BC6H decoder at its 8-bit resolution, on blocks made of two colour regions; and the round trip is closer to the source than the one-region mode's."""
    PIL = pytest.importorskip("PIL.Image")
    import io, struct
    rng = np.random.default_rng(7); n = 1024
    T = np.zeros((n, 16, 3), np.float32)
    for k in range(n):
        c0, c1 = rng.uniform(0, 1, 3), rng.uniform(0, 1, 3); a = rng.uniform(0, 2 * np.pi); off = rng.uniform(-1, 1)
        side = ((np.arange(16) % 4 - 1.5) * np.cos(a) + (np.arange(16) // 4 - 1.5) * np.sin(a)) > off
        T[k] = np.clip(np.where(side[:, None], c0, c1) * rng.uniform(0.9, 1.1, (16, 3)), 0, 1)
    T = T.astype(np.float16).astype(np.float32)
    blk = ptref.bc6_encode(T, quality=True)
    two = (blk[:, 0] & 31) != 3
    assert ((blk[:, 0] & 3) == 1).sum() > 100 and ((blk[:, 0] & 31) == 0xE).sum() > 100
    mine = ptref.bc6_decode(blk).astype(np.uint16).view(np.float16).astype(np.float32).reshape(n, 16, 3)
    hdr = b"DDS " + struct.pack("<7I", 124, 0x1007, 128, 128, 0, 0, 1) + b"\0" * 44 + struct.pack("<2I4s5I", 32, 4, b"DX10", 0, 0, 0, 0, 0) + struct.pack("<5I", 0x1000, 0, 0, 0, 0) + struct.pack("<5I", 95, 3, 0, 1, 0)
    im = np.asarray(PIL.open(io.BytesIO(hdr + blk.tobytes())).convert("RGB")).astype(int)
    pil = im.reshape(32, 4, 32, 4, 3).transpose(0, 2, 1, 3, 4).reshape(n, 16, 3)
    assert np.array_equal(np.floor(np.clip(mine, 0, 1) * 255).astype(int), pil)
    fast = ptref.bc6_decode(ptref.bc6_encode(T)).astype(np.uint16).view(np.float16).astype(np.float32).reshape(n, 16, 3)
    assert np.abs(mine[two] - T[two]).mean() < 0.8 * np.abs(fast[two] - T[two]).mean()


@pytest.mark.skipif(not HAVE_REF, reason="no /root/reference on this machine: the reference text cannot be compiled here")
def test_bc6_encoder_matches_live_reference_text():
    rng = np.random.default_rng(44)
    T = (rng.uniform(0, 1, (3000, 1, 3)) * 10 ** rng.uniform(-3, 3.5, (3000, 1, 1)) * rng.uniform(0.2, 1.8, (3000, 16, 3)) ** rng.integers(1, 4, (3000, 1, 1))).astype(np.float16).astype(np.float32)
    T[::17] = T[::17, :1]                                                     # flat blocks
    assert np.array_equal(ptref.bc6_encode(T), ptref.bc6_encode(T, reference=True))


def test_bc6_mode11_decode_against_an_independent_decoder():
    """What a BC6H_UF16 fetch returns for the encoder's blocks: the oracle's decode against Pillow's BC6H decoder — which outputs 8 bits (clamp to [0, 1], x 255, truncated),
    so this checks bit layout, unquantisation and interpolation at that resolution (the blocks are white noise: how close a round trip stays is checked on cubes below)."""
    PIL = pytest.importorskip("PIL.Image")
    import io, struct
    rng = np.random.default_rng(5); n = 1024
    T = (rng.uniform(0, 1.2, (n, 1, 3)) * rng.uniform(0.3, 1.0, (n, 16, 3))).astype(np.float16).astype(np.float32)
    blk = ptref.bc6_encode(T)
    mine = ptref.bc6_decode(blk).astype(np.uint16).view(np.float16).astype(np.float32).reshape(n, 16, 3)
    hdr = b"DDS " + struct.pack("<7I", 124, 0x1007, 128, 128, 0, 0, 1) + b"\0" * 44 + struct.pack("<2I4s5I", 32, 4, b"DX10", 0, 0, 0, 0, 0) + struct.pack("<5I", 0x1000, 0, 0, 0, 0) + struct.pack("<5I", 95, 3, 0, 1, 0)
    im = np.asarray(PIL.open(io.BytesIO(hdr + blk.tobytes())).convert("RGB")).astype(int)
    pil = im.reshape(32, 4, 32, 4, 3).transpose(0, 2, 1, 3, 4).reshape(n, 16, 3)
    assert np.array_equal(np.floor(np.clip(mine, 0, 1) * 255).astype(int), pil)


@pytest.mark.parametrize("name", ["sky_32_discs_bc6", "sky_64_hdr_sun_bc6", "sky_32_discs_bc6q", "sky_64_hdr_sun_bc6q"])
def test_compressed_cube_and_importance_map(name):
    """With compression on, the sampled cube is the reference-text bake sent through the encoder text and the decode (committed golden); the importance map is built from the
    UNCOMPRESSED cube (EnvMapBaker.cpp:635) and therefore equals the uncompressed case's."""
    g = np.load(GOLDEN); sc = CASES[name]; base = name.rsplit("_bc6", 1)[0]
    (cube, dim, levels), o = _cube(sc)
    assert np.array_equal(cube, g[name]) and (cube != g[base]).any(-1).mean() > 0.9
    h = _half(cube); assert np.all(h[:, 3] == 1.0) and np.isfinite(h).all() and (h >= 0).all()
    assert np.array_equal(o.env_importance(64), g[base + "_importance64"]) and np.array_equal(g[name + "_importance64"], g[base + "_importance64"])
    rel = np.abs(_half(cube)[:, :3] - _half(g[base])[:, :3]) / (_half(g[base])[:, :3] + 1e-3)
    assert np.median(rel) < 0.02
