"""GPU parity of the environment-cube bake (run with -m gpu): k_env_cube_base / k_env_cube_mip and the device cube fetch against the oracle and against
the committed reference-text cubes (tests/golden/env_cube_golden.npz), bit for bit; then the sizes the bench uses (2048: EnvMapBaker's resolution for an
image source)."""
import os, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "env_cube_golden.npz")


def _imports():
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    from oracle import ptref
    import pin_scenes
    return pt, scenes, ptref, pin_scenes


@pytest.mark.parametrize("name", ["cubesrc_32_discs", "cubesrc_64_bc6", "sky_16", "sky_32_discs", "sky_64_hdr_sun", "sky_32_discs_bc6", "sky_64_hdr_sun_bc6", "sky_32_discs_bc6q", "sky_64_hdr_sun_bc6q", "procsky_64_midday", "procsky_32_clock_image_discs_bc6"])
def test_device_cube_matches_reference_text_golden_and_oracle(name):
    pt, scenes, ptref, pin_scenes = _imports()
    sc = pin_scenes.env_cube_cases()[name]
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(scenes.default_settings())
    cube, dim, levels = g.env_cube()
    gold = np.load(GOLDEN)
    assert (dim, levels) == tuple(int(v) for v in gold[name + "_dim"])
    bad = (cube != gold[name]).any(-1)
    assert not bad.any(), "%s: %d of %d texels differ from the reference-text bake" % (name, int(bad.sum()), bad.size)
    o = ptref.Oracle(); o.set_scene(sc)
    assert np.array_equal(cube, o.env_cube()[0])


def test_device_cube_fetch_matches_oracle():
    """EnvMap::EvalLocal on the device (pt_probe kind 9) == the oracle's, on random directions and lods (fractional, negative, beyond the chain), on
    directions along face edges and corners, and on axis-aligned ones."""
    pt, scenes, ptref, pin_scenes = _imports()
    sc = pin_scenes.env_cube_cases()["sky_64_hdr_sun"]
    g = pt.PathTracer(test_hooks=True); g.set_scene(sc); g.set_settings(scenes.default_settings())
    o = ptref.Oracle(); o.set_scene(sc)
    rng = np.random.default_rng(11)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    lod = (rng.random(20000) * 6.0 - 1.0).astype(np.float32); lod[::7] = np.floor(lod[::7])
    edge = np.array([[1, 1, 0.3], [1, -1, 0.3], [1, 1, 1], [-1, 1, -1], [0, 0, 1], [0, 1, 0], [-1, 0, 0], [1, 0.999999, 0.2], [0.5, 0.5, 0.5000001], [1e-20, 1, 1e-20]], np.float32)
    rows = np.concatenate([np.concatenate([d, lod[:, None]], 1), np.concatenate([edge, np.zeros((len(edge), 1), np.float32)], 1),
                           np.concatenate([edge, np.full((len(edge), 1), 1.5, np.float32)], 1)]).astype(np.float32)
    got = g.probe(9, rows, (len(rows), 3))
    want = o.env_eval(rows)
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%d of %d fetches differ; first %s: %s vs %s" % (int(bad.sum()), len(rows), rows[bad][0], got[bad][0], want[bad][0])


def test_device_cube_2048_with_discs_matches_oracle_and_rebakes_on_change():
    """EnvMapBaker's resolution for an image source (2048, EnvMapBaker.cpp:374-375) with directional lights: 28 M texels equal the oracle's; changing the lights
    re-bakes (and the environment quad-tree lights with it); a frame rendered afterwards equals the oracle's."""
    pt, scenes, ptref, pin_scenes = _imports()
    make, S, w, h, first, n = pin_scenes.cases()["c2_sun_discs"]
    sc, cam = make(); sc = dict(sc); sc["env_cube_dim"] = 2048
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h)
    cube, dim, levels = g.env_cube()
    assert (dim, levels) == (2048, 9) and cube.shape[0] == sum(6 * (2048 >> l) ** 2 for l in range(9))
    assert np.array_equal(cube, o.env_cube()[0])
    g.render(first, n); o.render(first, n)
    assert np.array_equal(g.radiance().view(np.uint32), o.radiance().view(np.uint32))
    lights0 = g.lights()
    sc2 = dict(sc); sc2["env_directional_lights"] = sc["env_directional_lights"][:1] * np.array([1, 1, 1, 3.0, 1, 1, 1, 2.0], np.float32)
    g.set_scene(sc2); o.set_scene(sc2)
    assert np.array_equal(g.env_cube()[0], o.env_cube()[0]) and not np.array_equal(g.env_cube()[0], cube)
    lights1 = g.lights()
    assert lights0["lights"].shape == lights1["lights"].shape and not np.array_equal(lights0["lights"], lights1["lights"])      # the quad-tree lights are made from the cube
    g.reset_accumulation(); o.reset_accumulation(); g.render(first, n); o.render(first, n)
    assert np.array_equal(g.radiance().view(np.uint32), o.radiance().view(np.uint32))


def test_environment_bake_argument_validation():
    pt, scenes, ptref, pin_scenes = _imports()
    import ctypes
    g = pt.PathTracer()
    lights = (ctypes.c_float * (8 * 17))()
    assert g.L.pt_set_environment_bake(g.h, 100, None, 0) != 0            # not a power of two
    assert g.L.pt_set_environment_bake(g.h, 8, None, 0) != 0              # below the 8x8 mip floor
    assert g.L.pt_set_environment_bake(g.h, 16384, None, 0) != 0
    assert g.L.pt_set_environment_bake(g.h, 256, None, 2) != 0            # lights missing
    assert g.L.pt_set_environment_bake(g.h, 256, lights, 17) != 0         # EMB_MAXDIRLIGHTS
    assert g.L.pt_set_environment_bake(g.h, 0, lights, 16) == 0           # 0 keeps the resolution
    assert g.L.pt_set_environment_bake(g.h, 256, None, 0) == 0
    n = ctypes.c_uint32(7)
    assert g.L.pt_get_env_cube(g.h, ctypes.byref(n), None, None, None, 0) == 0 and n.value == 0      # no environment set: no cube


def test_compressed_cube_frames_and_switching():
    """pt_set_environment_compression: the frame with the BC6H cube equals the oracle's (and the reference-text golden), the light tables do not change (the importance map reads the
    uncompressed cube), switching back restores the uncompressed frame, quality 2 is refused."""
    pt, scenes, ptref, pin_scenes = _imports()
    make, S, w, h, first, n = pin_scenes.cases()["c2_sun_discs_bc6"]
    sc, cam = make(); camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h); g.render(first, n)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(first, n)
    on = g.radiance().copy()
    assert np.array_equal(on, o.radiance()) and np.array_equal(g.env_cube()[0], o.env_cube()[0])
    lights_on = g.light_tables() if hasattr(g, "light_tables") else None
    assert g.L.pt_set_environment_compression(g.h, 0) == 0
    g.reset_accumulation(); g.render(first, n); off = g.radiance().copy()
    sc0 = dict(sc); sc0["env_compression"] = 0
    o0 = ptref.Oracle(); o0.set_scene(sc0); o0.set_camera(camd); o0.set_settings(S); o0.resize(w, h); o0.render(first, n)
    assert np.array_equal(off, o0.radiance()) and not np.array_equal(on, off)
    assert g.L.pt_set_environment_compression(g.h, 1) == 0
    g.reset_accumulation(); g.render(first, n); assert np.array_equal(g.radiance(), on)
    assert g.L.pt_set_environment_compression(g.h, 2) == 0                                    # "Quality": the two-region modes
    g.reset_accumulation(); g.render(first, n); q = g.radiance().copy()
    sc2 = dict(sc); sc2["env_compression"] = 2
    o2 = ptref.Oracle(); o2.set_scene(sc2); o2.set_camera(camd); o2.set_settings(S); o2.resize(w, h); o2.render(first, n)
    assert np.array_equal(q, o2.radiance()) and np.array_equal(g.env_cube()[0], o2.env_cube()[0]) and not np.array_equal(q, on)
    assert g.L.pt_set_environment_compression(g.h, 3) == pt.PT_ERROR_INVALID_ARGUMENT


def test_cube_map_source_frame_and_switching(tmp_path):
    """pt_set_environment_cube (EnvMapBaker.hlsl BackgroundSourceType 2): a frame lit by an environment whose image is a cube map read back from a .dds file equals the oracle's,
    cube and light tables included; a lat-long image replaces the cube source and the other way round; dim 0 switches the environment off."""
    import struct
    pt, scenes, ptref, pin_scenes = _imports()
    make, S, w, h, first, n = pin_scenes.cases()["c2"]
    sc, cam = make(); sc = dict(sc); camd = scenes.bridge_camera(w, h, **cam)
    rgb, tw, cm = sc["env"]
    rng = np.random.default_rng(5); d = 40
    faces = np.concatenate([(rng.random((6, d, d, 3), np.float32) ** 3 * 6.0).astype(np.float16).astype(np.float32), np.ones((6, d, d, 1), np.float32)], axis=-1)
    hdr = b"DDS " + struct.pack("<7I", 124, 0x1007, d, d, d * 8, 0, 1) + b"\0" * 44 + struct.pack("<2I4s5I", 32, 4, struct.pack("<I", 113), 0, 0, 0, 0, 0) + struct.pack("<5I", 0x1008, 0xFE00, 0, 0, 0)
    (tmp_path / "sky.dds").write_bytes(hdr + faces.astype(np.float16).tobytes())
    got = pt.read_dds_cube(tmp_path / "sky.dds")
    assert np.array_equal(got, faces)
    sc_cube = dict(sc); sc_cube["env"] = None; sc_cube["env_cube_source"] = (got, tw, cm); sc_cube["env_cube_dim"] = 128
    g = pt.PathTracer(); g.set_scene(sc_cube); g.set_camera(camd); g.set_settings(S); g.resize(w, h); g.render(first, n)
    o = ptref.Oracle(); o.set_scene(sc_cube); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(first, n)
    a = g.radiance().copy()
    assert np.array_equal(g.env_cube()[0], o.env_cube()[0]) and np.array_equal(a, o.radiance()) and float(a[..., :3].sum()) > 0
    sc_ll = dict(sc); sc_ll["env_cube_dim"] = 128
    g.set_scene(sc_ll); g.reset_accumulation(); g.render(first, n); b = g.radiance().copy()
    o2 = ptref.Oracle(); o2.set_scene(sc_ll); o2.set_camera(camd); o2.set_settings(S); o2.resize(w, h); o2.render(first, n)
    assert np.array_equal(b, o2.radiance()) and not np.array_equal(a, b)
    g.set_scene(sc_cube); g.reset_accumulation(); g.render(first, n)
    assert np.array_equal(g.radiance(), a)
    assert g.L.pt_set_environment_cube(g.h, None, 0, None) == 0
    g.reset_accumulation(); g.render(first, n); dark = g.radiance().copy()
    sc_off = dict(sc); sc_off["env"] = None
    o3 = ptref.Oracle(); o3.set_scene(sc_off); o3.set_camera(camd); o3.set_settings(S); o3.resize(w, h); o3.render(first, n)
    assert np.array_equal(dark, o3.radiance())


def test_procedural_sky_cube_at_1024_and_a_frame_lit_by_it():
    """EnvMapBaker's resolution for a procedural sky (1024, EnvMapBaker.cpp:374-375): k_env_sky_lowres + the sky term of the base layer equal the oracle's cube
    (6.3 M texels x 8 levels, pinned to the reference text on the CPU side); a frame lit by the sky alone equals the oracle's; updating the constants
    without textures re-bakes; switching the sky off leaves no environment."""
    pt, scenes, ptref, pin_scenes = _imports()
    consts, _ = pt.procedural_sky_update(pt.PtProceduralSkyState(), 0.0, "==PROCEDURAL_SKY_MIDDAY==", force_instant=True)
    sc, cam = scenes.cornell_box("C2"); sc = dict(sc); sc["env"] = None
    sc["sky"] = {"consts": consts, "textures": pin_scenes._sky_textures()}; sc["env_cube_dim"] = 1024
    S = scenes.config_settings("C2")
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(S)
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(S)
    cube, dim, levels = g.env_cube(); want, d2, l2 = o.env_cube()
    assert (dim, levels) == (1024, 8) == (d2, l2)
    bad = (cube != want).any(-1)
    assert not bad.any(), "%d of %d texels differ" % (int(bad.sum()), bad.size)
    assert cube.view(np.float16).astype(np.float32).reshape(-1, 4)[:, :3].max() > 1000.0      # the sun disc
    w, h = 96, 64; camd = scenes.bridge_camera(w, h, **cam)
    g.set_camera(camd); g.resize(w, h); g.render(0, 2)
    o.set_camera(camd); o.resize(w, h); o.render(0, 2)
    a, b = g.radiance(), o.radiance()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a[..., :3].max() > 0
    # evening constants, textures kept: the cube follows
    c2, _ = pt.procedural_sky_update(pt.PtProceduralSkyState(), 0.0, "==PROCEDURAL_SKY_EVENING==", force_instant=True)
    g.set_procedural_sky(c2); o.set_procedural_sky(c2)
    cube2 = g.env_cube()[0]
    assert (cube2 != cube).any() and np.array_equal(cube2, o.env_cube()[0])
    g.set_procedural_sky(None); g.reset_accumulation(); g.render(0, 1)
    assert g.env_cube()[1] == 0                                                               # no image, no sky: the environment is off
    g.close(); o.close()
