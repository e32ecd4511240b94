"""GPU edge cases (run with -m gpu): inputs at the boundaries of the data contract, each compared with the oracle bit for bit.

  * empty scene (no geometry): every ray misses; environment only / nothing at all
  * frame sizes that are not a multiple of the 32x32 tile or of the 64-ray traversal chunk, down to 1x1, on 1 and 3 shards
  * degenerate settings: 0 bounces, NEE off, Russian roulette off, 1 NEE candidate
  * a stack-depth stress scene (300k mutually overlapping triangles) that drives the traversal stack through its LDS part into the
    global-memory tail (pt_traverse8.h: BVH8_STACK + T8_SPILL_DEPTH) — closest hits and visibility must still equal the oracle
  * degenerate triangles (zero area, duplicated vertices) and rays parallel to triangles / starting on them
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _imports():
    import rtxpt_amd as pt
    from rtxpt_amd import scenes, parallel
    from oracle import ptref
    return pt, scenes, parallel, ptref


def _both(sc, cam, S, w, h, first=0, count=2):
    pt, scenes, parallel, ptref = _imports()
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h); st = g.render(first, count)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(first, count)
    a, b = g.radiance(), o.radiance()
    assert not np.isnan(a).any()
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), "%d pixels differ" % int((a != b).any(-1).sum())
    c = o.counters()
    assert st["extendRays"] == c["extendRays"] and st["shadowRays"] == c["shadowRays"]
    return g, o, st


CAM = dict(pos=(0.0, 1.0, -4.0), direction=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0), fov_y=0.9)


def test_empty_scene_env_only_and_nothing():
    pt, scenes, parallel, ptref = _imports()
    b = scenes.SceneBuilder()
    b.add_material(scenes.make_material())
    b.set_environment(scenes.sky_equirect(128, 64))
    g, o, st = _both(b.finish(), CAM, scenes.default_settings(), 70, 45)
    assert st["hits"] == 0 and st["extendRays"] == 70 * 45 * 2 and st["shadowRays"] == 0
    assert g.radiance()[..., :3].max() > 0
    b2 = scenes.SceneBuilder(); b2.add_material(scenes.make_material())
    g2, o2, st2 = _both(b2.finish(), CAM, scenes.default_settings(), 33, 17)
    assert g2.radiance()[..., :3].max() == 0


@pytest.mark.parametrize("size", [(1, 1), (33, 17), (64, 1), (97, 65)])
def test_ragged_frame_sizes(size):
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    w, h = size
    g, o, st = _both(sc, cam, scenes.config_settings("C2"), w, h, first=3, count=2)
    # the same frame from 3 shards, reassembled (tiles are 32x32: most of these frames are a single partial tile)
    ref = g.radiance()
    camd = scenes.bridge_camera(w, h, **cam)
    img = np.zeros_like(ref)
    for r in range(3):
        gs = pt.PathTracer(shard_rank=r, shard_count=3); gs.set_scene(sc); gs.set_camera(camd); gs.set_settings(scenes.config_settings("C2")); gs.resize(w, h)
        gs.render(3, 2)
        px = parallel.shard_pixels(w, h, r, 3)                      # x << 16 | y
        yy, xx = (px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)
        assert gs.shard_info()[0] == px.size
        img[yy, xx] = gs.radiance()[yy, xx]
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("kw", [dict(bounceCount=0, diffuseBounceCount=0), dict(NEEEnabled=0), dict(enableRussianRoulette=0, bounceCount=3, diffuseBounceCount=3),
                                dict(NEECandidateSamples=1), dict(NEEType=0), dict(nestedDielectricsQuality=0), dict(enableLDSamplerForBSDF=0)])
def test_degenerate_settings(kw):
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    _both(sc, cam, scenes.default_settings(**kw), 96, 54, first=0, count=2)


@pytest.mark.parametrize("full", [0, 2, 3, 8, 100])
def test_nee_multiple_full_samples(full):
    """HandleNEE_MultipleSamples (PathTracerNEE.hlsli:277-301): grouped shadow queue, fp16 NEEResult accumulation in sample order; 100 clamps to 63."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    g, o, st = _both(sc, cam, scenes.default_settings(NEEFullSamples=full), 96, 54, first=1, count=2)
    if full == 0: assert st["shadowRays"] == 0
    # the bistro-like scene: alpha-tested occluders, emissive triangles + environment quads, stragglers split into sub-tree tasks
    if full in (2, 8):
        sc3, cam3 = scenes.bistro_like(scale=0.02, tex_size=128)
        _both(sc3, cam3, scenes.default_settings(NEEFullSamples=full, NEECandidateSamples=3), 128, 72, first=0, count=1)
    # switching back to one sample on the same context returns to the ungrouped queue
    S1 = scenes.default_settings(); g.set_settings(S1); g.reset_accumulation(); g.render(1, 2)
    o.set_settings(S1); o.reset_accumulation(); o.render(1, 2)
    assert (g.radiance().view(np.uint32) == o.radiance().view(np.uint32)).all()


def _overlap_scene(n_tris, seed):
    """n_tris large triangles through one region: every BVH level overlaps, so a central ray has to keep most children of every node."""
    pt, scenes, parallel, ptref = _imports()
    rng = np.random.default_rng(seed)
    c = rng.normal(scale=0.05, size=(n_tris, 1, 3)) + np.array([0.0, 0.0, 1.0]) * rng.uniform(0.0, 40.0, (n_tris, 1, 1))
    p = (c + rng.normal(scale=1.0, size=(n_tris, 3, 3)) * np.array([1.0, 1.0, 0.02])).astype(np.float32)
    b = scenes.SceneBuilder()
    m = b.add_material(scenes.make_material())
    b.begin_mesh(); b.add_geometry(p.reshape(-1, 3), np.arange(3 * n_tris, dtype=np.uint32), m); mesh = b.end_mesh(); b.add_instance(mesh)
    return b.finish()


def test_deep_stack_spills_to_global_memory_and_stays_exact():
    pt, scenes, parallel, ptref = _imports()
    sc = _overlap_scene(300000, 11)
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(scenes.default_settings())
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(scenes.default_settings())
    rng = np.random.default_rng(5)
    n = 20000
    org = np.concatenate([rng.normal(scale=0.3, size=(n, 2)), np.full((n, 1), -2.0)], 1)
    d = np.concatenate([rng.normal(scale=0.02, size=(n, 2)), np.ones((n, 1))], 1); d /= np.linalg.norm(d, axis=1, keepdims=True)
    # half of the rays look for the FARTHEST-first order: start behind the pile and shoot back
    org[n // 2:, 2] = 45.0; d[n // 2:, 2] *= -1.0
    rays = np.concatenate([org, np.zeros((n, 1)), d, np.full((n, 1), 1e15)], 1).astype(np.float32)
    hg, _ = g.trace_closest(rays)
    ho = o.trace_closest(rays)
    assert (hg.view(np.uint32) == ho.view(np.uint32)).all()
    assert (hg[:, 1].view(np.uint32) != 0xFFFFFFFF).mean() > 0.9
    vg, _ = g.trace_visibility(rays); vo = o.trace_visibility(rays)
    assert np.array_equal(vg, vo)
    st = g.build_stats()
    print("deep-stack scene: build %.1f ms" % st["buildMs"])


def test_degenerate_triangles_and_grazing_rays():
    pt, scenes, parallel, ptref = _imports()
    b = scenes.SceneBuilder()
    m = b.add_material(scenes.make_material())
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0],          # a regular triangle in z = 0
                    [2, 0, 0], [2, 0, 0], [2, 0, 0],          # a point
                    [3, 0, 0], [4, 0, 0], [5, 0, 0],          # a segment (zero area)
                    [0, 0, 1], [1, 0, 1], [0, 1, 1]], np.float32)
    b.begin_mesh(); b.add_geometry(pos, np.arange(12, dtype=np.uint32), m); mesh = b.end_mesh(); b.add_instance(mesh)
    sc = b.finish()
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(scenes.default_settings())
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(scenes.default_settings())
    rays = np.array([[0.25, 0.25, -1, 0, 0, 0, 1, 1e15],       # through both real triangles
                     [0.25, 0.25, 0, 0, 0, 0, 1, 1e15],        # starts ON the first triangle (t = 0 is rejected: t > tmin)
                     [-1, 0.25, 0, 0, 1, 0, 0, 1e15],          # in the plane of the triangle (det = 0)
                     [2, 0, -1, 0, 0, 0, 1, 1e15],             # at the point triangle
                     [4, 0, -1, 0, 0, 0, 1, 1e15],             # at the segment triangle
                     [0.25, 0.25, -1, 0, 0, 0, 1, 0.5],        # tmax in front of everything
                     [0, 0, -1, 0, 0, 0, 1, 1e15],             # exactly through a vertex
                     [0.5, 0.5, -1, 0, 0, 0, 1, 1e15]],        # exactly on the hypotenuse
                    np.float32)
    hg, _ = g.trace_closest(rays); ho = o.trace_closest(rays)
    assert (hg.view(np.uint32) == ho.view(np.uint32)).all()
    vg, _ = g.trace_visibility(rays); vo = o.trace_visibility(rays)
    assert np.array_equal(vg, vo)
    assert hg[0, 1].view(np.uint32) == 0 and hg[1, 1].view(np.uint32) == 3 and hg[5, 1].view(np.uint32) == 0xFFFFFFFF


def test_closest_hit_equals_exhaustive_loop_on_badly_conditioned_geometry():
    """The HIP traversal (every builder: PLOC + parallel re-insertion + cost-driven wide nodes on the device, binned SAH on the host, plain PLOC, Karras) against the oracle's exhaustive loop on the sliver / grazing-ray stress set: identical hit records, i.e. the
    closest hit does not depend on the tree above the triangles (pt_scene.h tri_box_accepts)."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    from oracle import ptref
    sc, rays = scenes.sliver_stress()
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(scenes.default_settings())
    want_bvh = o.trace_closest(rays)
    nb = 40000
    want_brute = o.trace_closest(rays[:nb], brute=True)
    assert np.array_equal(want_bvh[:nb].view(np.uint32), want_brute.view(np.uint32))
    vis_o = o.trace_visibility(rays)
    for builder in ("default", "sah", "ploc", "karras", "flag", "hostflag"):          # default: PLOC + parallel re-insertion + cost-driven wide nodes on the device (prefer fast trace); sah: round 2's host builder; flag / hostflag: PT_DEVICE_PREFER_FAST_BUILD / PT_DEVICE_HOST_SAH_BUILDER through the ABI
        if builder not in ("flag", "hostflag", "default"): os.environ["MI355PT_BVH_BUILDER"] = builder
        try:
            g = pt.PathTracer(prefer_fast_build=(builder == "flag"), host_sah_builder=(builder == "hostflag")); g.set_scene(sc); g.set_settings(scenes.default_settings())
            info = g.bvh_info()
            assert info["builder"] == {"default": 3, "sah": 2, "ploc": 0, "karras": 1, "flag": 0, "hostflag": 2}[builder] and info["builtOn"] == ("host" if builder in ("sah", "hostflag") else "device")
            if builder == "default": assert info["optimiserPasses"] == 12
            got, _ = g.trace_closest(rays)
            vis_g, _ = g.trace_visibility(rays)
        finally:
            os.environ.pop("MI355PT_BVH_BUILDER", None)
        same = (got.view(np.uint32) == want_bvh.view(np.uint32)).all(1)
        assert same.all(), "%s: %d of %d closest-hit records differ" % (builder, int((~same).sum()), same.size)
        assert np.array_equal(vis_g, vis_o), builder


def test_raw_buffer_abi_rejects_out_of_range_indices_and_mesh_ranges():
    """pt_set_geometry validates what the device would otherwise read out of bounds: an index beyond the geometry's vertex range, a mesh that names
    geometries outside the array. pt_set_materials refuses textures beyond 32768 texels per side (TexInfo holds 16 mip offsets)."""
    import ctypes
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C1")
    g = pt.PathTracer()
    bad = dict(sc); bad["indices"] = sc["indices"].copy(); bad["indices"][5] = 10 ** 6
    with pytest.raises(pt.PtError):
        g.set_scene(bad)
    bad = dict(sc); bad["meshes"] = sc["meshes"].copy(); bad["meshes"]["numGeometries"][0] = len(sc["geometries"]) + 3
    with pytest.raises(pt.PtError):
        g.set_scene(bad)
    g.set_scene(sc)                                    # the context is still usable
    g.set_camera(scenes.bridge_camera(32, 32, **cam)); g.set_settings(scenes.config_settings("C1")); g.resize(32, 32); g.render(0, 1)
    assert np.isfinite(g.radiance()).all()


def test_neeat_multi_sample_call_reports_summed_stats_and_feedback_guards():
    """pt_render(first, count > 1) with the baker in the loop traces one frame per sample: the stats are the sums over the frames (paths traced, ray counts), not those of the
    last one; and pt_get_light_feedback refuses cleanly while the planes it would read do not exist yet (NEE-AT switched on after a local-table frame, a reset, a resize)."""
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    sc, cam = scenes.bistro_like(scale=0.02, tex_size=64)
    S = scenes.default_settings(NEEType=2, useFp16Types=1); w, h = 96, 54
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
    t.set_neeat(True)
    with pytest.raises(Exception): t.light_feedback(0)                      # enabled, nothing rendered yet
    one = [t.render(s, 1) for s in range(3)]
    t.neeat_reset(); t.reset_accumulation()
    with pytest.raises(Exception): t.light_feedback(0)                      # after a reset
    three = t.render(0, 3)
    assert three["pathsTraced"] == 3 * w * h
    for k in ("extendRays", "shadowRays", "hits"): assert three[k] == sum(o[k] for o in one), k
    t.light_feedback(0)
    t.resize(64, 36)
    with pytest.raises(Exception): t.light_feedback(0)                      # another size: no frame of it yet
    t.render(0, 1); t.light_feedback(0)
    t.close()
