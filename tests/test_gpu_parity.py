"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs. Integer work (RNG, hit records, light tables) must be bit-exact. Radiance is floating point: the stated
tolerance is relative L2 <= 1e-6 over the linear RGBA32F accumulation buffer (north_star allows 1e-3); because both sides are
written to one arithmetic contract (single IEEE ops in fixed order, shared deterministic elementary functions) the observed
difference is exactly 0 and the tests additionally report the number of non-identical pixels."""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_L2_TOL = 1e-6


def _imports():
    import rtxpt_amd as pt
    from rtxpt_amd import scenes, parallel
    from oracle import ptref
    return pt, scenes, parallel, ptref


def rel_l2(a, b):
    a = a[..., :3].astype(np.float64); b = b[..., :3].astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def render_both(sc, cam, S, w, h, first, count):
    pt, scenes, parallel, ptref = _imports()
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h)
    stats = g.render(first, count)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(first, count)
    return g, o, stats


def check_images(g, o):
    a, b = g.radiance(), o.radiance()
    assert not np.isnan(a).any()
    d = rel_l2(a, b)
    nonident = int((np.abs(a[..., :3] - b[..., :3]).max(-1) > 0).sum())
    print("relL2 %.3e non-identical pixels %d / %d" % (d, nonident, a.shape[0] * a.shape[1]))
    assert d <= REL_L2_TOL
    return nonident


def test_library_is_the_hip_build():
    pt, *_ = _imports()
    L = pt.load_library()
    assert os.path.samefile(L._name, pt.LIB_PATH)
    import torch
    assert torch.cuda.is_available()


def test_c1_cornell_lambertian_256():
    """BASELINE configs[0]: Cornell 256x256, 1 spp, 2 bounces."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C1")
    g, o, st = render_both(sc, cam, scenes.config_settings("C1"), 256, 256, 0, 1)
    assert check_images(g, o) == 0
    c = o.counters()
    assert st["extendRays"] == c["extendRays"] and st["shadowRays"] == c["shadowRays"] and st["hits"] == c["hits"]


def test_c2_cornell_standard_bsdf_env_nee():
    """BASELINE configs[1] at reduced resolution (the oracle finishes in seconds): full StandardBSDF, nested dielectric glass box,
    metal box, sky environment with quad-tree NEE, Russian roulette, 8 bounces, 4 accumulated samples."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    g, o, st = render_both(sc, cam, scenes.config_settings("C2"), 480, 270, 0, 4)
    assert check_images(g, o) == 0
    c = o.counters()
    assert st["extendRays"] == c["extendRays"] and st["shadowRays"] == c["shadowRays"]


def test_bistro_like_small_textures_alpha_normalmaps():
    """C3 structure at 1/100 scale: textured + normal-mapped facades, alpha-tested foliage, 2000->20 emissive triangles, instances."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=0.01, tex_size=128)
    g, o, st = render_both(sc, cam, scenes.default_settings(), 320, 180, 0, 2)
    assert check_images(g, o) == 0
    lg, lo = g.lights(), o.lights()
    for k in ("lights", "lightsEx", "proxyCounters", "proxyIndices", "envLookup"):
        assert np.array_equal(lg[k], lo[k]), k
    assert np.array_equal(g.subinstances(), o.subinstances())


def test_firefly_filter_and_sample_offsets():
    """fireflyFilterThreshold != 0 path + non-zero first sample index + accumulation over two pt_render calls."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    S = scenes.default_settings(fireflyFilterThreshold=2.5, envMapDiffuseSampleMIPLevel=2.0, nestedDielectricsQuality=2)
    w, h = 200, 120
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h)
    g.render(7, 2); g.render(9, 1)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(7, 3)
    assert check_images(g, o) == 0


def test_hit_records_bit_exact_random_rays():
    """k_extend / k_shadow through the probe entry points vs the oracle's BVH on 200k random rays (SURVEY.md §7 gate 4)."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=0.02, tex_size=64)
    g = pt.PathTracer(); g.set_scene(sc); g.set_settings(scenes.default_settings())
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(scenes.default_settings())
    rng = np.random.default_rng(0x5EED0003)
    n = 200000
    org = rng.uniform((0.5, 0.05, 8.2), (119.5, 24.0, 31.8), (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.random(n) < 0.5, 1e15, rng.uniform(0.5, 30.0, n))
    rays = np.concatenate([org, np.zeros((n, 1)), d, tmax[:, None]], 1).astype(np.float32)
    hg, ms = g.trace_closest(rays)
    ho = o.trace_closest(rays)
    same = (hg.view(np.uint32) == ho.view(np.uint32)).all(1)
    print("closest-hit records identical: %d / %d (%.3f ms on GPU)" % (same.sum(), n, ms))
    assert same.all()
    vg, _ = g.trace_visibility(rays)
    vo = o.trace_visibility(rays)
    assert np.array_equal(vg, vo)
    assert 0.05 < vg.mean() < 0.95


def test_leaf_functions_bit_exact():
    """Device vs oracle: deterministic math, fp16 packing, sample streams, BSDF eval/sample, camera rays."""
    pt, scenes, parallel, ptref = _imports()
    L = ptref.lib()
    g = pt.PathTracer(test_hooks=True)      # pt_probe: the tests' build of the library (include/mi355pt_testhooks.h)
    rng = np.random.default_rng(3)
    n = 20000
    for fn, lo, hi in ((0, -30, 30), (1, -30, 30), (2, -140, 130), (3, 1e-30, 1e30), (4, -5, 5), (5, 1e-3, 50), (6, -1, 1), (7, 0, 4)):
        x = (10.0 ** rng.uniform(-30, 30, n) if fn == 3 else rng.uniform(lo, hi, n)).astype(np.float32)
        y = rng.uniform(-3, 3, n).astype(np.float32)
        inp = np.stack([np.full(n, fn, np.float32), x, y], 1)
        dev = g.probe(0, inp, (n,))
        ref = np.zeros(n, np.float32)
        L.ptref_dmath(fn, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), n, ref.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)), "dmath fn %d" % fn
    xs = np.concatenate([rng.uniform(-70000, 70000, n), 10.0 ** rng.uniform(-9, 5, n), [0.0, 65504.0, 65519.9, 65520.0, 5.9604645e-8, 2.9802322e-8]]).astype(np.float32)
    dev = g.probe(1, xs, (xs.size, 2))
    with np.errstate(over="ignore"):
        exp = xs.astype(np.float16)
    assert np.array_equal(dev[:, 0].view(np.uint32), exp.view(np.uint16).astype(np.uint32))
    assert np.array_equal(dev[:, 1], exp.astype(np.float32))
    m = 4000
    q = np.stack([rng.integers(0, 2**32, m), rng.integers(0, 12, m), rng.integers(0, 4096, m), rng.integers(0, 7, m), rng.integers(0, 5, m), np.full(m, 8)], 1).astype(np.uint32)
    q[q[:, 4] < 2, 5] = 4
    dev = g.probe(2, q, (m, 8))
    ref = np.zeros(8, np.float32)
    for i in range(m):
        ref[:] = 0
        L.ptref_sample_stream(int(q[i, 0]), int(q[i, 1]), int(q[i, 2]), int(q[i, 3]), int(q[i, 4]), int(q[i, 5]), ref.ctypes.data_as(ctypes.c_void_p))
        k = int(q[i, 5])
        assert np.array_equal(dev[i, :k].view(np.uint32), ref[:k].view(np.uint32)), q[i]
    # BSDF eval + sample
    m = 3000
    P = np.zeros((m, 24), np.float32)
    P[:, 0:3] = rng.random((m, 3)); P[:, 3:6] = rng.random((m, 1)) * 0.2; P[:, 6] = rng.random(m); P[:, 7] = (rng.random(m) < 0.3) * rng.random(m)
    P[:, 8:11] = rng.random((m, 3)); P[:, 11] = (rng.random(m) < 0.3) * rng.random(m); P[:, 12] = (rng.random(m) < 0.4) * rng.random(m); P[:, 13] = np.where(rng.random(m) < 0.5, 1 / 1.5, 1.5)
    P[:, 14] = rng.random(m) < 0.5; P[:, 15] = np.where(rng.random(m) < 0.3, 0, 2)
    wi = rng.normal(size=(m, 3)); wi[:, 2] = np.abs(wi[:, 2]) + 0.05; wi /= np.linalg.norm(wi, axis=1, keepdims=True); P[:, 16:19] = wi
    mode = (rng.random(m) < 0.5)
    wo = rng.normal(size=(m, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    P[:, 19:22] = np.where(mode[:, None], rng.random((m, 3)), wo); P[:, 22] = mode
    dev = g.probe(3, P, (m, 10))
    ref = np.zeros(10, np.float32)
    for i in range(m):
        ref[:] = 0
        L.ptref_bsdf_probe(P[i, :14].ctypes.data_as(ctypes.c_void_p), int(P[i, 14]), int(P[i, 15]), P[i, 16:19].copy().ctypes.data_as(ctypes.c_void_p),
                           P[i, 19:22].copy().ctypes.data_as(ctypes.c_void_p), int(P[i, 22]), ref.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(dev[i].view(np.uint32), ref.view(np.uint32)), (i, dev[i], ref)
    # camera rays (thin lens with aperture)
    cam = scenes.bridge_camera(640, 360, (1, 2, 3), (0.3, -0.2, 1), (0, 1, 0), 0.9, aperture_radius=0.05, focal_distance=4.0)
    g.set_camera(cam); g.set_settings(scenes.default_settings())
    o = ptref.Oracle(); o.set_camera(cam); o.set_settings(scenes.default_settings())
    q = np.stack([rng.integers(0, 640, 500), rng.integers(0, 360, 500), rng.integers(0, 64, 500)], 1).astype(np.uint32)
    dev = g.probe(4, q, (500, 6))
    for i in range(500):
        assert np.array_equal(dev[i].view(np.uint32), o.camera_ray(int(q[i, 0]), int(q[i, 1]), int(q[i, 2])).view(np.uint32))


def test_gltf_import_equals_raw_buffers(tmp_path):
    """pt_load_scene_gltf (Sample::LoadScene seam): the frame equals, bit for bit, the frame of the same buffers handed over through pt_set_* — on the GPU and
    on the oracle. The buffers are the ones the import produced (read back from the host-side import object), the vertex streams are the writer's inputs:
    normals / tangents survive the float round trip exactly (tests/gltf_writer.py), and the materials are what ImportFromDonut + FillData make of the
    document (no IoR / volume import, see test_gltf_transmission_material_is_a_refracting_solid)."""
    pt, scenes, parallel, ptref = _imports()
    from tests.gltf_writer import write_gltf
    sc, cam = scenes.cornell_box("C2")
    path = str(tmp_path / "cornell.gltf")
    write_gltf(sc, path)
    (tmp_path / "c.scene.json").write_text(json.dumps({"models": ["cornell.gltf"], "graph": [{"model": 0}]}))
    imp = pt.SceneImport(tmp_path / "c.scene.json")
    assert np.array_equal(imp.geometries, sc["geometries"]) and np.array_equal(imp.instances["meshIndex"], sc["instances"]["meshIndex"])
    assert np.array_equal(imp.instances["transform"], sc["instances"]["transform"])
    sc2 = dict(sc); sc2["materials"] = imp.materials.copy()
    S = scenes.config_settings("C2"); w, h = 160, 96
    camd = scenes.bridge_camera(w, h, **cam)
    a = pt.PathTracer(); a.set_scene(sc2); a.set_camera(camd); a.set_settings(S); a.resize(w, h); a.render(0, 2)
    b = pt.PathTracer(); b.load_scene_gltf(path)
    rgb, tw, cm = sc["env"]
    p = pt.PtEnvMapSceneParams((ctypes.c_float * 12)(*tw.tolist()), (ctypes.c_float * 3)(*(cm * np.float32(4.0)).tolist()), 1.0)      # intensity / c_envMapRadianceScale (Sample.cpp:1939)
    assert b.L.pt_set_environment(b.h, rgb.ctypes.data_as(ctypes.c_void_p), rgb.shape[1], rgb.shape[0], ctypes.byref(p)) == 0
    assert b.L.pt_set_environment_bake(b.h, 256, None, 0) == 0                     # the cube resolution set_scene uses for `a` (the C default is EnvMapBaker's 2048)
    b.set_camera(camd); b.set_settings(S); b.resize(w, h); b.render(0, 2)
    assert b.scene_info()["triangles"] == a.scene_info()["triangles"]
    assert np.array_equal(a.subinstances(), b.subinstances())
    assert np.array_equal(a.radiance(), b.radiance())
    o = ptref.Oracle(); o.set_scene(sc2); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(0, 2)
    assert np.array_equal(b.radiance(), o.radiance())


def test_gltf_animation_drives_pt_animate(tmp_path):
    """pt_gltf_animation_* -> pt_animate (Sample::Animate seam): the file's rest pose is what pt_load_scene_gltf built, and a frame rendered after
    pt_animate(instances at t) equals the oracle's frame of the scene with those instance transforms."""
    pt, scenes, parallel, ptref = _imports()
    from tests.gltf_writer import write_gltf
    sc, cam = scenes.cornell_box("C2")
    path = tmp_path / "cornell.gltf"
    write_gltf(sc, str(path))
    doc = json.loads(path.read_text()); k = len(doc["nodes"]) - 1
    t34 = sc["instances"]["transform"][k].reshape(3, 4)
    scl = np.linalg.norm(t34[:, :3].astype(np.float64), axis=0); ang = float(np.arctan2(t34[0, 2] / scl[2], t34[0, 0] / scl[0]))      # the tall box: scale, then a turn about y
    qy = lambda a: [0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2))]
    doc["nodes"][k] = {"mesh": doc["nodes"][k]["mesh"], "translation": [float(v) for v in t34[:, 3]], "rotation": qy(ang), "scale": [float(v) for v in scl]}
    keys = np.array([0.0, 1.0, 2.0], np.float32); tr = (t34[:, 3][None, :] + np.array([[0, 0, 0], [0.1, 0.05, 0], [0.1, 0.2, -0.1]], np.float32)).astype(np.float32)
    q = np.array([qy(ang), qy(ang + 0.6), qy(ang + 1.2)], np.float32)
    blob = keys.tobytes() + tr.tobytes() + q.tobytes(); (tmp_path / "anim.bin").write_bytes(blob)
    nb, nv, na = len(doc["buffers"]), len(doc["bufferViews"]), len(doc["accessors"])
    doc["buffers"].append({"uri": "anim.bin", "byteLength": len(blob)})
    doc["bufferViews"] += [{"buffer": nb, "byteOffset": 0, "byteLength": 12}, {"buffer": nb, "byteOffset": 12, "byteLength": 36}, {"buffer": nb, "byteOffset": 48, "byteLength": 48}]
    doc["accessors"] += [{"bufferView": nv, "componentType": 5126, "count": 3, "type": "SCALAR", "min": [0.0], "max": [2.0]}, {"bufferView": nv + 1, "componentType": 5126, "count": 3, "type": "VEC3"},
                         {"bufferView": nv + 2, "componentType": 5126, "count": 3, "type": "VEC4"}]
    doc["animations"] = [{"samplers": [{"input": na, "output": na + 1}, {"input": na, "output": na + 2}],
                          "channels": [{"sampler": 0, "target": {"node": k, "path": "translation"}}, {"sampler": 1, "target": {"node": k, "path": "rotation"}}]}]
    path.write_text(json.dumps(doc))
    an = pt.GltfAnimation(path)
    assert an.count == 1 and an.duration == 2.0
    rest = an.instances(0.0)
    assert np.array_equal(rest["meshIndex"], sc["instances"]["meshIndex"]) and np.array_equal(rest["transform"][:k], sc["instances"]["transform"][:k])
    assert np.allclose(rest["transform"][k], sc["instances"]["transform"][k], rtol=0, atol=1e-6)
    S = scenes.config_settings("C2"); w, h = 160, 96
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.load_scene_gltf(str(path))
    rgb, tw, cm = sc["env"]
    p = pt.PtEnvMapSceneParams((ctypes.c_float * 12)(*tw.tolist()), (ctypes.c_float * 3)(*(cm * np.float32(4.0)).tolist()), 1.0)
    assert g.L.pt_set_environment(g.h, rgb.ctypes.data_as(ctypes.c_void_p), rgb.shape[1], rgb.shape[0], ctypes.byref(p)) == 0
    assert g.L.pt_set_environment_bake(g.h, 256, None, 0) == 0
    g.set_camera(camd); g.set_settings(S); g.resize(w, h); g.render(0, 1); first = g.radiance().copy()
    inst = an.instances(1.4)
    assert not np.array_equal(inst["transform"][k], rest["transform"][k]) and np.array_equal(inst["transform"][:k], rest["transform"][:k])
    g.animate(instances=inst, rebuild=False); g.render(0, 2)
    assert not np.array_equal(g.radiance(), first)
    (tmp_path / "c.scene.json").write_text(json.dumps({"models": ["cornell.gltf"], "graph": [{"model": 0}]}))
    sc2 = dict(sc); sc2["materials"] = pt.SceneImport(tmp_path / "c.scene.json").materials.copy()
    o = ptref.Oracle(); o.set_scene(sc2); o.set_instances(inst); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(0, 2)
    assert np.array_equal(g.radiance(), o.radiance())


def test_analytic_light_proxy_tables_and_frames():
    """Analytic light proxies (PtInstanceDesc.analyticProxyLight -> SubInstanceData.AnalyticProxyLightIndex, LightsBaker.cpp:718-753): the sub-instance table and the frame equal
    the oracle's; the link survives pt_animate; a proxy index beyond the light list, or a flagged material without a link, is inert."""
    pt, scenes, parallel, ptref = _imports()
    make, S, w, h, first, n = _pin_cases()["c2_sphere_light_proxy"]
    sc, cam = make(); camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h); g.render(first, n)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(first, n)
    si = g.subinstances()
    assert np.array_equal(si, o.subinstances()) and (si[:, 3] != 0xFFFFFFFF).sum() >= 1
    assert np.array_equal(g.radiance(), o.radiance())
    inst = sc["instances"].copy(); inst["transform"][1][3] += 0.02                       # move the proxy instance: refit, the link stays
    g.animate(instances=inst, rebuild=False); g.reset_accumulation(); g.render(first, n)
    o2 = ptref.Oracle(); o2.set_scene(sc); o2.set_instances(inst); o2.set_camera(camd); o2.set_settings(S); o2.resize(w, h); o2.render(first, n)
    assert np.array_equal(g.radiance(), o2.radiance()) and np.array_equal(g.subinstances(), o2.subinstances())
    inst["analyticProxyLight"][1] = 99                                                   # no such light: inert
    g.animate(instances=inst, rebuild=False); g.render(first, 1); assert (g.subinstances()[:, 3] == 0xFFFFFFFF).all()


def test_refit_equals_rebuild_and_oracle():
    """pt_animate: instance motion -> LBVH refit; refit, full rebuild and the oracle (fresh SAH build) agree bit-for-bit."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=0.01, tex_size=64)
    S = scenes.default_settings(); w, h = 256, 144
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h); g.render(0, 1)
    inst = scenes.animate_instances(sc, 1.7)
    g.animate(instances=inst, rebuild=False); g.render(0, 2); a = g.radiance()
    assert g.build_stats()["refitMs"] > 0
    g.animate(instances=inst, rebuild=True); g.render(0, 2); b = g.radiance()
    assert np.array_equal(a, b)
    o = ptref.Oracle(); o.set_scene(sc); o.set_instances(inst); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(0, 2)
    assert rel_l2(a, o.radiance()) <= REL_L2_TOL
    assert np.array_equal(a[..., :3], o.radiance()[..., :3])
    # deformation: move vertices of one geometry (skinned-mesh style), refit
    pos = sc["positions"].copy(); pos[: pos.shape[0] // 50, 1] += 0.05
    g.animate(positions=pos, rebuild=False); g.render(0, 1); c = g.radiance()
    sc2 = dict(sc); sc2["positions"] = pos
    o2 = ptref.Oracle(); o2.set_scene(sc2); o2.set_instances(inst); o2.set_camera(camd); o2.set_settings(S); o2.resize(w, h); o2.render(0, 1)
    assert np.array_equal(c[..., :3], o2.radiance()[..., :3])


def test_c5_animated_frames_refit_nested_dielectrics():
    """BASELINE config C5 at a small scale: per frame the 40 clutter groups move rigidly and the banner mesh deforms (refit only), the camera
    looks at the nested-dielectric props; three consecutive frames equal an oracle that is rebuilt from scratch for every frame."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.bistro_like(scale=0.01, tex_size=64, animated=True)
    cam = dict(cam, pos=(20.0, 2.5, 20.0), direction=(1.0, -0.08, 0.02))
    S = scenes.default_settings(nestedDielectricsQuality=2); w, h = 192, 108
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h)
    for frame, t in enumerate((0.0, 0.4, 0.8)):
        inst, pos = scenes.animate_instances(sc, t), scenes.animate_positions(sc, t)
        g.animate(instances=inst, positions=pos, rebuild=False)
        g.reset_accumulation(); st = g.render(frame * 2, 2)
        sc_t = dict(sc); sc_t["positions"] = pos
        o = ptref.Oracle(); o.set_scene(sc_t); o.set_instances(inst); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(frame * 2, 2)
        assert np.array_equal(g.radiance()[..., :3], o.radiance()[..., :3]), "frame %d" % frame
        assert st["extendRays"] == o.counters()["extendRays"]
    assert g.build_stats()["refitMs"] > 0


def test_tile_shards_reassemble_bit_exact():
    """2-way pixel-tile sharding on one GPU: each shard traces only its tiles; pack -> (gather) -> unpack reproduces the
    single-context frame bit-for-bit (RNG keyed on absolute pixel + sample index)."""
    pt, scenes, parallel, ptref = _imports()
    import torch
    sc, cam = scenes.cornell_box("C2")
    S = scenes.config_settings("C2"); w, h = 200, 136
    camd = scenes.bridge_camera(w, h, **cam)
    full = pt.PathTracer(); full.set_scene(sc); full.set_camera(camd); full.set_settings(S); full.resize(w, h); full.render(0, 2)
    ref = full.radiance()
    ctxs = []
    for r in range(2):
        c = pt.PathTracer(shard_rank=r, shard_count=2); c.set_scene(sc); c.set_camera(camd); c.set_settings(S); c.resize(w, h); c.render(0, 2); ctxs.append(c)
    bufs = []
    for r, c in enumerate(ctxs):
        n, nbytes = c.shard_info()
        assert n == parallel.shard_pixels(w, h, r, 2).size
        t = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        c.pack_shard(t.data_ptr(), nbytes); torch.cuda.synchronize()
        bufs.append(t)
        own = parallel.shard_pixels(w, h, r, 2)
        assert np.array_equal(t.cpu().numpy(), ref[(own & 0xFFFF).astype(np.int64), (own >> 16).astype(np.int64)])
    ctxs[0].unpack_shard(bufs[1].data_ptr(), bufs[1].numel() * 4, 1)
    assert np.array_equal(ctxs[0].radiance(), ref)


def test_golden_fixture_c1_32():
    """Committed fixture (tests/golden/c1_32_radiance.npy, written by the oracle in the build container): no oracle call here."""
    pt, scenes, parallel, ptref = _imports()
    path = os.path.join(ROOT, "tests", "golden", "c1_32_radiance.npy")
    sc, cam = scenes.cornell_box("C1")
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(32, 32, **cam)); g.set_settings(scenes.config_settings("C1")); g.resize(32, 32); g.render(0, 4)
    assert np.array_equal(g.radiance(), np.load(path))


def test_full_size_properties_1080p():
    """BASELINE configs[1] at its full size (1920x1080, 4 spp, 8 bounces) through size-independent properties: accumulation is
    associative over pt_render calls (bit-exact), finite/non-negative radiance, ray budget <= 17 per path, and the frame
    and 8 complete pixel rows are bit-identical to the oracle."""
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    S = scenes.config_settings("C2"); w, h = 1920, 1080
    camd = scenes.bridge_camera(w, h, **cam)
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(w, h)
    st = g.render(0, 4); a = g.radiance()
    g.reset_accumulation(); g.render(0, 1); g.render(1, 2); g.render(3, 1); b = g.radiance()
    assert np.array_equal(a, b)
    assert np.isfinite(a).all() and (a >= 0).all() and np.all(a[..., 3] == 1.0)
    assert st["extendRays"] + st["shadowRays"] <= 17 * st["pathsTraced"] + 4 * st["pathsTraced"]
    # 8 full rows of the 1080p frame, bit-exact against the oracle (one oracle context per row keeps the accumulation weights aligned)
    for y0 in range(4, h - 8, 135):
        orow = ptref.Oracle(); orow.set_scene(sc); orow.set_camera(camd); orow.set_settings(S); orow.resize(w, h)
        orow.render(0, 4, rect=(0, y0, w, y0 + 1))
        assert np.array_equal(a[y0, :, :3], orow.radiance()[y0, :, :3]), y0


def _pin_cases():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pin_scenes
    return pin_scenes.cases()


@pytest.mark.parametrize("name", ["c1", "c2", "c2_firefly", "c2_nee3", "c2_nee_off", "c2_nested2_norr_nold", "c2_nested0_uniform", "c2_sphere_lights", "c2_exclude_from_nee", "c2_env_rotated_mip2", "c2_mirrored_room", "bistro_like", "bistro_like_material_zoo", "bistro_like_c5", "c2_spec_gloss", "bistro_like_spec_gloss", "c2_sphere_light_proxy", "c2_sun_discs_bc6"])
def test_product_matches_reference_integrator_golden(name):
    """The HIP path against frames rendered by the REFERENCE'S integrator source text (tests/golden/reference_integrator_golden.npz, made in the build
    container by compiling PathTracer.hlsli & co. over the oracle's scene services — tests/test_oracle_refpin_integrator.py). No oracle call here."""
    pt, scenes, parallel, ptref = _imports()
    make, S, w, h, first, n = _pin_cases()[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden.npz"))
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(S); t.resize(w, h); st = t.render(first, n)
    got, want = t.radiance(), g[name]
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert (st["extendRays"], st["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])


def _pin_cases_lp16():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pin_scenes
    return pin_scenes.cases_lp16()


@pytest.mark.parametrize("name", ["c1", "c2", "c2_firefly", "c2_nee3", "c2_nee_off", "c2_nested2_norr_nold", "c2_nested0_uniform", "c2_sphere_lights", "c2_exclude_from_nee", "c2_env_rotated_mip2", "c2_mirrored_room", "bistro_like", "bistro_like_material_zoo", "bistro_like_c5", "bistro_like_firefly", "bistro_like_material_zoo_firefly", "c2_spec_gloss", "bistro_like_spec_gloss", "c2_sphere_light_proxy", "c2_sun_discs_bc6"])
def test_product_lp16_matches_reference_integrator_golden(name):
    """PtSettings.useFp16Types = 1 — the reference's DEFAULT build (lp types in binary16; SampleUI.h:182, Sample.cpp:1035) and what pt_default_settings returns —
    against frames rendered by the reference's integrator text compiled that way (tests/golden/reference_integrator_golden_lp16.npz). No oracle call here."""
    pt, scenes, parallel, ptref = _imports()
    make, S, w, h, first, n = _pin_cases_lp16()[name]
    assert int(S["useFp16Types"]) == 1
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden_lp16.npz"))
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(S); t.resize(w, h); st = t.render(first, n)
    got, want = t.radiance(), g[name]
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s (lp16): %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert (st["extendRays"], st["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])


def _wide_cases():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pin_scenes
    return pin_scenes.wide_cases()


@pytest.mark.parametrize("name", ["c2_wide", "c2_wide_lp16", "bistro_like_wide", "bistro_like_wide_lp16", "bistro_like_material_zoo_wide", "bistro_like_material_zoo_wide_lp16",
                                  "bistro_like_c5_wide", "bistro_like_c5_wide_lp16"])
@pytest.mark.parametrize("tail", [0, 32768], ids=["wavefront", "tail_kernel"])
def test_product_matches_wide_reference_integrator_golden(name, tail):
    """256 x 144 x 4 samples per pin family and lp build, rendered by the reference's integrator text (tests/golden/reference_integrator_golden_wide.npz): the HIP
    path, with and without the tail kernel, against it — every pixel's bits and the ray counts. No oracle call here."""
    pt, scenes, parallel, ptref = _imports()
    make, S, w, h, first, n = _wide_cases()[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden_wide.npz"))
    sc, cam = make()
    t = pt.PathTracer(); t.set_tail_paths(tail); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(S); t.resize(w, h); st = t.render(first, n)
    got, want = t.radiance(), g[name]
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert (st["extendRays"], st["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])
    if tail: assert st["tailLaunches"] > 0


@pytest.mark.parametrize("name", ["bistro_like_xl", "bistro_like_xl_lp16"])
def test_product_matches_xl_reference_integrator_golden(name):
    """1280 x 720 x 4 samples of the reference's integrator text with the bench configuration's settings (tests/golden/reference_integrator_golden_xl.npz: every sixteenth row, a
    SHA-256 of the whole frame, the ray counts) — the HIP path as the product runs it (pipelined batches, straggler rounds, tail kernel). No oracle call here."""
    pt, scenes, parallel, ptref = _imports()
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pin_scenes
    make, S, w, h, first, n = pin_scenes.xl_cases()[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_integrator_golden_xl.npz"))
    sc, cam = make()
    t = pt.PathTracer(); t.set_tail_paths(32768); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(S); t.resize(w, h); st = t.render(first, n)
    rad = t.radiance()
    rows = rad[::pin_scenes.XL_ROW_STEP]
    bad = (rows.view(np.uint32) != g[name + "_rows"].view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels of the kept rows differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert np.array_equal(pin_scenes.frame_digest(rad), g[name + "_sha256"]), "%s: the frame's digest differs" % name
    assert (st["extendRays"], st["shadowRays"]) == tuple(int(v) for v in g[name + "_rays"])


def test_c_default_settings_are_the_reference_default_build():
    pt, scenes, parallel, ptref = _imports()
    t = pt.PathTracer()
    assert int(t.default_settings()["useFp16Types"]) == 1


@pytest.mark.parametrize("lp16", [False, True], ids=["fp32", "lp16"])
@pytest.mark.parametrize("name", ["c2", "c2_mirrored_room", "bistro_like", "bistro_like_material_zoo", "bistro_like_c5", "c2_spec_gloss", "bistro_like_spec_gloss", "c2_sphere_light_proxy", "c2_sun_discs_bc6"])
def test_device_load_surface_matches_oracle(name, lp16):
    """Bridge::loadSurface on the device (geometry gather, material evaluation with its lp types, textures, normal map, tangent frame, BSDF inputs, emissive light
    index) against the oracle's — which is pinned to PathTracerBridgeDonut.hlsli compiled from the reference in both builds of the lp types
    (tests/test_oracle_refpin_integrator.py::test_load_surface_matches_reference_text) — on 20 000 random hits per scene, 45 words each."""
    pt, scenes, parallel, ptref = _imports()
    make, S, w, h, first, n = (_pin_cases_lp16() if lp16 else _pin_cases())[name]
    sc, cam = make()
    g = pt.PathTracer(test_hooks=True); g.set_scene(sc); g.set_settings(S); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.resize(w, h)
    o = ptref.Oracle(lp16=lp16); o.set_scene(sc); o.set_settings(S); o.resize(8, 8)
    nt = o.num_tris()
    rng = np.random.default_rng(0x5F + len(name)); k = 20000
    prims = rng.integers(0, nt, k).astype(np.uint32); u = rng.uniform(0, 1, k); v = rng.uniform(0, 1, k) * (1 - u)
    d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rows7 = np.column_stack([u, v, d, rng.uniform(0, 0.5, k), rng.uniform(0, 0.01, k)]).astype(np.float32)
    want = o.surface_probe(prims, rows7)
    rows8 = np.zeros((k, 8), np.float32); rows8[:, 0] = prims.view(np.float32); rows8[:, 1:] = rows7
    got = g.probe(8, rows8, (k, 45), out_dtype=np.uint32)
    bad = (got != want)
    assert not bad.any(), "%d of %d surfaces differ; per word %s; first: hit %d device %s oracle %s" % (
        int(bad.any(1).sum()), k, {i: int(bad[:, i].sum()) for i in range(45) if bad[:, i].any()}, int(np.flatnonzero(bad.any(1))[0]),
        got[bad.any(1)][0][bad[bad.any(1)][0]].view(np.float32), want[bad.any(1)][0][bad[bad.any(1)][0]].view(np.float32))
