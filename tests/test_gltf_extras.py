"""glTF features a production asset folder can carry (verdict r04, "importer leftovers"), through pt_scene_json_import — host only, no device:
  * sparse accessors (glTF 2.0, 3.6.2.3): cgltf, which Donut's importer reads glTF with, resolves them, so the reference accepts such files; with and without a base bufferView;
  * KHR_lights_punctual: point / spot lights become LightsBaker's analytic lights (the records pt_convert_light makes of a PointLight / SpotLight leaf,
    Rtxpt/SampleCommon/ExtendedScene.cpp:54-143), directional lights go to the environment baker's list (Rtxpt/Sample.cpp:1361-1388), under the model node's transform, in scene-graph
    order, invisible ones dropped (Sample.cpp:567-573);
  * KHR_texture_transform is read past: PTMaterialData has no slot for it (Rtxpt/Shaders/PathTracer/Materials/MaterialPT.h:45-77), so the reference's shaders cannot apply one."""
import json
import math
import os
import struct
import sys

import numpy as np
import pytest

import rtxpt_amd as pt
from rtxpt_amd import scenes

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gltf_writer import write_gltf
from test_scene_json import trs_matrix


def _folder(tmp_path, patch, graph=None):
    media = tmp_path / "media"; (media / "Models").mkdir(parents=True, exist_ok=True)
    sc, cam = scenes.cornell_box("C2")
    path = media / "Models" / "m.gltf"
    write_gltf(sc, str(path))
    doc = json.loads(path.read_text()); blob = bytearray((media / "Models" / "m.bin").read_bytes())
    patch(doc, blob)
    doc["buffers"][0]["byteLength"] = len(blob)
    (media / "Models" / "m.bin").write_bytes(bytes(blob)); path.write_text(json.dumps(doc))
    (media / "t.scene.json").write_text(json.dumps({"models": ["Models/m.gltf"], "graph": graph or [{"name": "room", "model": 0}]}))
    return media, sc


def _append(doc, blob, data):
    while len(blob) % 4: blob += b"\0"
    doc["bufferViews"].append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}); blob += data
    return len(doc["bufferViews"]) - 1


def test_sparse_accessors(tmp_path):
    moved = {}

    def patch(doc, blob):
        prims = doc["meshes"][0]["primitives"]
        # primitive 0: its POSITION accessor keeps the base view, three vertices are replaced through u16 indices
        a = doc["accessors"][prims[0]["attributes"]["POSITION"]]
        idx = np.array([0, 2, 3], np.uint16); val = np.array([[9, 8, 7], [-1, -2, -3], [0.5, 0.25, 0.125]], np.float32)
        a["sparse"] = {"count": 3, "indices": {"bufferView": _append(doc, blob, idx.tobytes()), "componentType": 5123}, "values": {"bufferView": _append(doc, blob, val.tobytes())}}
        moved["p0"] = (idx, val)
        # primitive 1: no base view at all (zeros), every vertex but the last set through u8 indices with a byteOffset into a shared view
        a = doc["accessors"][prims[1]["attributes"]["POSITION"]]; n = a["count"]
        idx = np.arange(n - 1, dtype=np.uint8); val = (np.arange(3 * (n - 1), dtype=np.float32).reshape(-1, 3) + 0.5)
        view = _append(doc, blob, b"\xAA\xAA\xAA\xAA" + idx.tobytes())
        del a["bufferView"]; a.pop("byteOffset", None)
        a["sparse"] = {"count": int(n - 1), "indices": {"bufferView": view, "byteOffset": 4, "componentType": 5121}, "values": {"bufferView": _append(doc, blob, val.tobytes())}}
        moved["p1"] = (idx, val, n)
    media, sc = _folder(tmp_path, patch)
    imp = pt.SceneImport(media / "t.scene.json")
    g = imp.geometries
    vo, nv = int(g[0]["vertexOffset"]), int(g[0]["numVertices"])
    want = sc["positions"][int(sc["geometries"][0]["vertexOffset"]):][:nv].copy(); want[moved["p0"][0]] = moved["p0"][1]
    assert np.array_equal(imp.positions[vo:vo + nv], want)
    vo, nv = int(g[1]["vertexOffset"]), int(g[1]["numVertices"])
    idx, val, n = moved["p1"]; want = np.zeros((n, 3), np.float32); want[idx] = val
    assert nv == n and np.array_equal(imp.positions[vo:vo + nv], want)
    vo, nv = int(g[2]["vertexOffset"]), int(g[2]["numVertices"])      # an ordinary accessor next to them is untouched
    assert np.array_equal(imp.positions[vo:vo + nv], sc["positions"][int(sc["geometries"][2]["vertexOffset"]):][:nv])
    imp.close()


def test_malformed_sparse_accessors_are_refused(tmp_path):
    def patch(doc, blob):
        a = doc["accessors"][doc["meshes"][0]["primitives"][0]["attributes"]["POSITION"]]
        idx = np.array([0, 60000], np.uint16); val = np.zeros((2, 3), np.float32)      # an index beyond the accessor's count
        a["sparse"] = {"count": 2, "indices": {"bufferView": _append(doc, blob, idx.tobytes()), "componentType": 5123}, "values": {"bufferView": _append(doc, blob, val.tobytes())}}
    media, _ = _folder(tmp_path, patch)
    try:
        pt.SceneImport(media / "t.scene.json"); assert False, "accepted"
    except pt.PtError as e:
        assert e.code == 4      # PT_ERROR_IO


def test_khr_lights_punctual(tmp_path):
    q = (math.sin(0.3), 0.0, 0.0, math.cos(0.3))      # about x

    def patch(doc, blob):
        doc["extensions"] = {"KHR_lights_punctual": {"lights": [
            {"type": "point", "color": [1.0, 0.5, 0.25], "intensity": 30.0},
            {"type": "spot", "intensity": 80.0, "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.6}},
            {"type": "directional", "color": [1.0, 0.9, 0.8], "intensity": 4.0},
            {"type": "point", "intensity": 0.0},
            {"type": "spot", "color": [0.2, 0.4, 0.6]}]}}      # defaults: intensity 1, cone 0 .. pi / 4
        doc["extensionsUsed"].append("KHR_lights_punctual")
        n0 = len(doc["nodes"])
        doc["nodes"] += [{"name": "bulb", "translation": [0.1, 0.2, 0.3], "extensions": {"KHR_lights_punctual": {"light": 0}}},
                         {"name": "rig", "translation": [1.0, 0.0, 0.0], "rotation": list(q), "children": [n0 + 2, n0 + 3]},
                         {"name": "spot", "translation": [0.0, 1.0, 0.0], "extensions": {"KHR_lights_punctual": {"light": 1}}},
                         {"name": "sun", "extensions": {"KHR_lights_punctual": {"light": 2}}},
                         {"name": "dark", "extensions": {"KHR_lights_punctual": {"light": 3}}},
                         {"name": "default_spot", "extensions": {"KHR_lights_punctual": {"light": 4}}}]
        doc["scenes"][0]["nodes"] += [n0, n0 + 1, n0 + 4, n0 + 5]
    graph = [{"name": "room", "model": 0, "translation": [5.0, 0.0, 0.0], "scaling": 2.0},
             {"name": "own", "type": "PointLight", "translation": [0.0, 3.0, 0.0], "intensity": 7.0, "radius": 0.05}]
    media, _ = _folder(tmp_path, patch, graph)
    imp = pt.SceneImport(media / "t.scene.json")
    W = trs_matrix((5, 0, 0), s=(2, 2, 2)); rig = W @ trs_matrix((1, 0, 0), q)

    def dirz(M): z = -M[:3, 2]; return z / np.linalg.norm(z)
    want = [pt.convert_light("point", (W @ trs_matrix((0.1, 0.2, 0.3)))[:3, 3], (1.0, 0.5, 0.25), 30.0, 0.0, dirz(W), 180.0, 180.0),
            pt.convert_light("spot", (rig @ trs_matrix((0, 1, 0)))[:3, 3], (1, 1, 1), 80.0, 0.0, dirz(rig), math.degrees(0.2), math.degrees(0.6)),
            pt.convert_light("spot", W[:3, 3], (0.2, 0.4, 0.6), 1.0, 0.0, dirz(W), 0.0, 45.0),
            pt.convert_light("point", (0.0, 3.0, 0.0), (1, 1, 1), 7.0, 0.05)]      # the graph's own light comes after the model's (scene-graph order)
    assert imp.info["numLights"] == 4 and imp.info["lightsDropped"] == 1 and imp.info["directionalLights"] == 1
    for k, (b, e) in enumerate(want):
        got = imp.lights[k].view(np.float32), imp.lights_ex[k].view(np.float32); ref = b.view(np.float32), e.view(np.float32)
        # the composed transforms are formed in double on both sides but not by the same expression: compare as floats, field by field (packed fields: equal bits)
        assert np.array_equal(imp.lights[k][[3, 4, 5, 6, 7]], b[[3, 4, 5, 6, 7]]) or np.allclose(got[0], ref[0], rtol=1e-6, atol=1e-6), (k, imp.lights[k], b)
        assert np.allclose(got[0][:3], ref[0][:3], rtol=0, atol=1e-5), (k, got[0][:3], ref[0][:3])
    d = imp.directional_lights
    assert d.shape[0] == 1 and np.allclose(d[0, :4], [1.0, 0.9, 0.8, 4.0]) and np.allclose(d[0, 4:7], dirz(rig), atol=1e-6) and d[0, 7] == 0.0
    imp.close()


def test_gltf_embedded_cameras(tmp_path):
    """A glTF file's own perspective cameras join the graph's camera list in scene-graph order, under the model node's transform; they are plain PerspectiveCamera leaves:
    Sample::UpdateCameraFromScene (Sample.cpp:457-478) takes pose, fov and near plane from them and leaves the tone-mapping block alone. Orthographic ones are not listed."""
    q = (0.0, math.sin(0.4), 0.0, math.cos(0.4))      # about y

    def patch(doc, blob):
        doc["cameras"] = [{"type": "perspective", "name": "hero", "perspective": {"yfov": 0.7, "znear": 0.05, "zfar": 100.0, "aspectRatio": 1.5}},
                          {"type": "orthographic", "orthographic": {"xmag": 1.0, "ymag": 1.0, "znear": 0.1, "zfar": 10.0}},
                          {"type": "perspective", "perspective": {"yfov": 1.1, "znear": 0.2}}]
        n0 = len(doc["nodes"])
        doc["nodes"] += [{"name": "dolly", "translation": [0.0, 1.0, 2.0], "rotation": list(q), "children": [n0 + 1, n0 + 2]},
                         {"name": "cam_a", "translation": [0.5, 0.0, 0.0], "camera": 0}, {"name": "ortho", "camera": 1}, {"name": "cam_b", "camera": 2}]
        doc["scenes"][0]["nodes"] += [n0, n0 + 3]
    graph = [{"name": "first", "type": "PerspectiveCameraEx", "translation": [0.0, 2.0, 0.0], "verticalFov": 0.9, "exposureCompensation": 1.5},
             {"name": "room", "model": 0, "translation": [5.0, 0.0, 0.0], "scaling": 2.0}]
    media, _ = _folder(tmp_path, patch, graph)
    imp = pt.SceneImport(media / "t.scene.json")
    W = trs_matrix((5, 0, 0), s=(2, 2, 2)); dolly = W @ trs_matrix((0, 1, 2), q)
    cams = imp.cameras
    assert imp.info["numCameras"] == 3 and imp.info["selectedCamera"] == 2
    assert [bytes(c["name"]).split(b"\0")[0].decode() for c in cams] == ["first", "hero", "cam_b"]
    for c, M, fov, zn in ((cams[1], dolly @ trs_matrix((0.5, 0, 0)), 0.7, 0.05), (cams[2], W, 1.1, 0.2)):
        assert np.allclose(c["position"], M[:3, 3], atol=1e-6) and np.allclose(c["direction"], -M[:3, 2], atol=1e-6) and np.allclose(c["up"], M[:3, 1], atol=1e-6)
        assert c["verticalFov"] == np.float32(fov) and c["zNear"] == np.float32(zn) and int(c["exposureMask"]) == 0x80000000
    ui = pt.default_tone_mapping_parameters(); ui["exposureValueMin"] = -3.0
    imp.tone_mapping(ui, camera=0)
    assert ui["exposureCompensation"] == 1.5 and ui["exposureValueMin"] == -16.0       # a PerspectiveCameraEx: its keys, defaults for the missing ones
    ui["exposureValueMin"] = -3.0; imp.tone_mapping(ui)                                # the selected camera is a glTF camera: only SceneLoaded's two resets
    assert ui["exposureCompensation"] == 2.0 and ui["exposureValue"] == 0.0 and ui["exposureValueMin"] == -3.0
    imp.close()


def test_texture_transform_is_read_past(tmp_path):
    def patch(doc, blob):
        doc["extensionsUsed"].append("KHR_texture_transform")
        doc["materials"][0].setdefault("pbrMetallicRoughness", {})["baseColorTexture"] = {"index": 0, "extensions": {"KHR_texture_transform": {"offset": [0.5, 0.5], "scale": [2.0, 2.0], "rotation": 1.0}}}
    media, sc = _folder(tmp_path, patch)
    imp = pt.SceneImport(media / "t.scene.json")      # (the texture index names no texture: "not loaded", as for any missing image; the transform changes nothing)
    assert imp.info["numGeometries"] == len(sc["geometries"])
    imp.close()


@pytest.mark.gpu
def test_gltf_punctual_lights_reach_the_context(tmp_path):
    """pt_load_scene_gltf on a file with KHR_lights_punctual: the frame equals, bit for bit, the frame of the same scene with the same lights handed over by hand — the point / spot records
    through pt_set_lights (scene dict key "lights"), the directional light through Sample::UpdateLighting's host step (pt_env_bake_lights) into pt_set_environment_bake — under a ROTATED
    environment, so that the conversion the library performs at bake time (pt_set_scene_directional_lights) is seen to use the environment's orientation and the bake's cube size."""
    import ctypes
    q = (math.sin(0.4), 0.0, 0.0, math.cos(0.4))
    media = tmp_path / "m"; media.mkdir()
    sc, cam = scenes.cornell_box("C2")
    path = media / "lit.gltf"; write_gltf(sc, str(path))
    doc = json.loads(path.read_text())
    doc["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "point", "color": [1.0, 0.6, 0.3], "intensity": 3.0}, {"type": "spot", "intensity": 8.0, "spot": {"innerConeAngle": 0.3, "outerConeAngle": 0.7}},
                                                             {"type": "directional", "color": [0.9, 0.95, 1.0], "intensity": 2.0}]}}
    n0 = len(doc["nodes"])
    doc["nodes"] += [{"name": "bulb", "translation": [0.2, 0.4, 0.2], "extensions": {"KHR_lights_punctual": {"light": 0}}},
                     {"name": "spot", "translation": [0.3, 0.5, 0.1], "rotation": list(q), "extensions": {"KHR_lights_punctual": {"light": 1}}},
                     {"name": "sun", "rotation": list(q), "extensions": {"KHR_lights_punctual": {"light": 2}}}]
    doc["scenes"][0]["nodes"] += [n0, n0 + 1, n0 + 2]
    path.write_text(json.dumps(doc))
    (media / "c.scene.json").write_text(json.dumps({"models": ["lit.gltf"], "graph": [{"model": 0}]}))
    imp = pt.SceneImport(media / "c.scene.json")
    assert imp.info["numLights"] == 2 and imp.directional_lights.shape[0] == 1
    S = scenes.config_settings("C2"); w, h = 160, 96
    camd = scenes.bridge_camera(w, h, **cam)
    rgb, tw, cm = sc["env"]
    c_, s_ = math.cos(0.7), math.sin(0.7)
    rot = np.array([c_, 0, s_, 0, 0, 1, 0, 0, -s_, 0, c_, 0], np.float32)      # the environment rotated about y
    prm = pt.PtEnvMapSceneParams((ctypes.c_float * 12)(*rot.tolist()), (ctypes.c_float * 3)(*(cm * np.float32(4.0)).tolist()), 1.0)
    CUBE = 128

    def environment(t, baked_lights):
        assert t.L.pt_set_environment(t.h, rgb.ctypes.data_as(ctypes.c_void_p), rgb.shape[1], rgb.shape[0], ctypes.byref(prm)) == 0
        n = 0 if baked_lights is None else baked_lights.shape[0]
        assert t.L.pt_set_environment_bake(t.h, CUBE, baked_lights.ctypes.data_as(ctypes.c_void_p) if n else None, n) == 0
    # by hand: the imported buffers, the import object's light records, the directional light converted by the host step
    conv = np.zeros((1, 8), np.float32); wl = np.ascontiguousarray(imp.directional_lights, np.float32)
    f = pt.load_library().pt_env_bake_lights; f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    assert f(wl.ctypes.data_as(ctypes.c_void_p), 1, ctypes.byref(prm), CUBE, conv.ctypes.data_as(ctypes.c_void_p)) == 0
    assert not np.allclose(conv[0, 4:7], wl[0, 4:7])      # (the rotation does act on the direction)
    sc2 = dict(sc); sc2["materials"] = imp.materials.copy(); sc2["lights"] = (imp.lights.copy(), imp.lights_ex.copy()); sc2["env"] = None
    a = pt.PathTracer(); a.set_scene(sc2); environment(a, conv); a.set_camera(camd); a.set_settings(S); a.resize(w, h); a.render(0, 2)
    # through the loader
    b = pt.PathTracer(); environment(b, None); b.load_scene_gltf(str(path)); b.set_camera(camd); b.set_settings(S); b.resize(w, h); b.render(0, 2)
    la, lb = a.lights(), b.lights()
    assert np.array_equal(la["lights"], lb["lights"]) and np.array_equal(la["lightsEx"], lb["lightsEx"])
    assert np.array_equal(a.radiance(), b.radiance())
    # ... and the lights do something: without them the frame differs
    sc3 = dict(sc2); sc3["lights"] = None
    c = pt.PathTracer(); c.set_scene(sc3); environment(c, None); c.set_camera(camd); c.set_settings(S); c.resize(w, h); c.render(0, 2)
    assert not np.array_equal(a.radiance(), c.radiance())
    for t in (a, b, c): t.close()
    imp.close()


@pytest.mark.gpu
def test_scene_json_through_pt_load_scene_gltf(tmp_path):
    """pt_load_scene_gltf on a `.scene.json`: import + apply + the graph's directional light + the EnvironmentLight's image — here a cube-map .dds (EnvMapBaker's
    BackgroundSourceType 2) — with the environment UI block at identity, as after Sample::SceneLoaded. The frame equals the same steps taken by hand; a second folder names a
    lat-long .hdr; an unreadable image leaves the scene without one."""
    import ctypes, struct
    from test_hdr_images import write_hdr
    media = tmp_path / "m"; media.mkdir()
    sc, cam = scenes.cornell_box("C2")
    write_gltf(sc, str(media / "box.gltf"))
    rng = np.random.default_rng(8); d = 24
    faces = np.concatenate([(rng.random((6, d, d, 3), np.float32) ** 2 * 3.0).astype(np.float16).astype(np.float32), np.ones((6, d, d, 1), np.float32)], axis=-1)
    hdr = b"DDS " + struct.pack("<7I", 124, 0x1007, d, d, d * 8, 0, 1) + b"\0" * 44 + struct.pack("<2I4s5I", 32, 4, struct.pack("<I", 113), 0, 0, 0, 0, 0) + struct.pack("<5I", 0x1008, 0xFE00, 0, 0, 0)
    (media / "sky.dds").write_bytes(hdr + faces.astype(np.float16).tobytes())
    rgbe = rng.integers(100, 140, (16, 32, 4)).astype(np.uint8); write_hdr(media / "sky.hdr", rgbe, True)
    graph = lambda env: [{"model": 0}, {"name": "sun", "type": "DirectionalLight", "rotation": [math.sin(0.5), 0.0, 0.0, math.cos(0.5)], "color": [1.0, 0.9, 0.8], "irradiance": 1.5, "angularSize": 2.0},
                         {"type": "EnvironmentLight", "path": env}]
    for name in ("sky.dds", "sky.hdr", "nothing.exr"):
        (media / ("s_%s.scene.json" % name)).write_text(json.dumps({"models": ["box.gltf"], "graph": graph(name)}))
    S = scenes.config_settings("C2"); w, h = 128, 80
    camd = scenes.bridge_camera(w, h, **cam)
    frames = {}
    for name in ("sky.dds", "sky.hdr", "nothing.exr"):
        scene = media / ("s_%s.scene.json" % name)
        a = pt.PathTracer(); a.load_scene_gltf(str(scene)); a.set_camera(camd); a.set_settings(S); a.resize(w, h); a.render(0, 2)
        imp = pt.SceneImport(scene); assert imp.directional_lights.shape[0] == 1
        b = pt.PathTracer(); b.apply_scene_import(imp)
        dl = np.ascontiguousarray(imp.directional_lights, np.float32)
        assert b.L.pt_set_scene_directional_lights(b.h, dl.ctypes.data_as(ctypes.c_void_p), 1) == 0
        if name == "sky.dds": assert b.L.pt_set_environment_cube(b.h, np.ascontiguousarray(pt.read_dds_cube(media / name)).ctypes.data_as(ctypes.c_void_p), d, None) == 0
        elif name == "sky.hdr":
            img = np.ascontiguousarray(pt.read_float_image(media / name)); assert b.L.pt_set_environment(b.h, img.ctypes.data_as(ctypes.c_void_p), img.shape[1], img.shape[0], None) == 0
        b.set_camera(camd); b.set_settings(S); b.resize(w, h); b.render(0, 2)
        assert np.array_equal(a.radiance(), b.radiance()), name
        if name != "nothing.exr": assert np.array_equal(a.env_cube()[0], b.env_cube()[0]) and a.env_cube()[0].any()
        frames[name] = a.radiance().copy(); a.close(); b.close(); imp.close()
    assert not np.array_equal(frames["sky.dds"], frames["sky.hdr"]) and not np.array_equal(frames["sky.dds"], frames["nothing.exr"])
