"""RTXPT `.scene.json` asset folders through pt_scene_json_import — host only, no device. The leaves and what Sample::SceneLoaded does with
them follow the reference tree (ExtendedScene.cpp:104-143, 313-375; Sample.cpp:457-479, 520-640; MaterialsBaker.cpp:707-748, 857-864); the graph format is
Donut's (restated, see include/mi355pt.h). Checks are against independent numpy compositions, pt_convert_light / pt_material_from_json (each pinned to the
reference text elsewhere), and — on the GPU — against the same scene handed over through the raw-buffer entry points."""
import json
import math
import os
import sys

import numpy as np
import pytest

import rtxpt_amd as pt
from rtxpt_amd import scenes

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gltf_writer import write_gltf

NAMES = ["white", "red", "green", "light", "glass", "gold"]


def trs_matrix(t=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)
    M = np.eye(4); M[:3, :3] = R * np.asarray(s, np.float64)[None, :]; M[:3, 3] = t
    return M


def make_folder(tmp_path, graph, overrides=None, scene_name="test.scene.json", textures=None):
    media = tmp_path / "media"; (media / "Models").mkdir(parents=True, exist_ok=True)
    sc, cam = scenes.cornell_box("C2")
    sc["material_names"] = NAMES
    write_gltf(sc, str(media / "Models" / "cornell.gltf"))
    for rel, doc in (overrides or {}).items():
        f = media / "Materials" / rel; f.parent.mkdir(parents=True, exist_ok=True); f.write_text(json.dumps(doc))
    for rel, rgba in (textures or {}).items():
        f = media / rel; f.parent.mkdir(parents=True, exist_ok=True); pt.write_image(str(f), rgba)
    (media / scene_name).write_text(json.dumps({"models": ["Models/cornell.gltf"], "graph": graph}))
    return media, sc, cam


Q_Y90 = (0.0, math.sin(math.pi / 4), 0.0, math.cos(math.pi / 4))
GRAPH = [
    {"name": "room", "model": 0, "translation": [1.0, 2.0, 3.0], "rotation": list(Q_Y90), "scaling": 2.0},
    {"name": "group", "translation": [10.0, 0.0, 0.0], "children": [{"name": "room2", "model": 0, "scaling": [1.0, 0.5, 1.0]}, "Models/cornell.gltf"]},
    {"name": "Lights", "translation": [0.0, 1.0, 0.0], "children": [
        {"name": "spot", "type": "SpotLight", "translation": [0.2, 0.3, 0.1], "rotation": [0.7071067811865476, 0.0, 0.0, 0.7071067811865476], "color": [1.0, 0.8, 0.6], "intensity": 50.0,
         "radius": 0.05, "innerAngle": 20.0, "outerAngle": 45.0, "proxyMeshNodes": ["/room"]},
        {"name": "off", "type": "PointLight", "intensity": 0.0, "radius": 0.1},
        {"name": "bulb", "type": "PointLight", "translation": [0.1, 0.1, 0.1], "color": 0.5, "intensity": 20.0, "radius": 0.02},
        {"name": "sun", "type": "DirectionalLight", "irradiance": 3.0}]},
    {"name": "Cameras", "children": [
        {"name": "Default", "type": "PerspectiveCameraEx", "translation": [0.278, 0.273, -0.8], "rotation": [0.0, 1.0, 0.0, 0.0], "verticalFov": 0.6859, "zNear": 0.01,
         "enableAutoExposure": False, "exposureCompensation": 1.5, "exposureValue": -2.0},
        {"name": "Second", "type": "PerspectiveCamera", "translation": [0.0, 1.0, 5.0]}]},
    {"name": "Sky", "type": "EnvironmentLight", "path": "EnvironmentMaps/sky.exr", "radianceScale": [2.0, 2.0, 1.5], "rotation": 0.25},
    {"name": "Sky2", "type": "EnvironmentLight", "path": "ignored.exr"},
    {"name": "Settings", "type": "SampleSettings", "maxBounces": 7, "startingCamera": 0, "realtimeFireflyFilter": 0.5, "enableAnimations": True},
    {"name": "Game", "type": "GameSettings", "anything": 1},
]


def test_scene_graph_instances_lights_cameras(tmp_path):
    media, sc, cam = make_folder(tmp_path, GRAPH)
    imp = pt.SceneImport(media / "test.scene.json")
    I = imp.info
    base = sc["instances"]
    assert I["numModels"] == 1 and I["numInstances"] == 3 * len(base) and I["numMeshes"] == len(sc["meshes"]) and I["numGeometries"] == len(sc["geometries"])
    assert I["numMaterials"] == len(sc["materials"]) + 1 and I["materialOverrides"] == 0 and I["numTextures"] == 0
    # instances: node world (double) x glTF instance matrix, then fp32
    worlds = [trs_matrix((1, 2, 3), Q_Y90, (2, 2, 2)), trs_matrix((10, 0, 0)) @ trs_matrix(s=(1, 0.5, 1)), trs_matrix((10, 0, 0))]
    k = 0
    for W in worlds:
        for inst in base:
            m = np.eye(4); m[:3, :] = inst["transform"].reshape(3, 4).astype(np.float64)
            want = (W @ m)[:3, :].astype(np.float32).reshape(-1)
            got = imp.instances[k]["transform"]
            assert np.allclose(got, want, rtol=0, atol=1e-6 * max(1.0, np.abs(want).max())), (k, got, want)
            assert imp.instances[k]["meshIndex"] == inst["meshIndex"]
            k += 1
    assert np.array_equal(imp.geometries["materialIndex"], sc["geometries"]["materialIndex"])
    # lights: scene-graph order, invisible one dropped, directional not baked; records == pt_convert_light on the node's world position / -Z axis
    assert I["numLights"] == 2 and I["lightsDropped"] == 1 and I["directionalLights"] == 1 and I["lightProxies"] == 1
    Wl = trs_matrix((0, 1, 0)) @ trs_matrix((0.2, 0.3, 0.1), (0.7071067811865476, 0, 0, 0.7071067811865476))
    z = Wl[:3, 2]; d = -z / np.linalg.norm(z)
    b, e = pt.convert_light("spot", Wl[:3, 3].astype(np.float32), (1.0, 0.8, 0.6), 50.0, 0.05, d.astype(np.float32), 20.0, 45.0)
    assert np.array_equal(imp.lights[0], b) and np.array_equal(imp.lights_ex[0], e)
    assert np.allclose(d, (0, 1, 0), atol=1e-12)              # +90 degrees about X turns -Z into +Y
    b, e = pt.convert_light("point", (0.1, 1.1, 0.1), (0.5, 0.5, 0.5), 20.0, 0.02, (0.0, 0.0, -1.0), 180.0, 180.0)
    assert np.array_equal(imp.lights[1], b) and np.array_equal(imp.lights_ex[1], e)
    # the directional light is handed out as an EMB_DirectionalLight record (world space) for the environment-cube bake: Donut defaults colour 1, angularSize 0
    assert imp.directional_lights.shape == (1, 8) and np.array_equal(imp.directional_lights[0], np.array([1, 1, 1, 3.0, 0, 0, -1, 0], np.float32))
    # cameras (Sample::UpdateCameraFromScene): a half turn about Y makes the glTF-style -Z camera look along +Z
    assert I["numCameras"] == 2 and I["selectedCamera"] == 0
    c0, c1 = imp.cameras
    assert c0["name"] == b"Default" and np.allclose(c0["position"], (0.278, 0.273, -0.8)) and np.allclose(c0["direction"], (0, 0, 1), atol=1e-7) and np.allclose(c0["up"], (0, 1, 0), atol=1e-7)
    assert abs(c0["verticalFov"] - 0.6859) < 1e-7 and abs(c0["zNear"] - 0.01) < 1e-9
    assert c0["exposureMask"] == 0b00111 and c0["enableAutoExposure"] == 0 and c0["exposureCompensation"] == 1.5 and c0["exposureValue"] == -2.0
    assert c1["exposureMask"] == 0 and c1["verticalFov"] == 1.0 and c1["zNear"] == 1.0 and np.allclose(c1["direction"], (0, 0, -1)) and np.allclose(c1["position"], (0, 1, 5))
    # environment: first EnvironmentLight wins; settings: the seven SampleSettings keys
    assert I["hasEnvironment"] == 1 and I["envPath"] == b"EnvironmentMaps/sky.exr" and np.allclose(I["envRadianceScale"], (2, 2, 1.5)) and I["envRotation"] == 0.25 and I["envTextureIndex"] == -1
    assert I["settingsMask"] == (2 | 4 | 8 | 16) and I["maxBounces"] == 7 and I["startingCamera"] == 0 and I["realtimeFireflyFilter"] == 0.5 and I["enableAnimations"] == 1
    # tone-mapping block after the load (Sample.cpp:547-549 + UpdateCameraFromScene :467-478)
    ui = imp.tone_mapping(pt.default_tone_mapping_parameters(exposureValueMin=-3.0, whiteScale=7.0))
    assert ui["autoExposure"] == 0 and ui["exposureCompensation"] == 1.5 and ui["exposureValue"] == -2.0 and ui["exposureValueMin"] == -16.0 and ui["exposureValueMax"] == 16.0 and ui["whiteScale"] == 7.0
    ui = imp.tone_mapping(pt.default_tone_mapping_parameters(exposureCompensation=5.0, autoExposure=1), camera=1)      # a camera without exposure keys resets to the defaults
    assert ui["autoExposure"] == 0 and ui["exposureCompensation"] == 0.0 and ui["exposureValue"] == 0.0
    with pytest.raises(pt.PtError):
        imp.tone_mapping(camera=7)
    st = scenes.default_settings(); before = st.copy()
    imp.apply_settings(st)
    assert st["bounceCount"] == 7 and st["diffuseBounceCount"] == before["diffuseBounceCount"] and st["texLODBias"] == before["texLODBias"]
    imp.close()


def test_selected_camera_defaults_to_last(tmp_path):
    g = [n for n in GRAPH if not (isinstance(n, dict) and n.get("type") == "SampleSettings")]
    media, _, _ = make_folder(tmp_path, g)
    imp = pt.SceneImport(media / "test.scene.json")
    assert imp.info["settingsMask"] == 0 and imp.info["selectedCamera"] == 1
    g2 = g + [{"type": "SampleSettings", "startingCamera": 5}]
    media, _, _ = make_folder(tmp_path, g2)
    assert pt.SceneImport(media / "test.scene.json").info["selectedCamera"] == 1            # out of range: fall back to the last camera


def test_dds_next_to_png_wins(tmp_path):
    """PTMaterial::Read's loadTexture (MaterialsBaker.cpp:178-191): `x.dds` beside `x.png` replaces it — the BC7 files the reference's compression script writes; the sRGB flag
    stays the document's. A `.dds` named directly is read too."""
    import struct
    checker = np.zeros((4, 8, 4), np.uint8); checker[..., 3] = 255; checker[::2, ::2, :3] = 255
    doc = {"BaseTexture": {"path": "Textures/checker.png", "sRGB": True}, "EmissiveTexture": {"path": "Textures/glow.dds", "sRGB": False}}
    media, sc, _ = make_folder(tmp_path, [{"model": 0}], {"red.material.json": doc}, textures={"Textures/checker.png": checker})
    rng = np.random.default_rng(2); blocks = rng.integers(0, 256, 3 * 2 * 16, dtype=np.uint8); blocks[0::16] |= 1          # 12 x 8 BC7 (mode 0 blocks)
    hdr = lambda w, h, dx: b"DDS " + struct.pack("<7I", 124, 0x1007, h, w, 0, 0, 1) + b"\0" * 44 + struct.pack("<2I4s5I", 32, 4, b"DX10", 0, 0, 0, 0, 0) + struct.pack("<5I", 0x1000, 0, 0, 0, 0) + struct.pack("<5I", dx, 3, 0, 1, 0)
    (media / "Textures" / "checker.dds").write_bytes(hdr(12, 8, 98) + blocks.tobytes())
    glow = rng.integers(0, 256, (5, 3, 4), dtype=np.uint8); (media / "Textures" / "glow.dds").write_bytes(hdr(3, 5, 28) + glow.tobytes())
    imp = pt.SceneImport(media / "test.scene.json")
    assert imp.info["numTextures"] == 2 and imp.info["texturesNotLoaded"] == 0
    px, fmt = imp.texture(0); want, _ = pt.read_dds(media / "Textures" / "checker.dds")
    assert fmt == pt.PT_TEX_RGBA8_SRGB and px.shape == (8, 12, 4) and np.array_equal(px, want)
    assert imp.materials[1]["BaseOrDiffuseTextureIndex"] == scenes.pack_texture_word(0, 12, 8)
    px, fmt = imp.texture(1)
    assert fmt == pt.PT_TEX_RGBA8_UNORM and np.array_equal(px, glow) and imp.materials[1]["EmissiveTextureIndex"] == scenes.pack_texture_word(1, 3, 5)
    with pytest.raises(pt.PtError): imp.texture(2)


def test_material_overrides_follow_reference_candidate_order(tmp_path):
    checker = np.zeros((4, 8, 4), np.uint8); checker[..., 3] = 255; checker[::2, ::2, :3] = 255
    red_doc = {"BaseOrDiffuseColor": [0.1, 0.2, 0.3], "Roughness": 0.25, "EnableAlphaTesting": True, "AlphaCutoff": 0.3, "ExcludeFromNEE": True,
               "BaseTexture": {"path": "Textures/checker.png", "sRGB": True}, "NormalTexture": {"path": "Textures/missing.dds", "NormalMap": True}}
    overrides = {
        "test.scene/cornell.red.material.json": red_doc,                                    # candidate 0 beats the three below
        "test.scene/red.material.json": {"Roughness": 0.9},
        "cornell.red.material.json": {"Roughness": 0.8},
        "red.material.json": {"Roughness": 0.7},
        "cornell.green.material.json": {"Metalness": 1.0, "Roughness": 0.125},             # candidate 2 beats candidate 3
        "green.material.json": {"Metalness": 0.0},
        "gold.material.json": {"SkipRender": True},                                         # shared, unqualified
        "test.scene/white.material.json": "not an object",                                  # unreadable document: falls through to the next candidate
        "white.material.json": {"IoR": 1.33, "EnableTransmission": True, "TransmissionFactor": 0.5, "NestedPriority": 3},
    }
    media, sc, _ = make_folder(tmp_path, [{"model": 0}], overrides, textures={"Textures/checker.png": checker})
    imp = pt.SceneImport(media / "test.scene.json")
    I = imp.info
    assert I["materialOverrides"] == 4 and I["numTextures"] == 1 and I["texturesNotLoaded"] == 1 and I["skippedGeometries"] == 1
    word = scenes.pack_texture_word(0, 8, 4)
    want, info = pt.material_from_json(json.dumps(red_doc), (word, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF))
    assert imp.materials[1].tobytes() == want.tobytes() and info["enableAlphaTesting"] and info["excludeFromNEE"]
    assert imp.materials[1]["BaseOrDiffuseTextureIndex"] == word and imp.materials[1]["NormalTextureIndex"] == 0xFFFFFFFF
    assert imp.materials[2].tobytes() == pt.material_from_json(json.dumps(overrides["cornell.green.material.json"]))[0].tobytes()
    assert imp.materials[0].tobytes() == pt.material_from_json(json.dumps(overrides["white.material.json"]))[0].tobytes()
    # untouched materials keep the glTF import (emissive light, glass)
    assert np.allclose(imp.materials[3]["EmissiveColor"], sc["materials"][3]["EmissiveColor"]) and imp.materials[4]["TransmissionFactor"] == 1.0
    # geometry flags follow the document; the SkipRender material's geometry (the tall box) is gone together with its mesh and instance
    g = imp.geometries
    assert all(g["geomFlags"][g["materialIndex"] == 1] == 3) and all(g["geomFlags"][g["materialIndex"] == 0] == 0)
    assert I["numGeometries"] == len(sc["geometries"]) - 1 and I["numMeshes"] == len(sc["meshes"]) - 1 and I["numInstances"] == len(sc["instances"]) - 1
    assert 5 not in set(g["materialIndex"].tolist())
    # the scene-specialised folder is named after filename().stem(): "other.scene.json" -> Materials/other.scene/
    media2, _, _ = make_folder(tmp_path / "b", [{"model": 0}], overrides, scene_name="other.scene.json")
    imp2 = pt.SceneImport(media2 / "other.scene.json")
    assert imp2.materials[1].tobytes() == pt.material_from_json(json.dumps({"Roughness": 0.8}))[0].tobytes()


def test_euler_rotations_follow_donuts_rotation_quat(tmp_path):
    """A node's "euler" key (Donut's dm::rotationQuat, the function the reference's own GameMisc.cpp:53-61 calls for the same key): about the fixed x axis first, then y, then z;
    "rotation" wins when both are present."""
    a, b, c = 0.3, -1.1, 2.0
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]]); ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    media, sc, _ = make_folder(tmp_path, [{"model": 0, "euler": [a, b, c], "translation": [1, 2, 3]}, {"model": 0, "euler": [a, b, c], "rotation": [0, 0, 0, 1]}])
    imp = pt.SceneImport(media / "test.scene.json"); n = len(sc["instances"])
    R = rz @ ry @ rx
    for k in range(n):
        base = np.eye(4); base[:3] = sc["instances"]["transform"][k].reshape(3, 4)
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = [1, 2, 3]
        assert np.allclose(imp.instances["transform"][k].reshape(3, 4), (M @ base)[:3], atol=2e-6)
        assert np.allclose(imp.instances["transform"][n + k].reshape(3, 4), base[:3], atol=1e-7)


def test_media_path_argument_and_errors(tmp_path):
    media, _, _ = make_folder(tmp_path, [{"model": 0}], {"red.material.json": {"Roughness": 0.7}})
    elsewhere = tmp_path / "elsewhere"; (elsewhere / "Materials").mkdir(parents=True)
    (elsewhere / "Materials" / "red.material.json").write_text(json.dumps({"Roughness": 0.33}))
    assert abs(pt.SceneImport(media / "test.scene.json").materials[1]["Roughness"] - 0.7) < 1e-7
    assert abs(pt.SceneImport(media / "test.scene.json", elsewhere).materials[1]["Roughness"] - 0.33) < 1e-7
    for bad, code in (([{"model": 3}], 4), ([{"model": "nope.gltf"}], 4), (["nope.gltf"], 4)):
        m, _, _ = make_folder(tmp_path / "bad", bad)
        with pytest.raises(pt.PtError) as e:
            pt.SceneImport(m / "test.scene.json")
        assert e.value.code == code, (bad, e.value.code)
    # a light without radius converts to the point-type record, as in the reference (inert in the path tracer's light set, but it keeps its slot)
    m, _, _ = make_folder(tmp_path / "pointlight", [{"type": "SpotLight", "radius": 0.0, "intensity": 1.0}])
    pl = pt.SceneImport(m / "test.scene.json")
    assert pl.info["numLights"] == 1 and (int(pl.lights[0][3]) >> 24) & 0xF == 4
    (media / "broken.scene.json").write_text("{ \"models\": [")
    with pytest.raises(pt.PtError):
        pt.SceneImport(media / "broken.scene.json")
    with pytest.raises(pt.PtError):
        pt.SceneImport(media / "missing.scene.json")
    (media / "empty.scene.json").write_text(json.dumps({"models": [], "graph": []}))
    e = pt.SceneImport(media / "empty.scene.json")
    assert e.tone_mapping()["exposureCompensation"] == 2.0 and e.tone_mapping(pt.default_tone_mapping_parameters(exposureValue=3.0))["exposureValue"] == 0.0      # no camera: the "sensible defaults"
    assert e.info["numInstances"] == 0 and e.info["selectedCamera"] == -1 and e.info["hasEnvironment"] == 0


@pytest.mark.gpu
def test_scene_json_frame_equals_raw_buffer_frame(tmp_path):
    """The imported folder renders the same frame as the same scene handed over through pt_set_* (materials as the glTF import makes them, lights through
    pt_convert_light, camera from the scene's camera node)."""
    graph = [{"model": 0},
             {"name": "bulb", "type": "PointLight", "translation": [0.27, 0.4, 0.28], "color": [1.0, 0.9, 0.7], "intensity": 3.0, "radius": 0.03},
             {"name": "Default", "type": "PerspectiveCameraEx", "translation": [0.278, 0.273, -0.8], "rotation": [0.0, 1.0, 0.0, 0.0], "verticalFov": math.radians(39.3), "zNear": 0.01}]
    media, sc, cam = make_folder(tmp_path, graph)
    imp = pt.SceneImport(media / "test.scene.json")
    c = imp.cameras[imp.info["selectedCamera"]]
    W = H = 96
    camera = pt.bridge_camera(W, H, c["position"], c["direction"], c["up"], float(c["verticalFov"]), near_z=float(c["zNear"]), far_z=100.0, focal_distance=1.0)
    frames = []
    for use_import in (True, False):
        p = pt.PathTracer()
        if use_import:
            p.apply_scene_import(imp)
        else:
            p.load_scene_gltf(str(media / "Models" / "cornell.gltf"))
            b, e = pt.convert_light("point", (0.27, 0.4, 0.28), (1.0, 0.9, 0.7), 3.0, 0.03, (0.0, 0.0, -1.0), 180.0, 180.0)
            p._chk(p.L.pt_set_lights(p.h, pt._p(b), pt._p(e), 1), "pt_set_lights")
        p.set_settings(p.default_settings()); p.resize(W, H); p.set_camera(camera)
        p.render(0, 4)
        frames.append(p.radiance().copy()); p.close()
    assert np.isfinite(frames[0]).all() and frames[0][..., :3].max() > 0
    assert np.array_equal(frames[0], frames[1])


# ---- the reference-tree half of the import against the reference's own text
def _random_leaf_graph(rng):
    def maybe(p=0.6):
        return rng.random() < p

    def f(lo, hi):
        return float(np.float32(rng.uniform(lo, hi)))
    nodes = []
    for _ in range(int(rng.integers(0, 7))):
        kind = ["SpotLight", "PointLight", "DirectionalLight"][int(rng.integers(0, 3))]
        n = {"name": "l%d" % len(nodes), "type": kind}
        if maybe(): n["color"] = [f(0, 2), f(0, 2), f(0, 2)] if maybe(0.7) else f(0, 2)
        if maybe(0.8): n["intensity"] = 0.0 if maybe(0.15) else f(0.1, 100)
        if kind != "DirectionalLight" or maybe(0.2): n["radius"] = f(0.01, 0.5)
        elif maybe(): n["irradiance"] = 0.0 if maybe(0.3) else f(0.1, 5)
        if maybe(): n["innerAngle"] = f(1, 40)
        if maybe(): n["outerAngle"] = f(41, 120) * (-1.0 if maybe(0.2) else 1.0)
        if maybe(0.3): n["proxyMeshNodes"] = ["/a", "/b"][:int(rng.integers(1, 3))]
        if kind == "SpotLight" and "radius" not in n: n["radius"] = f(0.01, 0.5)       # the reference asserts on spots without radius
        nodes.append(n)
    for _ in range(int(rng.integers(0, 4))):
        n = {"name": "c%d" % len(nodes), "type": ["PerspectiveCamera", "PerspectiveCameraEx"][int(rng.integers(0, 2))], "translation": [f(-5, 5), f(-5, 5), f(-5, 5)]}
        if maybe(): n["verticalFov"] = f(0.2, 2.0)
        if maybe(): n["zNear"] = f(0.001, 1.0)
        if maybe(0.5): n["enableAutoExposure"] = bool(maybe(0.5)) if maybe(0.8) else 1
        for k in ("exposureCompensation", "exposureValue", "exposureValueMin", "exposureValueMax"):
            if maybe(0.4): n[k] = f(-8, 8)
        nodes.append(n)
    for _ in range(int(rng.integers(0, 3))):
        n = {"type": "EnvironmentLight", "path": "env%d.exr" % len(nodes)}
        if maybe(): n["radianceScale"] = [f(0, 4), f(0, 4), f(0, 4)] if maybe(0.5) else f(0, 4)
        if maybe(): n["rotation"] = f(0, 6.28)
        if maybe(0.3): n["textureIndex"] = int(rng.integers(0, 5))
        nodes.append(n)
    for _ in range(int(rng.integers(0, 3))):
        n = {"type": "SampleSettings"}
        if maybe(0.4): n["realtimeMode"] = bool(maybe(0.5))
        if maybe(0.4): n["enableAnimations"] = bool(maybe(0.5))
        if maybe(0.5): n["startingCamera"] = int(rng.integers(0, 4))
        if maybe(0.4): n["realtimeFireflyFilter"] = f(0.05, 2)
        if maybe(0.5): n["maxBounces"] = int(rng.integers(1, 40))
        if maybe(0.5): n["maxDiffuseBounces"] = int(rng.integers(0, 6))
        if maybe(0.4): n["textureMIPBias"] = f(-2, 2)
        nodes.append(n)
    nodes.append({"type": "GameSettings", "x": 1}); nodes.append({"type": "SomethingElse"})
    order = rng.permutation(len(nodes))
    nodes = [nodes[i] for i in order]
    # nest a few under a group node to exercise the traversal order
    if len(nodes) > 3:
        k = int(rng.integers(1, len(nodes) - 1))
        nodes = nodes[:k] + [{"name": "group", "children": nodes[k:k + 2]}] + nodes[k + 2:]
    return nodes


def test_scene_leaves_match_reference_text(tmp_path):
    from oracle import ptref
    if ptref.reference_scene_leaves(json.dumps({"graph": []})) is None:
        pytest.skip("oracle/_ref/librefpin_mat.so not built (no /root/reference here)")
    rng = np.random.default_rng(2026)
    (tmp_path / "m").mkdir()
    for case in range(300):
        graph = _random_leaf_graph(rng)
        doc = json.dumps({"models": [], "graph": graph})
        path = tmp_path / "m" / "leaves.scene.json"; path.write_text(doc)
        ref = ptref.reference_scene_leaves(doc)
        imp = pt.SceneImport(path); I = imp.info
        ctx = (case, doc)
        assert I["numLights"] == ref["numLights"] and I["directionalLights"] >= ref["directional"] and I["lightProxies"] == ref["proxies"], ctx
        assert len(imp.directional_lights) == ref["directional"], ctx            # the ones Sample::SceneLoaded's clean-up keeps (Sample.cpp:563-570)
        n = min(int(I["numLights"]), 16)
        assert np.array_equal(np.concatenate([imp.lights, imp.lights_ex], axis=1)[:n], ref["lights"][:n]), ctx
        assert I["numCameras"] == ref["numCameras"] and I["hasEnvironment"] == ref["hasEnvironment"], ctx
        if ref["hasEnvironment"]:
            assert bytes(I["envPath"]).split(b"\0")[0].decode() == ref["envPath"] and np.array_equal(I["envRadianceScale"], ref["envRadianceScale"]) and I["envRotation"] == ref["envRotation"] and I["envTextureIndex"] == ref["envTextureIndex"], ctx
        # settings as Sample::SceneLoaded applies them to the reference's UI defaults (BounceCount 20, DiffuseBounceCount 2, TexLODBias -1)
        st = scenes.default_settings(); st["bounceCount"], st["diffuseBounceCount"], st["texLODBias"] = 20, 2, -1.0
        imp.apply_settings(st)
        assert st["bounceCount"] == ref["bounceCount"] and st["diffuseBounceCount"] == ref["diffuseBounceCount"] and st["texLODBias"] == np.float32(ref["texLODBias"]), ctx
        assert (int(I["realtimeMode"]) if I["settingsMask"] & 1 else 1) == ref["realtimeMode"] and (int(I["enableAnimations"]) if I["settingsMask"] & 2 else 0) == ref["enableAnimations"], ctx
        if I["settingsMask"] & 8:
            assert I["realtimeFireflyFilter"] == np.float32(ref["realtimeFireflyFilterThreshold"]), ctx
        assert (int(I["startingCamera"]) + 1 if I["settingsMask"] & 4 else 0) == ref["selectedCameraIndex"], ctx
        # camera + tone-mapping block of the camera the reference ends up on
        if I["numCameras"]:
            c = imp.cameras[int(I["selectedCamera"])]
            assert np.array_equal(c["position"], ref["cameraPos"]) and np.array_equal(c["position"] + c["direction"], ref["cameraTarget"]) and np.array_equal(c["up"], ref["cameraUp"]), ctx
            assert c["verticalFov"] == np.float32(ref["verticalFov"]) and c["zNear"] == np.float32(ref["zNear"]), ctx
        ui = imp.tone_mapping()
        assert (int(ui["autoExposure"]), float(ui["exposureCompensation"]), float(ui["exposureValue"]), float(ui["exposureValueMin"]), float(ui["exposureValueMax"])) == \
            (ref["autoExposure"], ref["exposureCompensation"], ref["exposureValue"], ref["exposureValueMin"], ref["exposureValueMax"]), ctx
        imp.close()


@pytest.mark.gpu
def test_scene_json_with_overrides_renders(tmp_path):
    """An asset folder whose `.material.json` overrides bring a PNG texture, alpha testing and a SkipRender material goes through pt_scene_import_apply and renders:
    finite, different from the same folder without overrides, and the SkipRender box is gone (its pixels now show the wall behind it)."""
    checker = np.zeros((8, 8, 4), np.uint8); checker[..., 3] = 255; checker[::2, ::2, :3] = 255; checker[1::2, 1::2, :3] = 255
    overrides = {"test.scene/cornell.red.material.json": {"BaseOrDiffuseColor": [1.0, 1.0, 1.0], "Roughness": 0.8, "BaseTexture": {"path": "Textures/checker.png", "sRGB": True}},
                 "gold.material.json": {"SkipRender": True},
                 "green.material.json": {"BaseOrDiffuseColor": [0.1, 0.9, 0.1], "Metalness": 1.0, "Roughness": 0.2}}
    cam_node = {"name": "Default", "type": "PerspectiveCameraEx", "translation": [0.278, 0.273, -0.8], "rotation": [0.0, 1.0, 0.0, 0.0], "verticalFov": math.radians(39.3), "zNear": 0.01}
    W, H = 96, 96
    frames = []
    for ov, sub in ((overrides, "a"), (None, "b")):
        media, sc, cam = make_folder(tmp_path / sub, [{"model": 0}, cam_node], ov, textures={"Textures/checker.png": checker} if ov else None)
        imp = pt.SceneImport(media / "test.scene.json")
        assert imp.info["materialOverrides"] == (3 if ov else 0) and imp.info["numTextures"] == (1 if ov else 0)
        c = imp.cameras[0]
        p = pt.PathTracer(); p.apply_scene_import(imp)
        p.set_settings(imp.apply_settings(p.default_settings())); p.resize(W, H)
        p.set_camera(pt.bridge_camera(W, H, c["position"], c["direction"], c["up"], float(c["verticalFov"]), near_z=float(c["zNear"]), far_z=100.0, focal_distance=1.0))
        p.render(0, 8)
        frames.append(p.radiance().copy()); p.close(); imp.close()
    a, b = frames
    assert np.isfinite(a).all() and np.isfinite(b).all() and a[..., :3].max() > 0
    assert not np.array_equal(a, b)
    # the tall gold box (right half of the image, lower part) is metallic yellow in b and absent in a: the mean colour of that region moves away from gold
    region = (slice(40, 80), slice(50, 80))
    gold_b = b[region][..., :3].mean((0, 1)); gold_a = a[region][..., :3].mean((0, 1))
    assert gold_b[0] > 1.15 * gold_b[2] and abs(gold_a[0] / max(gold_a[2], 1e-6) - gold_b[0] / max(gold_b[2], 1e-6)) > 0.1


def test_scene_json_parser_survives_damaged_documents(tmp_path):
    """Truncated and byte-flipped scene / material documents must come back as an error code or a clean import, never as a crash."""
    media, _, _ = make_folder(tmp_path, GRAPH, {"red.material.json": {"Roughness": 0.7, "BaseTexture": {"path": "Textures/none.png"}}})
    good = (media / "test.scene.json").read_bytes()
    mat = (media / "Materials" / "red.material.json").read_bytes()
    rng = np.random.default_rng(99)
    ok = bad = 0
    for i in range(400):
        doc = bytearray(good)
        mode = i % 4
        if mode == 0:
            doc = doc[:int(rng.integers(0, len(doc)))]
        elif mode == 1:
            for _ in range(int(rng.integers(1, 6))):
                doc[int(rng.integers(0, len(doc)))] = int(rng.integers(32, 127))
        elif mode == 2:
            a = int(rng.integers(0, len(doc))); b = min(len(doc), a + int(rng.integers(1, 40))); del doc[a:b]
        else:
            m = bytearray(mat); m[int(rng.integers(0, len(m)))] = int(rng.integers(32, 127)); (media / "Materials" / "red.material.json").write_bytes(bytes(m))
        (media / "fuzz.scene.json").write_bytes(bytes(doc))
        try:
            imp = pt.SceneImport(media / "fuzz.scene.json"); imp.close(); ok += 1
        except pt.PtError as e:
            assert e.code in (1, 4, 5), e.code; bad += 1
    assert ok > 0 and bad > 0


def test_hostile_gltf_documents_return_error_codes(tmp_path):
    """File-controlled sizes never reach an allocation or a read unchecked, and no exception crosses the C ABI: a negative accessor count, offsets beyond the
    buffer, a wrong IHDR length, absurd PNG sizes, a truncated number at the end of the buffer and 100 000 nested brackets all come back as error codes
    (the host process would abort on an exception escaping through ctypes)."""
    import struct, zlib
    media, sc, cam = make_folder(tmp_path, [{"model": 0}])
    gltf = media / "Models" / "cornell.gltf"
    good = json.loads(gltf.read_text())

    def expect_error(doc_text, name="bad"):
        (media / "Models" / (name + ".gltf")).write_text(doc_text)
        (media / (name + ".scene.json")).write_text(json.dumps({"models": ["Models/%s.gltf" % name], "graph": [{"model": 0}]}))
        with pytest.raises(pt.PtError):
            pt.SceneImport(media / (name + ".scene.json"))
    for mutate in (lambda d: d["accessors"][0].__setitem__("count", -1),
                   lambda d: d["accessors"][0].__setitem__("count", 2 ** 31 - 1),
                   lambda d: d["accessors"][0].__setitem__("byteOffset", -8),
                   lambda d: d["bufferViews"][0].__setitem__("byteOffset", 1e18),
                   lambda d: d["bufferViews"][0].__setitem__("byteStride", -4)):
        d = json.loads(json.dumps(good)); mutate(d); expect_error(json.dumps(d))
    expect_error("[" * 100000)
    expect_error('{"asset": {"version": "2.0"}, "x": 12')                    # ends inside a number
    expect_error('{"a": "\\u12')                                           # ends inside an escape
    # PNG with a short IHDR / absurd dimensions is "texture not loaded", not a crash
    def png(ihdr):
        def chunk(t, b):
            return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b"")
    for k, ihdr in enumerate((struct.pack(">II", 4, 4), struct.pack(">IIBBBBB", 0x7FFFFFFF, 0x7FFFFFFF, 8, 6, 0, 0, 0), struct.pack(">IIBBBBB", 60000, 60000, 8, 6, 0, 0, 0))):
        (media / "Models" / ("t%d.png" % k)).write_bytes(png(ihdr))
        d = json.loads(json.dumps(good)); d["images"] = [{"uri": "t%d.png" % k}]; d["textures"] = [{"source": 0}]
        d["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
        (media / "Models" / "png.gltf").write_text(json.dumps(d))
        (media / "png.scene.json").write_text(json.dumps({"models": ["Models/png.gltf"], "graph": [{"model": 0}]}))
        imp = pt.SceneImport(media / "png.scene.json")
        assert imp.info["numTextures"] == 0
    # pt_material_from_json on the same kinds of damage
    for text in ("[" * 100000, '{"Roughness": 0.', '{"Name": "\\u'):
        with pytest.raises(pt.PtError):
            pt.material_from_json(text)


def test_directional_light_records_and_bake_helper(tmp_path):
    """DirectionalLight leaves -> EMB_DirectionalLight rows (colour, irradiance, the node's -Z, radians(clamp(angularSize, 0, 90))); pt_env_bake_lights is
    Sample::UpdateLighting's preprocessing (Sample.cpp:1361-1388): angular size raised to pi / (cubeDim / 2), direction into the environment's local frame."""
    graph = [{"name": "sun", "type": "DirectionalLight", "rotation": [0.7071067811865476, 0.0, 0.0, 0.7071067811865476], "color": [1.0, 0.9, 0.8], "irradiance": 2.5, "angularSize": 0.53},
             {"name": "wide", "type": "DirectionalLight", "rotation": list(Q_Y90), "angularSize": 400.0},
             {"name": "dark", "type": "DirectionalLight", "irradiance": 0.0}]
    path = tmp_path / "d.scene.json"; path.write_text(json.dumps({"models": [], "graph": graph}))
    imp = pt.SceneImport(path)
    assert imp.info["directionalLights"] == 3 and imp.info["lightsDropped"] == 1
    d = imp.directional_lights
    assert d.shape == (2, 8)
    assert np.array_equal(d[0, :4], np.array([1.0, 0.9, 0.8, 2.5], np.float32)) and np.allclose(d[0, 4:7], (0, 1, 0), atol=1e-7) and abs(d[0, 7] - math.radians(0.53)) < 1e-8
    assert np.allclose(d[1, 4:7], (-1, 0, 0), atol=1e-7) and abs(d[1, 7] - math.pi / 2) < 1e-6          # clamped to 90 degrees; +90 degrees about Y turns -Z into -X
    out = pt.env_bake_lights(d, 2048)
    assert np.array_equal(out[:, :7], d[:, :7])
    assert out[0, 7] == d[0, 7] > np.float32(3.141592654) / np.float32(1024.0) and out[1, 7] == d[1, 7]            # a 0.53 degree sun is wider than three 2048-cube texels
    assert pt.env_bake_lights(d, 256)[0, 7] == np.float32(3.141592654) / np.float32(128.0)                  # a 0.53 degree sun cannot be drawn into a 256 cube: widened
    # rotated environment: the baked direction is EnvMap::ToLocal(direction) = mul(dir, (float3x3)InvTransform), InvTransform = Transform^T for a rotation
    yaw = 0.7; R = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    T = np.concatenate([R, np.zeros((3, 1))], axis=1).astype(np.float32).reshape(-1)
    rot = pt.env_bake_lights(d, 2048, transform=T)
    assert np.allclose(rot[:, 4:7], d[:, 4:7].astype(np.float64) @ np.asarray(R, np.float64).T, atol=1e-6)
    assert np.allclose(np.linalg.norm(rot[:, 4:7], axis=1), 1.0, atol=1e-6)
    with pytest.raises(pt.PtError):
        pt.env_bake_lights(d, 0)


def test_proxy_mesh_nodes_link_instances_to_their_light(tmp_path):
    """A point / spot light's "proxyMeshNodes" (ExtendedScene.cpp:46, 246-263): Donut's SceneGraph::FindNode walks '/'-separated node names from the root, the model
    hangs below its graph node as a node named after the model file with the glTF nodes below it; the mesh instance found there stands in for the light
    (LightsBaker.cpp:718-753) — PtInstanceDesc.analyticProxyLight = light index + 1, counted from the lights that survive the visibility clean-up. Paths that end at a
    node without a mesh instance (the graph node itself), unknown paths and the proxies of a dropped light link nothing."""
    graph = [
        {"name": "room", "model": 0},
        {"name": "group", "children": [{"name": "room2", "model": 0}]},
        {"name": "Lights", "children": [
            {"name": "off", "type": "PointLight", "intensity": 0.0, "radius": 0.1, "proxyMeshNodes": ["/room/cornell.gltf/instance0"]},
            {"name": "bulb", "type": "PointLight", "color": 0.5, "intensity": 20.0, "radius": 0.02, "proxyMeshNodes": ["/group/room2/cornell.gltf/instance1", "/room", "/nowhere/x"]},
            {"name": "spot", "type": "SpotLight", "intensity": 5.0, "radius": 0.05, "proxyMeshNodes": ["/room/cornell.gltf/../cornell.gltf/instance2", "room/cornell.gltf/instance0"]}]},
    ]
    media, sc, cam = make_folder(tmp_path, graph)
    imp = pt.SceneImport(media / "test.scene.json")
    I = imp.info; n = len(sc["instances"])
    assert I["numLights"] == 2 and I["lightsDropped"] == 1 and I["lightProxies"] == 6 and I["lightProxiesResolved"] == 3
    link = imp.instances["analyticProxyLight"]
    want = np.zeros(2 * n, np.uint32); want[n + 1] = 1; want[2] = 2; want[0] = 2          # bulb = light 0 -> 1, spot = light 1 -> 2 (".." and a relative path resolve too)
    assert np.array_equal(link, want), (link, want)
