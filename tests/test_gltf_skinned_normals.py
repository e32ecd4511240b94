"""glTF skins, the normal / tangent half (pt_gltf_animation_normals -> pt_animate_normals; Donut's skinning rewrites normals with the positions, Sample.cpp:1170-1198): a two-joint bar
with per-vertex normals and tangents under a non-uniformly scaled mesh node, against an independent float64 numpy evaluation (inverse-transpose of the joint matrices for the
normals, the matrices themselves for the tangents, handedness kept), in pt_set_geometry's SNORM8 packing."""
import base64, json, math, os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import rtxpt_amd as pt
from test_gltf_skins import _mat_trs, _quat_z

MESH_T, MESH_S = (3.0, 0.5, -1.0), (2.0, 1.0, 0.5)


def _write(tmp_path):
    ys = np.linspace(0.0, 2.0, 5)
    P = np.array([[x, y, 0.0] for y in ys for x in (-0.1, 0.1)], np.float32)
    N = np.array([[0.6 * (1 if i % 2 else -1), 0.0, 0.8] for i in range(10)], np.float32)
    T = np.array([[0.8, 0.0, -0.6 * (1 if i % 2 else -1), 1.0 if i < 5 else -1.0] for i in range(10)], np.float32)
    w1 = np.clip((P[:, 1] - 0.5) / 1.0, 0, 1); W = np.zeros((10, 4), np.float32); W[:, 0] = 1 - w1; W[:, 1] = w1
    J = np.zeros((10, 4), np.uint8); J[:, 1] = 1
    I = np.array([[2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 1, 2 * k + 3, 2 * k + 2] for k in range(4)], np.uint16).reshape(-1)
    ibm = np.stack([np.eye(4), np.linalg.inv(_mat_trs((0, 1, 0), (0, 0, 0, 1), (1, 1, 1)))]).astype(np.float32)
    times = np.array([0.0, 1.0], np.float32); rots = np.array([_quat_z(0.0), _quat_z(math.pi / 3)], np.float32)
    blobs = [P.tobytes(), W.tobytes(), J.tobytes(), I.tobytes(), np.ascontiguousarray(ibm.transpose(0, 2, 1)).tobytes(), times.tobytes(), rots.tobytes(), N.tobytes(), T.tobytes()]
    offs, blob = [], b""
    for b_ in blobs: blob += b"\0" * ((-len(blob)) % 4); offs.append(len(blob)); blob += b_
    views = [{"buffer": 0, "byteOffset": o, "byteLength": len(b_)} for o, b_ in zip(offs, blobs)]
    acc = [{"bufferView": 0, "componentType": 5126, "count": 10, "type": "VEC3", "min": P.min(0).tolist(), "max": P.max(0).tolist()}, {"bufferView": 1, "componentType": 5126, "count": 10, "type": "VEC4"},
           {"bufferView": 2, "componentType": 5121, "count": 10, "type": "VEC4"}, {"bufferView": 3, "componentType": 5123, "count": 24, "type": "SCALAR"},
           {"bufferView": 4, "componentType": 5126, "count": 2, "type": "MAT4"}, {"bufferView": 5, "componentType": 5126, "count": 2, "type": "SCALAR"}, {"bufferView": 6, "componentType": 5126, "count": 2, "type": "VEC4"},
           {"bufferView": 7, "componentType": 5126, "count": 10, "type": "VEC3"}, {"bufferView": 8, "componentType": 5126, "count": 10, "type": "VEC4"}]
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1, 4]}],
           "nodes": [{"name": "bar", "mesh": 0, "skin": 0, "translation": list(MESH_T), "scale": list(MESH_S)}, {"name": "root", "children": [2]}, {"name": "j0", "children": [3]},
                     {"name": "j1", "translation": [0.0, 1.0, 0.0]}, {"name": "static", "mesh": 1, "translation": [-2.0, 0.0, 0.0]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 7, "TANGENT": 8, "JOINTS_0": 2, "WEIGHTS_0": 1}, "indices": 3}]},
                      {"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 7, "TANGENT": 8}, "indices": 3}]}],
           "skins": [{"joints": [2, 3], "inverseBindMatrices": 4}],
           "animations": [{"samplers": [{"input": 5, "output": 6, "interpolation": "LINEAR"}], "channels": [{"sampler": 0, "target": {"node": 3, "path": "rotation"}}]}],
           "accessors": acc, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]}
    f = tmp_path / "bar_normals.gltf"; f.write_text(json.dumps(doc)); return f, P, N, T, W


def _snorm8(v):      # the unpack of pt_set_geometry's packing: signed bytes / 127, clamped
    b = np.stack([(v >> (8 * k)) & 0xFF for k in range(4)], -1).astype(np.int64); b = np.where(b > 127, b - 256, b)
    return np.clip(b / 127.0, -1.0, 1.0)


def _expected(N, T, W, t):
    a = min(t, 1.0) * math.pi / 3
    mesh = _mat_trs(MESH_T, (0, 0, 0, 1), MESH_S); j0 = np.eye(4); j1 = _mat_trs((0, 1, 0), _quat_z(a), (1, 1, 1))
    ibm = [np.eye(4), np.linalg.inv(_mat_trs((0, 1, 0), (0, 0, 0, 1), (1, 1, 1)))]
    jm = [np.linalg.inv(mesh) @ j0 @ ibm[0], np.linalg.inv(mesh) @ j1 @ ibm[1]]
    n = sum(W[:, k:k + 1].astype(np.float64) * (N.astype(np.float64) @ np.linalg.inv(jm[k][:3, :3])) for k in range(2))            # n^T M^-1 == (M^-T n)^T
    tg = sum(W[:, k:k + 1].astype(np.float64) * (T[:, :3].astype(np.float64) @ jm[k][:3, :3].T) for k in range(2))
    return n / np.linalg.norm(n, axis=1, keepdims=True), tg / np.linalg.norm(tg, axis=1, keepdims=True)


def test_skinned_normals_and_tangents_match_an_independent_evaluation(tmp_path):
    f, P, N, T, W = _write(tmp_path)
    a = pt.GltfAnimation(f)
    for t in (0.0, 0.4, 1.0):
        nrm, tan = a.normals(t)
        assert nrm.shape == (20,) and tan.shape == (20,)
        en, et = _expected(N, T, W, t)
        assert np.abs(_snorm8(nrm[:10])[:, :3] - en).max() <= 1.0 / 127 + 1e-6, t          # one quantisation step of the SNORM8 packing
        assert np.abs(_snorm8(tan[:10])[:, :3] - et).max() <= 1.0 / 127 + 1e-6, t
        assert np.array_equal(np.sign(_snorm8(tan[:10])[:, 3]), np.sign(T[:, 3]))          # handedness kept
        # under the mesh node's non-uniform scale the posed normals stay perpendicular to the posed tangents' plane only with the inverse transpose; the unskinned copy keeps its bind pose
        bn, bt = a.normals(0.0)
        assert np.array_equal(nrm[10:], bn[10:]) and np.array_equal(tan[10:], bt[10:])
    n1, _ = a.normals(1.0); n0, _ = a.normals(0.0)
    assert (n1[5:10] != n0[5:10]).any()                                                     # the rotated joint's vertices did turn
    # a file without skins: the bind-pose streams
    a.close()


@pytest.mark.gpu
def test_device_shades_with_the_posed_normals(tmp_path):
    """pt_animate(positions) + pt_animate_normals == a scene set up with the posed streams from the start == the oracle on that scene"""
    from rtxpt_amd import scenes
    from oracle import ptref
    sc, cam = scenes.cornell_box("C2"); S = scenes.config_settings("C2"); w, h = 96, 64
    camd = scenes.bridge_camera(w, h, **cam)
    rng = np.random.default_rng(5)
    v = rng.normal(size=(len(sc["normals"]), 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    # perturb every vertex normal a little around its packed value
    def unpack(u): b = np.stack([(u >> (8 * k)) & 0xFF for k in range(3)], -1).astype(np.int64); b = np.where(b > 127, b - 256, b); return b / 127.0
    n0 = unpack(np.asarray(sc["normals"], np.uint32)); n1 = n0 + 0.25 * v; n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    q = np.clip(np.trunc(n1 * 127.0), -127, 127).astype(np.int64) & 0xFF
    posed = (q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16)).astype(np.uint32)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(w, h)
    t.render(0, 1)
    t.animate_normals(normals=posed)
    t.render(0, 2); got = t.radiance(); t.close()
    sc2 = dict(sc); sc2["normals"] = posed
    o = ptref.Oracle(); o.set_scene(sc2); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.render(0, 2)
    assert np.array_equal(got.view(np.uint32), o.radiance().view(np.uint32))
    t2 = pt.PathTracer(); t2.set_scene(sc2); t2.set_settings(S); t2.set_camera(camd); t2.resize(w, h); t2.render(0, 2)
    assert np.array_equal(got.view(np.uint32), t2.radiance().view(np.uint32)); t2.close()


def test_morph_targets_displace_normals_and_tangents(tmp_path):
    """NORMAL / TANGENT displacements of morph targets (glTF 2.0 3.7.2.2): n = normalize(base + SUM_i w_i dn_i), the node's weights, no skin"""
    rng = np.random.default_rng(11)
    P = rng.uniform(-1, 1, (6, 3)).astype(np.float32)
    N = rng.normal(size=(6, 3)); N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
    T = rng.normal(size=(6, 3)); T = T / np.linalg.norm(T, axis=1, keepdims=True); T = np.concatenate([T, np.array([[1.0], [-1.0]] * 3)], 1).astype(np.float32)
    DP = rng.uniform(-0.2, 0.2, (6, 3)).astype(np.float32); DN = rng.uniform(-0.6, 0.6, (6, 3)).astype(np.float32); DT = rng.uniform(-0.6, 0.6, (6, 3)).astype(np.float32)
    I = np.arange(6, dtype=np.uint16)
    blobs = [P.tobytes(), N.tobytes(), T.tobytes(), DP.tobytes(), DN.tobytes(), DT.tobytes(), I.tobytes()]
    offs, blob = [], b""
    for b_ in blobs: blob += b"\0" * ((-len(blob)) % 4); offs.append(len(blob)); blob += b_
    views = [{"buffer": 0, "byteOffset": o, "byteLength": len(b_)} for o, b_ in zip(offs, blobs)]
    v3 = lambda i: {"bufferView": i, "componentType": 5126, "count": 6, "type": "VEC3"}
    acc = [dict(v3(0), min=P.min(0).tolist(), max=P.max(0).tolist()), v3(1), {"bufferView": 2, "componentType": 5126, "count": 6, "type": "VEC4"}, v3(3), v3(4), v3(5), {"bufferView": 6, "componentType": 5123, "count": 6, "type": "SCALAR"}]
    prim = {"attributes": {"POSITION": 0, "NORMAL": 1, "TANGENT": 2}, "indices": 6, "targets": [{"POSITION": 3, "NORMAL": 4, "TANGENT": 5}]}
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1]}], "nodes": [{"name": "morphed", "mesh": 0, "weights": [0.75]}, {"name": "rest", "mesh": 1}],
           "meshes": [{"primitives": [prim], "weights": [0.0]}, {"primitives": [prim], "weights": [0.0]}],
           "accessors": acc, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]}
    f = tmp_path / "morph_normals.gltf"; f.write_text(json.dumps(doc))
    a = pt.GltfAnimation(f)
    nrm, tan = a.normals(0.0)
    en = N.astype(np.float64) + 0.75 * DN; en /= np.linalg.norm(en, axis=1, keepdims=True)
    et = T[:, :3].astype(np.float64) + 0.75 * DT; et /= np.linalg.norm(et, axis=1, keepdims=True)
    assert np.abs(_snorm8(nrm[:6])[:, :3] - en).max() <= 1.0 / 127 + 1e-6 and np.abs(_snorm8(tan[:6])[:, :3] - et).max() <= 1.0 / 127 + 1e-6
    assert np.array_equal(np.sign(_snorm8(tan[:6])[:, 3]), np.sign(T[:, 3]))
    assert np.abs(_snorm8(nrm[6:])[:, :3] - N).max() <= 1.0 / 127 + 1e-6                 # weight 0: the base streams
    assert np.allclose(a.positions(0.0)[:6], P.astype(np.float64) + 0.75 * DP, atol=2e-6)
    a.close()
