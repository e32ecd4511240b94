"""glTF skins evaluated on the host into the positions pt_animate takes (pt_gltf_animation_positions; the reference's skinned meshes are Donut SkinnedMeshInstances whose vertices
are rewritten every frame, Sample.cpp:1065, 1170-1198): a two-joint bar under a translated, scaled mesh node, the upper joint animated about Z, weights blending along the bar,
JOINTS_0 as unsigned bytes and WEIGHTS_0 as normalised unsigned bytes in one primitive and floats in the other — against an independent float64 numpy evaluation of the
specification's joint matrices with the mesh node's transform taken out (the instance keeps it)."""
import base64, json, math, struct
import numpy as np
import pytest

import rtxpt_amd as pt


def _quat_z(a): return [0.0, 0.0, math.sin(a / 2), math.cos(a / 2)]


def _mat_trs(t, q, s):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4); M[:3, :3] = R * np.asarray(s, np.float64)[None, :]; M[:3, 3] = t; return M


def _write(tmp_path):
    # a bar of 5 rungs (10 vertices, 8 triangles) along +Y from 0 to 2; joint 0 at the origin, joint 1 at y = 1; weights: below y = 0.5 joint 0, above 1.5 joint 1, linear between
    ys = np.linspace(0.0, 2.0, 5)
    P = np.array([[x, y, 0.0] for y in ys for x in (-0.1, 0.1)], np.float32)
    w1 = np.clip((P[:, 1] - 0.5) / 1.0, 0, 1); W = np.zeros((10, 4), np.float32); W[:, 0] = 1 - w1; W[:, 1] = w1
    J = np.zeros((10, 4), np.uint8); J[:, 1] = 1
    I = np.array([[2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 1, 2 * k + 3, 2 * k + 2] for k in range(4)], np.uint16).reshape(-1)
    ibm = np.stack([np.linalg.inv(_mat_trs((0, 0, 0), (0, 0, 0, 1), (1, 1, 1))), np.linalg.inv(_mat_trs((0, 1, 0), (0, 0, 0, 1), (1, 1, 1)))]).astype(np.float32)      # the joints' bind-pose worlds (mesh node at identity in bind pose)
    times = np.array([0.0, 1.0, 2.0], np.float32); rots = np.array([_quat_z(0.0), _quat_z(math.pi / 2), _quat_z(math.pi / 2)], np.float32)
    Wb = np.round(W * 255).astype(np.uint8)                                       # the second primitive: the same bar with normalised-byte weights (sums stay 255 here)
    blobs = [P.tobytes(), W.tobytes(), J.tobytes(), I.tobytes(), np.ascontiguousarray(ibm.transpose(0, 2, 1)).tobytes(), times.tobytes(), rots.tobytes(), Wb.tobytes()]
    offs, blob = [], b""
    for b_ in blobs: blob += b"\\0" * ((-len(blob)) % 4); offs.append(len(blob)); blob += b_
    views = [{"buffer": 0, "byteOffset": o, "byteLength": len(b_)} for o, b_ in zip(offs, blobs)]
    acc = [{"bufferView": 0, "componentType": 5126, "count": 10, "type": "VEC3", "min": P.min(0).tolist(), "max": P.max(0).tolist()}, {"bufferView": 1, "componentType": 5126, "count": 10, "type": "VEC4"},
           {"bufferView": 2, "componentType": 5121, "count": 10, "type": "VEC4"}, {"bufferView": 3, "componentType": 5123, "count": 24, "type": "SCALAR"},
           {"bufferView": 4, "componentType": 5126, "count": 2, "type": "MAT4"}, {"bufferView": 5, "componentType": 5126, "count": 3, "type": "SCALAR"}, {"bufferView": 6, "componentType": 5126, "count": 3, "type": "VEC4"},
           {"bufferView": 7, "componentType": 5121, "normalized": True, "count": 10, "type": "VEC4"}]
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1, 4]}],
           "nodes": [{"name": "bar", "mesh": 0, "skin": 0, "translation": [3.0, 0.5, -1.0], "scale": [2.0, 2.0, 2.0]},
                     {"name": "root", "children": [2], "translation": [0.0, 0.0, 0.0]}, {"name": "j0", "children": [3]}, {"name": "j1", "translation": [0.0, 1.0, 0.0]},
                     {"name": "static", "mesh": 1, "translation": [-2.0, 0.0, 0.0]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "JOINTS_0": 2, "WEIGHTS_0": 1}, "indices": 3}, {"attributes": {"POSITION": 0, "JOINTS_0": 2, "WEIGHTS_0": 7}, "indices": 3}]},
                      {"primitives": [{"attributes": {"POSITION": 0}, "indices": 3}]}],
           "skins": [{"joints": [2, 3], "inverseBindMatrices": 4, "skeleton": 1}],
           "animations": [{"samplers": [{"input": 5, "output": 6, "interpolation": "LINEAR"}], "channels": [{"sampler": 0, "target": {"node": 3, "path": "rotation"}}]}],
           "accessors": acc, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]}
    f = tmp_path / "bar.gltf"; f.write_text(json.dumps(doc)); return f, P, W, Wb


def _expected(P, W, t):
    a = min(t, 1.0) * math.pi / 2 if t <= 1.0 else math.pi / 2
    # slerp between the two key quaternions == rotation about Z by the interpolated angle
    mesh = _mat_trs((3.0, 0.5, -1.0), (0, 0, 0, 1), (2, 2, 2)); j0 = np.eye(4); j1 = j0 @ _mat_trs((0, 1, 0), _quat_z(a), (1, 1, 1))
    ibm = [np.eye(4), np.linalg.inv(_mat_trs((0, 1, 0), (0, 0, 0, 1), (1, 1, 1)))]
    jm = [np.linalg.inv(mesh) @ j0 @ ibm[0], np.linalg.inv(mesh) @ j1 @ ibm[1]]
    Ph = np.concatenate([P.astype(np.float64), np.ones((len(P), 1))], 1)
    return sum(W[:, k:k + 1].astype(np.float64) * (Ph @ jm[k].T)[:, :3] for k in range(2))


def test_skinned_positions_match_an_independent_evaluation(tmp_path):
    f, P, W, Wb = _write(tmp_path)
    a = pt.GltfAnimation(f)
    assert a.count == 1 and abs(a.duration - 2.0) < 1e-6
    for t in (0.0, 0.5, 1.0, 1.7):
        got = a.positions(t)
        assert got.shape == (30, 3)                                              # two primitives of the bar + the static mesh's copy
        assert np.allclose(got[:10], _expected(P, W, t), atol=2e-6), t           # float weights
        assert np.allclose(got[10:20], _expected(P, Wb.astype(np.float64) / 255.0, t), atol=2e-6), t      # normalised-byte weights
        assert np.array_equal(got[20:], P)                                       # the unskinned mesh keeps its vertices
    # t = 0 is NOT the bind pose here: the mesh node's own transform is taken out of the joint matrices (the instance applies it)
    rest = a.positions(0.0)[:10]
    assert np.allclose(rest, (P.astype(np.float64) - (3.0, 0.5, -1.0)) / 2.0, atol=2e-6)
    # the instance transforms are untouched by the skin: the bar's node keeps its translation and scale
    inst = a.instances(1.0)
    assert len(inst) == 2 and np.allclose(inst[0]["transform"].reshape(3, 4)[:, 3], (3.0, 0.5, -1.0))
    a.close()


def test_files_without_skins_return_their_vertices(tmp_path):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gltf_writer import write_gltf
    from rtxpt_amd import scenes
    sc, cam = scenes.cornell_box("C2")
    write_gltf(sc, str(tmp_path / "c.gltf"))
    a = pt.GltfAnimation(tmp_path / "c.gltf")
    assert np.array_equal(a.positions(0.3), sc["positions"])
    a.close()


def test_a_file_without_scenes_poses_its_parentless_nodes_only(tmp_path):
    """No "scenes" entry: the roots are the nodes that are nobody's child. Visiting every node as a root would reach joint j1 a second time with its LOCAL transform only and
    overwrite the world matrix recorded under its parent chain (root -> j0 -> j1) — the skin would then bend about the wrong pivot — and list the mesh nodes twice."""
    f, P, W, Wb = _write(tmp_path)
    doc = json.loads(f.read_text()); del doc["scenes"]; del doc["scene"]
    doc["nodes"][2]["translation"] = [0.0, 0.25, 0.0]                      # give the joints' parent chain a transform, so that "world" and "local" differ for j1
    g = tmp_path / "noscenes.gltf"; g.write_text(json.dumps(doc))
    doc2 = json.loads(f.read_text()); doc2["nodes"][2]["translation"] = [0.0, 0.25, 0.0]
    h = tmp_path / "withscenes.gltf"; h.write_text(json.dumps(doc2))
    a, b = pt.GltfAnimation(g), pt.GltfAnimation(h)
    for t in (0.0, 0.5, 1.0):
        assert np.array_equal(a.positions(t), b.positions(t)), t
        ia, ib = a.instances(t), b.instances(t)
        assert len(ia) == len(ib) == 2 and np.array_equal(ia["transform"], ib["transform"])
    a.close(); b.close()
