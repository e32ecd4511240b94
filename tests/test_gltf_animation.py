"""pt_gltf_animation_* (SURVEY.md N2 leftovers: glTF animations -> the instance transforms pt_animate takes), CPU only: a hand-built .gltf with a three-level node
hierarchy whose nodes are driven by LINEAR (translation, spherical rotation), STEP (scale) and CUBICSPLINE (translation) samplers, evaluated by the library and by
an independent numpy implementation of the glTF 2.0 interpolation rules; clamping outside the key range, a second animation, a matrix node, damaged files."""
import json, os, struct
import numpy as np
import pytest

import rtxpt_amd as pt


def _quat(axis, ang):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a); return np.concatenate([a * np.sin(ang / 2), [np.cos(ang / 2)]])


def _mat(t, q, s):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4); M[:3, :3] = R * np.asarray(s)[None, :]; M[:3, 3] = t; return M


KEYS = np.array([0.0, 0.5, 1.25, 2.0], np.float32)
ROT = np.stack([_quat((0, 1, 0), 0.0), _quat((0, 1, 0), 1.2), -_quat((1, 1, 0), 2.0), _quat((0, 0, 1), 3.0)]).astype(np.float32)      # (one key negated: slerp must take the short arc)
TRA = np.array([[0, 0, 0], [1, 2, 0], [1, 2, 3], [-2, 0, 1]], np.float32)
SCL = np.array([[1, 1, 1], [2, 1, 1], [1, 0.5, 3], [1, 1, 1]], np.float32)
rng = np.random.default_rng(4)
CUB = rng.normal(size=(4, 3, 3)).astype(np.float32)          # per key: in-tangent, value, out-tangent
KEYS2 = np.array([0.25, 1.0], np.float32); TRA2 = np.array([[5, 0, 0], [5, 4, 0]], np.float32)


def _write(tmp_path):
    blob = bytearray(); views = []; accs = []
    def add(arr, typ, ctype=5126):
        nonlocal blob
        while len(blob) % 4: blob += b"\0"
        data = np.ascontiguousarray(arr).tobytes(); views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}); blob += data
        a = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(np.asarray(arr).shape[0]), "type": typ}
        if typ == "VEC3": a["min"] = [float(v) for v in np.asarray(arr).reshape(-1, 3).min(0)]; a["max"] = [float(v) for v in np.asarray(arr).reshape(-1, 3).max(0)]
        accs.append(a); return len(accs) - 1
    pos = add(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), "VEC3"); idx = add(np.array([0, 1, 2], np.uint32), "SCALAR", 5125)
    tk = add(KEYS, "SCALAR"); ar = add(ROT, "VEC4"); at = add(TRA, "VEC3"); asc = add(SCL, "VEC3"); ac = add(CUB.reshape(12, 3), "VEC3"); tk2 = add(KEYS2, "SCALAR"); at2 = add(TRA2, "VEC3")
    M = _mat((3, 0, 1), _quat((1, 0, 0), 0.4), (1, 2, 1))
    nodes = [{"name": "root", "children": [1, 2], "translation": [0.5, 0, 0]},                                         # 0: rotation animated (LINEAR)
             {"name": "a", "mesh": 0, "children": [3], "rotation": [float(v) for v in _quat((0, 0, 1), 0.3)], "scale": [1, 1, 2]},      # 1: translation animated (CUBICSPLINE), rotation / scale static
             {"name": "b", "mesh": 0, "matrix": [float(v) for v in M.T.reshape(-1)]},                                    # 2: static matrix node
             {"name": "c", "mesh": 0, "translation": [0, 1, 0]}]                                                          # 3: scale animated (STEP); in animation 1: translation (LINEAR)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes,
           "meshes": [{"primitives": [{"attributes": {"POSITION": pos}, "indices": idx}]}], "buffers": [{"uri": "a.bin", "byteLength": len(blob)}], "bufferViews": views, "accessors": accs,
           "animations": [{"samplers": [{"input": tk, "output": ar}, {"input": tk, "output": ac, "interpolation": "CUBICSPLINE"}, {"input": tk, "output": asc, "interpolation": "STEP"}],
                           "channels": [{"sampler": 0, "target": {"node": 0, "path": "rotation"}}, {"sampler": 1, "target": {"node": 1, "path": "translation"}},
                                        {"sampler": 2, "target": {"node": 3, "path": "scale"}}, {"sampler": 0, "target": {"path": "rotation"}}]},      # (a channel without a node: ignored)
                          {"samplers": [{"input": tk2, "output": at2, "interpolation": "LINEAR"}], "channels": [{"sampler": 0, "target": {"node": 3, "path": "translation"}}]}]}
    (tmp_path / "a.bin").write_bytes(bytes(blob)); (tmp_path / "a.gltf").write_text(json.dumps(doc))
    return tmp_path / "a.gltf", M


def _slerp(a, b, u):
    a, b = a.astype(np.float64), b.astype(np.float64); d = float(a @ b)
    if d < 0: b, d = -b, -d
    if d > 0.9995: q = a + u * (b - a)
    else: th = np.arccos(d); q = (np.sin((1 - u) * th) * a + np.sin(u * th) * b) / np.sin(th)
    return q / np.linalg.norm(q)


def _segment(keys, t):
    if t <= keys[0]: return 0, 0, 0.0
    if t >= keys[-1]: return len(keys) - 1, len(keys) - 1, 0.0
    k = int(np.searchsorted(keys, t, side="right") - 1); return k, k + 1, float((t - keys[k]) / (keys[k + 1] - keys[k]))


def _want(t, M, anim=0):
    t = float(np.float32(t))
    rot0 = np.array([0, 0, 0, 1.0]); tr1 = np.zeros(3); sc3 = np.ones(3); tr3 = np.array([0, 1.0, 0])
    if anim == 0:
        k0, k1, u = _segment(KEYS, t)
        rot0 = _slerp(ROT[k0], ROT[k1], u) if k0 != k1 else ROT[k0].astype(np.float64) / np.linalg.norm(ROT[k0].astype(np.float64))
        if k0 == k1: tr1 = CUB[k0, 1].astype(np.float64)
        else:
            dt = float(KEYS[k1]) - float(KEYS[k0]); p0, m0, p1, m1 = CUB[k0, 1].astype(np.float64), CUB[k0, 2].astype(np.float64) * dt, CUB[k1, 1].astype(np.float64), CUB[k1, 0].astype(np.float64) * dt
            tr1 = (2 * u ** 3 - 3 * u ** 2 + 1) * p0 + (u ** 3 - 2 * u ** 2 + u) * m0 + (-2 * u ** 3 + 3 * u ** 2) * p1 + (u ** 3 - u ** 2) * m1
        sc3 = SCL[k0].astype(np.float64)
    elif anim == 1:
        k0, k1, u = _segment(KEYS2, t); tr3 = TRA2[k0].astype(np.float64) * (1 - u) + TRA2[k1].astype(np.float64) * u
    root = _mat((0.5, 0, 0), rot0, (1, 1, 1)); a = root @ _mat(tr1, _quat((0, 0, 1), 0.3), (1, 1, 2)); b = root @ M; c = a @ _mat(tr3, (0, 0, 0, 1), sc3)
    return np.stack([a[:3].reshape(-1), c[:3].reshape(-1), b[:3].reshape(-1)])          # the importer's order: depth first (a, its child c, then b)


def test_animation_matches_a_numpy_evaluation_of_the_gltf_rules(tmp_path):
    path, M = _write(tmp_path)
    an = pt.GltfAnimation(path)
    assert an.count == 2 and abs(an.duration - 2.0) < 1e-6
    for t in (-1.0, 0.0, 0.1, 0.5, 0.77, 1.25, 1.9, 2.0, 7.0):
        got = an.instances(t)
        assert got.shape == (3,) and np.all(got["meshIndex"] == 0)
        assert np.allclose(got["transform"], _want(t, M), rtol=0, atol=2e-6), t
    for t in (0.0, 0.6, 1.0, 3.0):
        assert np.allclose(an.instances(t, animation=1)["transform"], _want(t, M, anim=1), rtol=0, atol=2e-6), t
    assert np.allclose(an.instances(0.3, animation=9)["transform"], _want(0.3, M, anim=None), rtol=0, atol=2e-6)      # no such animation: the file's rest pose
    an.close()


def test_animation_of_a_file_without_animations_and_of_damaged_files(tmp_path):
    path, M = _write(tmp_path)
    doc = json.loads(path.read_text()); del doc["animations"]; (tmp_path / "static.gltf").write_text(json.dumps(doc))
    an = pt.GltfAnimation(tmp_path / "static.gltf")
    assert an.count == 0 and an.duration == 0.0 and an.instances(1.0).shape == (3,)
    doc = json.loads(path.read_text()); doc["animations"][0]["channels"][0]["sampler"] = 7; (tmp_path / "bad1.gltf").write_text(json.dumps(doc))
    with pytest.raises(pt.PtError): pt.GltfAnimation(tmp_path / "bad1.gltf")
    doc = json.loads(path.read_text()); doc["animations"][0]["samplers"][1]["interpolation"] = "LINEAR"; (tmp_path / "bad2.gltf").write_text(json.dumps(doc))     # 12 outputs for 4 keys without CUBICSPLINE
    with pytest.raises(pt.PtError): pt.GltfAnimation(tmp_path / "bad2.gltf")
    doc = json.loads(path.read_text()); doc["accessors"][2]["count"] = 4000; (tmp_path / "bad3.gltf").write_text(json.dumps(doc))
    with pytest.raises(pt.PtError): pt.GltfAnimation(tmp_path / "bad3.gltf")
    with pytest.raises(pt.PtError): pt.GltfAnimation(tmp_path / "missing.gltf")


def test_damaged_animation_documents_never_crash(tmp_path):
    """Random damage to the animation part of the document (indices, counts, types, paths, truncated buffers): an error code or a valid answer, never a crash."""
    path, M = _write(tmp_path); good = json.loads(path.read_text()); rng = np.random.default_rng(12); blob = (tmp_path / "a.bin").read_bytes()
    def mutate(v, depth=0):
        if isinstance(v, dict):
            k = list(v)[int(rng.integers(0, len(v)))] if v else None
            if k is None: return v
            if rng.random() < 0.3: v.pop(k)
            else: v[k] = mutate(v[k], depth + 1)
            return v
        if isinstance(v, list):
            if not v: return v
            i = int(rng.integers(0, len(v)))
            if rng.random() < 0.2: v.pop(i)
            else: v[i] = mutate(v[i], depth + 1)
            return v
        r = rng.random()
        return [-1, 0, 7, 10 ** 9, 2.5, "x", None, [], {}, True][int(rng.integers(0, 10))] if r < 0.8 else v
    ok = 0
    for k in range(300):
        doc = json.loads(json.dumps(good))
        for _ in range(int(rng.integers(1, 4))):
            key = ["animations", "accessors", "bufferViews", "nodes"][int(rng.integers(0, 4))]; doc[key] = mutate(doc[key])
        (tmp_path / "m.gltf").write_text(json.dumps(doc))
        if k % 10 == 0: (tmp_path / "a.bin").write_bytes(blob[: int(rng.integers(0, len(blob)))])
        else: (tmp_path / "a.bin").write_bytes(blob)
        try:
            an = pt.GltfAnimation(tmp_path / "m.gltf")
            for a in range(min(an.count, 3)):
                try: an.instances(float(rng.uniform(-1, 3)), animation=a)
                except pt.PtError: pass
            an.close(); ok += 1
        except pt.PtError:
            pass
    assert ok > 20          # (many mutations leave a loadable file)
