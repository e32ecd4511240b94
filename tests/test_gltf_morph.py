"""glTF morph targets evaluated on the host into the positions pt_animate takes (pt_gltf_animation_positions; the reference's deformed meshes are rewritten per frame and their
BLAS updated, Sample.cpp:1065, 1170-1198): p = base + SUM_i w_i target_i before the skin, the weights from the animation's "weights" channel (LINEAR, STEP and CUBICSPLINE),
else from the node, else from the mesh — against a float64 numpy evaluation of the specification."""
import base64, json
import numpy as np

import rtxpt_amd as pt


def _write(tmp_path, mode="LINEAR"):
    rng = np.random.default_rng(5)
    P = rng.uniform(-1, 1, (9, 3)).astype(np.float32)
    D0 = rng.uniform(-0.5, 0.5, (9, 3)).astype(np.float32); D1 = rng.uniform(-0.5, 0.5, (9, 3)).astype(np.float32)
    I = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8], np.uint16)
    times = np.array([0.0, 1.0, 3.0], np.float32)
    keys = np.array([[0.0, 1.0], [1.0, 0.25], [0.5, 0.5]], np.float32)                      # two weights per key
    if mode == "CUBICSPLINE":
        tin = rng.uniform(-1, 1, (3, 2)).astype(np.float32); tout = rng.uniform(-1, 1, (3, 2)).astype(np.float32)
        out = np.stack([tin, keys, tout], 1).reshape(-1)                                   # in-tangents, values, out-tangents per key
    else: tin = tout = None; out = keys.reshape(-1)
    blobs = [P.tobytes(), D0.tobytes(), D1.tobytes(), I.tobytes(), times.tobytes(), out.astype(np.float32).tobytes()]
    offs, blob = [], b""
    for b_ in blobs: blob += b"\0" * ((-len(blob)) % 4); offs.append(len(blob)); blob += b_
    views = [{"buffer": 0, "byteOffset": o, "byteLength": len(b_)} for o, b_ in zip(offs, blobs)]
    acc = [{"bufferView": 0, "componentType": 5126, "count": 9, "type": "VEC3", "min": P.min(0).tolist(), "max": P.max(0).tolist()},
           {"bufferView": 1, "componentType": 5126, "count": 9, "type": "VEC3"}, {"bufferView": 2, "componentType": 5126, "count": 9, "type": "VEC3"},
           {"bufferView": 3, "componentType": 5123, "count": 9, "type": "SCALAR"}, {"bufferView": 4, "componentType": 5126, "count": 3, "type": "SCALAR"},
           {"bufferView": 5, "componentType": 5126, "count": int(out.size), "type": "SCALAR"}]
    prim = {"attributes": {"POSITION": 0}, "indices": 3, "targets": [{"POSITION": 1}, {"POSITION": 2}]}
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1, 2, 3]}],
           "nodes": [{"name": "animated", "mesh": 0, "translation": [1.0, 2.0, 3.0]}, {"name": "node weights", "mesh": 1, "weights": [0.25, -0.5]},
                     {"name": "mesh weights", "mesh": 2}, {"name": "plain", "mesh": 3}],
           "meshes": [{"primitives": [prim], "weights": [0.0, 0.0]}, {"primitives": [prim], "weights": [1.0, 1.0]}, {"primitives": [prim], "weights": [0.5, 2.0]}, {"primitives": [{"attributes": {"POSITION": 0}, "indices": 3}]}],
           "animations": [{"samplers": [{"input": 4, "output": 5, "interpolation": mode}], "channels": [{"sampler": 0, "target": {"node": 0, "path": "weights"}}]}],
           "accessors": acc, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]}
    f = tmp_path / ("morph_%s.gltf" % mode); f.write_text(json.dumps(doc)); return f, P, D0, D1, times, keys, tin, tout


def _weights(mode, times, keys, tin, tout, t):
    times = times.astype(np.float64); keys = keys.astype(np.float64)
    if t <= times[0]: return keys[0]
    if t >= times[-1]: return keys[-1]
    k = int(np.searchsorted(times, t, side="right") - 1); dt = times[k + 1] - times[k]; u = (t - times[k]) / dt
    if mode == "STEP": return keys[k]
    if mode == "LINEAR": return keys[k] + u * (keys[k + 1] - keys[k])
    m0 = tout[k].astype(np.float64) * dt; m1 = tin[k + 1].astype(np.float64) * dt
    return (2 * u**3 - 3 * u**2 + 1) * keys[k] + (u**3 - 2 * u**2 + u) * m0 + (-2 * u**3 + 3 * u**2) * keys[k + 1] + (u**3 - u**2) * m1


def test_morphed_positions_follow_the_weights_channel(tmp_path):
    for mode in ("LINEAR", "STEP", "CUBICSPLINE"):
        f, P, D0, D1, times, keys, tin, tout = _write(tmp_path, mode)
        a = pt.GltfAnimation(f)
        assert a.count == 1 and abs(a.duration - 3.0) < 1e-6
        P64, A, B = P.astype(np.float64), D0.astype(np.float64), D1.astype(np.float64)
        for t in (0.0, 0.4, 1.0, 2.2, 3.0, 9.0):
            w = _weights(mode, times, keys, tin, tout, t)
            got = a.positions(t)
            assert got.shape == (36, 3)
            assert np.allclose(got[:9], P64 + w[0] * A + w[1] * B, atol=2e-6), (mode, t)      # the animated node: the channel's weights
            assert np.allclose(got[9:18], P64 + 0.25 * A - 0.5 * B, atol=2e-6)                  # the node's own weights override the mesh's
            assert np.allclose(got[18:27], P64 + 0.5 * A + 2.0 * B, atol=2e-6)                  # the mesh's default weights
            assert np.array_equal(got[27:], P)                                                  # no targets: the vertices as they are
        inst = a.instances(1.0)
        assert len(inst) == 4 and np.allclose(inst[0]["transform"].reshape(3, 4)[:, 3], (1.0, 2.0, 3.0))      # the weights channel leaves the node transforms alone
        a.close()


def test_a_target_count_mismatch_is_rejected_and_a_wrong_channel_length_is_ignored(tmp_path):
    f, P, D0, D1, times, keys, tin, tout = _write(tmp_path, "LINEAR")
    doc = json.loads(f.read_text())
    doc["accessors"][1]["count"] = 8                                                           # a target with fewer displacements than vertices
    g = tmp_path / "bad_target.gltf"; g.write_text(json.dumps(doc))
    try: pt.GltfAnimation(g); ok = True
    except Exception: ok = False
    assert not ok
    doc = json.loads(f.read_text())
    doc["meshes"][0]["primitives"][0]["targets"].append({"POSITION": 1})                        # three targets, the channel still carries two weights per key
    h = tmp_path / "short_channel.gltf"; h.write_text(json.dumps(doc))
    a = pt.GltfAnimation(h)
    got = a.positions(0.5)
    assert np.allclose(got[:9], P, atol=2e-6)                                                   # the channel does not fit: the mesh's default weights (zeros) apply
    a.close()
