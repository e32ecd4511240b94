"""Binary glTF (.glb) through the importer (CPU only): the same scene as .gltf + .bin and as one .glb — JSON chunk, BIN chunk, buffer 0 without uri — gives identical
geometry / instances / materials; an image stored in the BIN chunk (bufferView, image/jpeg) is decoded; truncated containers fail cleanly."""
import io, json, os, struct, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rtxpt_amd as pt
from rtxpt_amd import scenes
from gltf_writer import write_gltf


def _pack_glb(doc, blob):
    j = json.dumps(doc).encode(); j += b" " * (-len(j) % 4); blob = blob + b"\0" * (-len(blob) % 4)
    body = struct.pack("<II", len(j), 0x4E4F534A) + j + struct.pack("<II", len(blob), 0x004E4942) + blob
    return b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body


def test_glb_equals_gltf_and_embedded_jpeg(tmp_path):
    sc, cam = scenes.cornell_box("C2")
    write_gltf(sc, str(tmp_path / "c.gltf"))
    doc = json.loads((tmp_path / "c.gltf").read_text()); blob = (tmp_path / doc["buffers"][0]["uri"]).read_bytes()
    (tmp_path / "a.scene.json").write_text(json.dumps({"models": ["c.gltf"], "graph": [{"model": 0}]}))
    ref = pt.SceneImport(tmp_path / "a.scene.json")
    # the image goes behind the geometry in the BIN chunk
    PIL = pytest.importorskip("PIL.Image")
    y, x = np.mgrid[0:24, 0:40]; img = np.stack([x * 6, y * 10, (x + y) * 3], -1).astype(np.uint8)
    buf = io.BytesIO(); PIL.fromarray(img, "RGB").save(buf, "JPEG", quality=90); jpg = buf.getvalue()
    pad = -len(blob) % 4; blob2 = blob + b"\0" * pad + jpg
    d2 = json.loads(json.dumps(doc)); del d2["buffers"][0]["uri"]; d2["buffers"][0]["byteLength"] = len(blob2)
    d2["bufferViews"].append({"buffer": 0, "byteOffset": len(blob) + pad, "byteLength": len(jpg)})
    d2["images"] = [{"bufferView": len(d2["bufferViews"]) - 1, "mimeType": "image/jpeg"}]; d2["textures"] = [{"source": 0}]
    d2["materials"][0].setdefault("pbrMetallicRoughness", {})["baseColorTexture"] = {"index": 0}
    (tmp_path / "c.glb").write_bytes(_pack_glb(d2, blob2))
    (tmp_path / "b.scene.json").write_text(json.dumps({"models": ["c.glb"], "graph": [{"model": 0}]}))
    got = pt.SceneImport(tmp_path / "b.scene.json")
    assert np.array_equal(got.geometries, ref.geometries) and np.array_equal(got.instances, ref.instances) and got.info["numMeshes"] == ref.info["numMeshes"]
    m = got.materials.copy(); word = m["BaseOrDiffuseTextureIndex"][0]; m["BaseOrDiffuseTextureIndex"][0] = ref.materials["BaseOrDiffuseTextureIndex"][0]; m["Flags"][0] = ref.materials["Flags"][0]
    assert m.tobytes() == ref.materials.tobytes() and word == scenes.pack_texture_word(0, 40, 24) and got.materials["Flags"][0] & 0x8
    px, fmt = got.texture(0)
    assert np.array_equal(px[..., :3], np.asarray(PIL.open(io.BytesIO(jpg)).convert("RGB"))) and fmt == pt.PT_TEX_RGBA8_SRGB
    # damaged containers
    good = (tmp_path / "c.glb").read_bytes()
    for cut in (0, 8, 19, 40, len(good) // 2):
        (tmp_path / "t.glb").write_bytes(good[:cut]); (tmp_path / "t.scene.json").write_text(json.dumps({"models": ["t.glb"], "graph": [{"model": 0}]}))
        with pytest.raises(pt.PtError): pt.SceneImport(tmp_path / "t.scene.json")
