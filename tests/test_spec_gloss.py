"""Specular-glossiness materials (PTMaterialFlags_UseSpecularGlossModel; EvaluateSceneMaterialRTXPT, PathTracerBridgeDonut.hlsli:318-333): what Bistro ships.
The reference hands the two colours to Donut's ConvertSpecularGlossToMetalRough, which is outside its tree; product and oracle restate the Khronos conversion
it follows. Here: the oracle's loadSurface against an independent float64 evaluation of that published conversion, and the glTF extension import."""
import json, os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rtxpt_amd as pt
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes
from gltf_writer import write_gltf


def _khronos(diffuse, specular):
    """three.pbrUtilities.js of the KHR_materials_pbrSpecularGlossiness "convert-between-workflows" example, in float64."""
    eps, ds = 1e-6, 0.04
    br = lambda c: np.sqrt(0.299 * c[0] ** 2 + 0.587 * c[1] ** 2 + 0.114 * c[2] ** 2)
    one_minus = 1.0 - specular.max()
    d, s = br(diffuse), br(specular)
    if s < ds: m = 0.0
    else:
        a, b, c = ds, d * one_minus / (1 - ds) + s - 2 * ds, ds - s
        m = float(np.clip((-b + np.sqrt(max(b * b - 4 * a * c, 0.0))) / (2 * a), 0, 1))
    from_d = diffuse * (one_minus / (1 - ds) / max(1 - m, eps)); from_s = (specular - ds * (1 - m)) / max(m, eps)
    return np.clip(from_d + (from_s - from_d) * (m * m), 0, 1), m


@pytest.mark.parametrize("name", ["c2_spec_gloss", "bistro_like_spec_gloss"])
def test_reconstruction_follows_the_published_conversion(name):
    make, S, w, h, first, n = pin_scenes.cases()[name]
    sc, cam = make(); m = sc["materials"]
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(S); o.resize(8, 8)
    rng = np.random.default_rng(5); k = 4000
    prims = rng.integers(0, o.num_tris(), k).astype(np.uint32); u = rng.uniform(0, 1, k); v = rng.uniform(0, 1, k) * (1 - u)
    d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    P = o.surface_probe(prims, np.column_stack([u, v, d, np.full(k, 0.01), np.full(k, 0.001)]).astype(np.float32))
    mat = P[:, 23]; f = P.view(np.float32); seen = set(); metals = 0
    for i in range(k):
        mm = m[mat[i]]
        if not (mm["Flags"] & 1) or (mm["Flags"] & 0xC): continue          # spec-gloss, untextured (textured ones: the pinned call site multiplies the texels in)
        base, met = _khronos(mm["BaseOrDiffuseColor"].astype(np.float64), mm["SpecularColor"].astype(np.float64))
        assert np.allclose(f[i, 37:40], base, rtol=0, atol=3e-6) and abs(f[i, 36] - met) < 3e-6                      # StandardBSDFData.transmission carries the base colour
        assert np.allclose(f[i, 29:32], base * (1 - met), rtol=0, atol=3e-6)                                       # diffuse = lerp(baseColor, 0, metalness)
        F0 = ((float(mm["IoR"]) - 1) / (float(mm["IoR"]) + 1)) ** 2
        assert np.allclose(f[i, 33:36], F0 + (base - F0) * met, rtol=0, atol=3e-6)
        assert f[i, 32] == np.float32(1.0) - np.float32(1.0) * (np.float32(1.0) - mm["Roughness"])                  # roughness = 1 - glossTexel * (1 - Roughness), texel 1
        seen.add(int(mat[i])); metals += met > 0.5
    assert len(seen) >= (3 if name.startswith("c2") else 10) and metals > 0


def test_dielectric_specular_keeps_the_diffuse_colour():
    """specular = 0.04 grey is the metal-rough model's own dielectric F0: metalness 0 and base colour = diffuse colour (to rounding)."""
    base, met = _khronos(np.array([0.3, 0.5, 0.7]), np.array([0.04, 0.04, 0.04]))
    assert met < 1e-6 and np.allclose(base, [0.3, 0.5, 0.7], atol=1e-6)


def test_gltf_extension_import(tmp_path):
    """KHR_materials_pbrSpecularGlossiness as Donut's importer reads it (ImportFromDonut, MaterialsBaker.cpp:661-705): flag, colours, opacity, roughness = 1 - glossiness;
    the extension wins over pbrMetallicRoughness; defaults (white diffuse, white specular, glossiness 1)."""
    sc, cam = scenes.cornell_box("C2")
    write_gltf(sc, str(tmp_path / "c.gltf"))
    doc = json.loads((tmp_path / "c.gltf").read_text())
    doc["materials"][0].setdefault("extensions", {})["KHR_materials_pbrSpecularGlossiness"] = {"diffuseFactor": [0.2, 0.3, 0.4, 0.75], "specularFactor": [0.9, 0.6, 0.1], "glossinessFactor": 0.8}
    doc["materials"][1].setdefault("extensions", {})["KHR_materials_pbrSpecularGlossiness"] = {}
    doc["extensionsUsed"].append("KHR_materials_pbrSpecularGlossiness")
    (tmp_path / "c.gltf").write_text(json.dumps(doc))
    (tmp_path / "c.scene.json").write_text(json.dumps({"models": ["c.gltf"], "graph": [{"model": 0}]}))
    m = pt.SceneImport(tmp_path / "c.scene.json").materials
    assert m["Flags"][0] & 1 and m["Flags"][1] & 1 and not (m["Flags"][2] & 1)
    assert np.array_equal(m["BaseOrDiffuseColor"][0], np.array([0.2, 0.3, 0.4], np.float32)) and m["Opacity"][0] == np.float32(0.75)
    assert np.array_equal(m["SpecularColor"][0], np.array([0.9, 0.6, 0.1], np.float32)) and m["Roughness"][0] == np.float32(1.0) - np.float32(0.8)
    assert np.array_equal(m["BaseOrDiffuseColor"][1], np.ones(3, np.float32)) and np.array_equal(m["SpecularColor"][1], np.ones(3, np.float32)) and m["Roughness"][1] == 0 and m["Opacity"][1] == 1
    assert not (m["Flags"][0] & 0xC)                      # no textures referenced
