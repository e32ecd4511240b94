"""`.material.json` import (the part the reference tree defines completely): pt_material_from_json against an independent
Python restatement of PTMaterial defaults (MaterialsBaker.h:126-193), Read (MaterialsBaker.cpp:150-259) and FillData (:516-591). CPU only."""
import json
import os

import numpy as np
import pytest

import rtxpt_amd as pt

F_SPECGLOSS, F_MR_TEX, F_BASE_TEX, F_EMISSIVE_TEX, F_NORMAL_TEX, F_TRANS_TEX = 0x1, 0x4, 0x8, 0x10, 0x20, 0x80
F_METAL_RED, F_THIN, F_PSD_EXCLUDE, F_PROXY, F_IGNORE_TANGENT, F_MV_B0, F_MV_B1 = 0x100, 0x200, 0x400, 0x800, 1 << 12, 1 << 13, 1 << 14
DEFAULTS = dict(BaseOrDiffuseColor=[1, 1, 1], SpecularColor=[0, 0, 0], EmissiveColor=[0, 0, 0], EmissiveIntensity=1.0, Metalness=0.0, Roughness=0.0, Opacity=1.0,
                TransmissionFactor=0.0, DiffuseTransmissionFactor=0.0, NormalTextureScale=1.0, IoR=1.5, UseSpecularGlossModel=False, EnableBaseTexture=True,
                EnableOcclusionRoughnessMetallicTexture=True, EnableNormalTexture=True, EnableEmissiveTexture=True, EnableTransmissionTexture=True,
                EnableAlphaTesting=False, AlphaCutoff=0.5, EnableTransmission=False, MetalnessInRedChannel=False, ThinSurface=False, ExcludeFromNEE=False,
                PSDExclude=True, PSDDominantDeltaLobe=-1, PSDBlockMotionVectorsAtSurfaceType=0, NestedPriority=14, VolumeAttenuationDistance=3.402823466e+38,
                VolumeAttenuationColor=[1, 1, 1], ShadowNoLFadeout=0.0, EnableAsAnalyticLightProxy=False, IgnoreMeshTangentSpace=False,
                UseDonutEmissiveIntensity=False, SkipRender=False)
TEX = ["BaseTexture", "OcclusionRoughnessMetallicTexture", "NormalTexture", "EmissiveTexture", "TransmissionTexture"]


def fill_data(doc, words):
    m = dict(DEFAULTS); m.update({k: v for k, v in doc.items() if k in DEFAULTS})
    loaded = [bool(doc.get(t, {}).get("path")) and words[i] != 0xFFFFFFFF for i, t in enumerate(TEX)]
    f = 0
    f |= F_SPECGLOSS if m["UseSpecularGlossModel"] else 0
    f |= F_BASE_TEX if loaded[0] and m["EnableBaseTexture"] else 0
    f |= F_MR_TEX if loaded[1] and m["EnableOcclusionRoughnessMetallicTexture"] else 0
    f |= F_EMISSIVE_TEX if loaded[3] and m["EnableEmissiveTexture"] else 0
    f |= F_NORMAL_TEX if loaded[2] and m["EnableNormalTexture"] else 0
    f |= F_TRANS_TEX if loaded[4] and m["EnableTransmissionTexture"] and m["EnableTransmission"] else 0
    f |= F_METAL_RED if m["MetalnessInRedChannel"] else 0
    f |= F_THIN if (m["ThinSurface"] or not m["EnableTransmission"]) else 0
    f |= F_PSD_EXCLUDE if m["PSDExclude"] else 0
    f |= F_MV_B0 if m["PSDBlockMotionVectorsAtSurfaceType"] % 2 else 0
    f |= F_MV_B1 if m["PSDBlockMotionVectorsAtSurfaceType"] // 2 else 0
    f |= F_PROXY if m["EnableAsAnalyticLightProxy"] else 0
    f |= F_IGNORE_TANGENT if m["IgnoreMeshTangentSpace"] else 0
    f |= min(max(m["NestedPriority"], 0), 14) << 28
    f |= min(max(m["PSDDominantDeltaLobe"] + 1, 0), 7) << 24
    w = lambda i, bit: words[i] if (f & bit) and loaded[i] else 0xFFFFFFFF
    et = m["EnableTransmission"]
    return dict(Flags=f, BaseOrDiffuseColor=m["BaseOrDiffuseColor"], SpecularColor=m["SpecularColor"],
                EmissiveColor=[np.float32(c) * np.float32(m["EmissiveIntensity"]) for c in m["EmissiveColor"]],
                Roughness=m["Roughness"], Metalness=m["Metalness"], NormalTextureScale=m["NormalTextureScale"], TransmissionFactor=m["TransmissionFactor"] if et else 0.0,
                DiffuseTransmissionFactor=m["DiffuseTransmissionFactor"] if et else 0.0, Opacity=m["Opacity"], AlphaCutoff=m["AlphaCutoff"], IoR=m["IoR"],
                AttenuationColor=m["VolumeAttenuationColor"], AttenuationDistance=m["VolumeAttenuationDistance"], ShadowNoLFadeout=min(max(m["ShadowNoLFadeout"], 0.0), 0.25),
                BaseOrDiffuseTextureIndex=w(0, F_BASE_TEX), MetalRoughOrSpecularTextureIndex=w(1, F_MR_TEX), EmissiveTextureIndex=w(3, F_EMISSIVE_TEX),
                NormalTextureIndex=w(2, F_NORMAL_TEX), TransmissionTextureIndex=w(4, F_TRANS_TEX), _padding0=42, _padding1=42.0)


CASES = {
    "defaults": ({}, [0xFFFFFFFF] * 5),
    "glass_in_liquid": ({"version": 1, "EnableTransmission": True, "TransmissionFactor": 0.95, "IoR": 1.33, "ThinSurface": False, "NestedPriority": 3, "Roughness": 0.02,
                         "VolumeAttenuationDistance": 0.25, "VolumeAttenuationColor": [0.9, 0.4, 0.1], "PSDDominantDeltaLobe": 0, "PSDExclude": False}, [0xFFFFFFFF] * 5),
    "textured_lamp": ({"BaseTexture": {"sRGB": True, "NormalMap": False, "path": "textures/lamp_base.png"}, "NormalTexture": {"sRGB": False, "NormalMap": True, "path": "textures/lamp_n.png"},
                       "EmissiveTexture": {"sRGB": True, "path": "textures/lamp_e.png"}, "OcclusionRoughnessMetallicTexture": {"path": "textures/lamp_orm.png"},
                       "EmissiveColor": [1.0, 0.8, 0.5], "EmissiveIntensity": 250.0, "Metalness": 1.0, "Roughness": 0.35, "MetalnessInRedChannel": True, "EnableNormalTexture": False,
                       "ExcludeFromNEE": True, "ShadowNoLFadeout": 0.6, "PSDBlockMotionVectorsAtSurfaceType": 3, "IgnoreMeshTangentSpace": True},
                      [0x14090003, 0x14090004, 0x14090005, 0xFFFFFFFF, 0xFFFFFFFF]),
    "foliage": ({"BaseTexture": {"sRGB": True, "path": "leaf.png"}, "EnableAlphaTesting": True, "AlphaCutoff": 0.33, "DiffuseTransmissionFactor": 0.4, "EnableTransmission": False,
                 "TransmissionTexture": {"path": "leaf_t.png"}, "NestedPriority": 99, "PSDDominantDeltaLobe": 12, "SkipRender": True, "UseDonutEmissiveIntensity": True,
                 "EnableAsAnalyticLightProxy": True, "UseSpecularGlossModel": True, "SpecularColor": [0.04, 0.05, 0.06]}, [0x0A050007, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0x0A050009]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_material_json_matches_fill_data(name):
    doc, words = CASES[name]
    got, info = pt.material_from_json(json.dumps(doc), words)
    want = fill_data(doc, words)
    for k, v in want.items():
        g = got[k]
        if isinstance(v, (list, tuple)):
            assert np.array_equal(np.asarray(g, np.float32), np.asarray(v, np.float32)), k
        elif isinstance(v, float):
            assert np.float32(g) == np.float32(v), k
        else:
            assert int(g) == int(v), (k, hex(int(g)), hex(int(v)))
    assert info["enableAlphaTesting"] == bool(doc.get("EnableAlphaTesting", False)) and info["excludeFromNEE"] == bool(doc.get("ExcludeFromNEE", False))
    assert info["skipRender"] == bool(doc.get("SkipRender", False)) and info["useDonutEmissiveIntensity"] == bool(doc.get("UseDonutEmissiveIntensity", False))
    for i, t in enumerate(TEX):
        assert info["texturePath"][i] == doc.get(t, {}).get("path", "")
        assert bool(info["textureSRGB"][i]) == bool(doc.get(t, {}).get("sRGB", False)) and bool(info["textureNormalMap"][i]) == bool(doc.get(t, {}).get("NormalMap", False))


def test_material_json_rejects_garbage():
    with pytest.raises(pt.PtError):
        pt.material_from_json("{ not json", [0xFFFFFFFF] * 5)
    with pytest.raises(pt.PtError):
        pt.material_from_json("[1, 2, 3]", [0xFFFFFFFF] * 5)


def _random_document(rng):
    """A `.material.json` document with a random subset of fields, including out-of-range and wrongly typed ones."""
    doc = {}
    num = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    fields = {"BaseOrDiffuseColor": lambda: [num(0, 1) for _ in range(3)], "SpecularColor": lambda: [num(0, 1) for _ in range(3)], "EmissiveColor": lambda: [num(0, 4) for _ in range(3)],
              "EmissiveIntensity": lambda: num(0, 500), "Metalness": lambda: num(0, 1), "Roughness": lambda: num(0, 1), "Opacity": lambda: num(0, 1), "TransmissionFactor": lambda: num(0, 1),
              "DiffuseTransmissionFactor": lambda: num(0, 1), "NormalTextureScale": lambda: num(0.2, 2), "IoR": lambda: num(1, 2.5), "AlphaCutoff": lambda: num(0, 1),
              "VolumeAttenuationDistance": lambda: num(0.01, 10), "VolumeAttenuationColor": lambda: [num(0, 1) for _ in range(3)], "ShadowNoLFadeout": lambda: num(-0.1, 0.6),
              "PSDDominantDeltaLobe": lambda: int(rng.integers(-3, 12)), "PSDBlockMotionVectorsAtSurfaceType": lambda: int(rng.integers(0, 4)), "NestedPriority": lambda: int(rng.integers(0, 40))}
    for b in ("UseSpecularGlossModel", "EnableBaseTexture", "EnableOcclusionRoughnessMetallicTexture", "EnableNormalTexture", "EnableEmissiveTexture", "EnableTransmissionTexture", "EnableAlphaTesting",
              "EnableTransmission", "MetalnessInRedChannel", "ThinSurface", "ExcludeFromNEE", "PSDExclude", "EnableAsAnalyticLightProxy", "IgnoreMeshTangentSpace", "UseDonutEmissiveIntensity", "SkipRender"):
        fields[b] = lambda: bool(rng.integers(0, 2))
    for k, f in fields.items():
        if rng.random() < 0.55: doc[k] = f()
    if rng.random() < 0.3: doc["Roughness"] = "rough"                # wrong kind: the field keeps its default
    if rng.random() < 0.2: doc["BaseOrDiffuseColor"] = [0.5, 0.5]      # wrong length
    words, textures = [0xFFFFFFFF] * 5, {}
    for i, t in enumerate(TEX):
        if rng.random() < 0.5:
            name = "tex_%d_%d.png" % (i, int(rng.integers(0, 100)))
            doc[t] = {"path": "textures/" + name, "sRGB": bool(rng.integers(0, 2)), "NormalMap": bool(rng.integers(0, 2))}
            if rng.random() < 0.8:        # the loader found it
                words[i] = (int(rng.integers(4, 25)) << 24) | (int(rng.integers(1, 13)) << 16) | int(rng.integers(0, 4000)); textures[name] = words[i]
    return doc, words, textures


@pytest.mark.parametrize("seed", range(40))
def test_material_json_matches_reference_text(seed):
    """pt_material_from_json against the reference's own PTMaterial::Read + FillData + GetBindlessTextureIndex (Rtxpt/Materials/MaterialsBaker.{h,cpp} compiled as
    they stand over Donut / jsoncpp stand-ins, oracle/refpin/mat_stubs.h): all 128 bytes of PTMaterialData and the fields outside it."""
    from oracle import ptref
    rng = np.random.default_rng(0x3A7 + seed)
    doc, words, textures = (CASES[sorted(CASES)[seed]][0], CASES[sorted(CASES)[seed]][1], None) if seed < len(CASES) else _random_document(rng)
    if textures is None:
        textures = {os.path.basename(doc[t]["path"]): words[i] for i, t in enumerate(TEX) if t in doc and words[i] != 0xFFFFFFFF}
    ref = ptref.reference_material_from_json(json.dumps(doc), textures)
    if ref is None:
        pytest.skip("librefpin_mat.so not available (no /root/reference on this machine)")
    got, info = pt.material_from_json(json.dumps(doc), words)
    assert got.tobytes() == ref[0], [(n, got[n], np.frombuffer(ref[0], got.dtype)[0][n]) for n in got.dtype.names if np.asarray(got[n]).tobytes() != np.asarray(np.frombuffer(ref[0], got.dtype)[0][n]).tobytes()]
    assert [int(info["enableAlphaTesting"]), int(info["excludeFromNEE"]), int(info["skipRender"]), int(info["useDonutEmissiveIntensity"])] == ref[1]
    assert info["texturePath"] == ref[2]
    assert [(int(bool(a)), int(bool(b))) for a, b in zip(info["textureSRGB"], info["textureNormalMap"])] == [(int(bool(a)), int(bool(b))) for a, b in ref[3]]


def test_gltf_transmission_material_is_a_refracting_solid(tmp_path):
    """MaterialsBaker::ImportFromDonut (MaterialsBaker.cpp:660-705) never derives ThinSurface from a glTF document and imports neither the index of
    refraction (`//materialPT->IoR = material.ior`) nor a volume: a plain KHR_materials_transmission material is a refracting solid with the PTMaterial
    defaults (IoR 1.5, nested priority 14, white / FLT_MAX attenuation), whatever KHR_materials_ior / KHR_materials_volume say; a material without
    transmission is a thin surface (FillData: `ThinSurface || !EnableTransmission`)."""
    import json
    import numpy as np
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gltf_writer import write_gltf
    sc, cam = scenes.cornell_box("C2")
    gltf = tmp_path / "m.gltf"
    write_gltf(sc, str(gltf))
    doc = json.loads(gltf.read_text())
    doc["materials"] = [
        {"name": "glass", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.95, 1.0, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.05},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.9}}},
        {"name": "glass_with_ior_and_volume", "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.9},
                                                          "KHR_materials_volume": {"thicknessFactor": 0.0, "attenuationDistance": 0.25, "attenuationColor": [0.2, 0.4, 0.8]}}},
        {"name": "opaque", "pbrMetallicRoughness": {"roughnessFactor": 0.4}, "extensions": {"KHR_materials_ior": {"ior": 1.2}}},
    ] + doc["materials"][3:]
    gltf.write_text(json.dumps(doc))
    (tmp_path / "m.scene.json").write_text(json.dumps({"models": ["m.gltf"], "graph": [{"model": 0}]}))
    m = pt.SceneImport(tmp_path / "m.scene.json").materials
    THIN, PRIORITY14, PSD_EXCLUDE = 0x200, 14 << 28, 0x400
    for k in (0, 1):
        assert not (int(m[k]["Flags"]) & THIN) and (int(m[k]["Flags"]) >> 28) == 14 and int(m[k]["Flags"]) & PSD_EXCLUDE
        assert m[k]["IoR"] == np.float32(1.5) and m[k]["AttenuationDistance"] == np.float32(3.402823466e38) and np.all(m[k]["AttenuationColor"] == 1.0)
        assert m[k]["DiffuseTransmissionFactor"] == 0.0
    assert m[0]["TransmissionFactor"] == np.float32(0.9) and m[1]["TransmissionFactor"] == np.float32(1.0)
    assert int(m[2]["Flags"]) & THIN and m[2]["TransmissionFactor"] == 0.0 and m[2]["IoR"] == np.float32(1.5)
    # the same through the reference's own FillData (compiled from MaterialsBaker.cpp where the reference tree is present): a PTMaterial with these fields
    words = (0xFFFFFFFF,) * 5
    ref, _ = pt.material_from_json(json.dumps({"BaseOrDiffuseColor": [0.9, 0.95, 1.0], "Metalness": 0.0, "Roughness": 0.05, "TransmissionFactor": 0.9, "EnableTransmission": True}), words)
    for f in ("Flags", "IoR", "TransmissionFactor", "Roughness", "Metalness", "AttenuationDistance"):
        assert ref[f] == m[0][f], f
