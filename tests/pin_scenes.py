"""Cases shared by tests/golden/make_reference_integrator_golden.py, tests/test_oracle_refpin_integrator.py and the GPU fixture test: name ->
(scene factory, settings, width, height, first sample, sample count). Small frames: the point is which code runs, not how many pixels."""
import numpy as np
from rtxpt_amd import scenes


def with_sphere_lights(make):
    """The scene of `make` plus four analytic sphere lights (two of them spot-shaped), packed with the oracle's PackColor / oct encoding (pt_set_lights path)."""
    import numpy as np
    from oracle import ptref
    def build():
        sc, cam = make()
        pos = np.array([[0.15, 0.45, 0.2], [0.42, 0.5, 0.35], [0.28, 0.3, 0.1], [0.1, 0.2, 0.45]], np.float32)
        rad = np.array([[40, 30, 10], [5, 20, 40], [60, 60, 60], [30, 5, 5]], np.float32); radius = np.array([0.02, 0.03, 0.015, 0.025], np.float32)
        col = ptref.light_probe(0, rad.view(np.uint32))
        axis = ptref.light_probe(4, np.array([[0, -1, 0], [0.3, -0.9, 0.1], [0, -1, 0], [0, 0, 1]], np.float32).view(np.uint32))[:, 0]
        f16 = lambda x: np.asarray(x, np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
        base = np.zeros((4, 8), np.uint32); ex = np.zeros((4, 4), np.uint32)
        base[:, 0:3] = pos.view(np.uint32); spot = np.array([0, 1, 1, 0], np.uint32)
        base[:, 3] = col[:, 0] | (0 << 24) | (spot << 28) | (np.array([0, 0, 1, 0], np.uint32) << 30); base[:, 6] = f16(radius); base[:, 7] = col[:, 1]
        ex[:, 1] = axis; ex[:, 2] = f16([0, 0.7, 0.85, 0]) | (f16([0, 0.2, 0.1, 0]) << 16); ex[:, 3] = np.arange(4) + 100
        sc = dict(sc); sc["lights"] = (base, ex)
        return sc, cam
    return build


def with_light_proxy(make, instance=-2, light=0, radius=0.06):
    """The scene of `make` (which has analytic sphere lights) with one mesh instance standing in for one of them (analytic light proxies: LightsBaker.cpp:718-753,
    PathTracer.hlsli:636-648): the light is moved to the centre of that instance's bounds and given a radius that fills a good part of it, the instance's materials get
    EnableAsAnalyticLightProxy (other users of those materials carry the flag without a light: no effect), and instance.analyticProxyLight names the light."""
    import numpy as np
    def build():
        sc, cam = make()
        sc = dict(sc); inst = sc["instances"].copy(); mats = sc["materials"].copy(); base, ex = sc["lights"]; base = base.copy()
        k = instance % len(inst); m = sc["meshes"][inst["meshIndex"][k]]; T = inst["transform"][k].reshape(3, 4); pts = []
        for g in sc["geometries"][m["firstGeometry"]: m["firstGeometry"] + m["numGeometries"]]:
            p = sc["positions"][g["vertexOffset"]: g["vertexOffset"] + g["numVertices"]]; pts.append(p @ T[:, :3].T + T[:, 3]); mats["Flags"][g["materialIndex"]] |= 0x800
        pts = np.concatenate(pts); centre = ((pts.min(0) + pts.max(0)) * 0.5).astype(np.float32)
        base[light, 0:3] = centre.view(np.uint32); base[light, 6] = (base[light, 6] & 0xFFFF0000) | int(np.float16(radius).view(np.uint16))
        inst["analyticProxyLight"][k] = light + 1
        sc["instances"] = inst; sc["materials"] = mats; sc["lights"] = (base, ex)
        return sc, cam
    return build


def with_point_light_record(make):
    """The scene of `make` (which already has analytic lights) plus the record LightsBaker::ConvertLight makes of a point light WITHOUT radius: a point-type
    PolymorphicLight. The path tracer's light set has that type compiled out (PolymorphicLightPTConfig.h:17-22: "handled by sphere"), so the record is inert —
    no power, empty samples — but it still occupies a slot of the light buffer, which uniform light selection (NEEType 0) notices."""
    import numpy as np
    import rtxpt_amd as pt
    def build():
        sc, cam = make()
        base, ex = sc["lights"]
        b, e = pt.convert_light("point", (0.3, 0.4, 0.3), (1.0, 0.5, 0.25), 80.0, 0.0)
        assert (int(b[3]) >> 24) & 0xF == 4
        sc = dict(sc); sc["lights"] = (np.concatenate([base[:2], b[None, :], base[2:]]), np.concatenate([ex[:2], e[None, :], ex[2:]]))
        return sc, cam
    return build


def with_excluded_geometry(make, which=(-1, -2)):
    """The scene of `make` with some geometries flagged ExcludeFromNEE (shadow rays pass through them, AccelerationStructureUtil.h:35-104, BridgeDonut:981-989)."""
    def build():
        sc, cam = make()
        sc = dict(sc); g = sc["geometries"].copy()
        for k in which: g["geomFlags"][k] |= scenes.GEOMF_EXCLUDE_FROM_NEE
        sc["geometries"] = g
        return sc, cam
    return build


def with_rotated_environment(make, yaw=0.9, pitch=0.35, tint=(1.4, 0.8, 0.6)):
    """The scene of `make` with its environment rotated and tinted (EnvMapSceneParams Transform / InvTransform / ColorMultiplier, EnvMap.hlsli:54-93)."""
    import math
    import numpy as np
    def build():
        sc, cam = make()
        rgb, tw, cm = sc["env"]
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        R = (Ry @ Rx).astype(np.float32)
        sc = dict(sc); sc["env"] = (rgb, np.concatenate([R, np.zeros((3, 1), np.float32)], axis=1).reshape(-1).astype(np.float32), np.asarray(tint, np.float32))
        return sc, cam
    return build


def with_sun_discs(make, cube_dim=None, compression=0):
    """The scene of `make` with two directional lights baked into its environment cube (Sample::UpdateLighting hands the scene's DirectionalLights to
    EnvMapBaker::Update, Sample.cpp:1361-1388; EnvMapBaker.hlsl:166-192 draws them as anti-aliased discs): a small bright sun and a wide dim disc.
    Rows: colour rgb, intensity, direction the light travels in, angular size [rad]."""
    import numpy as np
    def build():
        sc, cam = make()
        sc = dict(sc)
        d0 = -np.array([0.35, 0.8, -0.45]) / np.linalg.norm([0.35, 0.8, -0.45]); d1 = -np.array([-0.5, 0.6, 0.62]) / np.linalg.norm([-0.5, 0.6, 0.62])
        sc["env_directional_lights"] = np.array([[1.0, 0.92, 0.8, 2.5, d0[0], d0[1], d0[2], 0.05], [0.3, 0.5, 1.0, 0.8, d1[0], d1[1], d1[2], 0.6]], np.float32)
        if cube_dim: sc["env_cube_dim"] = cube_dim
        if compression: sc["env_compression"] = compression
        return sc, cam
    return build


_SKY_TEX = []
def _sky_textures():
    if not _SKY_TEX: _SKY_TEX.extend(scenes.procedural_sky_textures())
    return _SKY_TEX


def env_cube_cases():
    """Environment-bake inputs (a tiny scene carrying the source image, the cube resolution and the directional lights): the smallest cube (16: two levels),
    a 32 cube with two discs, a 64 cube of a source with an HDR sun that exceeds the fp16 range after scaling (the clamp to HLF_MAX)."""
    import numpy as np
    def scene(src, dim, lights=None):
        sc, _ = scenes.cornell_box("C2")
        sc = dict(sc); rgb, tw, cm = sc["env"]
        sc["env"] = (src, tw, cm); sc["env_cube_dim"] = dim
        if lights is not None: sc["env_directional_lights"] = np.asarray(lights, np.float32)
        return sc
    d0 = -np.array([0.35, 0.8, -0.45]) / np.linalg.norm([0.35, 0.8, -0.45]); d1 = -np.array([-0.5, 0.6, 0.62]) / np.linalg.norm([-0.5, 0.6, 0.62]); d2 = np.array([0.0, 0.0, 1.0])
    lights = [[1.0, 0.92, 0.8, 2.5, d0[0], d0[1], d0[2], 0.09], [0.3, 0.5, 1.0, 0.8, d1[0], d1[1], d1[2], 0.6], [1.0, 0.2, 0.1, 0.3, d2[0], d2[1], d2[2], 3.0]]
    def procsky(preset, time, dim, image=False, **kw):      # the procedural sky as source (EnvMapBaker.hlsl:228-236, 247-265), alone or on top of an image and discs
        import rtxpt_amd as pt
        consts, _ = pt.procedural_sky_update(pt.PtProceduralSkyState(), time, preset, force_instant=True)
        sc = scene(scenes.sky_equirect(128, 64), dim, lights[:1] if image else None)
        if not image: sc["env"] = None
        sc["sky"] = {"consts": consts.as_array(), "textures": _sky_textures()}
        sc.update(kw); return sc
    def cubesrc(srcDim, dim, seed, nlights=0, **kw):      # a CUBE map as the source image (EnvMapBaker.hlsl SampleSource, BackgroundSourceType 2; pt_set_environment_cube): noisy HDR faces, every tap matters
        rng = np.random.default_rng(seed)
        faces = np.concatenate([(rng.random((6, srcDim, srcDim, 3), np.float32) ** 4 * 40.0).astype(np.float32), np.ones((6, srcDim, srcDim, 1), np.float32)], axis=-1)
        sc = scene(scenes.sky_equirect(128, 64), dim, lights[:nlights] if nlights else None); _, tw, cm = sc["env"]
        sc["env"] = None; sc["env_cube_source"] = (faces, tw, cm); sc.update(kw); return sc
    return {
        "cubesrc_32_discs": cubesrc(24, 32, 11, nlights=2),
        "cubesrc_64_bc6": cubesrc(80, 64, 12, env_compression=1),
        "procsky_64_midday": procsky("==PROCEDURAL_SKY_MIDDAY==", 0.0, 64),
        "procsky_32_clock_image_discs_bc6": procsky("==PROCEDURAL_SKY==", 41000.0, 32, image=True, env_compression=1),
        "sky_16": scene(scenes.sky_equirect(128, 64), 16),
        "sky_32_discs": scene(scenes.sky_equirect(256, 128), 32, lights),
        "sky_64_hdr_sun": scene(scenes.sky_equirect(512, 256, sun_radiance=4e5, sun_deg=3.0), 64, lights[:1]),
        # ... and through EnvMapBaker's BC6U compression ("Fast", its D3D12 default): BC6UCompress.hlsl's EncodeP1 + the BC6H_UF16 decode, every level
        "sky_32_discs_bc6": dict(scene(scenes.sky_equirect(256, 128), 32, lights), env_compression=1),
        "sky_64_hdr_sun_bc6": dict(scene(scenes.sky_equirect(512, 256, sun_radiance=4e5, sun_deg=3.0), 64, lights[:1]), env_compression=1),
        # ... and with "Quality" (QUALITY 1: the best of the 32 two-region partitions in modes 7.6 / 9.5 where its error estimate beats the one-region block)
        "sky_32_discs_bc6q": dict(scene(scenes.sky_equirect(256, 128), 32, lights), env_compression=2),
        "sky_64_hdr_sun_bc6q": dict(scene(scenes.sky_equirect(512, 256, sun_radiance=4e5, sun_deg=3.0), 64, lights[:1]), env_compression=2),
    }


def with_mirrored_instance(make, index=0, width=0.5528):
    """The scene of `make` with one instance mirrored in x (negative determinant: flipped winding for the light baker, LightsBaker.hlsl:669-683, and for
    the face normals) and stretched a little in y (non-uniform scale through the normal / tangent transforms)."""
    import numpy as np
    def build():
        sc, cam = make()
        sc = dict(sc); inst = sc["instances"].copy()
        inst["transform"][index] = np.array([-1, 0, 0, width, 0, 1.05, 0, 0, 0, 0, 1, 0], np.float32).reshape(inst["transform"][index].shape)
        sc["instances"] = inst
        return sc, cam
    return build


def with_material_zoo(make):
    """The bistro-like scene with material features no generator switches on by itself: metal-rough texture with metalness in the red channel,
    transmission texture (thick and thin), diffuse transmission, ShadowNoLFadeout, IgnoreMeshTangentSpace on a normal-mapped material, an emissive
    texture on the lamps (light baking + emissive hits), a normal texture scale != 1."""
    import numpy as np
    def build():
        sc, cam = make()
        sc = dict(sc); m = sc["materials"].copy()
        tex = lambda k: m["BaseOrDiffuseTextureIndex"][k]                  # packed texture words of the textured facade materials 0..23
        for k in (1, 9, 17): m["MetalRoughOrSpecularTextureIndex"][k] = tex(k + 1); m["Flags"][k] |= 0x4; m["Metalness"][k] = 0.8
        for k in (9, 17): m["Flags"][k] |= 0x100                          # MetalnessInRedChannel
        for k in (2, 10): m["TransmissionTextureIndex"][k] = tex(k + 2); m["Flags"][k] |= 0x80; m["TransmissionFactor"][k] = 0.7
        m["Flags"][10] &= ~np.uint32(0x200); m["IoR"][10] = 1.3           # not thin: refracting
        for k in (3, 11): m["DiffuseTransmissionFactor"][k] = 0.6
        for k in (4, 12, 25, 26): m["ShadowNoLFadeout"][k] = 0.15
        m["Flags"][5] |= (1 << 12); m["NormalTextureScale"][6] = 0.5; m["NormalTextureScale"][7] = 1.7
        for k in range(48, 64, 3): m["EmissiveTextureIndex"][k] = tex(k % 24); m["Flags"][k] |= 0x10
        sc["materials"] = m
        return sc, cam
    return build


def with_spec_gloss(make, textured=True):
    """The scene of `make` with most materials switched to the specular-glossiness model (KHR_materials_pbrSpecularGlossiness as Donut imports it, what Bistro
    ships): diffuse colour stays, specular colour from dark dielectric (below 0.04: metalness 0 branch) to coloured metal, roughness = 1 - glossiness, and —
    on textured materials — a specular-glossiness texture (rgb scales the specular colour, alpha the glossiness). Every fourth material stays metal-rough."""
    import numpy as np
    def build():
        sc, cam = make()
        sc = dict(sc); m = sc["materials"].copy(); rng = np.random.default_rng(77)
        for k in range(len(m)):
            if k % 4 == 3: continue
            m["Flags"][k] |= 0x1
            kind = k % 3
            spec = (rng.uniform(0.0, 0.035, 3), rng.uniform(0.04, 0.3, 3), rng.uniform(0.5, 1.0, 3))[kind]
            m["SpecularColor"][k] = spec.astype(np.float32)
            if kind == 2: m["BaseOrDiffuseColor"][k] = (m["BaseOrDiffuseColor"][k] * np.float32(0.05)).astype(np.float32)      # a metal: hardly any diffuse
            m["Roughness"][k] = np.float32(rng.uniform(0.05, 0.9)); m["Metalness"][k] = 0
            if textured and (m["Flags"][k] & 0x8) and k % 2 == 0:
                m["MetalRoughOrSpecularTextureIndex"][k] = m["BaseOrDiffuseTextureIndex"][(k + 1) % 24 if (m["Flags"][(k + 1) % 24] & 0x8) else k]; m["Flags"][k] |= 0x4
        sc["materials"] = m
        return sc, cam
    return build


def cases():
    c2 = lambda: scenes.cornell_box("C2")
    return {
        "c1": (lambda: scenes.cornell_box("C1"), scenes.config_settings("C1"), 48, 48, 0, 2),                      # Lambert, 2 bounces, no RR
        "c2": (c2, scenes.config_settings("C2"), 64, 36, 0, 2),                                                    # StandardBSDF mix, env, nested glass, RR
        "c2_firefly": (c2, scenes.default_settings(fireflyFilterThreshold=2.5), 64, 36, 3, 2),
        "c2_nee3": (c2, scenes.default_settings(NEEFullSamples=3), 64, 36, 0, 2),                                  # HandleNEE_MultipleSamples
        "c2_nee_off": (c2, scenes.default_settings(NEEEnabled=0), 64, 36, 0, 2),
        "c2_nested2_norr_nold": (c2, scenes.default_settings(nestedDielectricsQuality=2, enableRussianRoulette=0, enableLDSamplerForBSDF=0), 64, 36, 0, 2),
        "c2_nested0_uniform": (c2, scenes.default_settings(nestedDielectricsQuality=0, NEEType=0), 64, 36, 0, 2),
        "c2_sphere_lights": (with_sphere_lights(c2), scenes.default_settings(), 64, 36, 0, 2),                     # analytic lights (pt_set_lights): spheres, spot shaping
        "c2_point_light_record_uniform": (with_point_light_record(with_sphere_lights(c2)), scenes.default_settings(NEEType=0), 64, 36, 0, 2),   # inert point-type record in the buffer
        "c2_sphere_light_proxy": (with_light_proxy(with_sphere_lights(c2)), scenes.default_settings(), 64, 36, 4, 2),   # a mesh standing in for an analytic light: SphereLight::Eval + MIS on hit
        "c2_exclude_from_nee": (with_excluded_geometry(c2), scenes.default_settings(), 64, 36, 0, 2),              # ExcludeFromNEE geometry: invisible to shadow rays
        "c2_env_rotated_mip2": (with_rotated_environment(c2), scenes.default_settings(envMapDiffuseSampleMIPLevel=2.0), 64, 36, 5, 2),   # env transform + tint, diffuse-bounce env MIP 2 (the UI default)
        "c2_sun_discs": (with_sun_discs(c2), scenes.default_settings(), 64, 36, 2, 2),                             # directional lights baked into the environment cube
        "c2_sun_discs_bc6": (with_sun_discs(c2, compression=1), scenes.default_settings(envMapDiffuseSampleMIPLevel=2.0), 64, 36, 7, 2),   # the BC6H-compressed cube (EnvMapBaker's D3D12 default)
        "c2_mirrored_room": (with_mirrored_instance(c2), scenes.default_settings(), 64, 36, 0, 2),                 # negative-determinant instance holding the quad light
        "bistro_like": (lambda: scenes.bistro_like(scale=0.02, tex_size=128), scenes.default_settings(), 96, 54, 0, 2),      # alpha test, textures, normal maps, emissive triangles, env quads
        "bistro_like_material_zoo": (with_material_zoo(lambda: scenes.bistro_like(scale=0.02, tex_size=128)), scenes.default_settings(), 96, 54, 2, 2),
        "bistro_like_sun_discs_cube512": (with_sun_discs(lambda: scenes.bistro_like(scale=0.02, tex_size=128), cube_dim=512), scenes.default_settings(envMapDiffuseSampleMIPLevel=2.0), 96, 54, 6, 2),
        "c2_spec_gloss": (with_spec_gloss(c2), scenes.default_settings(), 64, 36, 1, 2),                            # PTMaterialFlags_UseSpecularGlossModel: metal-rough reconstruction
        "bistro_like_spec_gloss": (with_spec_gloss(lambda: scenes.bistro_like(scale=0.02, tex_size=128)), scenes.default_settings(), 96, 54, 3, 2),      # + specular-glossiness textures
        "bistro_like_c5": (lambda: scenes.bistro_like(scale=0.01, tex_size=64, animated=True), scenes.default_settings(), 96, 54, 0, 2),   # + nested-dielectric props
    }


def cases_lp16():
    """The same cases in the reference's DEFAULT build (useFp16Types = 1: lp types in binary16, SampleUI.h:182 / Sample.cpp:1035), plus two in which the
    firefly filter — whose arithmetic is all-half in that build (PathTracerHelpers.hlsli:206-213, the lpfloat3 overload of Average) — meets textured
    emitters, environment hits and many lamps."""
    out = {}
    for name, (make, S, w, h, first, n) in cases().items():
        S = S.copy(); S["useFp16Types"] = 1
        out[name] = (make, S, w, h, first, n)
    bl = lambda: scenes.bistro_like(scale=0.02, tex_size=128)
    out["bistro_like_firefly"] = (bl, scenes.default_settings(fireflyFilterThreshold=0.7, useFp16Types=1), 96, 54, 1, 2)
    out["bistro_like_material_zoo_firefly"] = (with_material_zoo(bl), scenes.default_settings(fireflyFilterThreshold=1.5, envMapDiffuseSampleMIPLevel=2.0, useFp16Types=1), 96, 54, 4, 2)
    return out


def wide_cases():
    """One notch wider than cases(): 256 x 144 pixels x 4 samples per pin family, in both builds of the lp types — 147 456 paths per frame instead of ~10 000, so that the
    rarely taken branches of the integrator text (late bounces, Russian roulette survivors, nested-dielectric stacks several levels deep, rejected interior hits, the
    firefly filter on long paths) are hit thousands of times instead of a handful. name -> (make, settings, w, h, first, n); names ending in _lp16 run the reference's
    default build (useFp16Types = 1). Fixture: tests/golden/reference_integrator_golden_wide.npz (make_reference_integrator_golden.py)."""
    c2 = lambda: scenes.cornell_box("C2")
    bl = lambda: scenes.bistro_like(scale=0.02, tex_size=128)
    fam = {
        "c2_wide": (c2, scenes.config_settings("C2"), 256, 144, 0, 4),
        "bistro_like_wide": (bl, scenes.config_settings("C3"), 256, 144, 0, 4),                                    # the bench configuration's settings: 8 bounces, emissive NEE, RR
        "bistro_like_material_zoo_wide": (with_material_zoo(bl), scenes.default_settings(fireflyFilterThreshold=1.5, envMapDiffuseSampleMIPLevel=2.0), 256, 144, 4, 4),
        "bistro_like_c5_wide": (lambda: scenes.bistro_like(scale=0.01, tex_size=64, animated=True), scenes.default_settings(nestedDielectricsQuality=2), 256, 144, 8, 4),   # nested-dielectric props (C5's scene), quality 2
    }
    out = {}
    for name, (make, S, w, h, first, n) in fam.items():
        S32 = S.copy(); S32["useFp16Types"] = 0; S16 = S.copy(); S16["useFp16Types"] = 1
        out[name] = (make, S32, w, h, first, n); out[name + "_lp16"] = (make, S16, w, h, first, n)
    return out


def xl_cases():
    """One more notch for the bench configuration's settings: 1280 x 720 x 4 samples (3 686 400 paths, a ninth of the 4K frame) of the reference's integrator text on the bistro-like
    scene at scale 0.3, both lp builds. The fixture keeps every sixteenth row and a SHA-256 of the whole frame (tests/golden/reference_integrator_golden_xl.npz)."""
    bl = lambda: scenes.bistro_like(scale=0.3, tex_size=256)
    S32 = scenes.config_settings("C3"); S32["useFp16Types"] = 0; S16 = S32.copy(); S16["useFp16Types"] = 1
    return {"bistro_like_xl": (bl, S32, 1280, 720, 0, 4), "bistro_like_xl_lp16": (bl, S16, 1280, 720, 0, 4)}


XL_ROW_STEP = 16


def frame_digest(rad):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(rad, np.float32).tobytes()).digest(), np.uint8).copy()


def neeat_cases():
    """NEE-AT, the path tracer's side (NEEType 2; SURVEY.md §8 row N4, first part): frames traced with screen-tile local samplers and / or temporal feedback.
    name -> (make, settings, w, h, first, n, dict(table_seed | None, jitter, ratio, ssc_threshold, feedback)). The tables are synthetic stand-ins for the
    baker's output (scenes.synthetic_local_light_tables); what is pinned is everything the path tracer does with them."""
    c2 = lambda: scenes.cornell_box("C2")
    bl = lambda: scenes.bistro_like(scale=0.02, tex_size=128)
    d = scenes.default_settings
    return {
        # the reference's defaults: ratio 0.65 (SampleUI.h:159), threshold 0.3 (LightsBaker.h:240), NEEFullSamples 1, feedback on
        "bistro_like_neeat": (bl, d(NEEType=2), 96, 54, 0, 2, dict(table_seed=5, jitter=(3, 5), ratio=0.65, ssc_threshold=0.3, feedback=True)),
        "bistro_like_neeat_lp16": (bl, d(NEEType=2, useFp16Types=1), 96, 54, 1, 2, dict(table_seed=6, jitter=(7, 0), ratio=0.65, ssc_threshold=0.3, feedback=True)),
        # local layer without feedback and with NEEFullSamples 3 (grouped shadow queue), nearly all candidates local
        "c2_neeat_table_only_nee3": (c2, d(NEEType=2, NEEFullSamples=3), 64, 36, 0, 2, dict(table_seed=7, jitter=(0, 0), ratio=0.95, ssc_threshold=0.3, feedback=False)),
        # feedback without a local layer (the first frames of a run, before the baker has produced tables), analytic lights, firefly filter
        "c2_sphere_lights_neeat_feedback_only": (with_sphere_lights(c2), d(NEEType=2, fireflyFilterThreshold=2.5), 64, 36, 2, 2, dict(table_seed=None, jitter=(0, 0), ratio=0.65, ssc_threshold=0.3, feedback=True)),
        # every vertex screen-space coherent (threshold above any cone-width ratio), no Russian roulette, one candidate sample (all global: (1 - 1) * ratio + 0.75 -> 0)
        "bistro_like_c5_neeat_all_ssc": (lambda: scenes.bistro_like(scale=0.01, tex_size=64, animated=True), d(NEEType=2, enableRussianRoulette=0, NEECandidateSamples=1), 96, 54, 0, 2, dict(table_seed=8, jitter=(1, 1), ratio=0.65, ssc_threshold=1e9, feedback=True)),
        # a mesh standing in for an analytic light (SphereLight::Eval on hit, MIS with both samplers' pdfs), rotated tinted environment at MIP 2, NEEFullSamples 2 without feedback
        "c2_sphere_light_proxy_neeat_nee2": (with_rotated_environment(with_light_proxy(with_sphere_lights(c2))), d(NEEType=2, NEEFullSamples=2, envMapDiffuseSampleMIPLevel=2.0), 64, 36, 4, 2,
                                             dict(table_seed=11, jitter=(5, 2), ratio=0.65, ssc_threshold=0.3, feedback=False)),
        # no vertex coherent (threshold 0): the table is bound but never sampled; many candidates
        "bistro_like_neeat_none_ssc": (bl, d(NEEType=2, NEECandidateSamples=9), 96, 54, 3, 1, dict(table_seed=9, jitter=(2, 6), ratio=0.5, ssc_threshold=0.0, feedback=True)),
    }


def neeat_table(opts, num_lights, w, h):
    if opts["table_seed"] is None: return None
    return scenes.synthetic_local_light_tables(num_lights, w, h, seed=opts["table_seed"], jitter=opts["jitter"])


def neeat_loop_cases():
    """NEE-AT with the light baker in the loop (pt_set_neeat): name -> (make, settings, w, h, frames, dict(global_feedback_weight, ratio, ssc_threshold, prefilter)).
    Every frame = LightsBaker's feedback passes, then one sample of the path tracer; the fixtures hold the reference text's output of every frame."""
    c2 = lambda: scenes.cornell_box("C2")
    bl = lambda: scenes.bistro_like(scale=0.02, tex_size=128)
    d = scenes.default_settings
    dflt = dict(global_feedback_weight=0.75, ratio=0.65, ssc_threshold=0.3, prefilter=True)
    return {
        "bistro_like_loop": (bl, d(NEEType=2), 96, 54, 4, dflt),                                                                   # the reference's defaults
        "c2_sphere_lights_loop_lp16": (with_sphere_lights(c2), d(NEEType=2, useFp16Types=1, fireflyFilterThreshold=2.5), 61, 35, 3, dflt),      # odd frame size: partial tiles, partial low-res pixels
        "bistro_like_c5_loop_nofilter": (lambda: scenes.bistro_like(scale=0.01, tex_size=64, animated=True), d(NEEType=2, NEECandidateSamples=3), 96, 54, 3,
                                         dict(global_feedback_weight=0.3, ratio=0.9, ssc_threshold=0.5, prefilter=False)),
        # with the host's world-to-clip matrix: the path tracer exports the clip depth of every path's last vertex and the baker's Reproject tests it (frame 1: everything
        # "disoccluded" against the cleared history, later frames: wherever two consecutive paths ended 1.5 x apart) — and the frustum importance boost on top
        "bistro_like_loop_depth_boost": (bl, d(NEEType=2, useFp16Types=1), 96, 54, 5, dict(dflt, view_projection=True, importance_boost=True)),
    }
