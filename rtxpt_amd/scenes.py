"""Deterministic synthetic scene generators + host-side data-contract helpers (numpy only).

The reference ships no scene assets, so the BASELINE.json configurations are generated here from fixed
seeds: the Cornell box (configs C1/C2) and a "bistro-like" street canyon (configs C3-C5).

Everything is expressed in the reference's GPU data contract so that the same arrays can be handed to the C-ABI
(`pt_set_geometry`, `pt_set_instances`, `pt_set_materials`, ...):
  * `PTMaterialData` 128 B            /root/reference/Rtxpt/Shaders/PathTracer/Materials/MaterialPT.h:45-77
  * vertex streams                    position float3, uv float2, normal / tangent RGBA8_SNORM, u32 indices
                                      (PathTracerBridgeDonut.hlsli:152-256; Packing.hlsli:127-167)
  * `PathTracerCameraData`            PathTracerShared.h:24-42, filled like `BridgeCamera` (PathTracerShared.h:109-141)
"""
import math
import numpy as np

SEED_BASE = 0x5EED0001

MATERIAL_DTYPE = np.dtype([
    ("BaseOrDiffuseColor", "<f4", 3), ("Flags", "<u4"),
    ("SpecularColor", "<f4", 3), ("_padding0", "<i4"),
    ("EmissiveColor", "<f4", 3), ("ShadowNoLFadeout", "<f4"),
    ("Opacity", "<f4"), ("Roughness", "<f4"), ("Metalness", "<f4"), ("NormalTextureScale", "<f4"),
    ("_padding1", "<f4"), ("AlphaCutoff", "<f4"), ("TransmissionFactor", "<f4"), ("BaseOrDiffuseTextureIndex", "<u4"),
    ("MetalRoughOrSpecularTextureIndex", "<u4"), ("EmissiveTextureIndex", "<u4"), ("NormalTextureIndex", "<u4"), ("OcclusionTextureIndex", "<u4"),
    ("TransmissionTextureIndex", "<u4"), ("IoR", "<f4"), ("ThicknessFactor", "<f4"), ("DiffuseTransmissionFactor", "<f4"),
    ("AttenuationColor", "<f4", 3), ("AttenuationDistance", "<f4"),
])
assert MATERIAL_DTYPE.itemsize == 128

GEOMETRY_DTYPE = np.dtype([("indexOffset", "<u4"), ("numIndices", "<u4"), ("vertexOffset", "<u4"), ("numVertices", "<u4"),
                           ("flags", "<u4"), ("materialIndex", "<u4"), ("geomFlags", "<u4"), ("_pad", "<u4")])
MESH_DTYPE = np.dtype([("firstGeometry", "<u4"), ("numGeometries", "<u4")])
INSTANCE_DTYPE = np.dtype([("transform", "<f4", 12), ("meshIndex", "<u4"), ("_pad", "<u4", 3)])
CAMERA_DTYPE = np.dtype([
    ("PosW", "<f4", 3), ("NearZ", "<f4"), ("DirectionW", "<f4", 3), ("PixelConeSpreadAngle", "<f4"),
    ("CameraU", "<f4", 3), ("FarZ", "<f4"), ("CameraV", "<f4", 3), ("FocalDistance", "<f4"),
    ("CameraW", "<f4", 3), ("AspectRatio", "<f4"), ("ViewportSize", "<u4", 2), ("ApertureRadius", "<f4"), ("_padding0", "<f4"),
    ("Jitter", "<f4", 2), ("_padding1", "<f4"), ("_padding2", "<f4"),
])
assert CAMERA_DTYPE.itemsize == 112
SETTINGS_DTYPE = np.dtype([
    ("bounceCount", "<u4"), ("diffuseBounceCount", "<u4"), ("perPixelJitterAAScale", "<f4"), ("texLODBias", "<f4"),
    ("fireflyFilterThreshold", "<f4"), ("envMapDiffuseSampleMIPLevel", "<f4"),
    ("NEEEnabled", "<u4"), ("NEEType", "<u4"), ("NEECandidateSamples", "<u4"), ("NEEFullSamples", "<u4"),
    ("enableRussianRoulette", "<u4"), ("nestedDielectricsQuality", "<u4"), ("enableLDSamplerForBSDF", "<u4"), ("diffuseBrdf", "<u4"),
    ("_pad", "<u4", 2),
])
assert SETTINGS_DTYPE.itemsize == 64

GEOM_HAS_UV, GEOM_HAS_NORMAL, GEOM_HAS_TANGENT = 1, 2, 4
GEOMF_ALPHA_TESTED, GEOMF_EXCLUDE_FROM_NEE = 1, 2
# MaterialPT.h:24-42
MF_UseMetalRoughOrSpecularTexture, MF_UseBaseOrDiffuseTexture, MF_UseEmissiveTexture, MF_UseNormalTexture = 0x4, 0x8, 0x10, 0x20
MF_UseTransmissionTexture, MF_ThinSurface, MF_NestedPriorityShift = 0x80, 0x200, 28
TEX_RGBA8_UNORM, TEX_RGBA8_SRGB, TEX_RGBA32F = 0, 1, 2


def default_settings(**kw):
    """Reference defaults for reference mode with the parity knobs of SURVEY.md §8a pinned
    (SampleUI.h:152-183,212-222; per-scene SampleSettings maxBounces override, ExtendedScene.cpp:354-363)."""
    s = np.zeros((), dtype=SETTINGS_DTYPE)
    s["bounceCount"] = 8
    s["diffuseBounceCount"] = 8
    s["perPixelJitterAAScale"] = 1.0          # AccumulationAA (Sample.cpp:1501)
    s["texLODBias"] = -1.0                    # SampleUI.h TexLODBias
    s["fireflyFilterThreshold"] = 0.0         # --disableFireflyFilters
    s["envMapDiffuseSampleMIPLevel"] = 0.0    # pinned to 0 (default 2 is a realtime perf knob)
    s["NEEEnabled"] = 1
    s["NEEType"] = 1                          # power: stationary distribution
    s["NEECandidateSamples"] = 5
    s["NEEFullSamples"] = 1
    s["enableRussianRoulette"] = 1
    s["nestedDielectricsQuality"] = 1
    s["enableLDSamplerForBSDF"] = 1
    s["diffuseBrdf"] = 2                      # Frostbite (BxDFConfig.hlsli:24)
    for k, v in kw.items():
        s[k] = v
    return s


def make_material(base=(0.8, 0.8, 0.8), emissive=(0, 0, 0), roughness=1.0, metalness=0.0, ior=1.5, transmission=0.0,
                  diff_transmission=0.0, thin=True, nested_priority=0, flags=0, alpha_cutoff=0.5, att_color=(1, 1, 1), att_dist=1e30,
                  base_tex=None, emissive_tex=None, normal_tex=None, mr_tex=None, shadow_nol_fadeout=0.0):
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["BaseOrDiffuseColor"] = base
    m["SpecularColor"] = (0, 0, 0)
    m["EmissiveColor"] = emissive
    m["Opacity"] = 1.0
    m["Roughness"] = roughness
    m["Metalness"] = metalness
    m["NormalTextureScale"] = 1.0
    m["AlphaCutoff"] = alpha_cutoff
    m["TransmissionFactor"] = transmission
    m["DiffuseTransmissionFactor"] = diff_transmission
    m["IoR"] = ior
    m["AttenuationColor"] = att_color
    m["AttenuationDistance"] = att_dist
    m["ShadowNoLFadeout"] = shadow_nol_fadeout
    f = int(flags)
    # thin-surface is forced on whenever transmission is disabled (MaterialsBaker.cpp:543-544)
    if thin or (transmission == 0.0 and diff_transmission == 0.0):
        f |= MF_ThinSurface
    f |= (int(nested_priority) & 0xF) << MF_NestedPriorityShift
    for tex, flag, field in ((base_tex, MF_UseBaseOrDiffuseTexture, "BaseOrDiffuseTextureIndex"), (emissive_tex, MF_UseEmissiveTexture, "EmissiveTextureIndex"),
                             (normal_tex, MF_UseNormalTexture, "NormalTextureIndex"), (mr_tex, MF_UseMetalRoughOrSpecularTexture, "MetalRoughOrSpecularTextureIndex")):
        if tex is not None:
            f |= flag
            m[field] = tex          # packed texture word, see pack_texture_word()
        else:
            m[field] = 0xFFFFFFFF
    m["OcclusionTextureIndex"] = 0xFFFFFFFF
    m["TransmissionTextureIndex"] = 0xFFFFFFFF
    m["Flags"] = f & 0xFFFFFFFF
    return m


def pack_texture_word(index, w, h):
    """MaterialsBaker.cpp:497-508: baseLOD<<24 | mipLevels<<16 | bindlessIndex, baseLOD = round(log2(w*h))."""
    mips = int(math.floor(math.log2(max(w, h)))) + 1
    base_lod = int(round(math.log2(w * h)))
    return (base_lod << 24) | (mips << 16) | (index & 0xFFFF)


def pack_snorm8(v):
    """Packing.hlsli:127-140 Pack_RGB8_SNORM / Pack_RGBA8_SNORM: int(clamp(v,-1,1)*127) & 0xff per byte."""
    v = np.asarray(v, dtype=np.float32)
    q = (np.clip(v, -1.0, 1.0) * np.float32(127.0)).astype(np.int32) & 0xFF
    out = q[..., 0].astype(np.uint32) | (q[..., 1].astype(np.uint32) << 8) | (q[..., 2].astype(np.uint32) << 16)
    if v.shape[-1] == 4:
        out |= q[..., 3].astype(np.uint32) << 24
    return out.astype(np.uint32)


def bridge_camera(width, height, pos, direction, up, fov_y, near_z=0.01, far_z=1e5, focal_distance=10.0, aperture_radius=0.0, jitter=(0.0, 0.0)):
    """float32 restatement of BridgeCamera (PathTracerShared.h:109-141); verified against the reference's own code in tests."""
    f32 = np.float32

    def norm(v):
        v = np.asarray(v, dtype=f32)
        l = np.sqrt(f32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), dtype=f32)
        return (v / l).astype(f32)

    def cross(a, b):
        return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32)

    cam = np.zeros((), dtype=CAMERA_DTYPE)
    aspect = f32(width) / f32(height)
    d = norm(direction)
    W = (d * f32(focal_distance)).astype(f32)
    U = norm(cross(W, np.asarray(up, dtype=f32)))
    V = norm(cross(U, W))
    t = f32(np.tan(f32(fov_y) * f32(0.5)))
    ulen = f32(f32(f32(focal_distance) * t) * aspect)
    vlen = f32(f32(focal_distance) * t)
    cam["PosW"] = np.asarray(pos, dtype=f32)
    cam["NearZ"] = near_z
    cam["FarZ"] = far_z
    cam["DirectionW"] = d
    cam["CameraW"] = W
    cam["CameraU"] = (U * ulen).astype(f32)
    cam["CameraV"] = (V * vlen).astype(f32)
    cam["FocalDistance"] = focal_distance
    cam["AspectRatio"] = aspect
    cam["ViewportSize"] = (width, height)
    cam["ApertureRadius"] = aperture_radius
    cam["PixelConeSpreadAngle"] = f32(np.arctan(f32(f32(2.0) * t) / f32(height)))
    cam["Jitter"] = (f32(jitter[0]), f32(-jitter[1]))
    return cam


class SceneBuilder:
    """Accumulates geometries/meshes/instances into the flat arrays of the data contract."""

    def __init__(self):
        self.indices, self.positions, self.uvs, self.normals, self.tangents = [], [], [], [], []
        self.geometries, self.meshes, self.instances, self.materials, self.textures = [], [], [], [], []
        self.nv = 0
        self.ni = 0
        self.env = None

    def add_material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def add_texture(self, pixels, fmt):
        """pixels: (h, w, 4) uint8 or float32. Returns the packed texture word for PTMaterialData."""
        h, w = pixels.shape[:2]
        self.textures.append((w, h, fmt, np.ascontiguousarray(pixels)))
        return pack_texture_word(len(self.textures) - 1, w, h)

    def begin_mesh(self):
        self._first_geom = len(self.geometries)

    def end_mesh(self):
        self.meshes.append((self._first_geom, len(self.geometries) - self._first_geom))
        return len(self.meshes) - 1

    def add_geometry(self, pos, idx, material, uv=None, normal=None, tangent=None, geom_flags=0):
        pos = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
        idx = np.asarray(idx, dtype=np.uint32).reshape(-1)
        n = pos.shape[0]
        flags = 0
        if uv is not None:
            flags |= GEOM_HAS_UV
            self.uvs.append(np.asarray(uv, dtype=np.float32).reshape(n, 2))
        else:
            self.uvs.append(np.zeros((n, 2), np.float32))
        if normal is None:      # flat normals need unshared vertices; callers that want them pass them explicitly
            self.normals.append(np.zeros(n, np.uint32))
        else:
            flags |= GEOM_HAS_NORMAL
            self.normals.append(pack_snorm8(np.asarray(normal, dtype=np.float32).reshape(n, 3)))
        if tangent is None:
            self.tangents.append(np.zeros(n, np.uint32))
        else:
            flags |= GEOM_HAS_TANGENT
            self.tangents.append(pack_snorm8(np.asarray(tangent, dtype=np.float32).reshape(n, 4)))
        self.positions.append(pos)
        self.indices.append(idx)
        self.geometries.append((self.ni, idx.size, self.nv, n, flags, material, geom_flags, 0))
        self.nv += n
        self.ni += idx.size

    def add_instance(self, mesh, transform=None):
        t = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32) if transform is None else np.asarray(transform, np.float32).reshape(12)
        self.instances.append((t, mesh, (0, 0, 0)))

    def set_environment(self, rgb, to_world=None, color_multiplier=(1, 1, 1)):
        self.env = (np.ascontiguousarray(rgb, dtype=np.float32), np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32) if to_world is None else np.asarray(to_world, np.float32),
                    np.asarray(color_multiplier, np.float32))

    def finish(self):
        sc = {
            "indices": np.concatenate(self.indices).astype(np.uint32), "positions": np.concatenate(self.positions).astype(np.float32),
            "uvs": np.concatenate(self.uvs).astype(np.float32), "normals": np.concatenate(self.normals).astype(np.uint32),
            "tangents": np.concatenate(self.tangents).astype(np.uint32),
            "geometries": np.array(self.geometries, dtype=GEOMETRY_DTYPE), "meshes": np.array(self.meshes, dtype=MESH_DTYPE),
            "instances": np.array(self.instances, dtype=INSTANCE_DTYPE), "materials": np.array(self.materials, dtype=MATERIAL_DTYPE),
            "textures": self.textures, "env": self.env,
        }
        return sc


def trs(translate=(0, 0, 0), rot_y=0.0, scale=(1, 1, 1)):
    c, s = math.cos(rot_y), math.sin(rot_y)
    sx, sy, sz = scale
    return np.array([c * sx, 0, s * sz, translate[0], 0, sy, 0, translate[1], -s * sx, 0, c * sz, translate[2]], np.float32)


def quad(p0, p1, p2, p3, uv_scale=1.0):
    """Two triangles (p0,p1,p2),(p0,p2,p3) with a flat normal = normalize(cross(p1-p0, p2-p0)); returns pos, idx, uv, normal, tangent."""
    p = np.array([p0, p1, p2, p3], np.float32)
    n = np.cross(p[1] - p[0], p[2] - p[0])
    n = n / np.linalg.norm(n)
    t = (p[1] - p[0]) / np.linalg.norm(p[1] - p[0])
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uv_scale
    return p, np.array([0, 1, 2, 0, 2, 3], np.uint32), uv, np.tile(n, (4, 1)), np.tile(np.append(t, 1.0), (4, 1))


def unit_cube():
    """Axis-aligned cube [-0.5,0.5]^3, 24 unshared vertices, outward normals, 12 triangles."""
    P, I, UV, N, T = [], [], [], [], []
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 1, 0), (0, 0, -1)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (0, 0, 1), (-1, 0, 0)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0))]
    for k, (n, a, b) in enumerate(faces):
        n, a, b = np.array(n, np.float32), np.array(a, np.float32), np.array(b, np.float32)
        # make (a, b, n) right-handed so that the winding is counter-clockwise seen from outside
        if np.dot(np.cross(a, b), n) < 0:
            a, b = b, a
        c = n * 0.5
        quad_p = [c - a * 0.5 - b * 0.5, c + a * 0.5 - b * 0.5, c + a * 0.5 + b * 0.5, c - a * 0.5 + b * 0.5]
        P += quad_p
        I += [4 * k + 0, 4 * k + 1, 4 * k + 2, 4 * k + 0, 4 * k + 2, 4 * k + 3]
        UV += [[0, 0], [1, 0], [1, 1], [0, 1]]
        N += [n] * 4
        T += [np.append(a, 1.0)] * 4
    return np.array(P, np.float32), np.array(I, np.uint32), np.array(UV, np.float32), np.array(N, np.float32), np.array(T, np.float32)


def sky_equirect(w=1024, h=512, horizon=(0.9, 0.95, 1.0), zenith=(0.25, 0.45, 0.9), sun_dir=(0.35, 0.8, -0.45), sun_radiance=5e4, sun_deg=1.0, ground=(0.15, 0.14, 0.13)):
    """Analytic gradient sky + sun disc as a lat-long float image. Row 0 = +Y; u from atan2(x,-z)."""
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    theta = v * math.pi
    phi = (2.0 * u - 1.0) * math.pi
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    d = np.stack([st * np.sin(phi)[None, :], np.broadcast_to(ct, (h, w)), -st * np.cos(phi)[None, :]], axis=-1)
    up = np.clip(d[..., 1], 0.0, 1.0)[..., None]
    img = np.asarray(horizon)[None, None, :] * (1.0 - up) + np.asarray(zenith)[None, None, :] * up
    img = np.where(d[..., 1:2] < 0.0, np.asarray(ground)[None, None, :], img)
    sd = np.asarray(sun_dir, np.float64)
    sd = sd / np.linalg.norm(sd)
    cosang = (d * sd[None, None, :]).sum(-1)
    img = np.where((cosang > math.cos(math.radians(sun_deg * 0.5)))[..., None], np.float64(sun_radiance), img)
    return img.astype(np.float32)


def cornell_box(variant="C1"):
    """Cornell box in metres (canonical data x 0.001). variant 'C1': all Lambertian; 'C2': StandardBSDF mix + sky env.
    Returns (scene dict, camera kwargs)."""
    s = 0.001
    b = SceneBuilder()
    lamb = variant == "C1"
    ior_l = 1.0 if lamb else 1.5       # IoR 1 -> F0 = 0 -> no specular lobe ("Lambertian only")
    white = b.add_material(make_material(base=(0.725, 0.71, 0.68), roughness=1.0 if lamb else 0.6, ior=ior_l))
    red = b.add_material(make_material(base=(0.63, 0.065, 0.05), roughness=1.0, ior=ior_l))
    green = b.add_material(make_material(base=(0.14, 0.45, 0.091), roughness=1.0, ior=ior_l))
    light = b.add_material(make_material(base=(0.78, 0.78, 0.78), emissive=(17.0, 12.0, 4.0), roughness=1.0, ior=ior_l))
    if lamb:
        short_m = tall_m = white
    else:
        short_m = b.add_material(make_material(base=(0.95, 0.95, 0.95), roughness=0.0, ior=1.5, transmission=1.0, thin=False, nested_priority=2,
                                               att_color=(0.8, 0.95, 0.85), att_dist=0.25))
        tall_m = b.add_material(make_material(base=(0.95, 0.75, 0.35), roughness=0.3, metalness=1.0, ior=1.5))
    X, Y, Z = 0.5528 * 1000 * s, 0.5488 * 1000 * s, 0.5592 * 1000 * s
    b.begin_mesh()
    # floor (normal +y), ceiling (-y), back wall (-z faces camera => normal -z), right wall (green, x=0, normal +x), left wall (red, x=X, normal -x)
    for quad_pts, mat in (
        (((0, 0, 0), (0, 0, Z), (X, 0, Z), (X, 0, 0)), white),
        (((0, Y, 0), (X, Y, 0), (X, Y, Z), (0, Y, Z)), white),
        (((0, 0, Z), (0, Y, Z), (X, Y, Z), (X, 0, Z)), white),
        (((0, 0, 0), (0, Y, 0), (0, Y, Z), (0, 0, Z)), green),
        (((X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0)), red),
    ):
        p, i, uv, n, t = quad(*quad_pts)
        b.add_geometry(p, i, mat, uv=uv, normal=n, tangent=t)
    # area light 130 x 105 just below the ceiling, facing down
    lx0, lx1, lz0, lz1, ly = 0.213, 0.343, 0.227, 0.332, Y - 0.0002
    p, i, uv, n, t = quad((lx0, ly, lz0), (lx1, ly, lz0), (lx1, ly, lz1), (lx0, ly, lz1))
    b.add_geometry(p, i, light, uv=uv, normal=n, tangent=t)
    room = b.end_mesh()
    b.add_instance(room)
    cp, ci, cuv, cn, ct = unit_cube()
    b.begin_mesh()
    b.add_geometry(cp, ci, short_m, uv=cuv, normal=cn, tangent=ct)
    short = b.end_mesh()
    b.begin_mesh()
    b.add_geometry(cp, ci, tall_m, uv=cuv, normal=cn, tangent=ct)
    tall = b.end_mesh()
    b.add_instance(short, trs((0.185, 0.0825 + 1e-4, 0.169), rot_y=-0.29, scale=(0.165, 0.165, 0.165)))
    b.add_instance(tall, trs((0.368, 0.165 + 1e-4, 0.351), rot_y=0.30, scale=(0.165, 0.33, 0.165)))
    if not lamb:
        b.set_environment(sky_equirect(), color_multiplier=(1, 1, 1))
    cam = dict(pos=(0.278, 0.273, -0.8), direction=(0, 0, 1), up=(0, 1, 0), fov_y=math.radians(39.3), near_z=0.01, far_z=100.0, focal_distance=1.0)
    return b.finish(), cam


def config_settings(name):
    """PtSettings for the BASELINE.json configurations."""
    if name == "C1":
        return default_settings(bounceCount=2, diffuseBounceCount=2, enableRussianRoulette=0, diffuseBrdf=0)
    return default_settings()
