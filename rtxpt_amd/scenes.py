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
INSTANCE_DTYPE = np.dtype([("transform", "<f4", 12), ("meshIndex", "<u4"), ("analyticProxyLight", "<u4"), ("_pad", "<u4", 2)])
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
    ("useFp16Types", "<u4"), ("_pad", "<u4"),
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
    s["useFp16Types"] = 0                     # lp types: 0 = fp32 build, 1 = binary16 — the reference's default (SampleUI.h:182), which pt_default_settings returns. The
    #                                           helper keeps 0 so that the fp32 fixtures stay what they are; the lp16 cases ask for 1 explicitly (tests/pin_scenes.py)
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
        self.instances.append((t, mesh, 0, (0, 0)))

    def set_environment(self, rgb, to_world=None, color_multiplier=(1, 1, 1)):
        self.env = (np.ascontiguousarray(rgb, dtype=np.float32), np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32) if to_world is None else np.asarray(to_world, np.float32),
                    np.asarray(color_multiplier, np.float32))

    def finish(self):
        def cat(parts, dtype, shape):          # an empty scene (no geometry at all) is legal: every ray misses
            return np.concatenate(parts).astype(dtype) if parts else np.zeros(shape, dtype)
        sc = {
            "indices": cat(self.indices, np.uint32, (0,)), "positions": cat(self.positions, np.float32, (0, 3)),
            "uvs": cat(self.uvs, np.float32, (0, 2)), "normals": cat(self.normals, np.uint32, (0,)),
            "tangents": cat(self.tangents, np.uint32, (0,)),
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


def sphere_patch(radius, half_width, n=12):
    """A square patch of a sphere of the given radius, bulging towards -z around the origin: (n + 1)^2 shared vertices with smooth normals (towards -z, away from the centre
    at (0, 0, radius)), 2 n^2 triangles wound counter-clockwise seen from -z. For the curvature-driven motion-vector block types (PathTracerBridgeDonut.hlsli:92-149, 704-716).
    Returns pos, idx, uv, normal, tangent like quad()."""
    P, UV, N, T, I = [], [], [], [], []
    for j in range(n + 1):
        for i in range(n + 1):
            x, y = (2.0 * i / n - 1.0) * half_width, (2.0 * j / n - 1.0) * half_width
            z = radius - math.sqrt(radius * radius - x * x - y * y)
            nrm = np.array([x, y, z - radius], np.float64); nrm /= np.linalg.norm(nrm)
            P.append((x, y, z)); N.append(nrm); UV.append((i / n, j / n)); T.append((1.0, 0.0, 0.0, 1.0))
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i, j * (n + 1) + i + 1, (j + 1) * (n + 1) + i + 1, (j + 1) * (n + 1) + i
            I += [a, d, c, a, c, b]
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


# ---------------------------------------------------------------------------------------------------------------------
# "bistro-like" street canyon. The real Bistro asset is not obtainable offline.
# ---------------------------------------------------------------------------------------------------------------------
def _value_noise(rng, size, octaves=(4, 8, 16, 32, 64, 128), persistence=0.6):
    img = np.zeros((size, size), np.float32)
    amp, total = 1.0, 0.0
    for o in octaves:
        g = rng.random((o, o), dtype=np.float32)
        # bilinear upsample with wrap
        x = (np.arange(size, dtype=np.float32) + 0.5) * (o / size) - 0.5
        x0 = np.floor(x).astype(np.int64)
        fx = (x - x0).astype(np.float32)
        x0m, x1m = x0 % o, (x0 + 1) % o
        rows = g[:, x0m] * (1 - fx)[None, :] + g[:, x1m] * fx[None, :]
        up = rows[x0m, :] * (1 - fx)[:, None] + rows[x1m, :] * fx[:, None]
        img += amp * up
        total += amp
        amp *= persistence
    return img / total


def _make_textures(rng, b, tex_size):
    """24 base-colour (sRGB) + 4 leaf RGBA (alpha) + 4 normal maps = 32 textures. Returns lists of packed texture words."""
    base_words, leaf_words, normal_words = [], [], []
    for i in range(24):
        n = _value_noise(rng, tex_size)
        n2 = _value_noise(rng, tex_size, octaves=(16, 64, 256) if tex_size >= 256 else (4, 8, 16))
        tint = 0.35 + 0.6 * rng.random(3)
        # brick / plank like banding
        yy = (np.arange(tex_size) // max(1, tex_size // (8 + 4 * (i % 4)))) % 2
        band = (0.85 + 0.15 * yy)[:, None]
        rgb = np.clip((0.45 + 0.55 * n)[..., None] * tint[None, None, :] * band[..., None] * (0.8 + 0.2 * n2)[..., None], 0, 1)
        px = np.concatenate([(rgb * 255).astype(np.uint8), np.full((tex_size, tex_size, 1), 255, np.uint8)], -1)
        base_words.append(b.add_texture(px, TEX_RGBA8_SRGB))
    for i in range(4):
        n = _value_noise(rng, tex_size, octaves=(8, 16, 32, 64))
        alpha = (n > 0.52 - 0.02 * i).astype(np.uint8) * 255
        g = np.clip(0.25 + 0.5 * _value_noise(rng, tex_size), 0, 1)
        rgb = np.stack([0.25 * g, 0.75 * g, 0.15 * g], -1)
        px = np.concatenate([(rgb * 255).astype(np.uint8), alpha[..., None]], -1)
        leaf_words.append(b.add_texture(px, TEX_RGBA8_SRGB))
    for i in range(4):
        hgt = _value_noise(rng, tex_size, octaves=(16, 32, 64))
        dx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)
        dy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
        nrm = np.stack([-dx * 6.0, -dy * 6.0, np.ones_like(hgt)], -1)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
        px = np.concatenate([((nrm * 0.5 + 0.5) * 255).astype(np.uint8), np.full((tex_size, tex_size, 1), 255, np.uint8)], -1)
        normal_words.append(b.add_texture(px, TEX_RGBA8_UNORM))
    return base_words, leaf_words, normal_words


def _quads(c, a, bb, uv_scale=None):
    """c, a, bb: (n,3) centre and half axes -> unshared quads. Returns pos (4n,3), idx (6n), uv (4n,2), normal (4n,3), tangent (4n,4)."""
    c, a, bb = (np.asarray(v, np.float32) for v in (c, a, bb))
    n = c.shape[0]
    pos = np.stack([c - a - bb, c + a - bb, c + a + bb, c - a + bb], 1).reshape(-1, 3)
    base = (np.arange(n, dtype=np.uint32) * 4)[:, None]
    idx = (base + np.array([0, 1, 2, 0, 2, 3], np.uint32)[None, :]).reshape(-1)
    nrm = np.cross(a, bb)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    tan = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-20)
    if uv_scale is None:
        uv_scale = np.ones((n, 2), np.float32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)[None, :, :] * np.asarray(uv_scale, np.float32)[:, None, :]
    return (pos, idx, uv.reshape(-1, 2), np.repeat(nrm, 4, 0), np.concatenate([np.repeat(tan, 4, 0), np.ones((4 * n, 1), np.float32)], 1))


def _boxes(c, h, yaw):
    """axis boxes (n) with centre c, half sizes h, yaw about y -> 6n quads."""
    n = c.shape[0]
    cy, sy = np.cos(yaw), np.sin(yaw)
    ex = np.stack([cy, np.zeros(n), -sy], 1).astype(np.float32)
    ez = np.stack([sy, np.zeros(n), cy], 1).astype(np.float32)
    eyv = np.tile(np.array([[0, 1, 0]], np.float32), (n, 1))
    hx, hy, hz = h[:, 0:1], h[:, 1:2], h[:, 2:3]
    C, A, B = [], [], []
    for (nv, hn, av, ha, bv, hb) in ((ex, hx, eyv, hy, ez, hz), (-ex, hx, ez, hz, eyv, hy), (eyv, hy, ez, hz, ex, hx), (-eyv, hy, ex, hx, ez, hz), (ez, hz, ex, hx, eyv, hy), (-ez, hz, eyv, hy, ex, hx)):
        C.append(c + nv * hn); A.append(av * ha); B.append(bv * hb)
    C, A, B = np.concatenate(C), np.concatenate(A), np.concatenate(B)
    # make winding outward: normal = cross(a,b) must point along nv
    nvs = np.concatenate([ex, -ex, eyv, -eyv, ez, -ez])
    flip = (np.cross(A, B) * nvs).sum(1) < 0
    A2 = np.where(flip[:, None], B, A); B2 = np.where(flip[:, None], A, B)
    return C, A2, B2


def sliver_stress(n_strips=400, seed=SEED_BASE + 9):
    """Geometry on which fp32 Moeller-Trumbore is badly conditioned: 120 m x 0.3 mm strips (aspect 4e5, as the roof tiles of bistro_like at scale 1),
    needle triangles, large walls, small axis-aligned quads; with rays that graze them. Used to check that the closest hit does not depend on the BVH
    (the hit definition of pt_scene.h: tri_box_accepts). Returns (scene dict, rays (n,8) float32: origin, tmin, direction, tmax)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    m = b.add_material(make_material(base=(0.7, 0.7, 0.7)))
    b.begin_mesh()
    tq = (np.arange(n_strips) + 0.5) / n_strips
    cq = np.stack([np.full(n_strips, 60.0), 25.0 + tq * 3.6, 8.0 - tq * 8.0], 1)
    aq = np.tile([[60.0, 0, 0]], (n_strips, 1)); hw = 0.6 * 8.0 / 30000
    bq = np.tile([[0, hw * 0.45, -hw]], (n_strips, 1))
    p, i, uv, n, t = _quads(cq, aq, bq); b.add_geometry(p, i, m, uv=uv, normal=n, tangent=t)
    # needles in random orientation
    k = n_strips
    c = rng.uniform((5, 1, 10), (115, 20, 30), (k, 3)); u = rng.normal(size=(k, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    w = np.cross(u, rng.normal(size=(k, 3))); w /= np.linalg.norm(w, axis=1, keepdims=True)
    p, i, uv, n, t = _quads(c, u * rng.uniform(2, 30, (k, 1)), w * rng.uniform(1e-4, 1e-3, (k, 1))); b.add_geometry(p, i, m, uv=uv, normal=n, tangent=t)
    # big walls + small axis-aligned quads
    p, i, uv, n, t = quad((0, 0, 8), (0, 25, 8), (120, 25, 8), (120, 0, 8)); b.add_geometry(p, i, m, uv=uv, normal=n, tangent=t)
    p, i, uv, n, t = quad((0, 0, 0), (0, 0, 40), (120, 0, 40), (120, 0, 0)); b.add_geometry(p, i, m, uv=uv, normal=n, tangent=t)
    c = rng.uniform((5, 1, 10), (115, 20, 30), (k, 3)); s = rng.uniform(0.03, 0.12, k)
    p, i, uv, n, t = _quads(c, np.stack([s, 0 * s, 0 * s], 1), np.stack([0 * s, 0 * s, s], 1)); b.add_geometry(p, i, m, uv=uv, normal=n, tangent=t)
    b.add_instance(b.end_mesh())
    sc = b.finish()
    # rays: towards random points on (and just beyond the ends of) the strips / needles, from far away, many of them grazing
    P = sc["positions"]; I = sc["indices"].reshape(-1, 3)
    nr = 400000
    tri = rng.integers(0, I.shape[0], nr)
    bary = rng.dirichlet((1, 1, 1), nr).astype(np.float32)
    overshoot = np.where(rng.random(nr) < 0.5, rng.uniform(-0.02, 0.02, nr), 0.0)[:, None]      # half the targets lie just outside an edge
    v0, v1, v2 = P[I[tri, 0]], P[I[tri, 1]], P[I[tri, 2]]
    target = v0 * bary[:, 0:1] + v1 * bary[:, 1:2] + v2 * bary[:, 2:3] + (v1 - v0) * overshoot
    nrm = np.cross(v1 - v0, v2 - v0); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    tang = (v1 - v0) / np.maximum(np.linalg.norm(v1 - v0, axis=1, keepdims=True), 1e-30)
    graze = 10.0 ** rng.uniform(-4, 0, nr)[:, None]                                             # |cos| between the ray and the plane normal
    dirv = nrm * graze + tang * np.sqrt(np.maximum(0.0, 1 - graze * graze)) * np.where(rng.random(nr) < 0.5, 1.0, -1.0)[:, None]
    dirv += rng.normal(size=(nr, 3)) * 1e-3 * (rng.random(nr) < 0.5)[:, None]
    dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
    dist = rng.uniform(0.5, 90.0, nr)[:, None]
    org = target - dirv * dist
    rays = np.concatenate([org, np.zeros((nr, 1)), dirv, np.full((nr, 1), 1e15)], 1).astype(np.float32)
    return sc, rays


def bistro_like(scale=1.0, seed=SEED_BASE + 3, tex_size=1024, animated=False):
    """Street canyon 120 x 40 x 25 m: ~2.8 M triangles at scale=1 (60 % long thin facade quads, 25 % alpha-tested foliage cards,
    15 % clutter boxes), 2000 emissive triangles, 64 materials, 32 textures, sky environment. `scale` shrinks the triangle counts
    (tests use small scales; the structure is identical). Returns (scene dict, camera kwargs)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    base_w, leaf_w, normal_w = _make_textures(rng, b, tex_size)
    mats = []
    for i in range(24):       # textured facade materials, 8 of them normal mapped, a few metallic
        mats.append(b.add_material(make_material(base=(1, 1, 1), roughness=float(0.35 + 0.55 * rng.random()), metalness=1.0 if i % 11 == 5 else 0.0, ior=1.5,
                                                 base_tex=base_w[i], normal_tex=normal_w[i % 4] if i < 8 else None)))
    for i in range(8):        # painted
        mats.append(b.add_material(make_material(base=tuple(0.2 + 0.7 * rng.random(3)), roughness=float(0.3 + 0.6 * rng.random()), ior=1.5)))
    leaf_mats = [b.add_material(make_material(base=(1, 1, 1), roughness=0.7, ior=1.4, base_tex=leaf_w[i], alpha_cutoff=0.5, diff_transmission=0.0)) for i in range(4)]
    clutter_mats = [b.add_material(make_material(base=tuple(0.15 + 0.75 * rng.random(3)), roughness=float(0.15 + 0.5 * rng.random()), metalness=1.0 if i % 3 == 0 else 0.0, ior=1.5)) for i in range(8)]
    glass_mats = [b.add_material(make_material(base=(0.9, 0.95, 0.97), roughness=0.0, ior=1.5, transmission=0.9, thin=True)) for i in range(4)]
    if animated:      # C5: two of the glass slots become a nested-dielectric pair (solid glass, priority 2, around a liquid, priority 1)
        b.materials[glass_mats[2]] = make_material(base=(0.97, 0.98, 0.98), roughness=0.0, ior=1.5, transmission=1.0, thin=False, nested_priority=2)
        b.materials[glass_mats[3]] = make_material(base=(0.95, 0.6, 0.3), roughness=0.0, ior=1.33, transmission=1.0, thin=False, nested_priority=1,
                                                  att_color=(0.9, 0.5, 0.2), att_dist=0.5)
    le = 10.0 ** rng.uniform(1.0, 4.0, 16)
    em_mats = [b.add_material(make_material(base=(0.8, 0.8, 0.8), emissive=tuple(float(le[i]) * np.array([1.0, 0.82 + 0.15 * rng.random(), 0.5 + 0.4 * rng.random()])), roughness=1.0, ior=1.5)) for i in range(16)]
    assert len(b.materials) == 64

    L, Wd, Hh = 120.0, 40.0, 25.0
    z0, z1 = 8.0, 32.0                      # facade planes; street between them
    n_facade = max(64, int(840000 * scale)); n_cards = max(64, int(350000 * scale)); n_boxes = max(16, int(35000 * scale)); n_em = max(8, int(1000 * scale))

    # --- static shell: ground, two big walls, end caps (one mesh)
    b.begin_mesh()
    p, i, uv, n, t = quad((0, 0, 0), (0, 0, Wd), (L, 0, Wd), (L, 0, 0), uv_scale=30.0); b.add_geometry(p, i, mats[0], uv=uv, normal=n, tangent=t)
    p, i, uv, n, t = quad((0, 0, z0), (0, Hh, z0), (L, Hh, z0), (L, 0, z0), uv_scale=20.0); b.add_geometry(p, i, mats[1], uv=uv, normal=n, tangent=t)       # normal +z
    p, i, uv, n, t = quad((0, 0, z1), (L, 0, z1), (L, Hh, z1), (0, Hh, z1), uv_scale=20.0); b.add_geometry(p, i, mats[2], uv=uv, normal=n, tangent=t)       # normal -z
    p, i, uv, n, t = quad((L, 0, z0), (L, Hh, z0), (L, Hh, z1), (L, 0, z1), uv_scale=8.0); b.add_geometry(p, i, mats[3], uv=uv, normal=n, tangent=t)        # far cap, normal -x
    shell = b.end_mesh(); b.add_instance(shell)

    # --- facade detail (60 % of the triangles): 8 "window bay" meshes of long thin quads — louvre shutters, balcony railings, window
    # frames, brick courses — instanced over every wall of the canyon and of the side alleys, plus roof tile strips.
    bay_w, bay_h = 3.0, 3.5
    walls = [((0.0, 0.0, z0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0), L), ((L, 0.0, z1), (-1.0, 0.0, 0.0), (0.0, 0.0, -1.0), L)]   # origin, along, normal, length
    for k in range(10):                       # side alleys: two facing walls each, 7 m deep behind the main facades
        xa = 6.0 + k * 12.0
        walls.append(((xa, 0.0, z0), (0.0, 0.0, -1.0), (1.0, 0.0, 0.0), 7.0)); walls.append(((xa + 3.0, 0.0, z0 - 7.0), (0.0, 0.0, 1.0), (-1.0, 0.0, 0.0), 7.0))
        walls.append(((xa + 3.0, 0.0, z1), (0.0, 0.0, 1.0), (-1.0, 0.0, 0.0), 7.0)); walls.append(((xa, 0.0, z1 + 7.0), (0.0, 0.0, -1.0), (1.0, 0.0, 0.0), 7.0))
    n_rows = int(Hh // bay_h)
    n_bays = sum(int(wl[3] // bay_w) for wl in walls) * n_rows
    n_roof = max(8, int(60000 * scale))
    per_bay = max(12, (n_facade - n_roof) // max(1, n_bays))
    n_slats = max(2, int(per_bay * 0.20))      # per shutter leaf, 4 leaves per bay
    n_bars = max(2, int(per_bay * 0.12))
    n_courses = max(2, per_bay - 4 * n_slats - n_bars - 10)
    bay_meshes = []
    for v in range(8):
        C, A, B, M = [], [], [], []
        def add(cq, aq, bq, m):
            C.append(np.atleast_2d(cq)); A.append(np.atleast_2d(aq)); B.append(np.atleast_2d(bq)); M.append(np.full(np.atleast_2d(cq).shape[0], m))
        # local frame: x along the wall, y up, z out of the wall (towards the street). Quads face +z.
        for wx in (0.75, 2.25):                # two windows per bay, each with two shutter leaves
            for leaf in (-1, 1):
                sx = wx + leaf * 0.48
                ys = 0.9 + (np.arange(n_slats) + 0.5) * (1.6 / n_slats)
                tilt = 0.5 + 0.1 * v
                hw = 0.45 * (1.6 / n_slats)
                cq = np.stack([np.full(n_slats, sx), ys, np.full(n_slats, 0.06)], 1)
                aq = np.tile(np.array([[0.2, 0, 0]]), (n_slats, 1))
                bq = np.tile(np.array([[0, hw * math.cos(tilt), hw * math.sin(tilt)]]), (n_slats, 1))
                add(cq, aq, bq, (3 * v + 1) % 32)
            # window frame (4 thin quads) + glass handled below
            add([[wx, 0.9, 0.03], [wx, 2.5, 0.03]], [[0.3, 0, 0]] * 2, [[0, 0.03, 0]] * 2, 24 + v % 8)
            add([[wx - 0.3, 1.7, 0.03], [wx + 0.3, 1.7, 0.03]], [[0.03, 0, 0]] * 2, [[0, 0.8, 0]] * 2, 24 + v % 8)
        xs = 0.15 + (np.arange(n_bars) + 0.5) * (2.7 / n_bars)          # balcony railing bars + rail
        add(np.stack([xs, np.full(n_bars, 0.45), np.full(n_bars, 0.45)], 1), np.tile([[0.4 * 2.7 / n_bars, 0, 0]], (n_bars, 1)), np.tile([[0, 0.45, 0]], (n_bars, 1)), (5 * v + 2) % 32)
        add([[1.5, 0.92, 0.45], [1.5, 0.0, 0.25]], [[1.4, 0, 0], [1.4, 0, 0]], [[0, 0.03, 0], [0, 0, -0.25]], 24 + (v + 3) % 8)
        ys = (np.arange(n_courses) + 0.5) * (bay_h / n_courses)           # brick courses / ledges: full-width thin strips
        add(np.stack([np.full(n_courses, 1.5), ys, np.full(n_courses, 0.012 + 0.004 * (v % 3))], 1), np.tile([[1.5, 0, 0]], (n_courses, 1)),
            np.tile([[0, 0.3 * bay_h / n_courses, 0]], (n_courses, 1)), (7 * v + 3) % 32)
        C, A, B, M = np.concatenate(C).astype(np.float32), np.concatenate(A).astype(np.float32), np.concatenate(B).astype(np.float32), np.concatenate(M)
        b.begin_mesh()
        for m in np.unique(M):
            sel = M == m
            sz = np.stack([np.linalg.norm(A[sel], axis=1) * 2, np.linalg.norm(B[sel], axis=1) * 2], 1)
            p, i, uv, n, t = _quads(C[sel], A[sel], B[sel], sz)
            b.add_geometry(p, i, mats[int(m)], uv=uv, normal=n, tangent=t)
        gq = _quads(np.array([[0.75, 1.7, 0.02], [2.25, 1.7, 0.02]], np.float32), np.array([[0.27, 0, 0]] * 2, np.float32), np.array([[0, 0.77, 0]] * 2, np.float32))
        b.add_geometry(gq[0], gq[1], glass_mats[v % 4], uv=gq[2], normal=gq[3], tangent=gq[4])
        bay_meshes.append(b.end_mesh())
    bay_idx = 0
    for (org, along, nrm, length) in walls:
        along = np.array(along, np.float32); nrm = np.array(nrm, np.float32); org = np.array(org, np.float32)
        for col in range(int(length // bay_w)):
            for row in range(n_rows):
                o = org + along * (col * bay_w) + np.array([0, row * bay_h, 0], np.float32)
                # local (x, y, z) -> world: x*along + y*up + z*normal  (row-major 3x4)
                tr = np.array([along[0], 0, nrm[0], o[0], along[1], 1, nrm[1], o[1], along[2], 0, nrm[2], o[2]], np.float32)
                b.add_instance(bay_meshes[(bay_idx * 5 + row) % 8], tr)
                bay_idx += 1
    # alley back walls + side walls (large quads behind the detail)
    b.begin_mesh()
    for (org, along, nrm, length) in walls[2:]:
        org = np.array(org, np.float32); along = np.array(along, np.float32)
        p0 = org; p1 = org + along * length
        # winding such that cross(p1-p0, up) == normal direction
        q = quad(tuple(p0), tuple(p1), tuple(p1 + np.array([0, Hh, 0], np.float32)), tuple(p0 + np.array([0, Hh, 0], np.float32)), uv_scale=6.0)
        if np.dot(q[3][0], np.array(nrm, np.float32)) < 0:
            q = quad(tuple(p1), tuple(p0), tuple(p0 + np.array([0, Hh, 0], np.float32)), tuple(p1 + np.array([0, Hh, 0], np.float32)), uv_scale=6.0)
        b.add_geometry(q[0], q[1], mats[4 + (int(org[0]) % 4)], uv=q[2], normal=q[3], tangent=q[4])
    b.add_instance(b.end_mesh())
    # roofs: sloped tile strips above both facade rows
    for sidez, sgn in ((z0, -1.0), (z1, 1.0)):
        rows_n = n_roof // 2
        tq = (np.arange(rows_n) + 0.5) / rows_n
        slope = 0.45
        cq = np.stack([np.full(rows_n, L / 2), Hh + tq * 8.0 * slope, sidez + sgn * tq * 8.0], 1)
        aq = np.tile([[L / 2, 0, 0]], (rows_n, 1)).astype(np.float32)
        hw = 0.6 * 8.0 / rows_n
        bq = np.tile([[0, hw * slope, sgn * hw]], (rows_n, 1)).astype(np.float32)
        if sgn > 0:
            aq = -aq                              # keep the normal pointing up/outwards
        p, i, uv, n, t = _quads(cq, aq, bq, np.tile([[40.0, 0.2]], (rows_n, 1)))
        b.begin_mesh(); b.add_geometry(p, i, mats[10 + (0 if sgn < 0 else 1)], uv=uv, normal=n, tangent=t); b.add_instance(b.end_mesh())

    # --- foliage: 8 tree meshes of alpha-tested cards, instanced along the pavements
    n_tree_meshes, n_tree_inst = 8, 25
    cards_per_tree = max(8, n_cards // (n_tree_meshes * n_tree_inst))
    tree_meshes = []
    for tm in range(n_tree_meshes):
        r = rng.random(cards_per_tree) ** (1 / 3) * 2.2
        dirv = rng.normal(size=(cards_per_tree, 3)); dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
        cc = dirv * r[:, None] * np.array([1.0, 0.8, 1.0]) + np.array([0, 4.6, 0])
        ua = rng.normal(size=(cards_per_tree, 3)); ua /= np.linalg.norm(ua, axis=1, keepdims=True)
        ub = np.cross(ua, rng.normal(size=(cards_per_tree, 3))); ub /= np.maximum(np.linalg.norm(ub, axis=1, keepdims=True), 1e-9)
        sz = rng.uniform(0.05, 0.12, cards_per_tree)[:, None]
        p, i, uv, n, t = _quads(cc, ua * sz, ub * sz)
        b.begin_mesh()
        b.add_geometry(p, i, leaf_mats[tm % 4], uv=uv, normal=n, tangent=t, geom_flags=GEOMF_ALPHA_TESTED)
        # trunk: one thin box
        tc, ta, tb = _boxes(np.array([[0, 2.0, 0]], np.float32), np.array([[0.12, 2.0, 0.12]], np.float32), np.zeros(1))
        p, i, uv, n, t = _quads(tc, ta, tb)
        b.add_geometry(p, i, mats[24 + tm % 8], uv=uv, normal=n, tangent=t)
        tree_meshes.append(b.end_mesh())
    tree_instances = []
    for k in range(n_tree_meshes * n_tree_inst):
        x = 2.0 + (L - 4.0) * ((k * 0.61803398875) % 1.0); z = (z0 + 2.2) if (k % 2 == 0) else (z1 - 2.2)
        s = float(0.8 + 0.5 * rng.random())
        tree_instances.append(len(b.instances))
        b.add_instance(tree_meshes[k % n_tree_meshes], trs((x, 0.0, z), rot_y=float(rng.uniform(0, 6.283)), scale=(s, s, s)))

    # --- clutter: boxes (crates, cars, kiosks) on the street, 40 rigid groups (animated in C5)
    n_groups = 40
    per_group = max(1, n_boxes // n_groups)
    clutter_instances = []
    for gidx in range(n_groups):
        bc = np.stack([rng.uniform(-4.0, 4.0, per_group), np.zeros(per_group), rng.uniform(-4.0, 4.0, per_group)], 1).astype(np.float32)
        bh = np.stack([rng.uniform(0.03, 0.25, per_group), rng.uniform(0.03, 0.4, per_group), rng.uniform(0.03, 0.25, per_group)], 1).astype(np.float32)
        bc[:, 1] = bh[:, 1] + rng.uniform(0, 1.2, per_group) * (rng.random(per_group) < 0.3)
        C, A, B = _boxes(bc, bh, rng.uniform(0, 6.283, per_group))
        msel = rng.integers(0, 8, C.shape[0] // 6); msel6 = np.tile(msel, 6)
        b.begin_mesh()
        for m in np.unique(msel6):
            s = msel6 == m
            p, i, uv, n, t = _quads(C[s], A[s], B[s])
            b.add_geometry(p, i, clutter_mats[int(m)], uv=uv, normal=n, tangent=t)
        mesh = b.end_mesh()
        clutter_instances.append(len(b.instances))
        b.add_instance(mesh, trs((float(rng.uniform(6, L - 6)), 0.0, float(rng.uniform(z0 + 6.0, z1 - 6.0))), rot_y=float(rng.uniform(0, 6.283))))

    # --- emissive: lamps and string lights (2 triangles each), 16 emissive materials with log-uniform radiance
    ex = rng.uniform(1, L - 1, n_em); ey = rng.uniform(2.5, 7.0, n_em); ez = rng.uniform(z0 + 0.6, z1 - 0.6, n_em)
    es = rng.uniform(0.03, 0.12, n_em)
    ec = np.stack([ex, ey, ez], 1)
    ea = np.stack([es, np.zeros(n_em), np.zeros(n_em)], 1); eb = np.stack([np.zeros(n_em), np.zeros(n_em), -es], 1)    # cross(a,b) = +? (es,0,0)x(0,0,-es) = (0*(-es)-0*0, 0*0-es*(-es), 0) = (0, es^2, 0) -> flip to face down
    eb = -eb; ea, eb = eb, ea
    emsel = rng.integers(0, 16, n_em)
    b.begin_mesh()
    for m in np.unique(emsel):
        s = emsel == m
        p, i, uv, n, t = _quads(ec[s], ea[s], eb[s])
        b.add_geometry(p, i, em_mats[int(m)], uv=uv, normal=n, tangent=t)
    b.add_instance(b.end_mesh())

    deform = None
    if animated:
        # --- C5: 20 nested-dielectric props (a glass box holding a liquid box) along the street
        cp, ci, cuv, cn, ct = unit_cube()
        b.begin_mesh()
        b.add_geometry(cp, ci, glass_mats[2], uv=cuv, normal=cn, tangent=ct)
        b.add_geometry(cp * np.array([0.8, 0.7, 0.8], np.float32) + np.array([0, -0.1, 0], np.float32), ci, glass_mats[3], uv=cuv, normal=cn, tangent=ct)
        prop_mesh = b.end_mesh()
        for k in range(20):
            sz = float(rng.uniform(0.5, 1.2))
            b.add_instance(prop_mesh, trs((float(rng.uniform(8, L - 8)), 0.5 * sz + 0.01, float(rng.uniform(z0 + 3.0, z1 - 3.0))), rot_y=float(rng.uniform(0, 6.283)), scale=(sz, sz, sz)))
        # --- C5: one deforming mesh of ~50 k triangles (a banner across the street, displaced every frame -> vertex refit)
        ng = max(4, int(round(158 * math.sqrt(max(scale, 1e-4)))))
        gu, gv = np.meshgrid(np.linspace(0, 1, ng + 1, dtype=np.float32), np.linspace(0, 1, ng + 1, dtype=np.float32), indexing="ij")
        pos = np.stack([30.0 + 0.0 * gu, 6.0 + 6.0 * gv, z0 + 1.0 + (z1 - z0 - 2.0) * gu], -1).reshape(-1, 3).astype(np.float32)
        ii = (np.arange(ng)[:, None] * (ng + 1) + np.arange(ng)[None, :]).reshape(-1)
        idx = np.stack([ii, ii + 1, ii + ng + 2, ii, ii + ng + 2, ii + ng + 1], 1).reshape(-1).astype(np.uint32)
        nrm = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (pos.shape[0], 1))
        tan = np.tile(np.array([[0.0, 0.0, 1.0, 1.0]], np.float32), (pos.shape[0], 1))
        b.begin_mesh()
        first_vertex = b.nv
        b.add_geometry(pos, idx, mats[24], uv=np.stack([gu, gv], -1).reshape(-1, 2) * 4.0, normal=nrm, tangent=tan)
        b.add_instance(b.end_mesh())
        deform = dict(first_vertex=first_vertex, rest=pos.copy(), u=gu.reshape(-1).copy(), v=gv.reshape(-1).copy())

    b.set_environment(sky_equirect(sun_dir=(0.25, 0.75, -0.35), sun_radiance=2e4), color_multiplier=(1, 1, 1))
    sc = b.finish()
    sc["anim"] = dict(clutter_instances=clutter_instances, tree_instances=tree_instances, deform=deform)
    cam = dict(pos=(4.0, 1.7, 20.0), direction=(1.0, 0.12, 0.05), up=(0, 1, 0), fov_y=math.radians(60.0), near_z=0.05, far_z=1000.0, focal_distance=10.0)
    return sc, cam


def icosphere_mesh(subdivisions):
    """(float32 positions on the unit sphere, uint32 triangle index array (n, 3)) of an icosphere with 20 * 4^subdivisions triangles and shared vertices"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    V = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]; F = list(f)
    for _ in range(subdivisions):
        mid = {}; F2 = []
        def m(a, b):
            k = (a, b) if a < b else (b, a)
            if k not in mid:
                p = V[a] + V[b]; V.append(p / np.linalg.norm(p)); mid[k] = len(V) - 1
            return mid[k]
        for a, b, c in F:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            F2 += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        F = F2
    return np.array(V, np.float32), np.array(F, np.uint32)


def closed_icosphere(subdivisions=5, transform=None):
    """A closed, indexed (shared-vertex) triangle mesh: an icosphere of radius 1 with 20 * 4^subdivisions triangles, one instance under `transform` (default: a rotation about y
    with a non-uniform scale and an offset, so that the world-space vertices carry rounding). For the watertightness measurement (tests/test_gpu_watertight.py): from inside,
    every ray must hit. Returns (scene dict, world-space float64 vertices as the float32 transform gives them, triangle index array)."""
    P, I = icosphere_mesh(subdivisions)
    b = SceneBuilder()
    mat = b.add_material(make_material(base=(0.7, 0.7, 0.7), roughness=1.0))
    b.begin_mesh(); b.add_geometry(P, I.reshape(-1), mat, normal=P); mesh = b.end_mesh()
    T = trs((0.37, -0.21, 0.53), rot_y=0.731, scale=(1.7, 0.9, 1.3)) if transform is None else np.asarray(transform, np.float32)
    b.add_instance(mesh, T)
    M = T.reshape(3, 4).astype(np.float64)
    W = P.astype(np.float64) @ M[:, :3].T + M[:, 3]
    return b.finish(), W, I


def animate_positions(sc, t):
    """C5 deforming mesh ("1 skinned-like mesh of 50 k tris displaced per frame"): the banner ripples along x; same
    topology, so the library refits. Returns the full position array for pt_animate."""
    d = sc["anim"]["deform"]
    pos = sc["positions"].copy()
    p = d["rest"].copy()
    p[:, 0] += (0.35 * np.sin(6.0 * d["u"] + 1.3 * t) * np.sin(3.0 * d["v"] + 0.7 * t)).astype(np.float32)
    pos[d["first_vertex"]:d["first_vertex"] + p.shape[0]] = p
    return pos


def animated_vertex_ranges(sc):
    """The vertices animate_positions moves, as (first, count) pairs for pt_animate_ranges: the banner mesh."""
    d = sc["anim"]["deform"]
    return [(int(d["first_vertex"]), int(d["rest"].shape[0]))]


def animate_instances(sc, t):
    """Rigid keyframed motion of the clutter groups (game props = TLAS instance transforms only)."""
    inst = sc["instances"].copy()
    for k, idx in enumerate(sc["anim"]["clutter_instances"]):
        tr = inst["transform"][idx].copy()
        tr[3] += 0.8 * math.sin(0.7 * t + k)
        tr[11] += 0.5 * math.cos(0.5 * t + 1.3 * k)
        inst["transform"][idx] = tr
    return inst


def previous_pose(sc, seed=0):
    """A previous frame for any scene (tests of the motion vectors' object term): every other instance shifted and turned a little about y, and the vertices of every third
    geometry displaced by a smooth wave — (instances, positions) shaped like sc["instances"] / sc["positions"], for pt_set_previous_pose / Oracle.set_previous_pose."""
    rng = np.random.default_rng(1000 + seed)
    inst = sc["instances"].copy()
    for i in range(0, len(inst), 2):
        T = inst["transform"][i].reshape(3, 4).astype(np.float64)
        a = float(rng.uniform(-0.06, 0.06)); c, s_ = math.cos(a), math.sin(a)
        R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
        T[:, :3] = R @ T[:, :3]; T[:, 3] = R @ T[:, 3] + rng.uniform(-0.05, 0.05, 3)
        inst["transform"][i] = T.astype(np.float32).reshape(inst["transform"][i].shape)
    pos = sc["positions"].copy()
    for g in range(0, len(sc["geometries"]), 3):
        first, n = int(sc["geometries"]["vertexOffset"][g]), int(sc["geometries"]["numVertices"][g])
        p = pos[first:first + n]
        p[:, 1] += (0.02 * np.sin(7.0 * p[:, 0] + 3.0 * p[:, 2] + g)).astype(np.float32)
    return inst, pos


def procedural_sky_textures(seed=SEED_BASE + 21, clouds=(64, 64, 16)):
    """Synthetic stand-ins for the four look-up textures of the procedural sky (the reference loads q2rtx_env/{transmittance,inscatter,irradiance}_earth.dds and
    clouds.dds, which no checkout carries): the sizes precomputed_sky.hlsli expects — transmittance 256x64, in-scatter (8*32)x128x32, irradiance 64x16 — smooth,
    positive and of plausible magnitude, and a tileable cloud volume. float32 RGBA arrays: (h, w, 4) or (d, h, w, 4)."""
    rng = np.random.default_rng(seed)
    u = (np.arange(256, dtype=np.float64) + 0.5) / 256.0; v = (np.arange(64, dtype=np.float64) + 0.5) / 64.0
    depth = (0.15 + 3.0 * (1.0 - u[None, :]) ** 2) * (1.0 - 0.6 * v[:, None])                       # optical depth: long near the horizon (u -> 0), short at altitude
    tr = np.exp(-depth[..., None] * np.array([0.35, 0.8, 1.9])[None, None, :])
    transmittance = np.concatenate([tr, np.ones((64, 256, 1))], -1).astype(np.float32)
    x = (np.arange(256, dtype=np.float64) + 0.5) / 256.0; y = (np.arange(128, dtype=np.float64) + 0.5) / 128.0; z = (np.arange(32, dtype=np.float64) + 0.5) / 32.0
    X, Y, Z = x[None, None, :], y[None, :, None], z[:, None, None]
    mus = (X * 8.0) % 1.0                                                                            # eight nu slices side by side, mu_s inside each
    horizon = np.exp(-((Y - 0.5) / 0.08) ** 2)
    ray = 60.0 * (0.02 + 0.25 * horizon + 0.05 * Y) * (0.2 + 0.8 * mus) * (1.0 - 0.7 * Z)
    sc = ray[..., None] * np.array([0.18, 0.42, 1.0])[None, None, None, :]
    mie = 60.0 * (0.01 + 0.3 * horizon) * (0.2 + 0.8 * mus) * (1.0 - 0.7 * Z) * 0.18                         # alpha: the red channel of the single Mie scattering
    scattering = np.concatenate([sc, mie[..., None] * np.ones((32, 128, 256, 1))], -1).astype(np.float32)
    iu = (np.arange(64, dtype=np.float64) + 0.5) / 64.0; iv = (np.arange(16, dtype=np.float64) + 0.5) / 16.0
    irr = 40.0 * (0.02 + 0.3 * np.clip(iu[None, :] * 2.0 - 0.8, 0.0, 1.0)) * (1.0 + 0.2 * iv[:, None])
    irradiance = np.concatenate([irr[..., None] * np.array([0.7, 0.85, 1.0])[None, None, :], np.ones((16, 64, 1))], -1).astype(np.float32)
    cw, ch, cd = clouds
    cx = np.arange(cw)[None, None, :] / cw; cy = np.arange(ch)[None, :, None] / ch; cz = np.arange(cd)[:, None, None] / cd
    vol = np.zeros((cd, ch, cw, 4))
    for c in range(2):                                                                               # tileable: integer frequencies only
        acc = np.zeros((cd, ch, cw))
        for k in range(6):
            fx, fy, fz = rng.integers(1, 6, 3); ph = rng.uniform(0, 2 * np.pi, 3)
            acc += np.sin(2 * np.pi * fx * cx + ph[0]) * np.sin(2 * np.pi * fy * cy + ph[1]) * np.cos(2 * np.pi * fz * cz + ph[2]) / (1.0 + k)
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        vol[..., c] = acc * np.sin(np.pi * np.clip(cz + 0.5 / cd, 0, 1)) if c == 0 else acc              # density falls off towards the layer's bottom and top
    vol[..., 3] = 1.0
    return [transmittance, scattering, irradiance, vol.astype(np.float32)]


def synthetic_local_light_tables(num_lights, width, height, seed=1, hot=6, jitter=(0, 0)):
    """A stand-in for what LightsBaker's feedback passes write (NEE-AT local samplers, LightsBaker.hlsl:1129-1855 — not built here): for every screen tile of
    8 x 8 pixels, 128 proxies drawn from a few tile-specific "hot" lights plus a uniform tail, sorted by light index and packed as light << 9 | (count - 1)
    (LightingTypes.hlsli:172-175). Returns uint32 [tilesY, tilesX, 128]; the resolution covers the frame for any tile jitter below 8."""
    import numpy as np
    rng = np.random.default_rng(seed)
    tx, ty = (width + 7 + 7) // 8, (height + 7 + 7) // 8
    out = np.zeros((ty, tx, 128), np.uint32)
    for y in range(ty):
        for x in range(tx):
            hots = rng.integers(0, num_lights, size=hot)
            picks = np.where(rng.random(128) < 0.75, hots[rng.integers(0, hot, size=128)], rng.integers(0, num_lights, size=128))
            picks = np.sort(picks.astype(np.uint32))
            lights, counts = np.unique(picks, return_counts=True)
            cnt = dict(zip(lights.tolist(), counts.tolist()))
            out[y, x] = np.array([(int(l) << 9) | (cnt[int(l)] - 1) for l in picks], np.uint32)
    return out


def view_projection(width, height, pos, direction, up, fov_y, near_z=0.01, **_unused):
    """A view-projection matrix of the kind Donut's PlanarView::GetViewProjectionMatrix returns (row vectors: clip = [x y z 1] @ M; reverse-Z, infinite far plane), for the
    camera arguments bridge_camera takes: what a host passes to pt_set_light_importance_boost. float32 [4, 4]."""
    f = np.asarray(direction, np.float64); f = f / np.linalg.norm(f)
    r = np.cross(np.asarray(up, np.float64), f); r = r / np.linalg.norm(r)
    u = np.cross(f, r); p = np.asarray(pos, np.float64)
    view = np.array([[r[0], u[0], f[0], 0.0], [r[1], u[1], f[1], 0.0], [r[2], u[2], f[2], 0.0], [-r.dot(p), -u.dot(p), -f.dot(p), 1.0]])
    ys = 1.0 / np.tan(0.5 * fov_y); xs = ys * height / width
    proj = np.array([[xs, 0, 0, 0], [0, ys, 0, 0], [0, 0, 0, 1.0], [0, 0, near_z, 0]])
    return (view @ proj).astype(np.float32)


# ---- stable planes (realtime mode's pre-pass): the host record (include/mi355pt.h PtStablePlanesParams) and the 80-byte plane record (StablePlanes.hlsli:41-76)
STABLE_PLANES_PARAMS_DTYPE = np.dtype([("activeStablePlaneCount", "<u4"), ("maxStablePlaneVertexDepth", "<u4"), ("allowPrimarySurfaceReplacement", "<u4"), ("subSampleCount", "<u4"),
                                       ("matWorldToClip", "<f4", 16), ("matWorldToClipNoOffset", "<f4", 16), ("prevMatWorldToClipNoOffset", "<f4", 16), ("clipToWindowScale", "<f4", 2), ("_pad", "<f4", 2)])
STABLE_PLANE_DTYPE = np.dtype([("RayOrigin", "<f4", 3), ("LastRayTCurrent", "<f4"), ("RayDir", "<f4", 3), ("SceneLength", "<f4"), ("PackedThpAndMVs", "<u4", 3), ("VertexIndexAndRoughness", "<u4"),
                               ("DenoiserPackedBSDFEstimate", "<u4", 3), ("PackedNormal", "<u4"), ("PackedNoisyRadianceAndSpecAvg", "<u4", 2), ("FlagsAndVertexIndex", "<u4"), ("PackedCounters", "<u4")])
assert STABLE_PLANES_PARAMS_DTYPE.itemsize == 224 and STABLE_PLANE_DTYPE.itemsize == 80


def stable_planes_params(width, height, world_to_clip, prev_world_to_clip=None, active_planes=3, max_vertex_depth=14, allow_psr=True, sub_samples=1):
    """The per-frame record of the stable-plane passes (Sample.cpp:1509-1540): Donut's matWorldToClip (the view-projection with the jitter offset; here the same matrix as the
    un-jittered one), last frame's matWorldToClipNoOffset (None: the camera did not move) and clipToWindowScale = (w / 2, -h / 2)."""
    p = np.zeros((), STABLE_PLANES_PARAMS_DTYPE)
    p["activeStablePlaneCount"] = active_planes; p["maxStablePlaneVertexDepth"] = max_vertex_depth; p["allowPrimarySurfaceReplacement"] = 1 if allow_psr else 0; p["subSampleCount"] = sub_samples
    m = np.asarray(world_to_clip, np.float32).reshape(16)
    p["matWorldToClip"] = m; p["matWorldToClipNoOffset"] = m
    p["prevMatWorldToClipNoOffset"] = m if prev_world_to_clip is None else np.asarray(prev_world_to_clip, np.float32).reshape(16)
    p["clipToWindowScale"] = (0.5 * width, -0.5 * height)
    return p


def stable_planes_address(x, y, plane, width, height):
    """GenericTSPixelToAddress (Utils.hlsli:337-356): index of pixel (x, y)'s record of `plane` in the stable-plane buffer (8 x 8 tiles, Morton order inside a tile)."""
    line = ((width + 7) // 8) * 8; plane_stride = line * ((height + 7) // 8) * 8
    xi, yi = x % 8, y % 8
    def spread(v): v = (v | (v << 2)) & 0x33; return (v | (v << 1)) & 0x55
    return (x - xi) * 8 + (y - yi) * line + (spread(xi) | (spread(yi) << 1)) + plane * plane_stride


MF_PSDExclude, MF_PSDBlockMVsB0, MF_PSDBlockMVsB1, MF_PSDDominantDeltaLobeP1Shift = 0x400, 1 << 13, 1 << 14, 24


def stable_planes_zoo(auto_mv=None):
    """auto_mv (None: the scene of the committed fixtures, unchanged): adds curved surfaces whose materials use the AUTOMATIC motion-vector block types of Bridge::loadSurface
    (PathTracerBridgeDonut.hlsli:704-716) — a gently curved mirror (AutoLow: the pixel curvature straddles the jittered threshold), a nearly flat thin glass pane whose 8-bit vertex
    normals make some triangles flat and others stepped (AutoHigh), a small solid glass sphere (AutoLow, always above the threshold). auto_mv = "off" / "full" builds the same
    geometry with block type 0 / 3 on all three (what the automatic types must differ from).
    A small scene that exercises every branch of the stable-plane build pass: an open box under the sky with a perfect mirror (primary surface replacement), solid glass
    (a transmission and a reflection plane, nested priorities, volume absorption), a thin pane whose dominant lobe is transmission, a pane that blocks motion vectors at its surface,
    a mirror excluded from the decomposition, a rough metal, an emitter that is also seen through the delta paths. Returns (scene dict, camera kwargs)."""
    b = SceneBuilder()
    floor = b.add_material(make_material(base=(0.6, 0.6, 0.55), roughness=0.7))
    wall = b.add_material(make_material(base=(0.2, 0.45, 0.7), roughness=1.0))
    mirror = b.add_material(make_material(base=(0.9, 0.9, 0.95), roughness=0.0, metalness=1.0))
    mirror_ex = b.add_material(make_material(base=(0.9, 0.6, 0.3), roughness=0.0, metalness=1.0, flags=MF_PSDExclude))
    rough_metal = b.add_material(make_material(base=(0.95, 0.75, 0.35), roughness=0.35, metalness=1.0))
    glass = b.add_material(make_material(base=(0.95, 0.97, 0.95), roughness=0.0, transmission=1.0, thin=False, nested_priority=2, att_color=(0.7, 0.95, 0.8), att_dist=0.3,
                                         flags=2 << MF_PSDDominantDeltaLobeP1Shift))
    glass_in = b.add_material(make_material(base=(0.97, 0.9, 0.9), roughness=0.0, ior=1.33, transmission=1.0, thin=False, nested_priority=4, att_color=(0.95, 0.6, 0.6), att_dist=0.2,
                                            flags=1 << MF_PSDDominantDeltaLobeP1Shift))
    pane = b.add_material(make_material(base=(0.9, 0.95, 1.0), roughness=0.0, transmission=0.9, thin=True, flags=1 << MF_PSDDominantDeltaLobeP1Shift))
    pane_block = b.add_material(make_material(base=(1.0, 0.9, 0.9), roughness=0.0, transmission=1.0, thin=True, flags=MF_PSDBlockMVsB0 | MF_PSDBlockMVsB1 | (1 << MF_PSDDominantDeltaLobeP1Shift)))
    lamp = b.add_material(make_material(base=(0.8, 0.8, 0.8), emissive=(9.0, 7.0, 3.0), roughness=1.0))
    coated = b.add_material(make_material(base=(0.1, 0.5, 0.2), roughness=0.0, ior=1.5))      # a dielectric with a delta coat over a diffuse base: delta + non-delta lobes
    b.begin_mesh()
    for pts, mat in ((((-1, 0, -1), (-1, 0, 2), (1, 0, 2), (1, 0, -1)), floor), (((-1, 0, 2), (-1, 1.2, 2), (1, 1.2, 2), (1, 0, 2)), mirror),
                     (((-1, 0, -1), (-1, 1.2, -1), (-1, 1.2, 2), (-1, 0, 2)), wall), (((1, 0, -1), (1, 0, 2), (1, 1.2, 2), (1, 1.2, -1)), mirror_ex),
                     (((-0.3, 1.19, 0.6), (0.3, 1.19, 0.6), (0.3, 1.19, 1.2), (-0.3, 1.19, 1.2)), lamp),
                     (((-0.95, 0.05, 0.2), (-0.95, 0.9, 0.2), (-0.45, 0.9, 0.5), (-0.45, 0.05, 0.5)), pane),
                     (((0.45, 0.05, 0.5), (0.45, 0.9, 0.5), (0.95, 0.9, 0.2), (0.95, 0.05, 0.2)), pane_block),
                     (((-0.25, 0.001, 0.1), (-0.25, 0.001, 0.5), (0.25, 0.001, 0.5), (0.25, 0.001, 0.1)), coated)):
        p, i, uv, n, t = quad(*pts)
        b.add_geometry(p, i, mat, uv=uv, normal=n, tangent=t)
    room = b.end_mesh(); b.add_instance(room)
    cp, ci, cuv, cn, ct = unit_cube()
    for mat, xf in ((glass, trs((-0.35, 0.2501, 1.1), rot_y=0.4, scale=(0.25, 0.25, 0.25))), (glass_in, trs((-0.35, 0.2501, 1.1), rot_y=0.1, scale=(0.12, 0.12, 0.12))),
                    (rough_metal, trs((0.4, 0.2001, 1.3), rot_y=-0.5, scale=(0.2, 0.2, 0.2))), (mirror, trs((0.1, 0.1501, 0.75), rot_y=0.8, scale=(0.15, 0.15, 0.15)))):
        b.begin_mesh(); b.add_geometry(cp, ci, mat, uv=cuv, normal=cn, tangent=ct); m = b.end_mesh(); b.add_instance(m, xf)
    if auto_mv is not None:
        lo, hi = {"auto": (MF_PSDBlockMVsB0, MF_PSDBlockMVsB1), "off": (0, 0), "full": (MF_PSDBlockMVsB0 | MF_PSDBlockMVsB1,) * 2}[auto_mv]
        mirror_lo = b.add_material(make_material(base=(0.92, 0.92, 0.85), roughness=0.0, metalness=1.0, flags=lo))
        pane_hi = b.add_material(make_material(base=(0.95, 1.0, 0.95), roughness=0.0, transmission=1.0, thin=True, flags=hi | (1 << MF_PSDDominantDeltaLobeP1Shift)))
        glass_lo = b.add_material(make_material(base=(0.9, 0.95, 1.0), roughness=0.0, transmission=1.0, thin=False, nested_priority=3, att_color=(0.8, 0.9, 1.0), att_dist=0.4,
                                                flags=lo | (2 << MF_PSDDominantDeltaLobeP1Shift)))
        for (pp, ii, uu, nn, tt), mat, xf in ((sphere_patch(1.0, 0.3, 10), mirror_lo, trs((0.0, 0.62, 1.6), rot_y=0.15)), (sphere_patch(8.0, 0.22, 14), pane_hi, trs((0.05, 0.42, 0.25), rot_y=-0.2))):
            b.begin_mesh(); b.add_geometry(pp, ii, mat, uv=uu, normal=nn, tangent=tt); m = b.end_mesh(); b.add_instance(m, xf)
        sp, si = icosphere_mesh(2)
        b.begin_mesh(); b.add_geometry(sp, si.reshape(-1), glass_lo, normal=sp); m = b.end_mesh(); b.add_instance(m, trs((0.55, 0.2, 0.9), scale=(0.12, 0.12, 0.12)))
    b.set_environment(sky_equirect(256, 128), color_multiplier=(1, 1, 1))
    cam = dict(pos=(0.0, 0.55, -0.9), direction=(0.0, -0.1, 1.0), up=(0, 1, 0), fov_y=math.radians(55.0), near_z=0.01, far_z=100.0, focal_distance=1.0)
    return b.finish(), cam
