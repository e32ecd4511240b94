// ORACLE (test infrastructure only) — scene data contract, textures, environment map, BVH + intersection.
//
// Data contract (mirrors the buffers the reference binds at Rtxpt/Sample.cpp:2319-2384):
//   * PTMaterialData 128 B        Rtxpt/Shaders/PathTracer/Materials/MaterialPT.h:45-77 (flags :24-42)
//   * SubInstanceData 32 B        Rtxpt/Shaders/SubInstanceData.h:23-46
//   * GeometryData / InstanceData Donut structs (absent): index u32, position float3, uv float2,
//                                 normal/tangent SNORM8x4 — fetched as in PathTracerBridgeDonut.hlsli:152-256
// What the DXR driver does (BLAS/TLAS build + traversal, Sample.cpp:1061-1079,1200-1240; BridgeDonut:993-1055) is replaced
// by an explicit binned-SAH BVH2 over world-space triangles with a Moeller-Trumbore test. Any exact BVH returns the same
// closest hit; ties on t are broken towards the lower global primitive index so that the result is traversal-order free.
// Texture filtering (fixed-function in the reference) is restated as explicit wrap-mode bilinear/trilinear on a box-filtered
// mip chain; the environment is sampled from the source lat-long image (the EnvMapBaker cube conversion is out of scope).
#pragma once
#include <functional>
#include <cstdio>
#include "lights.h"
#include "neeat.h"
#include "envcube.h"
#include "sky.h"
#include <vector>
#include <algorithm>

namespace ptref {

// ---- MaterialPT.h:24-42
enum : uint {
    PTMaterialFlags_UseSpecularGlossModel = 0x1, PTMaterialFlags_UseMetalRoughOrSpecularTexture = 0x4,
    PTMaterialFlags_UseBaseOrDiffuseTexture = 0x8, PTMaterialFlags_UseEmissiveTexture = 0x10, PTMaterialFlags_UseNormalTexture = 0x20,
    PTMaterialFlags_UseTransmissionTexture = 0x80, PTMaterialFlags_MetalnessInRedChannel = 0x100, PTMaterialFlags_ThinSurface = 0x200,
    PTMaterialFlags_PSDExclude = 0x400, PTMaterialFlags_EnableAsAnalyticLightProxy = 0x800, PTMaterialFlags_IgnoreMeshTangentSpace = 1u << 12,
    PTMaterialFlags_NestedPriorityMask = 0xF0000000u, PTMaterialFlags_NestedPriorityShift = 28,
};
// ---- MaterialPT.h:45-77 (128 bytes)
struct PTMaterialData {
    float3 BaseOrDiffuseColor; uint Flags;
    float3 SpecularColor; int _padding0;
    float3 EmissiveColor; float ShadowNoLFadeout;
    float Opacity, Roughness, Metalness, NormalTextureScale;
    float _padding1, AlphaCutoff, TransmissionFactor; uint BaseOrDiffuseTextureIndex;
    uint MetalRoughOrSpecularTextureIndex, EmissiveTextureIndex, NormalTextureIndex, OcclusionTextureIndex;
    uint TransmissionTextureIndex; float IoR, ThicknessFactor, DiffuseTransmissionFactor;
    float3 AttenuationColor; float AttenuationDistance;
};
static_assert(sizeof(PTMaterialData) == 128, "PTMaterialData must be 128 bytes");

// ---- SubInstanceData.h:23-46
struct SubInstanceData {
    enum : uint { Flags_AlphaTested = 1u << 16, Flags_ExcludeFromNEE = 1u << 17, Flags_AlphaOffsetOffset = 24 };
    uint FlagsAndAlphaInfo, GlobalGeometryIndex_PTMaterialDataIndex, EmissiveLightMappingOffset, AnalyticProxyLightIndex;
    // SUBINSTANCEDATA_EXTENDED words (SubInstanceData.h:37-42): here element (not byte) offsets into the shared streams
    uint IndexBufferIndex_VertexBufferIndex, IndexOffset, TexCoord1Offset, padding0;
    float AlphaCutoff() const { return (float)(FlagsAndAlphaInfo >> Flags_AlphaOffsetOffset) / 255.0f; }
    uint AlphaTextureIndex() const { return FlagsAndAlphaInfo & 0xFFFFu; }
};

// geometry = one glTF primitive; vertex streams share one vertex base
struct GeometryDesc {
    uint indexOffset, numIndices;      // into the u32 index array
    uint vertexOffset, numVertices;    // into the vertex streams
    uint flags;                        // bit0 has uv, bit1 has normals, bit2 has tangents
    uint materialIndex;
    uint geomFlags;                    // bit0 alpha-tested, bit1 exclude-from-NEE (AccelerationStructureUtil.h:35-104)
    uint _pad;
};
enum : uint { GEOM_HAS_UV = 1, GEOM_HAS_NORMAL = 2, GEOM_HAS_TANGENT = 4, GEOMF_ALPHA_TESTED = 1, GEOMF_EXCLUDE_FROM_NEE = 2 };
struct MeshDesc { uint firstGeometry, numGeometries; };
struct InstanceDesc { float3x4 transform; uint meshIndex; uint analyticProxyLight; uint _pad[2]; };      // analyticProxyLight: 0 = none, k + 1 = stands in for analytic light k

// ---- textures: RGBA float texels, full mip chain (box filter), wrap addressing
struct Texture {
    uint w, h, mipLevels;
    std::vector<std::vector<float4> > mips;
    const float4& texel(uint mip, int x, int y) const {
        uint mw = std::max(1u, w >> mip), mh = std::max(1u, h >> mip);
        int xi = x % (int)mw; if (xi < 0) xi += mw;
        int yi = y % (int)mh; if (yi < 0) yi += mh;
        return mips[mip][(size_t)yi * mw + xi];
    }
};
static inline float srgb_to_linear(float c) { return (c <= 0.04045f) ? c / 12.92f : dm_pow((c + 0.055f) / 1.055f, 2.4f); }
static inline void build_mips(Texture& t) {
    uint lv = 1; { uint m = std::max(t.w, t.h); while (m > 1) { m >>= 1; lv++; } }
    t.mipLevels = lv; t.mips.resize(lv);
    for (uint l = 1; l < lv; l++) {
        uint pw = std::max(1u, t.w >> (l - 1)), ph = std::max(1u, t.h >> (l - 1));
        uint mw = std::max(1u, t.w >> l), mh = std::max(1u, t.h >> l);
        t.mips[l].resize((size_t)mw * mh);
        for (uint y = 0; y < mh; y++) for (uint x = 0; x < mw; x++) {
            uint x0 = std::min(2 * x, pw - 1), x1 = std::min(2 * x + 1, pw - 1), y0 = std::min(2 * y, ph - 1), y1 = std::min(2 * y + 1, ph - 1);
            const std::vector<float4>& p = t.mips[l - 1];
            float4 s = (p[(size_t)y0 * pw + x0] + p[(size_t)y0 * pw + x1]) + (p[(size_t)y1 * pw + x0] + p[(size_t)y1 * pw + x1]);
            t.mips[l][(size_t)y * mw + x] = s * 0.25f;
        }
    }
}
// bilinear at an integer mip, wrap addressing, texel centres at (i+0.5)/dim
static inline float4 sample_bilinear(const Texture& t, uint mip, float2 uv) {
    uint mw = std::max(1u, t.w >> mip), mh = std::max(1u, t.h >> mip);
    float fx = uv.x * (float)mw - 0.5f, fy = uv.y * (float)mh - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float ax = fx - flx, ay = fy - fly;
    // keep the integer conversion in range for huge |uv|
    flx = flx - floorf(flx / (float)mw) * (float)mw; fly = fly - floorf(fly / (float)mh) * (float)mh;
    int x0 = (int)flx, y0 = (int)fly;
    float4 a = lerp4(t.texel(mip, x0, y0), t.texel(mip, x0 + 1, y0), ax);
    float4 b = lerp4(t.texel(mip, x0, y0 + 1), t.texel(mip, x0 + 1, y0 + 1), ax);
    return lerp4(a, b, ay);
}
// Texture2D.SampleLevel(sampler, uv, lambda) with trilinear filtering
static inline float4 sample_trilinear(const Texture& t, float2 uv, float lambda) {
    float maxl = (float)(t.mipLevels - 1);
    float l = clampf(lambda, 0.0f, maxl);
    float l0 = floorf(l);
    uint m0 = (uint)l0, m1 = std::min(m0 + 1, t.mipLevels - 1);
    float f = l - l0;
    float4 a = sample_bilinear(t, m0, uv);
    if (f == 0.0f || m1 == m0) return a;
    float4 b = sample_bilinear(t, m1, uv);
    return lerp4(a, b, f);
}

// Texture2D.SampleGrad through an anisotropic sampler (Donut's m_AnisotropicWrapSampler, maxAnisotropy 16), as the emissive-triangle bake asks for it
// (LightsBaker.hlsl:647). What a texture unit does with two gradients is implementation defined; restated in the formulation of EXT_texture_filter_anisotropic
// (UNPINNED: there is no reference text for it): Px, Py = the gradients' lengths in texels, N = min(ceil(Pmax / Pmin), 16) trilinear taps at LOD log2(Pmax / N),
// spaced evenly along the longer gradient and averaged. The bake's own gradients are collinear with a 2 : 1 length ratio (its long gradient is -shortEdge / 3: the
// UV edges sum to zero), so it always takes two taps, one level finer than a single tap at the longer gradient's LOD (tests/test_emissive_bake_anisotropy.py).
static inline float4 sample_grad_anisotropic(const Texture& t, float2 uv, float2 gx, float2 gy) {
    const float lx = length(make_float2(gx.x * (float)t.w, gx.y * (float)t.h)), ly = length(make_float2(gy.x * (float)t.w, gy.y * (float)t.h));
    const float pmax = fmaxf_(lx, ly), pmin = fminf_(lx, ly);
    const float2 major = (lx >= ly) ? gx : gy;
    float n = (pmin > 0.f) ? ceilf(pmax / pmin) : 16.0f;
    n = clampf(n, 1.0f, 16.0f);
    const float lod = (pmax > 0.f) ? dm_log2(clampf(pmax / n, FLT_MIN_, FLT_MAX_)) : 0.0f;
    float4 sum = make_float4(0, 0, 0, 0);
    const uint taps = (uint)n;
    for (uint i = 0; i < taps; i++) {
        const float o = ((float)i + 0.5f) / n - 0.5f;
        sum = sum + sample_trilinear(t, make_float2(uv.x + major.x * o, uv.y + major.y * o), lod);
    }
    return make_float4(sum.x / n, sum.y / n, sum.z / n, sum.w / n);
}

// ---- environment: the host hands over a lat-long RGB image (row 0 = +Y pole); the path tracer samples the CUBE EnvMapBaker makes of it (envcube.h);
// EnvMap.hlsli:54-93 semantics for transform / multiplier
struct EnvMap {
    bool enabled; Texture tex; float3x4 toWorld, toLocal; float3 colorMultiplier;
    uint cubeDim = 2048; std::vector<EnvDirectionalLight> dirLights; std::vector<uint2> cubeTexels; EnvCube cube; bool cubeDirty = true;
    // BC6H round trip (EnvMapBaker.cpp:593-633): `cube` is what the path tracer samples — the decoded compressed cube when cubeCompression != 0 — while the importance map keeps
    // reading the uncompressed texels (`cubeSource`, :635)
    uint cubeCompression = 0; std::vector<uint2> cubeTexelsSource; EnvCube cubeSource;
    // the procedural sky as (additional) source of the bake (EnvMapBaker.hlsl:228-236, 247-265; sky.h): constants, the four look-up textures, the half-resolution cloud pre-pass cube
    bool skyEnabled = false; ProceduralSkyContext sky; std::vector<float4> skyTex[4]; std::vector<uint2> skyLowResTexels; EnvCube skyLowRes;
    // the image as a CUBE map instead of a lat-long image (ptref_set_environment_cube; EnvMapBaker.cpp:399-411, EnvMapBaker.hlsl BackgroundSourceType 2): 6 x dim x dim RGBA16F texels
    std::vector<uint2> imageCube; uint imageCubeDim = 0;
    EnvCube imageCubeView() const { EnvCube v; memset(&v, 0, sizeof(v)); v.texels = imageCube.data(); v.dim = imageCubeDim; v.mipLevels = 1u; return v; }
    bool hasImage() const { return tex.w != 0u || imageCubeDim != 0u; }
    float3 ToLocal(float3 dir) const { return mul_vec_mat3(dir, toLocal); }
    float3 ToWorld(float3 dir) const { return mul_vec_mat3(dir, toWorld); }
    // SampleSource (EnvMapBaker.hlsl:98-110): the equirectangular source through a linear sampler, wrap in u, clamp in v, mip 0
    float3 SampleSource(float3 direction) const {
        if (imageCubeDim) return xyz(env_cube_sample_level(imageCubeView(), direction, 0.0f));      // BackgroundSourceType 2: t_SrcCubemapEnvMap.SampleLevel(s_Linear, direction, 0)
        if (!tex.w) return make_float3(0.f, 0.f, 0.f);      // BackgroundSourceType 0: no image (a procedural sky alone)
        float2 uv = world_to_latlong_map(direction);
        float mh = (float)tex.h;
        uv.y = clampf(uv.y, 0.5f / mh, 1.0f - 0.5f / mh);
        return xyz(sample_bilinear(tex, 0, uv));
    }
    float3 EvalLocal(float3 localDir, float lod) const { return xyz(env_cube_sample_level(cube, localDir, lod)) * colorMultiplier; }      // EnvMap.hlsli:82-85
};

// ---- the scene
struct Triangle { float3 v0, v1, v2; uint subInstance, triIndex, flags; float pad; };   // world-space vertices (shared vertices of a mesh are the same floats in every triangle that uses them); flags bit0 = non-opaque (alpha tested), bit1 = exclude from NEE; pad: tri_box_accepts
struct Scene {
    std::vector<uint> indices; std::vector<float3> positions; std::vector<float2> uvs; std::vector<uint> normals, tangents;
    std::vector<GeometryDesc> geometries; std::vector<MeshDesc> meshes; std::vector<InstanceDesc> instances;
    std::vector<PTMaterialData> materials; std::vector<Texture> textures;
    EnvMap env;
    // the pose of the frame before (ptref_set_previous_pose; Donut's InstanceData.prevTransform and GeometryData.prevPositionOffset): empty = the scene did not move
    std::vector<float3> prevPositions; std::vector<InstanceDesc> prevInstances;
    // derived
    std::vector<uint> instFirstSubInstance;                 // InstanceData.firstGeometryInstanceIndex
    std::vector<uint2> subInstToInstGeom;                   // subInstance -> (instanceIndex, global geometry index)
    std::vector<SubInstanceData> subInstances;
    std::vector<Triangle> tris;
    // lights
    std::vector<PolymorphicLightInfo> lights; std::vector<PolymorphicLightInfoEx> lightsEx;
    std::vector<uint> proxyCounters, proxyIndices, envLookup; uint envLookupDim; std::vector<float> lightWeights;      // lightWeights: ComputeWeight per light, kept for the per-frame proxy rebuild of NEE-AT
    std::vector<PolymorphicLightInfoFull> analyticLights;   // supplied by the host (pt_set_lights)
    LightTable lightTable; LightFrustumBoost lightBoost = {};      // lightBoost: ImportanceBooster's frustum term (mul 0: off)
    // NEE-AT inputs (ptref_set_local_light_sampling): the screen-tile local samplers as the host hands them in, and the feedback switch
    std::vector<uint> localTable; uint localResX = 0, localResY = 0, localJitterX = 0, localJitterY = 0; float localRatio = 0.f, sscThreshold = 0.f; bool feedbackRequired = false;
    float* depthExport = nullptr; uint depthWidth = 0; float clipZ[4] = {0, 0, 0, 0}, clipW[4] = {0, 0, 0, 0}; bool haveClip = false; float worldToClip[16] = {0};      // ptref_set_view_projection; the plane is the run's (NeeAtState)
    void bindLocalSampling() {
        LightTable& T = lightTable;
        T.LocalSamplingBuffer = localResX ? localTable.data() : nullptr; T.LocalResX = localResX; T.LocalResY = localResY; T.LocalJitterX = localJitterX; T.LocalJitterY = localJitterY;
        T.DepthExport = depthExport; T.DepthWidth = depthWidth; memcpy(T.ClipZ, clipZ, 16); memcpy(T.ClipW, clipW, 16);
        T.LocalToGlobalSampleRatio = localResX ? localRatio : 0.f; T.ScreenSpaceVsWorldSpaceThreshold = sscThreshold; T.TemporalFeedbackRequired = feedbackRequired ? 1u : 0u;
    }
    // BVH2
    struct Node { float3 bmin; uint leftFirst; float3 bmax; uint count; };   // count==0: inner (children leftFirst, leftFirst+1)
    std::vector<Node> nodes; std::vector<uint> triOrder;
    int bruteForce = 0;                                     // ptref_set_brute_force: every query tests every triangle (the definition the BVH must reproduce; diagnostics)
};

static_assert(sizeof(SubInstanceData) == 32, "SubInstanceData must be 32 bytes");
struct HitInfo { float t; uint prim; float u, v; };      // prim = global triangle index, 0xFFFFFFFF = miss

// ---- The hit definition, first half: a WATERTIGHT ray / triangle test (Woop, Benthin, Wald: "Watertight Ray/Triangle Intersection", JCGT 2013), both sides, tmin < t < tmax;
// (u, v) = the DXR barycentrics of vertices 1 and 2. DXR promises that a ray cannot slip between two triangles that share an edge or a vertex (what Bridge::traceScatterRay /
// traceVisibilityRay inherit from the API, PathTracerBridgeDonut.hlsli:993-1055). The vertices are translated to the ray origin and sheared into the ray's own frame (kz = the
// axis of the largest |d|, kx / ky the next two in cyclic order): a vertex's 2D position then depends on the vertex and the ray only, never on the triangle it is tested for.
// The three edge functions are formed WITHOUT fused products — a * b - c * d negates exactly when the edge is walked the other way, and round(p) - round(q) has the sign of
// p - q whenever it is not zero — and a zero is resolved by the exact residuals of the two products (fmaf(a, b, -p) is the error of p, exactly): every triangle around an edge or
// a vertex sees the same signs, a point on the boundary belongs to both sides, nothing falls between. Sz is the ray's correctly rounded reciprocal (ray_safe_rcp: the traversal
// holds it anyway; kz is the dominant axis, so the clamp never acts). Same text on both sides of the parity fence (rtxpt_amd/csrc/pt_scene.h).
static inline float ray_safe_rcp(float d) {                // correctly rounded 1/d with |d| clamped away from 0 (no inf, no NaN in the slab arithmetic)
    float a = fabsf(d);
    float s = (a < 7.888609e-31f) ? 7.888609e-31f : a;
    return 1.0f / ((d < 0.0f) ? -s : s);
}
static inline float wt_edge(float ax, float ay, float bx, float by) {      // the edge function ax * by - ay * bx with an exact sign
    const float p = ax * by, q = ay * bx;
    float e = p - q;
    if (e == 0.0f) e = fmaf(ax, by, -p) - fmaf(ay, bx, -q);                // p == q: the difference of the two rounding errors IS the exact value
    return e;
}
static inline bool intersect_tri_wt(float3 v0, float3 v1, float3 v2, float3 o, float3 d, float tmin, float tmax, float& t, float& u, float& v) {
    const float adx = fabsf(d.x), ady = fabsf(d.y), adz = fabsf(d.z);
    const int kz = (adz > adx && adz > ady) ? 2 : ((ady > adx) ? 1 : 0);      // ties go to the lower axis
    const float3 A = v0 - o, B = v1 - o, C = v2 - o;
    float Akx, Aky, Akz, Bkx, Bky, Bkz, Ckx, Cky, Ckz, dkx, dky, dkz;
    if (kz == 2) { Akx = A.x; Aky = A.y; Akz = A.z; Bkx = B.x; Bky = B.y; Bkz = B.z; Ckx = C.x; Cky = C.y; Ckz = C.z; dkx = d.x; dky = d.y; dkz = d.z; }
    else if (kz == 1) { Akx = A.z; Aky = A.x; Akz = A.y; Bkx = B.z; Bky = B.x; Bkz = B.y; Ckx = C.z; Cky = C.x; Ckz = C.y; dkx = d.z; dky = d.x; dkz = d.y; }
    else { Akx = A.y; Aky = A.z; Akz = A.x; Bkx = B.y; Bky = B.z; Bkz = B.x; Ckx = C.y; Cky = C.z; Ckz = C.x; dkx = d.y; dky = d.z; dkz = d.x; }
    const float Sz = ray_safe_rcp(dkz), Sx = dkx * Sz, Sy = dky * Sz;
    const float Ax = fmaf(-Sx, Akz, Akx), Ay = fmaf(-Sy, Akz, Aky), Bx = fmaf(-Sx, Bkz, Bkx), By = fmaf(-Sy, Bkz, Bky), Cx = fmaf(-Sx, Ckz, Ckx), Cy = fmaf(-Sy, Ckz, Cky);
    const float U = wt_edge(Cx, Cy, Bx, By), V = wt_edge(Ax, Ay, Cx, Cy), W = wt_edge(Bx, By, Ax, Ay);
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float Az = Sz * Akz, Bz = Sz * Bkz, Cz = Sz * Ckz;
    const float T = fmaf(W, Cz, fmaf(V, Bz, U * Az));
    const float inv = 1.0f / det;
    t = T * inv; u = V * inv; v = W * inv;
    return (t > tmin) && (t < tmax);
}
// The second half of the hit definition (see rtxpt_amd/csrc/pt_scene.h for the argument): whether a box above the triangle lets the ray through must not decide the closest
// hit. A hit only counts if t lies in the slab interval of the triangle's own padded bounding box, computed as every box test above it computes its interval:
// (plane - o) * inv, inv = ray_safe_rcp(d). Monotone rounding + nested boxes => no conservative BVH can cull an accepted hit. (The watertight test places t within a few
// roundings of the triangle's plane, the pad is 10 - 100 x wider: the box never takes back what the first half found — tests/test_gpu_watertight.py counts escapes: 0.)
static inline float tri_pad(float3 mn, float3 mx, float scenePad) {
    float3 e = mx - mn;
    return 2e-5f * fmaxf_(e.x, fmaxf_(e.y, e.z)) + scenePad;
}
static inline float scene_pad(float3 smn, float3 smx) { return 2e-6f * length(smx - smn); }
static inline bool tri_box_accepts(const Triangle& tr, float3 o, float3 inv, float t) {
    float3 mn = min3v(tr.v0, min3v(tr.v1, tr.v2)) - make_float3(tr.pad), mx = max3v(tr.v0, max3v(tr.v1, tr.v2)) + make_float3(tr.pad);
    float ax = (mn.x - o.x) * inv.x, bx = (mx.x - o.x) * inv.x, ay = (mn.y - o.y) * inv.y, by = (mx.y - o.y) * inv.y, az = (mn.z - o.z) * inv.z, bz = (mx.z - o.z) * inv.z;
    float tn = fmaxf_(fmaxf_(fminf_(ax, bx), fminf_(ay, by)), fminf_(az, bz));
    float tf = fminf_(fminf_(fmaxf_(ax, bx), fmaxf_(ay, by)), fmaxf_(az, bz));
    return (tn <= t) && (t <= tf);
}
static inline bool intersect_tri(const Triangle& tr, float3 o, float3 d, float tmin, float tmax, float& t, float& u, float& v) {
    if (!intersect_tri_wt(tr.v0, tr.v1, tr.v2, o, d, tmin, tmax, t, u, v)) return false;
    return tri_box_accepts(tr, o, make_float3(ray_safe_rcp(d.x), ray_safe_rcp(d.y), ray_safe_rcp(d.z)), t);
}

// BridgeDonut:929-971 AlphaTestImpl — base-colour texture alpha at mip 0 vs the 8-bit quantised cutoff
static inline bool AlphaTest(const Scene& sc, const Triangle& tr, float u, float v) {
    const SubInstanceData& si = sc.subInstances[tr.subInstance];
    if ((si.FlagsAndAlphaInfo & SubInstanceData::Flags_AlphaTested) == 0) return true;
    const GeometryDesc& g = sc.geometries[si.GlobalGeometryIndex_PTMaterialDataIndex >> 16];
    const uint* idx = &sc.indices[g.indexOffset + tr.triIndex * 3];
    float2 t0 = sc.uvs[g.vertexOffset + idx[0]], t1 = sc.uvs[g.vertexOffset + idx[1]], t2 = sc.uvs[g.vertexOffset + idx[2]];
    float b0 = 1.0f - (u + v);
    float2 tc = (t0 * b0 + t1 * u) + t2 * v;
    const Texture& tex = sc.textures[si.AlphaTextureIndex()];
    float opacity = sample_bilinear(tex, 0, tc).w;
    return opacity >= si.AlphaCutoff();
}

static inline bool slab(const Scene::Node& n, float3 o, float3 id, float tmax) {
    float tx1 = (n.bmin.x - o.x) * id.x, tx2 = (n.bmax.x - o.x) * id.x;
    float tmn = fminf_(tx1, tx2), tmx = fmaxf_(tx1, tx2);
    float ty1 = (n.bmin.y - o.y) * id.y, ty2 = (n.bmax.y - o.y) * id.y;
    tmn = fmaxf_(tmn, fminf_(ty1, ty2)); tmx = fminf_(tmx, fmaxf_(ty1, ty2));
    float tz1 = (n.bmin.z - o.z) * id.z, tz2 = (n.bmax.z - o.z) * id.z;
    tmn = fmaxf_(tmn, fminf_(tz1, tz2)); tmx = fminf_(tmx, fmaxf_(tz1, tz2));
    // generous conservative margins: the oracle must never cull a triangle the exact test would accept
    return (tmx * 1.00001f + 1e-6f >= tmn * 0.99999f - 1e-6f) && (tmn * 0.99999f - 1e-6f < tmax) && (tmx * 1.00001f + 1e-6f > 0.0f);
}

static inline HitInfo trace_closest_bruteforce(const Scene& sc, float3 o, float3 d, float tmin, float tmax);
// diagnostics (bruteForce == 2): the slab numbers of every node on the way to the triangle the BVH query missed
static void debug_report_miss(const Scene& sc, float3 o, float3 d, float tmax, uint gotPrim, float gotT, uint wantPrim, float wantT) {
    fprintf(stderr, "BVH != brute force: o %.9g %.9g %.9g d %.9g %.9g %.9g tmax %.9g | bvh prim %u t %.9g | brute prim %u t %.9g\n", o.x, o.y, o.z, d.x, d.y, d.z, tmax, gotPrim, gotT, wantPrim, wantT);
    if (wantPrim == 0xFFFFFFFFu) return;
    float3 id = make_float3(ray_safe_rcp(d.x), ray_safe_rcp(d.y), ray_safe_rcp(d.z));
    std::vector<uint> chain; std::vector<std::pair<uint, int>> st; st.push_back({0u, 0});      // depth-first search for the leaf holding wantPrim
    std::vector<uint> path;
    std::function<bool(uint)> find = [&](uint ni) -> bool {
        const Scene::Node& n = sc.nodes[ni]; path.push_back(ni);
        if (n.count) { for (uint i = 0; i < n.count; i++) if (sc.triOrder[n.leftFirst + i] == wantPrim) return true; path.pop_back(); return false; }
        if (find(n.leftFirst) || find(n.leftFirst + 1)) return true;
        path.pop_back(); return false;
    };
    find(0);
    for (uint ni : path) {
        const Scene::Node& n = sc.nodes[ni];
        fprintf(stderr, "   node %u box [%.9g %.9g %.9g | %.9g %.9g %.9g] slab(tmax=wantT*1.001) %d tx %.9g %.9g ty %.9g %.9g tz %.9g %.9g\n", ni, n.bmin.x, n.bmin.y, n.bmin.z, n.bmax.x, n.bmax.y, n.bmax.z,
                (int)slab(n, o, id, wantT * 1.001f), (n.bmin.x - o.x) * id.x, (n.bmax.x - o.x) * id.x, (n.bmin.y - o.y) * id.y, (n.bmax.y - o.y) * id.y, (n.bmin.z - o.z) * id.z, (n.bmax.z - o.z) * id.z);
    }
    const Triangle& tr = sc.tris[wantPrim];
    fprintf(stderr, "   tri v0 %.9g %.9g %.9g v1 %.9g %.9g %.9g v2 %.9g %.9g %.9g\n", tr.v0.x, tr.v0.y, tr.v0.z, tr.v1.x, tr.v1.y, tr.v1.z, tr.v2.x, tr.v2.y, tr.v2.z);
}
// closest hit (BridgeDonut:1029-1055 traceScatterRay): RAY_FLAG_NONE, alpha test on non-opaque candidates
static inline HitInfo trace_closest(const Scene& sc, float3 o, float3 d, float tmin, float tmax, uint64_t* nodeVisits = 0, uint64_t* triTests = 0) {
    HitInfo h; h.t = tmax; h.prim = 0xFFFFFFFFu; h.u = h.v = 0;
    if (sc.nodes.empty()) return h;
    if (sc.bruteForce == 1) {
        for (uint p = 0; p < sc.tris.size(); p++) {
            float t, u, v;
            if (!intersect_tri(sc.tris[p], o, d, tmin, tmax, t, u, v)) continue;
            if (!(t < h.t || (t == h.t && p < h.prim))) continue;
            if ((sc.tris[p].flags & 1u) && !AlphaTest(sc, sc.tris[p], u, v)) continue;
            h.t = t; h.prim = p; h.u = u; h.v = v;
        }
        return h;
    }
    float3 id = make_float3(ray_safe_rcp(d.x), ray_safe_rcp(d.y), ray_safe_rcp(d.z));
    uint stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const Scene::Node& n = sc.nodes[stack[--sp]];
        if (nodeVisits) (*nodeVisits)++;
        if (!slab(n, o, id, h.t)) continue;
        if (n.count) {
            for (uint i = 0; i < n.count; i++) {
                uint p = sc.triOrder[n.leftFirst + i];
                const Triangle& tr = sc.tris[p];
                float t, u, v;
                if (triTests) (*triTests)++;
                if (!intersect_tri(tr, o, d, tmin, tmax, t, u, v)) continue;
                if (!(t < h.t || (t == h.t && p < h.prim))) continue;
                if ((tr.flags & 1u) && !AlphaTest(sc, tr, u, v)) continue;
                h.t = t; h.prim = p; h.u = u; h.v = v;
            }
        } else { stack[sp++] = n.leftFirst; stack[sp++] = n.leftFirst + 1; }
    }
    if (sc.bruteForce == 2) {                              // diagnostics: report every query the BVH answers differently from the exhaustive loop
        HitInfo b = trace_closest_bruteforce(sc, o, d, tmin, tmax);
        if (b.prim != h.prim || b.t != h.t) debug_report_miss(sc, o, d, tmax, h.prim, h.t, b.prim, b.t);
    }
    return h;
}
// any hit (BridgeDonut:993-1027 traceVisibilityRay): returns true when VISIBLE (nothing committed)
static inline bool trace_visibility(const Scene& sc, float3 o, float3 d, float tmin, float tmax, uint64_t* nodeVisits = 0, uint64_t* triTests = 0) {
    if (sc.nodes.empty()) return true;
    if (sc.bruteForce == 1) {
        for (uint p = 0; p < sc.tris.size(); p++) {
            const Triangle& tr = sc.tris[p];
            float t, u, v;
            if (!intersect_tri(tr, o, d, tmin, tmax, t, u, v)) continue;
            if (tr.flags & 1u) { if (tr.flags & 2u) continue; if (!AlphaTest(sc, tr, u, v)) continue; }
            return false;
        }
        return true;
    }
    float3 id = make_float3(ray_safe_rcp(d.x), ray_safe_rcp(d.y), ray_safe_rcp(d.z));
    uint stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const Scene::Node& n = sc.nodes[stack[--sp]];
        if (nodeVisits) (*nodeVisits)++;
        if (!slab(n, o, id, tmax)) continue;
        if (n.count) {
            for (uint i = 0; i < n.count; i++) {
                const Triangle& tr = sc.tris[sc.triOrder[n.leftFirst + i]];
                float t, u, v;
                if (triTests) (*triTests)++;
                if (!intersect_tri(tr, o, d, tmin, tmax, t, u, v)) continue;
                if (tr.flags & 1u) {                       // non-opaque candidate: AlphaTestVisibilityRay (BridgeDonut:981-989)
                    if (tr.flags & 2u) continue;          // ExcludeFromNEE
                    if (!AlphaTest(sc, tr, u, v)) continue;
                }
                return false;
            }
        } else { stack[sp++] = n.leftFirst; stack[sp++] = n.leftFirst + 1; }
    }
    if (sc.bruteForce == 2) {
        for (uint p = 0; p < sc.tris.size(); p++) {
            const Triangle& tr = sc.tris[p];
            float t, u, v;
            if (!intersect_tri(tr, o, d, tmin, tmax, t, u, v)) continue;
            if (tr.flags & 1u) { if (tr.flags & 2u) continue; if (!AlphaTest(sc, tr, u, v)) continue; }
            debug_report_miss(sc, o, d, tmax, 0xFFFFFFFFu, tmax, p, t); break;
        }
    }
    return true;
}
static inline HitInfo trace_closest_bruteforce(const Scene& sc, float3 o, float3 d, float tmin, float tmax) {
    HitInfo h; h.t = tmax; h.prim = 0xFFFFFFFFu; h.u = h.v = 0;
    for (uint p = 0; p < sc.tris.size(); p++) {
        float t, u, v;
        if (!intersect_tri(sc.tris[p], o, d, tmin, tmax, t, u, v)) continue;
        if (!(t < h.t || (t == h.t && p < h.prim))) continue;
        if ((sc.tris[p].flags & 1u) && !AlphaTest(sc, sc.tris[p], u, v)) continue;
        h.t = t; h.prim = p; h.u = u; h.v = v;
    }
    return h;
}

// ---- scene finalisation: sub-instances, world-space triangles, BVH
void finalize_geometry(Scene& sc);        // scene.cpp part of ptref_api.cpp
void build_bvh(Scene& sc);
void bake_lights(Scene& sc, bool neeEnabled);

} // namespace ptref
