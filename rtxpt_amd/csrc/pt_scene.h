// mi355pt — device-side scene view: HBM data layout, texture / environment sampling, BVH traversal.
//
// HBM layout (all resident for the life of the scene; sized for 288 GB, nothing is streamed):
//   vertex streams   indices u32 | positions float3 | uvs float2 | normals,tangents RGBA8_SNORM      (object space, as uploaded)
//   GeometryDesc 32 B, InstanceDesc 64 B, SubInstanceData 32 B (Rtxpt/Shaders/SubInstanceData.h:23-46), PTMaterialData 128 B
//   TriRecord 48 B   the three world-space vertices, one 16-byte group per AXIS (x0 x1 x2 | prim, y0 y1 y2 | flags, z0 z1 z2 | pad), in BVH leaf order (one contiguous run per leaf)
//   BvhNode 64 B     BVH2 node holding BOTH child boxes (one 64 B fetch decides both children) + two child references
//   primInfo 8 B     global primitive id -> (subInstance, triangle index)
//   texel pool       RGBA32F texels of every mip of every texture (mips built on the host in float so that device == oracle)
//   light table      PolymorphicLightInfo 32 B (+Ex 16 B), proxy counters, proxy index array, env lookup map
// The reference gets traversal from the DXR driver (Rtxpt/Shaders/PathTracerBridgeDonut.hlsli:993-1055); here it is explicit.
#pragma once
#include "pt_lights.h"
#include "pt_envcube.h"
#include "pt_sky.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

// ---- MaterialPT.h:24-42
enum : uint {
    PTMaterialFlags_UseSpecularGlossModel = 0x1, PTMaterialFlags_UseMetalRoughOrSpecularTexture = 0x4,
    PTMaterialFlags_UseBaseOrDiffuseTexture = 0x8, PTMaterialFlags_UseEmissiveTexture = 0x10, PTMaterialFlags_UseNormalTexture = 0x20,
    PTMaterialFlags_UseTransmissionTexture = 0x80, PTMaterialFlags_MetalnessInRedChannel = 0x100, PTMaterialFlags_ThinSurface = 0x200,
    PTMaterialFlags_EnableAsAnalyticLightProxy = 0x800, PTMaterialFlags_IgnoreMeshTangentSpace = 1u << 12, PTMaterialFlags_NestedPriorityShift = 28,
};
// MaterialPT.h:45-77
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
// SubInstanceData.h:23-46 (extended words hold element offsets into the shared streams)
struct SubInstanceData {
    enum : uint { Flags_AlphaTested = 1u << 16, Flags_ExcludeFromNEE = 1u << 17, Flags_AlphaOffsetOffset = 24 };
    uint FlagsAndAlphaInfo, GlobalGeometryIndex_PTMaterialDataIndex, EmissiveLightMappingOffset, AnalyticProxyLightIndex;
    uint IndexBufferIndex_VertexBufferIndex, IndexOffset, TexCoord1Offset, padding0;
    float AlphaCutoff() const { return (float)(FlagsAndAlphaInfo >> Flags_AlphaOffsetOffset) / 255.0f; }
    uint AlphaTextureIndex() const { return FlagsAndAlphaInfo & 0xFFFFu; }
};
static_assert(sizeof(SubInstanceData) == 32, "SubInstanceData must be 32 bytes");
struct GeometryDesc { uint indexOffset, numIndices, vertexOffset, numVertices, flags, materialIndex, geomFlags, _pad; };
enum : uint { GEOM_HAS_UV = 1, GEOM_HAS_NORMAL = 2, GEOM_HAS_TANGENT = 4, GEOMF_ALPHA_TESTED = 1, GEOMF_EXCLUDE_FROM_NEE = 2 };
struct MeshDesc { uint firstGeometry, numGeometries; };
struct InstanceDesc { float3x4 transform; uint meshIndex; uint analyticProxyLight; uint _pad[2]; };      // analyticProxyLight: 0 = none, k + 1 = stands in for analytic light k (SubInstanceData.AnalyticProxyLightIndex)
static_assert(sizeof(InstanceDesc) == 64, "InstanceDesc must be 64 bytes");

// One 16-byte group per axis: the watertight test (intersect_tri_wt) wants the vertices' coordinates in the ray's axis order (kx, ky, kz), and with this layout that permutation
// is the ORDER OF THREE LOADS (group kx, group ky, group kz) instead of eighteen selects. Vertices are stored as such — a vertex shared by two triangles is the same three floats
// in both records, which is what the test's edge decisions rest on (the round-4 record held v0, v1 - v0, v2 - v0).
struct TriRecord { float x[3]; uint prim; float y[3]; uint flags; float z[3]; float pad; };    // flags bit0 non-opaque, bit1 exclude from NEE; pad: see tri_box_accepts
static inline TriRecord tri_record(float3 p0, float3 p1, float3 p2, uint prim, uint flags, float pad) {
    TriRecord t; t.x[0] = p0.x; t.x[1] = p1.x; t.x[2] = p2.x; t.prim = prim; t.y[0] = p0.y; t.y[1] = p1.y; t.y[2] = p2.y; t.flags = flags; t.z[0] = p0.z; t.z[1] = p1.z; t.z[2] = p2.z; t.pad = pad; return t;
}
static inline float3 tri_v0(const TriRecord& t) { return make_float3(t.x[0], t.y[0], t.z[0]); }
static inline float3 tri_v1(const TriRecord& t) { return make_float3(t.x[1], t.y[1], t.z[1]); }
static inline float3 tri_v2(const TriRecord& t) { return make_float3(t.x[2], t.y[2], t.z[2]); }
static inline void tri_bounds(const TriRecord& t, float3& mn, float3& mx) {      // the unpadded box of the three vertices
    mn = make_float3(fminf_(t.x[0], fminf_(t.x[1], t.x[2])), fminf_(t.y[0], fminf_(t.y[1], t.y[2])), fminf_(t.z[0], fminf_(t.z[1], t.z[2])));
    mx = make_float3(fmaxf_(t.x[0], fmaxf_(t.x[1], t.x[2])), fmaxf_(t.y[0], fmaxf_(t.y[1], t.y[2])), fmaxf_(t.z[0], fmaxf_(t.z[1], t.z[2])));
}
static_assert(sizeof(TriRecord) == 48, "TriRecord must be 48 bytes");
struct BvhNode { float3 lmin, lmax, rmin, rmax; uint left, right, _pad0, _pad1; };           // child ref: bit31 = leaf (first<<3 | count-1)
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");
#define PT_T8_LANES 2             // lanes per ray in the traversal kernels (pt_traverse8p.h; the four-lane kernel of rounds 1-3 is in the history)
#ifndef PT_BVH_MAX_LEAF
#define PT_BVH_MAX_LEAF 2         // triangles per leaf = lanes per ray: a leaf is tested in one round. With pairs, 2 instead of 4: k_extend 48.3 -> 43.6 ms, k_shadow 14.0 -> 12.2 ms
#endif                                                  // (1: 50.6 / 13.6 ms; 3: 46.1 / 13.3; 6: 51.9 / 15.3 — profiles/r03u_leafsize_ab*.txt)
static const uint BVH_LEAF_BIT = 0x80000000u, BVH_EMPTY = 0xFFFFFFFFu, BVH_MAX_LEAF = PT_BVH_MAX_LEAF, BVH_STACK = 64;
// BVH8 node, 128 B = one cache line: the 8 lanes of a ray's lane group each fetch one 12 B child slot plus the shared 16 B header, so a
// whole node costs one line lookup per group instead of four 16 B gathers per lane. Child boxes are 8-bit quantised relative to the node
// origin with power-of-two scales (conservative: decoded lo <= true lo, decoded hi >= true hi, verified with the decode arithmetic itself).
struct Bvh8Child { uint ref, q0, q1; };                       // q0 = qlo.x | qlo.y<<8 | qlo.z<<16 | qhi.x<<24, q1 = qhi.y | qhi.z<<8
struct Bvh8Node { float ox, oy, oz; uint exps; Bvh8Child c[8]; uint _pad[4]; };   // exps = ex | ey<<8 | ez<<16 | childCount<<24 (scale = 2^(e-127)); _pad[0..2] = the three scales as floats
typedef uint u32x4 __attribute__((ext_vector_type(4)));      // 16-byte aligned vector loads (global_load_dwordx4)
typedef uint u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));   // v_pk_*_f32 operands
struct __attribute__((packed, aligned(4))) u32x3p { uint x, y, z; };   // 12-byte child slot load (global_load_dwordx3)
typedef float f32x4 __attribute__((ext_vector_type(4)));
static_assert(sizeof(Bvh8Node) == 128, "Bvh8Node must be 128 bytes");
#ifndef PT_T8_CHUNK
#define PT_T8_CHUNK 32      // rays a wave parks in LDS per chunk fetch (one per lane pair)
#endif
#ifndef PT_BVH8_STACK
#define PT_BVH8_STACK 13         // (7 blocks of 256 threads per CU: 14 x 8 B x 128 pairs + the chunk parking lot per block; 12 entries: -0.6 %, 8: -6 %)
#endif
// traversal launch geometry (pt_traverse8p.h): 256-thread blocks, 2 lanes per ray -> 128 rays in flight per block, each with an LDS stack of
// BVH8_STACK entries (odd stride: quads land on different banks) and a T8_SPILL_DEPTH-entry tail in global memory (DeviceScene::travSpill)
static const uint BVH8_STACK = PT_BVH8_STACK, BVH8_STACK_STRIDE = PT_BVH8_STACK + 1;
static const uint T8_BLOCK = 256, T8_CHUNK = PT_T8_CHUNK, T8_LANES = PT_T8_LANES, T8_GROUPS_PER_WAVE = 64 / PT_T8_LANES, T8_GROUPS_PER_BLOCK = T8_BLOCK / T8_LANES, T8_SPILL_DEPTH = 96;
#ifndef PT_T8_MAX_BLOCKS
#define PT_T8_MAX_BLOCKS (256 * 7 * 3)  // upper bound of a traversal grid (sizes the stack-tail memory). The GPU holds 256 x 7 blocks (LDS). A frame
                                                                          // of several pipelined batches launches exactly that many per batch — 2x / 3x / 4x were 0.2 / 1.7 / 3.7 % slower on the full frame and
                                                                          // 4-7 % on one rank of an 8-way shard (profiles/r03w_maxblocks_ab.txt): the other batches fill the gaps —, a launch that has the GPU to
                                                                          // itself three times that: the blocks that finish early are replaced (TravAux::maxBlocks)
#endif
static const uint T8_MAX_BLOCKS = PT_T8_MAX_BLOCKS;     // persistent waves stride over 64-ray chunks

// per leaf-order triangle slot: what the alpha test needs, resolved at build time (texture coordinates of the 3 vertices, alpha texture, cutoff);
// tex == ~0: not alpha tested. Saves the primInfo -> subInstance -> index -> uv chain (7 dependent loads) inside the traversal loop.
// Since round 3 the record also says where the opacity values are (size and offset of the texture's ALPHA PLANE, below): the test is one record fetch and one
// fetch of four opacities, not record -> TexInfo -> texels.
struct AlphaRec { float2 t0, t1, t2; uint tex; float cutoff; uint wh; uint plane; uint fmt; uint _pad; };      // wh = w | h << 16 of mip 0; plane: byte offset into the alpha pool; fmt: 0 = u8, 1 = f32
static_assert(sizeof(AlphaRec) == 48, "AlphaRec must be 48 bytes");
// The alpha plane of a texture: the .w channel of its mip 0 on its own — one BYTE per texel where every opacity is k / 255 (8-bit sources: the float the shading
// path reads is (float)k / 255.0f, and the test forms exactly that quotient again), a float per texel otherwise. 16 x (4 x) smaller than the RGBA32F texels and
// contiguous: the alpha masks of a scene stay in the L2 while the traversal runs, and the four opacities of a test come from two lines instead of four.
struct AlphaPlane { uint wh, offset, fmt, _pad; };

struct TexInfo { uint w, h, mipLevels, _pad; unsigned long long base; uint mipOffset[16]; };   // offsets in texels relative to base

// Flat shading record, one 128 B line per global primitive (instance triangle): everything Bridge::loadSurface gathers through
// primInfo -> subInstToInstGeom -> {instance, subInstance, geometry} -> indices -> 4 vertex streams (five dependent hops, ~20 scattered loads) sits in
// one line that the hit's primitive id addresses directly, so all of it is in flight at once. The vertex data stay in OBJECT space and the instance is
// looked up by index (1 365 instances: cache resident), so the arithmetic of loadSurface — and a rigid animation — is untouched; deformed vertices rewrite
// the records (k_shade_tris, 0.3 ms at 2.8 M triangles). The two light links live in the sub-instance (re-baked with the lights) and are read only by
// emissive / proxy hits.
struct ShadeTri {
    uint instance, subInstance, triangleIndex, materialAndFlags;      // words 0-3; materialAndFlags = material index | GeometryDesc::flags << 16
    float3 p0, p1, p2;            // words  4-12: object-space positions
    float2 t0, t1, t2;            // words 13-18: texture coordinates (zero without GEOM_HAS_UV)
    float3 n0, n1, n2;            // words 19-27: the vertex normals as loadSurface uses them — normalize(Unpack_RGB8_SNORM(packed)), negated where they point away from the flat normal
                                  //   normalize(cross(p1 - p0, p2 - p0)) (BridgeDonut:655-668) — formed once per record by k_shade_tris with loadSurface's own expressions instead of
                                  //   per hit (three unpacks, square roots and nine divisions: ~200 of k_shade's 4 600 VALU instructions per vertex); zero without GEOM_HAS_NORMAL
    uint g0, g1, g2;              // words 28-30: RGBA8_SNORM tangents
    uint _pad;
};
static_assert(sizeof(ShadeTri) == 128, "ShadeTri must be one 128-byte line");

struct DeviceScene {
    const uint* indices; const float* positions; const float2* uvs; const uint* normals; const uint* tangents;
    const float* prevPositions; const InstanceDesc* prevInstances;      // the previous frame's pose (pt_set_motion_history / pt_set_previous_pose), or null: the stable-plane build pass's object motion
    const GeometryDesc* geometries; const InstanceDesc* instances; const SubInstanceData* subInstances; const uint2* subInstToInstGeom;
    const PTMaterialData* materials; uint materialCount;
    const TexInfo* textures; const float4* texels;
    TexInfo envTex; uint envEnabled; float3x4 envToWorld, envToLocal; float3 envColorMultiplier;      // envTex: the lat-long source (read by the cube bake only)
    EnvCube envCube;           // what the path tracer samples: EnvMapBaker's RGBA16F cube + mips (pt_envcube.h)
    const ProceduralSkyContext* sky;      // device copy of the procedural sky (constants + the four look-up textures), or null: the cube is baked from the image alone (pt_sky.h)
    EnvCube skyLowRes;         // the sky's half-resolution cloud pre-pass cube (one level; EnvMapBaker.cpp:318-324 m_cubemapLowRes)
    EnvCube envCubeSource;     // the uncompressed cube the importance map is built from (EnvMapBaker.cpp:635); the same texels as envCube unless the BC6H round trip is on
    LightTable lights;
    const BvhNode* nodes; const Bvh8Node* nodes8; const TriRecord* tris; const uint2* primInfo; uint numTris, rootIsValid;
    const AlphaRec* alphaRecs; // one per TriRecord slot (leaf order)
    const AlphaPlane* alphaPlanes; const unsigned char* alphaPool;      // per texture (pt_api.hip upload_textures); read by k_alpha_records and the traversal's alpha test
    const ShadeTri* shadeTris; // one per global primitive id (pt_build.hip k_shade_tris)
    const uint* primToSlot;    // global primitive id -> leaf-order slot (probes; the traversal's resolve pass takes it from TravAux)
    uint2* travSpill;          // T8_MAX_BLOCKS x T8_GROUPS_PER_BLOCK x T8_SPILL_DEPTH stack-tail entries
    const uint2* envImageCube; uint envImageCubeDim, _padEnvImageCube;      // the environment image given as a CUBE map instead of a lat-long image (pt_set_environment_cube): 6 x dim x dim RGBA16F texels, read by the cube bake only
};

struct HitInfo { float t; uint prim; float u, v; };
// a sub-tree of one ray's traversal handed to a follow-up launch (straggler splitting, pt_traverse8.h): 16 B
struct TravTask { uint tag, ref, tbits, _pad; };
struct TravTaskOut { TravTask* tasks; uint* count; uint capacity; };

// ---- texture sampling: wrap addressing, texel centres at (i+0.5)/dim, bilinear per mip, linear between mips
static inline const float4& tex_texel(const DeviceScene& sc, const TexInfo& t, uint mip, int x, int y) {
    uint mw = t.w >> mip; if (mw < 1u) mw = 1u;
    uint mh = t.h >> mip; if (mh < 1u) mh = 1u;
    // wrap: sample_bilinear hands over x0, x0+1 of an already wrapped x0, i.e. coordinates within one period of [0, dim), where a conditional
    // add/subtract equals the oracle's modulo (an integer `%` is ~35 VALU instructions on CDNA). The final clamp only matters for NaN/Inf
    // texture coordinates and keeps the fetch in bounds.
    int xi = x, yi = y;
    if (xi < 0) xi += (int)mw; else if (xi >= (int)mw) xi -= (int)mw;
    if (yi < 0) yi += (int)mh; else if (yi >= (int)mh) yi -= (int)mh;
    xi = xi < 0 ? 0 : (xi >= (int)mw ? (int)mw - 1 : xi);
    yi = yi < 0 ? 0 : (yi >= (int)mh ? (int)mh - 1 : yi);
    return sc.texels[t.base + t.mipOffset[mip] + (unsigned long long)yi * mw + (uint)xi];
}
static inline float4 sample_bilinear(const DeviceScene& sc, const TexInfo& t, uint mip, float2 uv) {
    uint mw = t.w >> mip; if (mw < 1u) mw = 1u;
    uint mh = t.h >> mip; if (mh < 1u) mh = 1u;
    float fx = uv.x * (float)mw - 0.5f, fy = uv.y * (float)mh - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float ax = fx - flx, ay = fy - fly;
    // the wrap: flx - floorf(flx / mw) * mw. For a power-of-two side the quotient is an exact scaling: the product with the exact reciprocal (2^-k from 2^k by exponent arithmetic) is the
    // same float as the correctly rounded division (as in alpha_test_slot); k_shade is bound by its VALU instructions and a trilinear fetch holds four of these divisions
    const float fmw = (float)mw, fmh = (float)mh;
    if (((mw & (mw - 1u)) | (mh & (mh - 1u))) == 0u) { flx = flx - floorf(flx * asfloat(0x7F000000u - asuint(fmw))) * fmw; fly = fly - floorf(fly * asfloat(0x7F000000u - asuint(fmh))) * fmh; }
    else { flx = flx - floorf(flx / fmw) * fmw; fly = fly - floorf(fly / fmh) * fmh; }
    int x0 = (int)flx, y0 = (int)fly;
    float4 a = lerp4(tex_texel(sc, t, mip, x0, y0), tex_texel(sc, t, mip, x0 + 1, y0), ax);
    float4 b = lerp4(tex_texel(sc, t, mip, x0, y0 + 1), tex_texel(sc, t, mip, x0 + 1, y0 + 1), ax);
    return lerp4(a, b, ay);
}
static inline float4 sample_trilinear(const DeviceScene& sc, const TexInfo& t, float2 uv, float lambda) {
    float maxl = (float)(t.mipLevels - 1);
    float l = clampf(lambda, 0.0f, maxl);
    float l0 = floorf(l);
    uint m0 = (uint)l0, m1 = m0 + 1; if (m1 > t.mipLevels - 1) m1 = t.mipLevels - 1;
    float f = l - l0;
    float4 a = sample_bilinear(sc, t, m0, uv);
    if (f == 0.0f || m1 == m0) return a;
    float4 b = sample_bilinear(sc, t, m1, uv);
    return lerp4(a, b, f);
}
// Texture2D.SampleGrad through an anisotropic sampler (Donut's m_AnisotropicWrapSampler, maxAnisotropy 16), as the emissive-triangle bake asks for it
// (LightsBaker.hlsl:647). What a texture unit does with two gradients is implementation defined; restated in the formulation of EXT_texture_filter_anisotropic
// (UNPINNED: there is no reference text for it): Px, Py = the gradients' lengths in texels, N = min(ceil(Pmax / Pmin), 16) trilinear taps at LOD log2(Pmax / N),
// spaced evenly along the longer gradient and averaged. The bake's own gradients are collinear with a 2 : 1 length ratio (its long gradient is -shortEdge / 3: the
// UV edges sum to zero), so it always takes two taps, one level finer than a single tap at the longer gradient's LOD (tests/test_emissive_bake_anisotropy.py).
static inline float4 sample_grad_anisotropic(const DeviceScene& sc, const TexInfo& t, float2 uv, float2 gx, float2 gy) {
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
        sum = sum + sample_trilinear(sc, t, make_float2(uv.x + major.x * o, uv.y + major.y * o), lod);
    }
    return make_float4(sum.x / n, sum.y / n, sum.z / n, sum.w / n);
}
// SampleSource (EnvMapBaker.hlsl:98-110): the equirectangular source through a linear sampler (wrap in u, clamp in v), mip 0 — or the cube-map source through the cube fetch
static inline float3 env_sample_source(const DeviceScene& sc, float3 direction) {
    if (sc.envImageCubeDim) {                                  // BackgroundSourceType 2: t_SrcCubemapEnvMap.SampleLevel(s_Linear, direction, 0)
        EnvCube src; src.texels = sc.envImageCube; src.dim = sc.envImageCubeDim; src.mipLevels = 1u; src._pad = 0u; src.mipOffset[0] = 0u;
        return xyz(env_cube_sample_level(src, direction, 0.0f));
    }
    if (!sc.envTex.w) return make_float3(0.f, 0.f, 0.f);      // BackgroundSourceType 0: no image (a procedural sky alone)
    float2 uv = world_to_latlong_map(direction);
    float mh = (float)sc.envTex.h;
    uv.y = clampf(uv.y, 0.5f / mh, 1.0f - 0.5f / mh);
    return xyz(sample_bilinear(sc, sc.envTex, 0, uv));
}
// EnvMap::EvalLocal (EnvMap.hlsli:82-85): TextureCube.SampleLevel on the baked cube
static inline float3 env_eval_local(const DeviceScene& sc, float3 localDir, float lod) {
    return xyz(env_cube_sample_level(sc.envCube, localDir, lod)) * sc.envColorMultiplier;
}

// ---- The hit definition, first half: a WATERTIGHT ray / triangle test (Woop, Benthin, Wald: "Watertight Ray/Triangle Intersection", JCGT 2013), both sides, tmin < t < tmax;
// (u, v) = the DXR barycentrics of vertices 1 and 2. DXR promises that a ray cannot slip between two triangles that share an edge or a vertex (what Bridge::traceScatterRay /
// traceVisibilityRay inherit from the API, PathTracerBridgeDonut.hlsli:993-1055). The vertices are translated to the ray origin and sheared into the ray's own frame (kz = the
// axis of the largest |d|, kx / ky the next two in cyclic order): a vertex's 2D position then depends on the vertex and the ray only, never on the triangle it is tested for.
// The three edge functions are formed WITHOUT fused products — a * b - c * d negates exactly when the edge is walked the other way, and round(p) - round(q) has the sign of
// p - q whenever it is not zero — and a zero is resolved by the exact residuals of the two products (fmaf(a, b, -p) is the error of p, exactly): every triangle around an edge or
// a vertex sees the same signs, a point on the boundary belongs to both sides, nothing falls between. Sz is the ray's correctly rounded reciprocal (ray_safe_rcp: the traversal
// holds it anyway; kz is the dominant axis, so the clamp never acts). Same text on both sides of the parity fence (the CPU restatement the tests check against holds it too); the traversal's leaf block (pt_traverse8p.h) performs the same operations on the same operands, with the axis
// permutation done by the order of its loads.
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
// The second half of the hit definition. Whether a box above the triangle lets the ray through must not decide the closest hit (a 120 m x 0.3 mm sliver seen at a grazing angle
// is found by the triangle test at a t that a conservative fp32 box test may or may not admit: found at 4K on C3 in round 1, 8 pixels of 2 M where two BVHs disagreed). So a hit
// only counts if it lies inside the triangle's own padded bounding box AS THE RAY SEES IT: the slab interval [tn, tf] of that box, computed with exactly the arithmetic every BVH
// node above it uses ((plane - o) * inv with inv = ray_safe_rcp(d), correctly rounded), must contain t. Rounding is monotone, every ancestor box contains this box (pad grows with
// the extent, see tri_pad / pad_box), hence every ancestor's interval contains [tn, tf] and therefore t: no conservative BVH over these boxes can cull an accepted hit, and the
// result equals the exhaustive loop bit for bit. (The watertight test places t within a few
// roundings of the triangle's plane, the pad is 10 - 100 x wider: the box never takes back what the first half found — tests/test_gpu_watertight.py counts escapes: 0.)
static inline float tri_pad(float3 mn, float3 mx, float scenePad) {
    float3 e = mx - mn;
    return 2e-5f * fmaxf_(e.x, fmaxf_(e.y, e.z)) + scenePad;
}
static inline float scene_pad(float3 smn, float3 smx) { return 2e-6f * length(smx - smn); }
static inline bool tri_box_accepts(const TriRecord& tr, float3 o, float3 inv, float t) {
    float3 mn, mx; tri_bounds(tr, mn, mx); mn = mn - make_float3(tr.pad); mx = mx + make_float3(tr.pad);
    float ax = (mn.x - o.x) * inv.x, bx = (mx.x - o.x) * inv.x, ay = (mn.y - o.y) * inv.y, by = (mx.y - o.y) * inv.y, az = (mn.z - o.z) * inv.z, bz = (mx.z - o.z) * inv.z;
    float tn = fmaxf_(fmaxf_(fminf_(ax, bx), fminf_(ay, by)), fminf_(az, bz));
    float tf = fminf_(fminf_(fmaxf_(ax, bx), fmaxf_(ay, by)), fmaxf_(az, bz));
    return (tn <= t) && (t <= tf);
}
static inline bool intersect_tri(const TriRecord& tr, float3 o, float3 d, float tmin, float tmax, float& t, float& u, float& v) {
    if (!intersect_tri_wt(tri_v0(tr), tri_v1(tr), tri_v2(tr), o, d, tmin, tmax, t, u, v)) return false;
    return tri_box_accepts(tr, o, make_float3(ray_safe_rcp(d.x), ray_safe_rcp(d.y), ray_safe_rcp(d.z)), t);
}

// AlphaTestImpl (BridgeDonut:929-971)
static inline bool alpha_test(const DeviceScene& sc, uint prim, float u, float v) {
    uint2 pi = sc.primInfo[prim];
    const SubInstanceData& si = sc.subInstances[pi.x];
    if ((si.FlagsAndAlphaInfo & SubInstanceData::Flags_AlphaTested) == 0) return true;
    const uint* idx = sc.indices + si.IndexOffset + pi.y * 3;
    float2 t0 = sc.uvs[si.TexCoord1Offset + idx[0]], t1 = sc.uvs[si.TexCoord1Offset + idx[1]], t2 = sc.uvs[si.TexCoord1Offset + idx[2]];
    float b0 = 1.0f - (u + v);
    float2 tc = (t0 * b0 + t1 * u) + t2 * v;
    float opacity = sample_bilinear(sc, sc.textures[si.AlphaTextureIndex()], 0, tc).w;
    return opacity >= si.AlphaCutoff();
}

#ifndef PT_ALPHA_LUT
#define PT_ALPHA_LUT 1      // 1: the opacity k / 255 of a byte plane comes from a 256-entry constant table, 0: from a division (A/B: the four divisions cost 4 % of k_extend)
#endif
#if PT_ALPHA_LUT
// k / 255.0f for k = 0..255 as a table (the quotients are formed by the compiler: IEEE division, the same floats the upload computes)
struct Unorm8Table { float v[256]; constexpr Unorm8Table() : v() { for (int k = 0; k < 256; k++) v[k] = (float)k / 255.0f; } };
__device__ __constant__ const Unorm8Table kUnorm8Table{};
static inline float unorm8_to_float(unsigned char k) { return kUnorm8Table.v[k]; }
#endif
// the same test from the build-time record of triangle slot `slot` (identical arithmetic on identical operands: only the .w channel of
// sample_bilinear at mip 0 is evaluated; the texture coordinates, cutoff and the place of the opacities come from the AlphaRec instead of the vertex streams / TexInfo)
static inline bool alpha_test_slot(const DeviceScene& sc, uint slot, float u, float v) {
    const AlphaRec r = sc.alphaRecs[slot];
    if (r.tex == 0xFFFFFFFFu) return true;
    float b0 = 1.0f - (u + v);
    float2 uv = (r.t0 * b0 + r.t1 * u) + r.t2 * v;
    uint mw = r.wh & 0xFFFFu; if (mw < 1u) mw = 1u;
    uint mh = r.wh >> 16; if (mh < 1u) mh = 1u;
    float fx = uv.x * (float)mw - 0.5f, fy = uv.y * (float)mh - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float ax = fx - flx, ay = fy - fly;
    // the wrap of sample_bilinear: flx - floorf(flx / mw) * mw. For a power-of-two side the quotient is an exact scaling, so the product with the exact reciprocal
    // (exponent arithmetic: 2^-k from 2^k) is the same float as the correctly rounded division, which costs 12 instructions in a VALU-bound loop
    const float fmw = (float)mw, fmh = (float)mh;
    if (((mw & (mw - 1u)) | (mh & (mh - 1u))) == 0u) {
        flx = flx - floorf(flx * asfloat(0x7F000000u - asuint(fmw))) * fmw; fly = fly - floorf(fly * asfloat(0x7F000000u - asuint(fmh))) * fmh;
    } else { flx = flx - floorf(flx / fmw) * fmw; fly = fly - floorf(fly / fmh) * fmh; }
    int x0 = (int)flx, y0 = (int)fly, x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 += (int)mw; if (x1 >= (int)mw) x1 -= (int)mw;
    if (y0 < 0) y0 += (int)mh; if (y1 >= (int)mh) y1 -= (int)mh;
    x0 = x0 < 0 ? 0 : (x0 >= (int)mw ? (int)mw - 1 : x0); x1 = x1 < 0 ? 0 : (x1 >= (int)mw ? (int)mw - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 >= (int)mh ? (int)mh - 1 : y0); y1 = y1 < 0 ? 0 : (y1 >= (int)mh ? (int)mh - 1 : y1);
    const uint r0 = (uint)y0 * mw, r1 = (uint)y1 * mw;
    float w00, w10, w01, w11;
    if (r.fmt == 0u) {          // opacities k / 255: the quotient the texture upload formed (correctly rounded division on both sides)
        const unsigned char* a = sc.alphaPool + r.plane;
#if PT_ALPHA_LUT
        w00 = unorm8_to_float(a[r0 + (uint)x0]); w10 = unorm8_to_float(a[r0 + (uint)x1]); w01 = unorm8_to_float(a[r1 + (uint)x0]); w11 = unorm8_to_float(a[r1 + (uint)x1]);
#else
        w00 = (float)a[r0 + (uint)x0] / 255.0f; w10 = (float)a[r0 + (uint)x1] / 255.0f; w01 = (float)a[r1 + (uint)x0] / 255.0f; w11 = (float)a[r1 + (uint)x1] / 255.0f;
#endif
    } else {
        const float* a = reinterpret_cast<const float*>(sc.alphaPool + r.plane);
        w00 = a[r0 + (uint)x0]; w10 = a[r0 + (uint)x1]; w01 = a[r1 + (uint)x0]; w11 = a[r1 + (uint)x1];
    }
    float opacity = lerpf(lerpf(w00, w10, ax), lerpf(w01, w11, ax), ay);
    return opacity >= r.cutoff;
}

#pragma clang force_cuda_host_device end
} // namespace ptk
