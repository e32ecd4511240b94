// ORACLE pin (test infrastructure only): what the integrator translation unit needs beyond hlsl_shim.h — the compile-time configuration the
// reference's pipeline baker passes as shader macros (Sample.cpp:988-1040; reference mode with the §8a parity knobs), and stand-ins for GPU
// resource types and the debug context. The switches that are shader macros in the reference but run-time settings in the oracle (diffuse BRDF model,
// Russian roulette, firefly filter, nested-dielectrics quality, LD sampler, NEE) come in as -D flags: one library per combination (oracle/ptref.py refpin_pt).
#pragma once
#define row_major
#ifndef PATH_TRACER_MODE
#define PATH_TRACER_MODE                                PATH_TRACER_MODE_REFERENCE      // -DPATH_TRACER_MODE=1 / 2: the stable-plane build / fill pass of the realtime mode (oracle/ptref.py refpin_pt)
#endif
#define NON_PATH_TRACING_PASS                           0
#define __SHADER_TARGET_MAJOR                           0
#define __SHADER_TARGET_MINOR                           0
#define ENABLE_DEBUG_SURFACE_VIZ                        0
#define ENABLE_DEBUG_LINES_VIZ                          0
#ifndef PT_NEE_ENABLED
#define PT_NEE_ENABLED                                  1
#endif
#define PT_USE_RESTIR_DI                                0
#define PT_USE_RESTIR_GI                                0
#define RTXPT_USE_APPROXIMATE_MIS                       0
#define RTXPT_DISCARD_NON_NEE_LIGHTING                  0
#define RTXPT_DISCARD_NEE_LIGHTING                      0
#ifndef RTXPT_LP_TYPES_USE_16BIT_PRECISION
#define RTXPT_LP_TYPES_USE_16BIT_PRECISION              0
#endif
#ifndef RTXPT_ENABLE_LOW_DISCREPANCY_SAMPLER_FOR_BSDF
#define RTXPT_ENABLE_LOW_DISCREPANCY_SAMPLER_FOR_BSDF   1
#endif
#define NEEAT_BAKER_ONLY                                0
#ifndef PT_ENABLE_RUSSIAN_ROULETTE
#define PT_ENABLE_RUSSIAN_ROULETTE                      1
#endif
#ifndef RTXPT_FIREFLY_FILTER
#define RTXPT_FIREFLY_FILTER                            1
#endif
#ifndef RTXPT_NESTED_DIELECTRICS_QUALITY
#define RTXPT_NESTED_DIELECTRICS_QUALITY                1
#endif
namespace hl {
struct RayDesc { float3 Origin; float TMin; float3 Direction; float TMax; };
// resources that the driver binds carry a pointer (+ pitch); unbound ones read as zero / swallow writes
template <class T> struct Texture2D { const T* p = nullptr; uint w = 0, h = 0; const ptref::Texture* tex = nullptr;      // tex: a material texture, filtered by the oracle's explicit trilinear fetch
    const ptref::SkyTexture* lut = nullptr;                                                                                 // lut: a look-up texture of the procedural sky (linear, wrap: ptref::sky_sample2d)
    T Sample(SamplerState, float2 uv) const { return sample_level(uv, 0.f); } T SampleLevel(SamplerState, float2 uv, float lod) const { return sample_level(uv, lod); } T SampleGrad(SamplerState, float2, float2, float2) const { return T(); }
    T sample_level(float2 uv, float lod) const;
    T Load(int3 c) const { return (p && (uint)c.x < w && (uint)c.y < h) ? p[(uint)c.y * w + (uint)c.x] : T(); }          // out-of-range loads return 0 (D3D)
    T Load(uint3 c) const { return Load(int3((int)c.x, (int)c.y, (int)c.z)); } T operator[](uint2 c) const { return Load(int3((int)c.x, (int)c.y, 0)); }
    void GetDimensions(uint& ow, uint& oh) const { ow = tex ? tex->w : w; oh = tex ? tex->h : h; } void GetDimensions(uint, uint& ow, uint& oh, uint& l) const { GetDimensions(ow, oh); l = tex ? tex->mipLevels : 1; } };
template <class T> T Texture2D<T>::sample_level(float2, float) const { return T(); }
template <> inline float4 Texture2D<float4>::sample_level(float2 uv, float lod) const { if (lut) { ptref::float4 c = ptref::sky_sample2d(*lut, uv.x, uv.y); return float4(c.x, c.y, c.z, c.w); } if (!tex) return float4(); ptref::float4 c = ptref::sample_trilinear(*tex, ptref::make_float2(uv.x, uv.y), lod); return float4(c.x, c.y, c.z, c.w); }
struct Texture3D { const ptref::SkyTexture* lut = nullptr;                                                                  // the sky's in-scatter and cloud volumes (linear, wrap: ptref::sky_sample3d)
    float4 SampleLevel(SamplerState, float3 uvw, float) const { if (!lut) return float4(); ptref::float4 c = ptref::sky_sample3d(*lut, ptref::make_float3(uvw.x, uvw.y, uvw.z)); return float4(c.x, c.y, c.z, c.w); } };
template <class T> struct TextureCube { T (*fetch)(const void*, float3, float) = nullptr; const void* ctx = nullptr;
    T SampleLevel(SamplerState, float3 dir, float lod) const { return fetch ? fetch(ctx, dir, lod) : T(); } };
template <class T> struct RWTexture2D { T* p = nullptr; uint w = 0; T dummy = T(); uint h = 0;      // h != 0: bounds-checked like a UAV (out-of-range writes dropped, reads 0)
    bool in(uint2 c) const { return p && (h == 0 || (c.x < w && c.y < h)); }
    T& operator[](uint2 c) { if (in(c)) return p[c.y * w + c.x]; dummy = T(); return dummy; } T operator[](uint2 c) const { return in(c) ? p[c.y * w + c.x] : T(); } void GetDimensions(uint& ow, uint& oh) const { ow = w; oh = 1; } };
template <class T> struct RWTexture2DArray { T* p = nullptr; uint w = 0, h = 0; T dummy = T();      // bound (the stable-plane header): [slice][y][x]
    T& operator[](uint3 c) { return p ? p[((size_t)c.z * h + c.y) * w + c.x] : dummy; } T operator[](uint3 c) const { return p ? p[((size_t)c.z * h + c.y) * w + c.x] : dummy; } };
template <class T> struct RWTexture3D { T dummy; T& operator[](uint3) { return dummy; } };
template <class T> struct StructuredBuffer { const T* p = nullptr; const T& operator[](uint i) const { return p[i]; } };
template <class T> struct RWStructuredBuffer { T* p = nullptr; T& operator[](uint i) const { return p[i]; } };
template <class T> struct Buffer { const T* p = nullptr; uint n = 0; T operator[](uint i) const { return (n && i >= n) ? T() : p[i]; } };      // n != 0: loads beyond the buffer return 0 (D3D typed-buffer behaviour)
template <class T> struct RWBuffer { T* p = nullptr; T& operator[](uint i) const { return p[i]; } };
struct ByteAddressBuffer { const unsigned char* p = nullptr;
    uint Load(uint o) const { uint v; memcpy(&v, p + o, 4); return v; } uint2 Load2(uint o) const { return uint2(Load(o), Load(o + 4)); } uint3 Load3(uint o) const { return uint3(Load(o), Load(o + 4), Load(o + 8)); } };
struct RWByteAddressBuffer { uint* p = nullptr; uint Load(uint o) const { return p[o / 4]; } void Store(uint o, uint v) const { p[o / 4] = v; } };
struct RaytracingAccelerationStructure {};
static const uint RAY_FLAG_NONE = 0, RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH = 4, RAY_FLAG_CULL_NON_OPAQUE = 0x80;
template <uint F, uint G = 0> struct RayQuery {};
struct GeometryInstanceIDPin { uint inst = 0, geom = 0; uint getInstanceIndex() const { return inst; } uint getGeometryIndex() const { return geom; } };
struct TriangleHit { GeometryInstanceIDPin instanceID; uint primitiveIndex = 0; float2 barycentrics; };
struct PackedHitInfo { uint4 d; };
// PathTracerDebug.hlsli's context: the integrator only ever calls into it behind ENABLE_DEBUG_* switches that are off here
struct DebugConstantsPin { uint exploreDeltaTree = 0; };
struct DebugContext { DebugConstantsPin constants; uint2 pixelPos; bool IsDebugPixel() const { return false; } bool IsDebugPixel(uint2) const { return false; } void Reset(uint) {} void Reset(uint2, int) {} void SetPickedMaterial(uint) {}
    template <class... A> void DrawDebugViz(A...) {} };
static inline void DebugCross(float3, float, float4) {}
static inline bool isfinite(float v) { uint u; memcpy(&u, &v, 4); return (u & 0x7F800000u) != 0x7F800000u; }      // HLSL isfinite (PathTracerSample.hlsl FirstHitFromVBuffer)
static inline float max3(float a, float b, float c) { return max(a, max(b, c)); }              // Utils/ColorHelpers.hlsli:19-27
static inline float max3(float3 v) { return max3(v.x, v.y, v.z); }
// DXR system values of the closest-hit shader HandleHit runs in: set by the driver loop before each call
static thread_local uint g_hitInstanceIndex = 0, g_hitGeometryIndex = 0, g_hitPrimitiveIndex = 0;
static inline uint InstanceIndex() { return g_hitInstanceIndex; } static inline uint GeometryIndex() { return g_hitGeometryIndex; } static inline uint PrimitiveIndex() { return g_hitPrimitiveIndex; }
} // namespace hl
