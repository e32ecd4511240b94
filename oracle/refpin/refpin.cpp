// ORACLE pin (test infrastructure only): compiles the reference's OWN host-compilable sources, from where they lie
// under /root/reference (include path set by oracle/Makefile; nothing is copied into this repo), behind a C API so that
// tests can check the oracle's restatement against real reference code and generate golden vectors
// (tests/golden/make_refpin_golden.py). Sources compiled:
//   Rtxpt/Shaders/PathTracer/Utils/NoiseAndSequences.hlsli   (Hash32, Hash32Combine, Hash32ToFloat; C++ branch :461-524 SobolC, PrecomputeSobol)
//   Rtxpt/Shaders/PathTracer/PathTracerShared.h               (struct layouts; BridgeCamera :109-141)
//   Rtxpt/Shaders/PathTracer/Lighting/PolymorphicLight.h      (packed light record layouts, type codes)
//   Rtxpt/Shaders/PathTracer/Materials/MaterialPT.h           (PTMaterialData layout + flag values)
//   Rtxpt/Shaders/SubInstanceData.h                           (SubInstanceData layout)
// The reference headers expect HLSL-style vector types / donut::math; the minimal shim below provides just those names.
#include <cstdint>
#include <cmath>
#include <cstring>
#include <cstddef>

typedef uint32_t uint;
struct uint2 { uint x, y; uint2() : x(0), y(0) {} uint2(uint a, uint b) : x(a), y(b) {} };
struct uint3 { uint x, y, z; uint3() : x(0), y(0), z(0) {} uint3(uint a, uint b, uint c) : x(a), y(b), z(c) {} };
static inline uint2& operator+=(uint2& a, uint2 b) { a.x += b.x; a.y += b.y; return a; }
static inline float saturate(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
static inline float pow(float a, float b) { return powf(a, b); }
struct float2 { float x, y; float2() : x(0), y(0) {} float2(float a, float b) : x(a), y(b) {} };
struct float3 { float x, y, z; float3() : x(0), y(0), z(0) {} float3(float a, float b, float c) : x(a), y(b), z(c) {} };
struct float4 { float x, y, z, w; };
struct float3x4 { float m[12]; };
static inline float3 operator*(float3 a, float b) { return float3(a.x * b, a.y * b, a.z * b); }
static inline float3& operator*=(float3& a, float b) { a = a * b; return a; }
static inline float2 operator*(float2 a, float2 b) { return float2(a.x * b.x, a.y * b.y); }
static inline float3 pow(float3 a, float e) { return float3(powf(a.x, e), powf(a.y, e), powf(a.z, e)); }
static inline float3 abs(float3 a) { return float3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline float3 operator+(float3 a, float b) { return float3(a.x + b, a.y + b, a.z + b); }
namespace donut { namespace math {
    typedef ::float3 float3; typedef ::float2 float2; typedef ::uint2 uint2; typedef ::float3x4 float3x4; typedef ::float4 float4; typedef ::uint3 uint3;
    static inline float3 normalize(float3 v) { float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); return float3(v.x / l, v.y / l, v.z / l); }
    static inline float3 cross(float3 a, float3 b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
} }

#include "PathTracer/Utils/NoiseAndSequences.hlsli"
#include "PathTracer/PathTracerShared.h"
#include "PathTracer/Lighting/PolymorphicLight.h"
#include "PathTracer/Materials/MaterialPT.h"
#define STATIC_ASSERT(x) static_assert(x, #x)
#include "SubInstanceData.h"

extern "C" {
uint32_t refpin_hash32(uint32_t x) { return Hash32(x); }
uint32_t refpin_hash32_combine(uint32_t s, uint32_t v) { return Hash32Combine(s, v); }
float refpin_hash32_to_float(uint32_t x) { return Hash32ToFloat(x); }
uint32_t refpin_sobol(uint32_t index, uint32_t dim) { return SobolC(index, dim); }
void refpin_bridge_camera(uint32_t w, uint32_t h, float aspect, const float* pos, const float* dir, const float* up, float fovY, float nearZ, float farZ,
                          float focalDistance, float apertureRadius, const float* jitter, void* out112) {
    PathTracerCameraData d = BridgeCamera(w, h, aspect, float3(pos[0], pos[1], pos[2]), float3(dir[0], dir[1], dir[2]), float3(up[0], up[1], up[2]), fovY, nearZ, farZ,
                                          focalDistance, apertureRadius, float2(jitter[0], jitter[1]));
    memcpy(out112, &d, sizeof(d));
}
float refpin_eval_mis(int heuristic, float n0, float p0, float n1, float p1) { return EvalMIS((MISHeuristic)heuristic, n0, p0, n1, p1); }
// struct sizes / offsets / constants as the reference compiles them
void refpin_layout(uint32_t* out) {
    int i = 0;
    out[i++] = sizeof(PathTracerCameraData);
    out[i++] = sizeof(PathTracerConstants);
    out[i++] = sizeof(PolymorphicLightInfo);
    out[i++] = sizeof(PolymorphicLightInfoEx);
    out[i++] = sizeof(PTMaterialData);
    out[i++] = sizeof(SubInstanceData);
    out[i++] = (uint32_t)offsetof(PTMaterialData, IoR);
    out[i++] = (uint32_t)offsetof(PTMaterialData, Volume);
    out[i++] = (uint32_t)offsetof(PTMaterialData, BaseOrDiffuseTextureIndex);
    out[i++] = (uint32_t)offsetof(PathTracerCameraData, ViewportSize);
    out[i++] = (uint32_t)offsetof(PathTracerCameraData, Jitter);
    out[i++] = PTMaterialFlags_ThinSurface;
    out[i++] = PTMaterialFlags_UseBaseOrDiffuseTexture;
    out[i++] = PTMaterialFlags_UseEmissiveTexture;
    out[i++] = PTMaterialFlags_UseNormalTexture;
    out[i++] = PTMaterialFlags_UseMetalRoughOrSpecularTexture;
    out[i++] = PTMaterialFlags_UseTransmissionTexture;
    out[i++] = PTMaterialFlags_NestedPriorityShift;
    out[i++] = (uint32_t)PolymorphicLightType::kTriangle;
    out[i++] = (uint32_t)PolymorphicLightType::kEnvironmentQuad;
    out[i++] = kPolymorphicLightTypeShift;
    out[i++] = (uint32_t)SubInstanceData::Flags_AlphaTested;
    out[i++] = (uint32_t)SubInstanceData::Flags_ExcludeFromNEE;
    out[i++] = PATH_TRACER_MAX_PAYLOAD_SIZE;
}
}
