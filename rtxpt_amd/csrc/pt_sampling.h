// mi355pt device/host leaf library — sampling and mapping helpers
// Part of the PRODUCT path (libmi355pt.so). Written to the arithmetic contract stated in pt_vec.h so that the HIP kernels
// reproduce the reference estimator bit-for-bit against the independent CPU oracle used by the tests.
// Reference anchors are cited per function (paths relative to /root/reference/Rtxpt/Shaders/PathTracer/ unless noted).
// Restates Rtxpt/Shaders/PathTracer/Utils/Math/MathHelpers.hlsli:185-229 (equal-area octahedral maps),
// :238-246 (sample_disk), :288-316 (concentric disk / cosine hemisphere), :436-448 (perp_stark),
// Rtxpt/Shaders/PathTracer/Utils/Geometry.hlsli:17-40 (BranchlessONB, SampleTriangleUniform), :79-82 (pdfAtoW),
// Rtxpt/Shaders/PathTracer/PathTracerHelpers.hlsli:29-42 (ComputeRayOrigin).
#pragma once
#include "pt_dmath.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

// MathHelpers.hlsli:185-200
static inline float2 ndir_to_oct_equal_area_unorm(float3 n) {
    float r = sqrtf_(1.f - fabsf(n.z));
    float phi = dm_atan2(fabsf(n.y), fabsf(n.x));
    float2 p;
    p.y = r * phi * K_2_PI;
    p.x = r - p.y;
    if (n.z < 0.f) { float2 q = make_float2(1.f - p.y, 1.f - p.x); p = q; }
    p.x *= signf_(n.x); p.y *= signf_(n.y);
    return make_float2(saturate(p.x * 0.5f + 0.5f), saturate(p.y * 0.5f + 0.5f));
}
// MathHelpers.hlsli:207-229
static inline float3 oct_to_ndir_equal_area_unorm(float2 p) {
    p = make_float2(p.x * 2.f - 1.f, p.y * 2.f - 1.f);
    float d = 1.f - (fabsf(p.x) + fabsf(p.y));
    float r = 1.f - fabsf(d);
    float phi = (r > 0.f) ? ((fabsf(p.y) - fabsf(p.x)) / r + 1.f) * K_PI_4 : 0.f;
    float f = r * sqrtf_(2.f - r * r);
    float s, c; dm_sincos(phi, s, c);
    float x = f * signf_(p.x) * c;
    float y = f * signf_(p.y) * s;
    float z = signf_(d) * (1.f - r * r);
    return make_float3(x, y, z);
}
// MathHelpers.hlsli:238-246
static inline float2 sample_disk(float2 u) {
    float r = sqrtf_(u.x);
    float phi = K_2PI * u.y;
    float s, c; dm_sincos(phi, s, c);
    return make_float2(r * c, r * s);
}
// MathHelpers.hlsli:288-305
static inline float2 sample_disk_concentric(float2 u) {
    u = make_float2(2.f * u.x - 1.f, 2.f * u.y - 1.f);
    if (u.x == 0.f && u.y == 0.f) return u;
    float phi, r;
    if (fabsf(u.x) > fabsf(u.y)) { r = u.x; phi = (u.y / u.x) * K_PI_4; }
    else { r = u.y; phi = K_PI_2 - (u.x / u.y) * K_PI_4; }
    float s, c; dm_sincos(phi, s, c);
    return make_float2(r * c, r * s);
}
// MathHelpers.hlsli:311-317
static inline float3 sample_cosine_hemisphere_concentric(float2 u, float& pdf) {
    float2 d = sample_disk_concentric(u);
    float z = sqrtf_(fmaxf_(0.f, 1.f - dot(d, d)));
    pdf = z * K_1_PI;
    return make_float3(d.x, d.y, z);
}
// MathHelpers.hlsli:436-448
static inline float3 perp_stark(float3 u) {
    float3 a = abs3(u);
    uint uyx = (a.x - a.y) < 0 ? 1u : 0u;
    uint uzx = (a.x - a.z) < 0 ? 1u : 0u;
    uint uzy = (a.y - a.z) < 0 ? 1u : 0u;
    uint xm = uyx & uzx;
    uint ym = (1u ^ xm) & uzy;
    uint zm = 1u ^ (xm | ym);
    return normalize(cross(u, make_float3((float)xm, (float)ym, (float)zm)));
}
// Geometry.hlsli:17-25
static inline void BranchlessONB(float3 normal, float3& tangent, float3& bitangent) {
    float sign = (normal.z >= 0) ? 1.f : -1.f;
    float a = -1.0f / (sign + normal.z);
    float b = normal.x * normal.y * a;
    tangent = make_float3(1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
    bitangent = make_float3(b, sign + normal.y * normal.y * a, -normal.y);
}
// Geometry.hlsli:33-40
static inline float3 SampleTriangleUniform(float2 rnd) {
    float sqrtx = sqrtf_(rnd.x);
    return make_float3(1 - sqrtx, sqrtx * (1 - rnd.y), sqrtx * rnd.y);
}
// Geometry.hlsli:79-82
static inline float pdfAtoW(float pdfA, float distance_, float cosTheta) {
    return pdfA * sq(distance_) / fmaxf_(cosTheta, 2e-9f);
}
// PathTracerHelpers.hlsli:29-42 (Waechter & Binder, Ray Tracing Gems ch. 6)
static inline float ComputeRayOrigin1(float p, float n) {
    const float origin = 1.f / 16.f, fScale = 3.f / 65536.f, iScale = 3 * 256.f;
    int iOff = (int)(n * iScale);
    float iPos = asfloat(asint(p) + ((p < 0.f) ? -iOff : iOff));
    float fOff = n * fScale;
    return (fabsf(p) < origin) ? (p + fOff) : iPos;
}
static inline float3 ComputeRayOrigin(float3 worldPosition, float3 faceNormal) {
    return make_float3(ComputeRayOrigin1(worldPosition.x, faceNormal.x), ComputeRayOrigin1(worldPosition.y, faceNormal.y),
                       ComputeRayOrigin1(worldPosition.z, faceNormal.z));
}
// Utils.hlsli:392-437 — only the Balance heuristic is used by NEE (LightSampler.hlsli:26)
static inline float EvalMIS_Balance(float n0, float p0, float n1, float p1) {
    float q0 = n0 * p0, q1 = n1 * p1;
    return saturate(q0 / (q0 + q1));
}

#pragma clang force_cuda_host_device end
} // namespace ptk
