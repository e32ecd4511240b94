// Host side of the analytic lights: what LightsBaker::ConvertLight (Rtxpt/Lighting/LightsBaker.cpp:456-556) does with a Donut PointLight / SpotLight
// (+ RTXPT's LightExtension) before the record reaches the light buffer — pt_convert_light produces the PolymorphicLightInfo(+Ex) pair that
// pt_set_lights takes. Restated with the host helpers it uses: packLightColor :397-413 (libm log2f / exp2f, R8G8B8 with round-half-up), the host
// NDirToOctUnorm32 :415-437 and the truncating fp32ToFp16 :439-455 (this one does NOT round to nearest, unlike the shader-side f32tof16).
#include "../../include/mi355pt.h"
#include <cmath>
#include <cstring>
#include <cstdint>
#include <algorithm>

namespace {
const uint32_t kTypeShift = 24, kShapingEnableBit = 1u << 28, kShapingUseMinFalloff = 1u << 30;          // PolymorphicLight.h:19-24
const uint32_t kSphere = 0, kPoint = 4;                                                                    // PolymorphicLightType
const float kMinLog2Radiance = -8.f, kMaxLog2Radiance = 40.f, PI_f = 3.141592654f;                         // PolymorphicLight.h:26-27; donut/core/math/basics.h
inline float saturate(float v) { return std::min(std::max(v, 0.f), 1.f); }
inline uint32_t floatToUInt(float v, float scale) { return (uint32_t)floorf(v * scale + 0.5f); }
inline uint32_t FLOAT3_to_R8G8B8_UNORM(float x, float y, float z) {
    return (floatToUInt(saturate(x), 0xFF) & 0xFF) | ((floatToUInt(saturate(y), 0xFF) & 0xFF) << 8) | ((floatToUInt(saturate(z), 0xFF) & 0xFF) << 16);
}
void packLightColor(const float c[3], PolymorphicLightInfo& li) {
    float maxRadiance = std::max(c[0], std::max(c[1], c[2]));
    if (maxRadiance <= 0.f) return;
    float logRadiance = (::log2f(maxRadiance) - kMinLog2Radiance) / (kMaxLog2Radiance - kMinLog2Radiance);
    logRadiance = saturate(logRadiance);
    uint32_t packedRadiance = std::min(uint32_t(ceilf(logRadiance * 65534.f)) + 1, 0xffffu);
    float unpackedRadiance = ::exp2f((float(packedRadiance - 1) / 65534.f) * (kMaxLog2Radiance - kMinLog2Radiance) + kMinLog2Radiance);
    li.ColorTypeAndFlags |= FLOAT3_to_R8G8B8_UNORM(c[0] / unpackedRadiance, c[1] / unpackedRadiance, c[2] / unpackedRadiance);
    li.LogRadiance |= packedRadiance;
}
uint32_t NDirToOctUnorm32(const float n3in[3]) {
    float s = fabsf(n3in[0]) + fabsf(n3in[1]) + fabsf(n3in[2]);
    float n3[3] = {n3in[0] / s, n3in[1] / s, n3in[2] / s};
    float nx = n3[0], ny = n3[1];
    if (!(n3[2] >= 0.0f)) { float wx = (1.0f - fabsf(ny)) * ((nx >= 0.0f) ? 1.0f : -1.0f), wy = (1.0f - fabsf(nx)) * ((ny >= 0.0f) ? 1.0f : -1.0f); nx = wx; ny = wy; }
    nx = nx * 0.5f + 0.5f; ny = ny * 0.5f + 0.5f;                       // Encode_Oct
    float px = saturate(nx * 0.5f + 0.5f), py = saturate(ny * 0.5f + 0.5f);      // (and once more in NDirToOctUnorm32, as the reference's host copy does)
    return uint32_t(px * 0xfffe) | (uint32_t(py * 0xfffe) << 16);
}
uint16_t fp32ToFp16(float v) {
    union FU { uint32_t ui; float f; }; FU multiple; multiple.ui = 0x07800000u;          // 2^-112
    FU biased; biased.f = v * multiple.f;
    const uint32_t u = biased.ui, sign = u & 0x80000000u, body = u & 0x0fffffffu;
    return (uint16_t)(sign >> 16 | body >> 13) & 0xFFFF;
}
inline float radians(float deg) { return deg * (PI_f / 180.f); }
}

extern "C" int32_t pt_convert_light(const PtAnalyticLightDesc* l, PolymorphicLightInfo* base, PolymorphicLightInfoEx* ex) {
    if (!l || !base || !ex || l->type > 1u) return PT_ERROR_INVALID_ARGUMENT;
    PolymorphicLightInfo p; memset(&p, 0, sizeof(p)); PolymorphicLightInfoEx e; memset(&e, 0, sizeof(e));
    // Donut hands position and direction over in double precision; the direction is normalised there and then narrowed (`float3(normalize(GetDirection()))`)
    double d[3] = {l->direction[0], l->direction[1], l->direction[2]};
    double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float dir[3] = {(float)(d[0] / dl), (float)(d[1] / dl), (float)(d[2] / dl)};
    if (l->type == 1u) {                 // LightType_Spot
        if (l->radius == 0.f) {          // (the reference asserts here: "not tested with radius == 0")
            float flux[3] = {l->color[0] * l->intensity, l->color[1] * l->intensity, l->color[2] * l->intensity};
            p.ColorTypeAndFlags = kPoint << kTypeShift | ((l->outerAngle < 0) ? kShapingUseMinFalloff : 0u);
            packLightColor(flux, p);
            memcpy(p.Center, l->position, 12);
            p.Direction1 = NDirToOctUnorm32(dir);
            p.Direction2 = fp32ToFp16(radians(fabsf(l->outerAngle)));
            p.Direction2 |= (uint32_t)fp32ToFp16(radians(l->innerAngle)) << 16;
        } else {
            float projectedArea = PI_f * (l->radius * l->radius);
            float radiance[3] = {l->color[0] * l->intensity / projectedArea, l->color[1] * l->intensity / projectedArea, l->color[2] * l->intensity / projectedArea};
            float softness = saturate(1.f - l->innerAngle / fabsf(l->outerAngle));
            p.ColorTypeAndFlags = kSphere << kTypeShift | ((l->outerAngle < 0) ? kShapingUseMinFalloff : 0u);
            p.ColorTypeAndFlags |= kShapingEnableBit;
            packLightColor(radiance, p);
            memcpy(p.Center, l->position, 12);
            p.Scalars = fp32ToFp16(l->radius);
            if (fabsf(l->outerAngle) > 0) {
                p.ColorTypeAndFlags |= kShapingEnableBit;
                e.PrimaryAxis = NDirToOctUnorm32(dir);
                e.CosConeAngleAndSoftness = fp32ToFp16(cosf(radians(fabsf(l->outerAngle))));
                e.CosConeAngleAndSoftness |= (uint32_t)fp32ToFp16(softness) << 16;
            }
            packLightColor(radiance, p);          // (twice in the reference: the OR makes the second call idempotent)
        }
    } else {                             // LightType_Point
        if (l->radius == 0.f) {
            float flux[3] = {l->color[0] * l->intensity, l->color[1] * l->intensity, l->color[2] * l->intensity};
            p.ColorTypeAndFlags = kPoint << kTypeShift;
            packLightColor(flux, p);
            memcpy(p.Center, l->position, 12);
            p.Direction2 = fp32ToFp16(PI_f) | (uint32_t)fp32ToFp16(0.0f) << 16;
        } else {
            float projectedArea = PI_f * (l->radius * l->radius);
            float radiance[3] = {l->color[0] * l->intensity / projectedArea, l->color[1] * l->intensity / projectedArea, l->color[2] * l->intensity / projectedArea};
            p.ColorTypeAndFlags = kSphere << kTypeShift;
            packLightColor(radiance, p);
            memcpy(p.Center, l->position, 12);
            p.Scalars = fp32ToFp16(l->radius);
        }
    }
    *base = p; *ex = e;
    return PT_OK;
}

// Sample::UpdateLighting (Rtxpt/Sample.cpp:1361-1388): the scene's directional lights as EnvMapBaker takes them. The angular size is raised to what the cube
// can resolve (pi / (cubeDim / 2)), and the direction goes into the environment's local frame (`rotationTransform.transformVector`, which is what the
// shader's EnvMap::ToLocal computes with InvTransform: EnvMap.hlsli:78-81) so that the disc keeps its world direction when the environment is rotated.
extern "C" int32_t pt_env_bake_lights(const PtEnvDirectionalLight* world, uint32_t n, const PtEnvMapSceneParams* params, uint32_t cubeDim, PtEnvDirectionalLight* out) {
    if ((n && (!world || !out)) || !cubeDim) return PT_ERROR_INVALID_ARGUMENT;
    const float minAngularSize = PI_f / ((float)cubeDim / 2.0f);
    for (uint32_t i = 0; i < n; i++) {
        PtEnvDirectionalLight l = world[i];
        l.AngularSize = std::max(l.AngularSize, minAngularSize);
        if (params) {                                 // InvTransform = transpose of the rotation in Transform (row r of Transform: params->Transform[4 r .. 4 r + 2])
            const float* T = params->Transform; const float x = world[i].Direction[0], y = world[i].Direction[1], z = world[i].Direction[2];
            for (int j = 0; j < 3; j++) l.Direction[j] = (x * T[4 * j + 0] + y * T[4 * j + 1]) + z * T[4 * j + 2];       // mul(dir, (float3x3)InvTransform), InvTransform[i][j] = Transform[j][i]
        }
        out[i] = l;
    }
    return PT_OK;
}
