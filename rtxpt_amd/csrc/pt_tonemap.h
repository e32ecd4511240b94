// mi355pt — display path after accumulation, device + host:
//   tone mapping  Rtxpt/ToneMapper/ToneMapping.ps.hlsli:31-176, constants Rtxpt/ToneMapper/ToneMapping_cb.h:17-45,
//   colour transform / manual exposure  Rtxpt/ToneMapper/ToneMappingPasses.cpp:428-441 (defaults ToneMappingPasses.h:36-53),
//   LDR target = SRGBA8_UNORM (Rtxpt/SampleCommon/RenderTargets.cpp:241): linear -> sRGB on write, round to nearest 8-bit.
// Arithmetic contract: single fp32 operations in the written order, dm_pow for pow().
#pragma once
#include "pt_vec.h"
#include "pt_dmath.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

enum ToneMapOperator : uint { TM_Linear = 0, TM_Reinhard = 1, TM_ReinhardModified = 2, TM_HejiHableAlu = 3, TM_HableUc2 = 4, TM_Aces = 5 };

struct ToneMapParams {            // 80 bytes; mirrors ToneMappingConstants with the colour transform as the 3x3 used by mul(color, M)
    float whiteScale, whiteMaxLuminance; uint toneMapOperator, clamped;
    uint autoExposure; float avgLuminance, autoExposureLumValueMin, autoExposureLumValueMax;
    float colorTransform[9];     // row-major M, result_j = sum_i color_i * M[i][j]
    uint enabled, _pad0, _pad1;
};
static_assert(sizeof(ToneMapParams) == 80, "ToneMapParams must be 80 bytes");

static inline float tm_luminance(float3 c) { return (c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f; }          // calcLuminance, ToneMapping.ps.hlsli:31-34
static inline float3 tm_uc2(float3 c) {                                                                        // applyUc2Curve :72-84
    const float A = 0.22f, B = 0.3f, C = 0.1f, D = 0.2f, E = 0.01f, F = 0.3f;
    float3 num = c * (c * A + make_float3(C * B)) + make_float3(D * E);
    float3 den = c * (c * A + make_float3(B)) + make_float3(D * F);
    return make_float3(num.x / den.x, num.y / den.y, num.z / den.z) - make_float3(E / F);
}
static inline float3 tm_operator(const ToneMapParams& p, float3 c) {                                           // toneMap :113-132
    switch (p.toneMapOperator) {
    case TM_Reinhard: { float l = tm_luminance(c); float r = l / (l + 1.0f); return c * (r / l); }                                        // :43-48
    case TM_ReinhardModified: { float l = tm_luminance(c); float r = (l * (1.0f + l / (p.whiteMaxLuminance * p.whiteMaxLuminance))) * (1.0f + l); return c * (r / l); }   // :51-56 (as written in the reference)
    case TM_HejiHableAlu: {                                                                                                              // :60-67
        float3 x = make_float3(fmaxf_(0.0f, c.x - 0.004f), fmaxf_(0.0f, c.y - 0.004f), fmaxf_(0.0f, c.z - 0.004f));
        float3 num = x * (x * 6.2f + make_float3(0.5f)), den = x * (x * 6.2f + make_float3(1.7f)) + make_float3(0.06f);
        return make_float3(dm_pow(num.x / den.x, 2.2f), dm_pow(num.y / den.y, 2.2f), dm_pow(num.z / den.z, 2.2f)); }
    case TM_HableUc2: { float3 v = tm_uc2(c * 2.0f); float ws = 1.0f / tm_uc2(make_float3(p.whiteScale)).x; return v * ws; }            // :86-94
    case TM_Aces: {                                                                                                                      // :96-111
        float3 x = c * 0.6f; const float A = 2.51f, B = 0.03f, C = 2.43f, D = 0.59f, E = 0.14f;
        float3 num = x * (x * A + make_float3(B)), den = x * (x * C + make_float3(D)) + make_float3(E);
        return make_float3(saturate(num.x / den.x), saturate(num.y / den.y), saturate(num.z / den.z)); }
    default: return c;                                                                                                                    // Linear :37-40
    }
}
static inline float3 tm_apply(const ToneMapParams& p, float3 c) {                                              // applyToneMapping :136-174
    if (p.autoExposure) {                                                                                      // TONEMAPPING_AUTOEXPOSURE_CPU == 1: avgLuminance comes from the host
        float s = clampf(0.042f / p.avgLuminance, p.autoExposureLumValueMin, p.autoExposureLumValueMax);
        c = c * s;
    }
    if (p.enabled) {
        const float* M = p.colorTransform;
        c = make_float3((c.x * M[0] + c.y * M[3]) + c.z * M[6], (c.x * M[1] + c.y * M[4]) + c.z * M[7], (c.x * M[2] + c.y * M[5]) + c.z * M[8]);
        c = tm_operator(p, c);
        if (p.clamped) c = make_float3(saturate(c.x), saturate(c.y), saturate(c.z));
    }
    return c;
}
// SRGBA8_UNORM store: D3D linear -> sRGB transfer, then UNORM8 round-to-nearest (NaN -> 0)
static inline uint tm_srgb8(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 1.0f) return 255u;
    float s = (v <= 0.0031308f) ? v * 12.92f : 1.055f * dm_pow(v, 1.0f / 2.4f) - 0.055f;
    return (uint)(s * 255.0f + 0.5f);
}
static inline uint tm_pixel(const ToneMapParams& p, float4 radiance) {       // -> R | G<<8 | B<<16 | A<<24
    float3 c = tm_apply(p, make_float3(radiance.x, radiance.y, radiance.z));
    float a = saturate(radiance.w);
    return tm_srgb8(c.x) | (tm_srgb8(c.y) << 8) | (tm_srgb8(c.z) << 16) | ((uint)(a * 255.0f + 0.5f) << 24);
}

// ---- auto exposure, luminance capture (ToneMappingPasses.cpp:78-97, 225-288): the colour target is drawn into a power-of-two-lowered R32F target as
// log2(max(1e-4, luminance)) (luminance_ps.hlsl:10-26, linear sampler at the quad's uv), the mip chain is generated and the last (1x1) mip is read back;
// avgLuminance = exp2 of it. The sampler's bilinear filter and Donut's MipMapGenPass (2x2 averages) are restated in fp32; texture units use fixed-point
// weights, so parity with the reference here is to tolerance, not bit-exact.
static inline uint tm_pow2_floor(uint v) { uint r = 1u; while ((r << 1) != 0u && (r << 1) <= v) r <<= 1; return r; }
static inline float tm_log_luminance_texel(const float4* accum, uint W, uint H, uint LW, uint LH, uint x, uint y) {
    float u = ((float)x + 0.5f) / (float)LW, v = ((float)y + 0.5f) / (float)LH;
    float sx = u * (float)W - 0.5f, sy = v * (float)H - 0.5f;
    float bx = floorf(sx), by = floorf(sy), fx = sx - bx, fy = sy - by;
    int x0 = (int)bx, y0 = (int)by, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > (int)W - 1 ? (int)W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > (int)W - 1 ? (int)W - 1 : x1);          // clamp addressing
    y0 = y0 < 0 ? 0 : (y0 > (int)H - 1 ? (int)H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > (int)H - 1 ? (int)H - 1 : y1);
    float4 a = accum[(size_t)y0 * W + x0], b = accum[(size_t)y0 * W + x1], c = accum[(size_t)y1 * W + x0], d = accum[(size_t)y1 * W + x1];
    float3 top = make_float3(a.x, a.y, a.z) * (1.0f - fx) + make_float3(b.x, b.y, b.z) * fx, bot = make_float3(c.x, c.y, c.z) * (1.0f - fx) + make_float3(d.x, d.y, d.z) * fx;
    float3 col = top * (1.0f - fy) + bot * fy;
    float lum = (col.x * 0.299f + col.y * 0.587f) + col.z * 0.114f;
    return dm_log2(fmaxf_(0.0001f, lum));
}
static inline float tm_mip_texel(const float* src, uint w, uint h, uint x, uint y) {            // one texel of the next mip: 2x2 average, edge texels repeat when a side is already 1
    uint x0 = 2u * x, y0 = 2u * y, x1 = x0 + 1u < w ? x0 + 1u : w - 1u, y1 = y0 + 1u < h ? y0 + 1u : h - 1u;
    return ((src[(size_t)y0 * w + x0] + src[(size_t)y0 * w + x1]) * 0.5f + (src[(size_t)y1 * w + x0] + src[(size_t)y1 * w + x1]) * 0.5f) * 0.5f;
}

#pragma clang force_cuda_host_device end
} // namespace ptk
