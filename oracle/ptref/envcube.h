// ORACLE (test infrastructure only) — the baked environment cube, restated. Twin of rtxpt_amd/csrc/pt_envcube.h.
#pragma once
#include "dmath.h"
#include "vec.h"

namespace ptref {

// Environment cube: what EnvMapBaker turns the lat-long source into and what the path tracer samples (Rtxpt/Lighting/Distant/EnvMapBaker.hlsl:71-118,
// 166-246, 268-371; EnvMapBaker.cpp:298-343, 425-620; Rtxpt/Shaders/PathTracer/Lighting/EnvMap.hlsli:54-93): a cube of RGBA16F texels with a solid-angle
// weighted mip chain down to 8x8, the scene's directional lights rasterised into it as anti-aliased discs, radiance scaled by c_envMapRadianceScale = 1/4
// (Sample.cpp:88; the host compensates in EnvMapSceneParams::ColorMultiplier, :1936-1948) and clamped to the fp16 range. The reference may additionally
// BC6H-compress the cube (lossy; on by default on D3D12, off on Vulkan): restated too, as a round trip through its encoder and the BC6H decode at bake time (below).
// Layout in memory: RGBA16F texels packed into uint2 (x | y<<16, z | w<<16), [mip][face][y][x]; faces +X -X +Y -Y +Z -Z.
// What the texture unit does is implementation-defined and restated as: face = major axis (ties x, y, z in that order), bilinear taps clamped to the face
// (no filtering across face edges), linear interpolation between the two nearest mips.
struct EnvDirectionalLight { float4 ColorIntensity; float3 Direction; float AngularSize; };      // EMB_DirectionalLight: W/sr in .a, Direction = light's incoming direction
static_assert(sizeof(EnvDirectionalLight) == 32, "EnvDirectionalLight layout");
struct EnvCube { const uint2* texels; uint dim, mipLevels, _pad; uint mipOffset[12]; };
static const float kEnvMapRadianceScale = 0.25f;          // Sample.cpp:88

static inline uint env_cube_mip_levels(uint dim) { uint l = 0; while ((dim >> l) > 8u) l++; return l + 1u; }      // uint(log2(dim / 4) + 0.5): 2048 -> 9 (2048 .. 8)

// EnvMapBaker.hlsl:71-92
static inline float3 CubemapGetDirectionFor(uint face, float2 uv) {
    float cx = (uv.x * 2.0f) - 1.0f;
    float cy = 1.0f - (uv.y * 2.0f);
    float3 dir;
    const float l = sqrtf_(cx * cx + cy * cy + 1.0f);
    switch (face) {
    case 0: dir = make_float3(1.0f, cy, -cx); break;
    case 1: dir = make_float3(-1.0f, cy, cx); break;
    case 2: dir = make_float3(cx, 1.0f, -cy); break;
    case 3: dir = make_float3(cx, -1.0f, cy); break;
    case 4: dir = make_float3(cx, cy, 1.0f); break;
    case 5: dir = make_float3(-cx, cy, -1.0f); break;
    default: dir = make_float3(0.f); break;
    }
    return dir * (1.0f / l);
}
// the inverse, i.e. cube-map addressing: which face a direction looks at and where
static inline void env_cube_face_uv(float3 d, uint& face, float2& uv) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z), cx, cy;
    if (ax >= ay && ax >= az) { face = d.x >= 0.f ? 0u : 1u; cx = (d.x >= 0.f ? -d.z : d.z) / ax; cy = d.y / ax; }
    else if (ay >= az) { face = d.y >= 0.f ? 2u : 3u; cx = d.x / ay; cy = (d.y >= 0.f ? -d.z : d.z) / ay; }
    else { face = d.z >= 0.f ? 4u : 5u; cx = (d.z >= 0.f ? d.x : -d.x) / az; cy = d.y / az; }
    uv = make_float2((cx + 1.0f) * 0.5f, (1.0f - cy) * 0.5f);
}
// :112-158 (filament's CubemapUtils)
static inline float SphereQuadrantArea(float x, float y) { return dm_atan2(x * y, sqrtf_(x * x + y * y + 1.0f)); }
static inline float4 CubemapTexelSolidAngle4(float cubeDim, uint x, uint y) {      // texels (x,y) (x,y+1) (x+1,y) (x+1,y+1): "00 01 10 11"
    const float iDim = 1.0f / cubeDim;
    float s = (((float)x + 0.5f) * 2.0f * iDim) - 1.0f, t = (((float)y + 0.5f) * 2.0f * iDim) - 1.0f;
    const float x0 = s - iDim, y0 = t - iDim, x1 = s + iDim, y1 = t + iDim, x2 = s + iDim * 3.0f, y2 = t + iDim * 3.0f;
    float sqa00 = SphereQuadrantArea(x0, y0), sqa01 = SphereQuadrantArea(x0, y1), sqa10 = SphereQuadrantArea(x1, y0), sqa11 = SphereQuadrantArea(x1, y1), sqa20 = SphereQuadrantArea(x2, y0),
          sqa21 = SphereQuadrantArea(x2, y1), sqa02 = SphereQuadrantArea(x0, y2), sqa12 = SphereQuadrantArea(x1, y2), sqa22 = SphereQuadrantArea(x2, y2);
    return make_float4(fmaxf_(1e-6f, fabsf(sqa00 - sqa01 - sqa10 + sqa11)), fmaxf_(1e-6f, fabsf(sqa01 - sqa02 - sqa11 + sqa12)),
                       fmaxf_(1e-6f, fabsf(sqa10 - sqa11 - sqa20 + sqa21)), fmaxf_(1e-6f, fabsf(sqa11 - sqa12 - sqa21 + sqa22)));
}
// :166-192: an anti-aliased disc of the light's angular size; acos / cos / pow through the deterministic library
static inline float3 EnvComputeLightContribution(uint px, uint py, uint face, const EnvDirectionalLight& light, uint cubeDim) {
    const float fadeRangeInTexels = 1.1f, h = 0.5f * fadeRangeInTexels, fd = (float)cubeDim;
    float3 nd = -light.Direction;
    float3 d0 = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f + -h) / fd, ((float)py + 0.5f + -h) / fd));
    float3 d1 = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f + h) / fd, ((float)py + 0.5f + -h) / fd));
    float3 d2 = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f + -h) / fd, ((float)py + 0.5f + h) / fd));
    float3 d3 = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f + h) / fd, ((float)py + 0.5f + h) / fd));
    float a0 = dot(nd, d0), a1 = dot(nd, d1), a2 = dot(nd, d2), a3 = dot(nd, d3);
    float dotMin = fminf_(fminf_(a0, a1), fminf_(a2, a3)), dotMax = fmaxf_(fmaxf_(a0, a1), fmaxf_(a2, a3));
    float angleMin = dm_acos(clampf(dotMax, -1.0f, 1.0f)), angleMax = dm_acos(clampf(dotMin, -1.0f, 1.0f));
    float pixelCoverage = saturate(((light.AngularSize * 0.5f) - angleMin) / (angleMax - angleMin + 1e-24f));
    pixelCoverage = dm_pow(pixelCoverage, 4.0f);
    float lightSolidAngle = 2.0f * K_PI * (1.0f - dm_cos(light.AngularSize * 0.5f));
    return xyz(light.ColorIntensity) * pixelCoverage * (light.ColorIntensity.w / lightSolidAngle);
}
// MathHelpers.hlsli:92-99
static inline float2 world_to_latlong_map(float3 dir) {
    float3 p = normalize(dir);
    return make_float2(dm_atan2(p.x, -p.z) * K_1_2PI + 0.5f, dm_acos(p.y) * K_1_PI);
}
static inline uint2 env_pack_rgba16f(float4 v) { return make_uint2((f32tof16(v.y) << 16) | f32tof16(v.x), (f32tof16(v.w) << 16) | f32tof16(v.z)); }
static inline float4 env_unpack_rgba16f(uint2 t) { return make_float4(f16tof32(t.x & 0xffffu), f16tof32(t.x >> 16), f16tof32(t.y & 0xffffu), f16tof32(t.y >> 16)); }
static inline float4 env_round_rgba16f(float4 v) { return env_unpack_rgba16f(env_pack_rgba16f(v)); }      // what a store to the RGBA16_FLOAT cube keeps

// ---- BC6H round trip of the cube (EnvMapBaker.cpp:593-633: on D3D12 the reference compresses every mip of the finished RGBA16F cube with BC6UCompress.hlsl — Narkowicz'
// GPURealTimeBC6H, "Fast" = QUALITY 0 = one-region mode 11 only, m_compressionQuality = 1 by default — and the path tracer samples the compressed cube; the importance
// map keeps reading the uncompressed one, :635). EncodeP1 is restated below (BC6UCompress.hlsl:58-69, 117-141, 163-167, 179-202, 235-277, 324-416; log2 / exp2 / rcp through
// the deterministic library); the decode is the BC6H_UF16 rule of the D3D11 functional specification for mode 11 (10-bit endpoints, no transform, 4-bit indices, the index
// of texel 0 one bit short): unquantise ((c << 16) + 0x8000) >> 10 with 0 and 1023 pinned to 0 and 0xFFFF, interpolate (a (64 - w) + b w + 32) >> 6, finish (x 31) >> 6.
static inline float bc6_half_bits(float x) { return (float)f32tof16(x); }                              // `float v = f32tof16(x)`: the half's bit pattern as a number
static inline uint bc6_index4(float texelPos, float endPoint0Pos, float endPoint1Pos) {               // ComputeIndex4
    float r = (texelPos - endPoint0Pos) / (endPoint1Pos - endPoint0Pos);
    return (uint)clampf(r * 14.93333f + 0.03333f + 0.5f, 0.0f, 15.0f);
}
static inline float3 bc6_quantize10(float3 x) { return make_float3((bc6_half_bits(x.x) * 1024.0f) / (0x7bff + 1.0f), (bc6_half_bits(x.y) * 1024.0f) / (0x7bff + 1.0f), (bc6_half_bits(x.z) * 1024.0f) / (0x7bff + 1.0f)); }
static inline float3 bc6_log2p1(float3 v) { return make_float3(dm_log2(v.x + 1.0f), dm_log2(v.y + 1.0f), dm_log2(v.z + 1.0f)); }
static inline float3 bc6_exp2m1(float3 v) { return make_float3(dm_exp2(v.x) - 1.0f, dm_exp2(v.y) - 1.0f, dm_exp2(v.z) - 1.0f); }
static inline float bc6_sel_min(float cur, float texel, float block) { float c = (texel == block) ? cur : texel; return fminf_(cur, c); }
static inline float bc6_sel_max(float cur, float texel, float block) { float c = (texel == block) ? cur : texel; return fmaxf_(cur, c); }
static inline float bc6_calc_msle(float3 a, float3 b);
static inline float3 bc6_finish_unquantize(float3 e0, float3 e1, float weight);
static inline void bc6_encode_p1(const float3 texels[16], uint block[4], float* blockMSLE = nullptr) {
    float3 blockMin = texels[0], blockMax = texels[0];
    for (uint i = 1; i < 16; ++i) { blockMin = min3v(blockMin, texels[i]); blockMax = max3v(blockMax, texels[i]); }
    const float3 blockMinNonInset = blockMin, blockMaxNonInset = blockMax;
    {   // InsetColorBBoxP1
        float3 rmin = blockMax, rmax = blockMin;
        for (uint i = 0; i < 16; ++i) {
            rmin = make_float3(bc6_sel_min(rmin.x, texels[i].x, blockMin.x), bc6_sel_min(rmin.y, texels[i].y, blockMin.y), bc6_sel_min(rmin.z, texels[i].z, blockMin.z));
            rmax = make_float3(bc6_sel_max(rmax.x, texels[i].x, blockMax.x), bc6_sel_max(rmax.y, texels[i].y, blockMax.y), bc6_sel_max(rmax.z, texels[i].z, blockMax.z));
        }
        float3 logRMax = bc6_log2p1(rmax), logRMin = bc6_log2p1(rmin), logMax = bc6_log2p1(blockMax), logMin = bc6_log2p1(blockMin);
        float3 ext = (logMax - logMin) * (1.0f / 32.0f);
        logMin = logMin + min3v(logRMin - logMin, ext);
        logMax = logMax - min3v(logMax - logRMax, ext);
        blockMin = bc6_exp2m1(logMin); blockMax = bc6_exp2m1(logMax);
    }
    {   // OptimizeEndpointsP1
        float3 dir = blockMax - blockMin; dir = dir / ((dir.x + dir.y) + dir.z);
        float e0 = bc6_half_bits(dot(blockMin, dir)), e1 = bc6_half_bits(dot(blockMax, dir));
        float3 alphaTexelSum = make_float3(0.f), betaTexelSum = make_float3(0.f); float alphaBetaSum = 0.0f, alphaSqSum = 0.0f, betaSqSum = 0.0f;
        for (int i = 0; i < 16; i++) {
            float texelPos = bc6_half_bits(dot(texels[i], dir));
            uint texelIndex = bc6_index4(texelPos, e0, e1);
            float beta = saturate((float)texelIndex / 15.0f), alpha = 1.0f - beta;
            float3 texelF16 = make_float3(bc6_half_bits(texels[i].x), bc6_half_bits(texels[i].y), bc6_half_bits(texels[i].z));
            alphaTexelSum = alphaTexelSum + texelF16 * alpha; betaTexelSum = betaTexelSum + texelF16 * beta;
            alphaBetaSum += alpha * beta; alphaSqSum += alpha * alpha; betaSqSum += beta * beta;
        }
        float det = alphaSqSum * betaSqSum - alphaBetaSum * alphaBetaSum;
        if (fabsf(det) > 0.00001f) {
            float detRcp = 1.0f / det;
            float3 a = (alphaTexelSum * betaSqSum - betaTexelSum * alphaBetaSum) * detRcp, b = (betaTexelSum * alphaSqSum - alphaTexelSum * alphaBetaSum) * detRcp;
            auto back = [](float v) { return f16tof32((uint)clampf(v, 0.0f, 65504.0f)); };               // f16tof32(clamp(.., 0, HALF_MAX)): the float is converted to the uint bit pattern
            float3 mn = make_float3(back(a.x), back(a.y), back(a.z)), mx = make_float3(back(b.x), back(b.y), back(b.z));
            blockMin = make_float3(clampf(mn.x, blockMinNonInset.x, blockMaxNonInset.x), clampf(mn.y, blockMinNonInset.y, blockMaxNonInset.y), clampf(mn.z, blockMinNonInset.z, blockMaxNonInset.z));
            blockMax = make_float3(clampf(mx.x, blockMinNonInset.x, blockMaxNonInset.x), clampf(mx.y, blockMinNonInset.y, blockMaxNonInset.y), clampf(mx.z, blockMinNonInset.z, blockMaxNonInset.z));
        }
    }
    float3 dir = blockMax - blockMin; dir = dir / ((dir.x + dir.y) + dir.z);
    float3 endpoint0 = bc6_quantize10(blockMin), endpoint1 = bc6_quantize10(blockMax);
    float e0 = bc6_half_bits(dot(blockMin, dir)), e1 = bc6_half_bits(dot(blockMax, dir));
    if (bc6_index4(bc6_half_bits(dot(texels[0], dir)), e0, e1) > 7u) { float t = e0; e0 = e1; e1 = t; float3 t3 = endpoint0; endpoint0 = endpoint1; endpoint1 = t3; }
    uint idx[16];
    for (uint i = 0; i < 16; ++i) idx[i] = bc6_index4(bc6_half_bits(dot(texels[i], dir)), e0, e1);
    if (blockMSLE) {          // the error estimate QUALITY 1 compares the two-region encodings with (BC6UCompress.hlsl:373-388): from the unfloored 10-bit endpoints
        const float3 u0 = make_float3((endpoint0.x * 65536.0f + 0x8000) / 1024.0f, (endpoint0.y * 65536.0f + 0x8000) / 1024.0f, (endpoint0.z * 65536.0f + 0x8000) / 1024.0f);
        const float3 u1 = make_float3((endpoint1.x * 65536.0f + 0x8000) / 1024.0f, (endpoint1.y * 65536.0f + 0x8000) / 1024.0f, (endpoint1.z * 65536.0f + 0x8000) / 1024.0f);
        float msle = 0.0f;
        for (uint i = 0; i < 16; ++i) { float weight = floorf(((float)idx[i] * 64.0f) / 15.0f + 0.5f); msle += bc6_calc_msle(texels[i], bc6_finish_unquantize(u0, u1, weight)); }
        *blockMSLE = msle;
    }
    uint x = 0x03u, y = 0u, z = 0u, w = 0u;
    x |= (uint)endpoint0.x << 5; x |= (uint)endpoint0.y << 15; x |= (uint)endpoint0.z << 25; y |= (uint)endpoint0.z >> 7;
    y |= (uint)endpoint1.x << 3; y |= (uint)endpoint1.y << 13; y |= (uint)endpoint1.z << 23; z |= (uint)endpoint1.z >> 9;
    z |= idx[0] << 1; z |= idx[1] << 4; z |= idx[2] << 8; z |= idx[3] << 12; z |= idx[4] << 16; z |= idx[5] << 20; z |= idx[6] << 24; z |= idx[7] << 28;
    for (uint i = 8; i < 16; ++i) w |= idx[i] << (4u * (i - 8u));
    block[0] = x; block[1] = y; block[2] = z; block[3] = w;
}
static inline uint bc6_unquantize10(uint c) { return c == 0u ? 0u : (c == 1023u ? 0xFFFFu : ((c << 16) + 0x8000u) >> 10); }
static inline void bc6_decode_mode11(const uint block[4], uint halfBits[16][3]) {        // -> the half bit patterns a BC6H_UF16 fetch returns (alpha reads 1)
    const uint x = block[0], y = block[1], z = block[2], w = block[3];
    const uint e0[3] = {(x >> 5) & 1023u, (x >> 15) & 1023u, ((x >> 25) | (y << 7)) & 1023u}, e1[3] = {(y >> 3) & 1023u, (y >> 13) & 1023u, ((y >> 23) | (z << 9)) & 1023u};
    const uint weights[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};
    for (uint i = 0; i < 16; ++i) {
        const uint id = i == 0u ? (z >> 1) & 7u : (i < 8u ? (z >> (4u * i)) & 15u : (w >> (4u * (i - 8u))) & 15u), wt = weights[id];
        for (uint c = 0; c < 3; ++c) { const uint a = bc6_unquantize10(e0[c]), b = bc6_unquantize10(e1[c]); halfBits[i][c] = (((a * (64u - wt) + b * wt + 32u) >> 6) * 31u) >> 6; }
    }
}
// ---- QUALITY 1 ("Quality" in the reference's UI, m_compressionQuality = 2): after EncodeP1 the 32 two-region partitions are scored (EvaluateP2Pattern: squared distance
// of every texel to its region's bounding-box diagonal), the best one is encoded in modes "7.6" (2-bit mode field 01: 7-bit base, three 6-bit deltas) and "9.5" (mode
// 01110: 9-bit base, 5-bit deltas) and replaces the one-region block if its error estimate (MSLE) is lower (BC6UCompress.hlsl:58-69, 71-105, 107-135, 137-141, 157-177,
// 278-322, 417-730, 790-808). The decode of the two modes is the BC6H_UF16 rule of the D3D11 functional specification: delta endpoints wrap within the base's width,
// 3-bit indices with weights {0, 9, 18, 27, 37, 46, 55, 64}, the anchor texels of the two regions one bit short.
static inline float bc6_calc_msle(float3 a, float3 b) {                                                  // CalcMSLE, LUMINANCE_WEIGHTS 1
    float3 delta = make_float3(dm_log2((b.x + 1.0f) / (a.x + 1.0f)), dm_log2((b.y + 1.0f) / (a.y + 1.0f)), dm_log2((b.z + 1.0f) / (a.z + 1.0f)));
    float3 deltaSq = delta * delta;
    deltaSq = deltaSq * make_float3(0.299f, 0.587f, 0.114f);
    return deltaSq.x + deltaSq.y + deltaSq.z;
}
static inline uint bc6_pattern_fixup_id(uint i) { uint ret = 15u; ret = ((3441033216u >> i) & 0x1u) ? 2u : ret; ret = ((845414400u >> i) & 0x1u) ? 8u : ret; return ret; }
static inline uint bc6_pattern(uint p, uint i) {
    const uint enc16[16] = {2290666700u, 3972591342u, 4276930688u, 3967876808u, 4293707776u, 3892379264u, 4278255592u, 4026597360u, 9369360u, 147747072u, 1930428556u, 2362323200u, 823134348u, 913073766u, 267393000u, 966553998u};
    const uint p2 = p / 2u, p3 = p - p2 * 2u;
    uint enc = p2 < 16u ? enc16[p2] : 0u;
    enc = p3 ? enc >> 16 : enc;
    return (enc >> i) & 0x1u;
}
static inline float3 bc6_quantize_bits(float3 x, float scale) { return make_float3((bc6_half_bits(x.x) * scale) / (0x7bff + 1.0f), (bc6_half_bits(x.y) * scale) / (0x7bff + 1.0f), (bc6_half_bits(x.z) * scale) / (0x7bff + 1.0f)); }      // Quantize7 / 9 / 10
static inline float3 bc6_unquantize_bits(float3 x, float scale) { return make_float3((x.x * 65536.0f + 0x8000) / scale, (x.y * 65536.0f + 0x8000) / scale, (x.z * 65536.0f + 0x8000) / scale); }   // Unquantize7 / 9 / 10
static inline float3 bc6_finish_unquantize(float3 e0, float3 e1, float weight) {                         // FinishUnquantize: f16tof32(uint3(comp))
    float3 comp = ((e0 * (64.0f - weight) + e1 * weight) + make_float3(32.0f)) * (31.0f / 4096.0f);
    return make_float3(f16tof32((uint)comp.x), f16tof32((uint)comp.y), f16tof32((uint)comp.z));
}
static inline uint bc6_index3(float texelPos, float endPoint0Pos, float endPoint1Pos) {                   // ComputeIndex3
    float r = (texelPos - endPoint0Pos) / (endPoint1Pos - endPoint0Pos);
    return (uint)clampf(r * 6.98182f + 0.00909f + 0.5f, 0.0f, 7.0f);
}
static inline float3 bc6_floor3(float3 v) { return make_float3(floorf(v.x), floorf(v.y), floorf(v.z)); }
static inline float3 bc6_sign_extend(float3 v1, uint mask, uint signFlag) {                              // SignExtend: int3 -> masked two's complement -> float3
    int x = (int)v1.x, y = (int)v1.y, z = (int)v1.z;
    x = (int)(((uint)x & mask) | (x < 0 ? signFlag : 0u)); y = (int)(((uint)y & mask) | (y < 0 ? signFlag : 0u)); z = (int)(((uint)z & mask) | (z < 0 ? signFlag : 0u));
    return make_float3((float)x, (float)y, (float)z);
}
// EncodeP1's error estimate for the block it produced (BC6UCompress.hlsl:373-384): recomputed from the block's own fields — endpoint floats are not kept by bc6_encode_p1, so
// the estimate is formed by bc6_encode_p1_msle below, which repeats EncodeP1 up to the estimate (same operations, same order)
static inline void bc6_region_bounds(const float3 texels[16], uint pattern, float3& p0Min, float3& p0Max, float3& p1Min, float3& p1Max) {
    p0Min = make_float3(65504.0f); p0Max = make_float3(0.0f); p1Min = make_float3(65504.0f); p1Max = make_float3(0.0f);
    for (uint i = 0; i < 16; ++i) {
        if (bc6_pattern(pattern, i) == 0u) { p0Min = min3v(p0Min, texels[i]); p0Max = max3v(p0Max, texels[i]); }
        else { p1Min = min3v(p1Min, texels[i]); p1Max = max3v(p1Max, texels[i]); }
    }
}
static inline float bc6_dist_to_line_sq(float3 PointOnLine, float3 LineDirection, float3 Point) { float3 w = Point - PointOnLine; float3 x = w - dot(w, LineDirection) * LineDirection; return dot(x, x); }
static inline float bc6_evaluate_p2_pattern(uint pattern, const float3 texels[16]) {                       // EvaluateP2Pattern
    float3 p0Min, p0Max, p1Min, p1Max; bc6_region_bounds(texels, pattern, p0Min, p0Max, p1Min, p1Max);
    float3 p0Dir = normalize(p0Max - p0Min), p1Dir = normalize(p1Max - p1Min);
    float sq = 0.0f;
    for (uint i = 0; i < 16; ++i) sq += (bc6_pattern(pattern, i) == 0u) ? bc6_dist_to_line_sq(p0Min, p0Dir, texels[i]) : bc6_dist_to_line_sq(p1Min, p1Dir, texels[i]);
    return sq;
}
static inline void bc6_optimize_endpoints_p2(const float3 texels[16], uint pattern, uint patternSelector, float3& blockMin, float3& blockMax) {      // OptimizeEndpointsP2
    float3 dir = blockMax - blockMin; dir = dir / ((dir.x + dir.y) + dir.z);
    float e0 = bc6_half_bits(dot(blockMin, dir)), e1 = bc6_half_bits(dot(blockMax, dir));
    float3 alphaTexelSum = make_float3(0.f), betaTexelSum = make_float3(0.f); float alphaBetaSum = 0.0f, alphaSqSum = 0.0f, betaSqSum = 0.0f;
    for (int i = 0; i < 16; i++) {
        if (bc6_pattern(pattern, (uint)i) != patternSelector) continue;
        float texelPos = bc6_half_bits(dot(texels[i], dir));
        uint texelIndex = bc6_index3(texelPos, e0, e1);
        float beta = saturate((float)texelIndex / 7.0f), alpha = 1.0f - beta;
        float3 texelF16 = make_float3(bc6_half_bits(texels[i].x), bc6_half_bits(texels[i].y), bc6_half_bits(texels[i].z));
        alphaTexelSum = alphaTexelSum + texelF16 * alpha; betaTexelSum = betaTexelSum + texelF16 * beta;
        alphaBetaSum += alpha * beta; alphaSqSum += alpha * alpha; betaSqSum += beta * beta;
    }
    float det = alphaSqSum * betaSqSum - alphaBetaSum * alphaBetaSum;
    if (fabsf(det) > 0.00001f) {
        float detRcp = 1.0f / det;
        float3 a = (alphaTexelSum * betaSqSum - betaTexelSum * alphaBetaSum) * detRcp, b = (betaTexelSum * alphaSqSum - alphaTexelSum * alphaBetaSum) * detRcp;
        auto back = [](float v) { return f16tof32((uint)clampf(v, 0.0f, 65504.0f)); };
        blockMin = make_float3(back(a.x), back(a.y), back(a.z)); blockMax = make_float3(back(b.x), back(b.y), back(b.z));
    }
}
// EncodeP2Pattern: replaces block / blockMSLE when the two-region encoding's estimate is lower
static inline void bc6_encode_p2_pattern(uint block[4], float& blockMSLE, uint pattern, const float3 texels[16]) {
    float3 p0Min, p0Max, p1Min, p1Max; bc6_region_bounds(texels, pattern, p0Min, p0Max, p1Min, p1Max);
    bc6_optimize_endpoints_p2(texels, pattern, 0u, p0Min, p0Max);
    bc6_optimize_endpoints_p2(texels, pattern, 1u, p1Min, p1Max);
    float3 p0Dir = p0Max - p0Min, p1Dir = p1Max - p1Min;
    p0Dir = p0Dir / ((p0Dir.x + p0Dir.y) + p0Dir.z); p1Dir = p1Dir / ((p1Dir.x + p1Dir.y) + p1Dir.z);
    float p0E0 = bc6_half_bits(dot(p0Min, p0Dir)), p0E1 = bc6_half_bits(dot(p0Max, p0Dir)), p1E0 = bc6_half_bits(dot(p1Min, p1Dir)), p1E1 = bc6_half_bits(dot(p1Max, p1Dir));
    const uint fixupID = bc6_pattern_fixup_id(pattern);
    if (bc6_index3(bc6_half_bits(dot(texels[0], p0Dir)), p0E0, p0E1) > 3u) { float t = p0E0; p0E0 = p0E1; p0E1 = t; float3 t3 = p0Min; p0Min = p0Max; p0Max = t3; }
    if (bc6_index3(bc6_half_bits(dot(texels[fixupID], p1Dir)), p1E0, p1E1) > 3u) { float t = p1E0; p1E0 = p1E1; p1E1 = t; float3 t3 = p1Min; p1Min = p1Max; p1Max = t3; }
    uint indices[16];
    for (uint i = 0; i < 16; ++i) {
        uint p0Index = bc6_index3(bc6_half_bits(dot(texels[i], p0Dir)), p0E0, p0E1), p1Index = bc6_index3(bc6_half_bits(dot(texels[i], p1Dir)), p1E0, p1E1);
        indices[i] = bc6_pattern(pattern, i) == 0u ? p0Index : p1Index;
    }
    float3 e760 = bc6_floor3(bc6_quantize_bits(p0Min, 128.0f)), e761 = bc6_floor3(bc6_quantize_bits(p0Max, 128.0f)), e762 = bc6_floor3(bc6_quantize_bits(p1Min, 128.0f)), e763 = bc6_floor3(bc6_quantize_bits(p1Max, 128.0f));
    float3 e950 = bc6_floor3(bc6_quantize_bits(p0Min, 512.0f)), e951 = bc6_floor3(bc6_quantize_bits(p0Max, 512.0f)), e952 = bc6_floor3(bc6_quantize_bits(p1Min, 512.0f)), e953 = bc6_floor3(bc6_quantize_bits(p1Max, 512.0f));
    e761 = e761 - e760; e762 = e762 - e760; e763 = e763 - e760;
    e951 = e951 - e950; e952 = e952 - e950; e953 = e953 - e950;
    e761 = clamp3(e761, -31.0f, 31.0f); e762 = clamp3(e762, -31.0f, 31.0f); e763 = clamp3(e763, -31.0f, 31.0f);
    e951 = clamp3(e951, -15.0f, 15.0f); e952 = clamp3(e952, -15.0f, 15.0f); e953 = clamp3(e953, -15.0f, 15.0f);
    const float3 u760 = bc6_unquantize_bits(e760, 128.0f), u761 = bc6_unquantize_bits(e760 + e761, 128.0f), u762 = bc6_unquantize_bits(e760 + e762, 128.0f), u763 = bc6_unquantize_bits(e760 + e763, 128.0f);
    const float3 u950 = bc6_unquantize_bits(e950, 512.0f), u951 = bc6_unquantize_bits(e950 + e951, 512.0f), u952 = bc6_unquantize_bits(e950 + e952, 512.0f), u953 = bc6_unquantize_bits(e950 + e953, 512.0f);
    float msle76 = 0.0f, msle95 = 0.0f;
    for (uint i = 0; i < 16; ++i) {
        const bool r0 = bc6_pattern(pattern, i) == 0u;
        float weight = floorf(((float)indices[i] * 64.0f) / 7.0f + 0.5f);
        msle76 += bc6_calc_msle(texels[i], bc6_finish_unquantize(r0 ? u760 : u762, r0 ? u761 : u763, weight));
        msle95 += bc6_calc_msle(texels[i], bc6_finish_unquantize(r0 ? u950 : u952, r0 ? u951 : u953, weight));
    }
    e761 = bc6_sign_extend(e761, 0x1Fu, 0x20u); e762 = bc6_sign_extend(e762, 0x1Fu, 0x20u); e763 = bc6_sign_extend(e763, 0x1Fu, 0x20u);
    e951 = bc6_sign_extend(e951, 0xFu, 0x10u); e952 = bc6_sign_extend(e952, 0xFu, 0x10u); e953 = bc6_sign_extend(e953, 0xFu, 0x10u);
    const float p2MSLE = fminf_(msle76, msle95);
    if (!(p2MSLE < blockMSLE)) return;
    blockMSLE = p2MSLE;
    uint x = 0u, y = 0u, z = 0u, w = 0u;
    if (p2MSLE == msle76) {      // 7.6
        x = 0x1u;
        x |= ((uint)e762.y & 0x20u) >> 3; x |= ((uint)e763.y & 0x10u) >> 1; x |= ((uint)e763.y & 0x20u) >> 1; x |= (uint)e760.x << 5;
        x |= ((uint)e763.z & 0x01u) << 12; x |= ((uint)e763.z & 0x02u) << 12; x |= ((uint)e762.z & 0x10u) << 10; x |= (uint)e760.y << 15;
        x |= ((uint)e762.z & 0x20u) << 17; x |= ((uint)e763.z & 0x04u) << 21; x |= ((uint)e762.y & 0x10u) << 20; x |= (uint)e760.z << 25;
        y |= ((uint)e763.z & 0x08u) >> 3; y |= ((uint)e763.z & 0x20u) >> 4; y |= ((uint)e763.z & 0x10u) >> 2; y |= (uint)e761.x << 3;
        y |= ((uint)e762.y & 0x0Fu) << 9; y |= (uint)e761.y << 13; y |= ((uint)e763.y & 0x0Fu) << 19; y |= (uint)e761.z << 23; y |= ((uint)e762.z & 0x07u) << 29;
        z |= ((uint)e762.z & 0x08u) >> 3; z |= (uint)e762.x << 1; z |= (uint)e763.x << 7;
    } else {                     // 9.5
        x = 0xEu;
        x |= (uint)e950.x << 5; x |= ((uint)e952.z & 0x10u) << 10; x |= (uint)e950.y << 15; x |= ((uint)e952.y & 0x10u) << 20; x |= (uint)e950.z << 25;
        y |= (uint)e950.z >> 7; y |= ((uint)e953.z & 0x10u) >> 2; y |= (uint)e951.x << 3; y |= ((uint)e953.y & 0x10u) << 4; y |= ((uint)e952.y & 0x0Fu) << 9;
        y |= (uint)e951.y << 13; y |= ((uint)e953.z & 0x01u) << 18; y |= ((uint)e953.y & 0x0Fu) << 19; y |= (uint)e951.z << 23; y |= ((uint)e953.z & 0x02u) << 27; y |= (uint)e952.z << 29;
        z |= ((uint)e952.z & 0x08u) >> 3; z |= (uint)e952.x << 1; z |= ((uint)e953.z & 0x04u) << 4; z |= (uint)e953.x << 7; z |= ((uint)e953.z & 0x08u) << 9;
    }
    z |= pattern << 13;
    if (fixupID == 15u) {
        z |= indices[0] << 18; z |= indices[1] << 20; z |= indices[2] << 23; z |= indices[3] << 26; z |= indices[4] << 29;
        w |= indices[5] << 0; w |= indices[6] << 3; w |= indices[7] << 6; w |= indices[8] << 9; w |= indices[9] << 12; w |= indices[10] << 15; w |= indices[11] << 18;
        w |= indices[12] << 21; w |= indices[13] << 24; w |= indices[14] << 27; w |= indices[15] << 30;
    } else if (fixupID == 2u) {
        z |= indices[0] << 18; z |= indices[1] << 20; z |= indices[2] << 23; z |= indices[3] << 25; z |= indices[4] << 28; z |= indices[5] << 31;
        w |= indices[5] >> 1; w |= indices[6] << 2; w |= indices[7] << 5; w |= indices[8] << 8; w |= indices[9] << 11; w |= indices[10] << 14; w |= indices[11] << 17;
        w |= indices[12] << 20; w |= indices[13] << 23; w |= indices[14] << 26; w |= indices[15] << 29;
    } else {
        z |= indices[0] << 18; z |= indices[1] << 20; z |= indices[2] << 23; z |= indices[3] << 26; z |= indices[4] << 29;
        w |= indices[5] << 0; w |= indices[6] << 3; w |= indices[7] << 6; w |= indices[8] << 9; w |= indices[9] << 11; w |= indices[10] << 14; w |= indices[11] << 17;
        w |= indices[12] << 20; w |= indices[13] << 23; w |= indices[14] << 26; w |= indices[15] << 29;
    }
    block[0] = x; block[1] = y; block[2] = z; block[3] = w;
}
// CSMain with QUALITY 1 (BC6UCompress.hlsl:787-808): the one-region block, then the best-scoring partition
static inline void bc6_encode_quality(const float3 texels[16], uint block[4]) {
    float blockMSLE = 0.0f; bc6_encode_p1(texels, block, &blockMSLE);
    float bestScore = bc6_evaluate_p2_pattern(0u, texels); uint bestPattern = 0u;
    for (uint p = 1u; p < 32u; ++p) { float score = bc6_evaluate_p2_pattern(p, texels); if (score < bestScore) { bestPattern = p; bestScore = score; } }
    bc6_encode_p2_pattern(block, blockMSLE, bestPattern, texels);
}
static inline uint bc6_unquantize_n(uint c, uint bits) { return c == 0u ? 0u : (c == (1u << bits) - 1u ? 0xFFFFu : ((c << 16) + 0x8000u) >> bits); }
static inline uint bc6_block_bits(const uint block[4], uint first, uint n) {                              // n <= 8 bits starting at bit `first` of the 128-bit block
    unsigned long long lo = (unsigned long long)block[first >> 5] | ((first >> 5) < 3u ? (unsigned long long)block[(first >> 5) + 1u] << 32 : 0ull);
    return (uint)(lo >> (first & 31u)) & ((1u << n) - 1u);
}
// any block the two encoders above produce: mode 11 (0x03), "7.6" (2-bit field 01), "9.5" (0x0E) -> the half bit patterns a BC6H_UF16 fetch returns
static inline void bc6_decode(const uint block[4], uint halfBits[16][3]) {
    const uint x = block[0], y = block[1], z = block[2];
    if ((x & 0x1Fu) == 0x03u) { bc6_decode_mode11(block, halfBits); return; }
    uint bits, dbits, e[4][3];      // e[0], e[1]: region 0; e[2], e[3]: region 1 (x = r, y = g, z = b)
    if ((x & 0x3u) == 0x1u) {       // 7.6: the inverse of the packing above
        bits = 7u; dbits = 6u;
        e[0][0] = (x >> 5) & 127u; e[0][1] = (x >> 15) & 127u; e[0][2] = (x >> 25) & 127u;
        e[1][0] = (y >> 3) & 63u; e[1][1] = (y >> 13) & 63u; e[1][2] = (y >> 23) & 63u;
        e[2][0] = (z >> 1) & 63u; e[3][0] = (z >> 7) & 63u;
        e[2][1] = ((y >> 9) & 0x0Fu) | (((x >> 24) & 1u) << 4) | (((x >> 2) & 1u) << 5);
        e[3][1] = ((y >> 19) & 0x0Fu) | (((x >> 3) & 1u) << 4) | (((x >> 4) & 1u) << 5);
        e[2][2] = ((y >> 29) & 0x07u) | ((z & 1u) << 3) | (((x >> 14) & 1u) << 4) | (((x >> 22) & 1u) << 5);
        e[3][2] = ((x >> 12) & 1u) | (((x >> 13) & 1u) << 1) | (((x >> 23) & 1u) << 2) | ((y & 1u) << 3) | (((y >> 2) & 1u) << 4) | (((y >> 1) & 1u) << 5);
    } else {                        // 9.5
        bits = 9u; dbits = 5u;
        e[0][0] = (x >> 5) & 511u; e[0][1] = (x >> 15) & 511u; e[0][2] = ((x >> 25) | (y << 7)) & 511u;
        e[1][0] = (y >> 3) & 31u; e[1][1] = (y >> 13) & 31u; e[1][2] = (y >> 23) & 31u;
        e[2][0] = (z >> 1) & 31u; e[3][0] = (z >> 7) & 31u;
        e[2][1] = ((y >> 9) & 0x0Fu) | (((x >> 24) & 1u) << 4);
        e[3][1] = ((y >> 19) & 0x0Fu) | (((y >> 8) & 1u) << 4);
        e[2][2] = ((y >> 29) & 0x07u) | ((z & 1u) << 3) | (((x >> 14) & 1u) << 4);
        e[3][2] = ((y >> 18) & 1u) | (((y >> 28) & 1u) << 1) | (((z >> 6) & 1u) << 2) | (((z >> 12) & 1u) << 3) | (((y >> 2) & 1u) << 4);
    }
    const uint mask = (1u << bits) - 1u, sign = 1u << (dbits - 1u);
    for (uint k = 1; k < 4; ++k) for (uint c = 0; c < 3; ++c) { const uint d = e[k][c]; const uint delta = (d & sign) ? (d | ~((1u << dbits) - 1u)) : d; e[k][c] = (e[0][c] + delta) & mask; }      // transformed endpoints wrap within the base's width
    const uint pattern = (z >> 13) & 31u, fixupID = bc6_pattern_fixup_id(pattern);
    const uint weights[8] = {0, 9, 18, 27, 37, 46, 55, 64};
    uint pos = 82u;
    for (uint i = 0; i < 16; ++i) {
        const uint n = (i == 0u || i == fixupID) ? 2u : 3u, id = bc6_block_bits(block, pos, n), wt = weights[id]; pos += n;
        const uint r = bc6_pattern(pattern, i) * 2u;
        for (uint c = 0; c < 3; ++c) { const uint a = bc6_unquantize_n(e[r][c], bits), b = bc6_unquantize_n(e[r + 1u][c], bits); halfBits[i][c] = (((a * (64u - wt) + b * wt + 32u) >> 6) * 31u) >> 6; }
    }
}
// one 4x4 block of a cube level through the encoder and the decoder, in place (texel (bx*4 + i%4, by*4 + i/4) is texels[i], CSMain's gather order)
static inline void env_cube_bc6_round_trip_block(uint2* level, uint dim, uint face, uint bx, uint by, uint quality = 1u) {      // quality: 1 = QUALITY 0 ("Fast"), 2 = QUALITY 1 ("Quality")
    float3 texels[16];
    for (uint i = 0; i < 16; ++i) { float4 t = env_unpack_rgba16f(level[((size_t)face * dim + (by * 4u + i / 4u)) * dim + (bx * 4u + i % 4u)]); texels[i] = xyz(t); }
    uint block[4], hb[16][3]; if (quality >= 2u) bc6_encode_quality(texels, block); else bc6_encode_p1(texels, block); bc6_decode(block, hb);
    for (uint i = 0; i < 16; ++i) level[((size_t)face * dim + (by * 4u + i / 4u)) * dim + (bx * 4u + i % 4u)] = make_uint2(hb[i][0] | (hb[i][1] << 16), hb[i][2] | (0x3C00u << 16));
}
static inline float4 env_cube_texel(const EnvCube& c, uint mip, uint face, int x, int y) {
    int d = (int)(c.dim >> mip);
    x = x < 0 ? 0 : (x >= d ? d - 1 : x); y = y < 0 ? 0 : (y >= d ? d - 1 : y);
    return env_unpack_rgba16f(c.texels[c.mipOffset[mip] + ((size_t)face * (uint)d + (uint)y) * (uint)d + (uint)x]);
}
static inline float4 env_cube_bilinear(const EnvCube& c, uint mip, uint face, float2 uv) {
    float d = (float)(c.dim >> mip);
    float fx = uv.x * d - 0.5f, fy = uv.y * d - 0.5f, flx = floorf(fx), fly = floorf(fy), ax = fx - flx, ay = fy - fly;
    int x0 = (int)flx, y0 = (int)fly;
    float4 a = lerp4(env_cube_texel(c, mip, face, x0, y0), env_cube_texel(c, mip, face, x0 + 1, y0), ax);
    float4 b = lerp4(env_cube_texel(c, mip, face, x0, y0 + 1), env_cube_texel(c, mip, face, x0 + 1, y0 + 1), ax);
    return lerp4(a, b, ay);
}
// TextureCube::SampleLevel(linear sampler, dir, lod)
static inline float4 env_cube_sample_level(const EnvCube& c, float3 dir, float lod) {
    uint face; float2 uv; env_cube_face_uv(dir, face, uv);
    float l = clampf(lod, 0.0f, (float)(c.mipLevels - 1u)), l0 = floorf(l), f = l - l0;
    uint m0 = (uint)l0, m1 = m0 + 1u; if (m1 > c.mipLevels - 1u) m1 = c.mipLevels - 1u;
    float4 a = env_cube_bilinear(c, m0, face, uv);
    if (f == 0.0f || m1 == m0) return a;
    return lerp4(a, env_cube_bilinear(c, m1, face, uv), f);
}

} // namespace ptref
