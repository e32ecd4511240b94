// mi355pt — the baked environment cube. Part of the PRODUCT path (libmi355pt.so); written to the arithmetic contract stated in pt_vec.h.
#pragma once
#include "pt_dmath.h"
#include "pt_vec.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

// Environment cube: what EnvMapBaker turns the lat-long source into and what the path tracer samples (Rtxpt/Lighting/Distant/EnvMapBaker.hlsl:71-118,
// 166-246, 268-371; EnvMapBaker.cpp:298-343, 425-620; Rtxpt/Shaders/PathTracer/Lighting/EnvMap.hlsli:54-93): a cube of RGBA16F texels with a solid-angle
// weighted mip chain down to 8x8, the scene's directional lights rasterised into it as anti-aliased discs, radiance scaled by c_envMapRadianceScale = 1/4
// (Sample.cpp:88; the host compensates in EnvMapSceneParams::ColorMultiplier, :1936-1948) and clamped to the fp16 range. The reference may additionally
// BC6H-compress the cube (lossy, disabled on Vulkan): the uncompressed RGBA16F path is restated.
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

#pragma clang force_cuda_host_device end
} // namespace ptk
