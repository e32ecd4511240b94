// ORACLE (test infrastructure only) — polymorphic lights + global (power) light sampler (SURVEY.md §8a rows a11,a18,a19).
// Follows:
//   Lighting/PolymorphicLight.h:18-79                         packed 32 B + 16 B light records, type codes
//   Lighting/PolymorphicLight.hlsli:93-259 (SphereLight), :399-503 (TriangleLight), :562-640 (EnvironmentQuadLight),
//     :642-676 (CalcSample dispatch + shaping), :752-792 (UnpackRadiance / UnpackColor / PackColor)
//   Lighting/LightShaping.hlsli:16-42,76-99                    spot shaping
//   Utils/Utils.hlsli:115-152                                  Encode/Decode_Oct, NDirToOctUnorm32/OctToNDirUnorm32
//   Lighting/LightSampler.hlsli:100-146,318-345,400-432        SampleGlobal(+PDF), LoadLight, BSDF-side MIS
// (paths relative to /root/reference/Rtxpt/Shaders/PathTracer/)
#pragma once
#include "sampling.h"

namespace ptref {

static const uint kPolymorphicLightTypeShift = 24, kPolymorphicLightTypeMask = 0xf;
static const uint kPolymorphicLightShapingEnableBit = 1u << 28;
static const uint kPolymorphicLightShapingUseMinFalloff = 1u << 30;
static const float kPolymorphicLightMinLog2Radiance = -8.f, kPolymorphicLightMaxLog2Radiance = 40.f;
static const float kMinSpotlightFalloff = 0.0001f;
static const float DISTANT_LIGHT_DISTANCE = 100000.0f;
static const float FLT_EPSILON_MINI = 2e-9f, MAX_SOLID_ANGLE_PDF = 1e10f;
static const uint RTXPT_INVALID_LIGHT_INDEX = 0xFFFFFFFFu;
enum PolymorphicLightType : uint { kSphere = 0, kTriangle, kDirectional, kEnvironment, kPoint, kEnvironmentQuad };

struct PolymorphicLightInfo {   // 32 B
    float3 Center; uint ColorTypeAndFlags;
    uint Direction1, Direction2, Scalars, LogRadiance;
    bool HasLightShaping() const { return (ColorTypeAndFlags & kPolymorphicLightShapingEnableBit) != 0; }
};
struct PolymorphicLightInfoEx { // 16 B
    uint IesProfileIndex, PrimaryAxis, CosConeAngleAndSoftness, UniqueID;
};
struct PolymorphicLightInfoFull { PolymorphicLightInfo Base; PolymorphicLightInfoEx Extended; };

struct PolymorphicLightSample {
    float3 Position, Normal, Radiance; float SolidAnglePdf; bool LightSampleableByBSDF;
};

// Utils.hlsli:127-152
static inline float3 Decode_Oct(float2 f) {
    f = make_float2(f.x * 2.0f - 1.0f, f.y * 2.0f - 1.0f);
    float3 n = make_float3(f.x, f.y, 1.0f - fabsf(f.x) - fabsf(f.y));
    float t = saturate(-n.z);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return normalize(n);
}
static inline float2 Encode_Oct(float3 n) {
    float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    n = n / s;
    float2 xy = make_float2(n.x, n.y);
    if (!(n.z >= 0.0f)) xy = make_float2((1.0f - fabsf(n.y)) * ((n.x >= 0.0f) ? 1.0f : -1.0f), (1.0f - fabsf(n.x)) * ((n.y >= 0.0f) ? 1.0f : -1.0f));
    return make_float2(xy.x * 0.5f + 0.5f, xy.y * 0.5f + 0.5f);
}
static inline uint NDirToOctUnorm32(float3 n) {
    float2 p = Encode_Oct(n);
    p = make_float2(saturate(p.x * 0.5f + 0.5f), saturate(p.y * 0.5f + 0.5f));
    return (uint)(p.x * 65534.0f) | ((uint)(p.y * 65534.0f) << 16);
}
static inline float3 OctToNDirUnorm32(uint pUnorm) {
    float2 p;
    p.x = saturate((float)(pUnorm & 0xffffu) / 65534.0f);
    p.y = saturate((float)(pUnorm >> 16) / 65534.0f);
    p = make_float2(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f);
    return Decode_Oct(p);
}

// PolymorphicLight.hlsli:752-792
static inline float UnpackLightRadiance(uint logRadiance) {
    return (logRadiance == 0) ? 0.f
        : dm_exp2(((float)(logRadiance - 1) / 65534.0f) * (kPolymorphicLightMaxLog2Radiance - kPolymorphicLightMinLog2Radiance) + kPolymorphicLightMinLog2Radiance);
}
static inline float3 UnpackLightColor(const PolymorphicLightInfo& b) {
    float3 color = Unpack_R8G8B8_UFLOAT(b.ColorTypeAndFlags);
    float radiance = UnpackLightRadiance(b.LogRadiance & 0xffffu);
    return color * radiance;
}
static inline void PackLightColor(float3 radiance, PolymorphicLightInfo& b) {
    float intensity = fmaxf_(radiance.x, fmaxf_(radiance.y, radiance.z));
    if (intensity > 0.0f) {
        float logRadiance = saturate((dm_log2(intensity) - kPolymorphicLightMinLog2Radiance) / (kPolymorphicLightMaxLog2Radiance - kPolymorphicLightMinLog2Radiance));
        uint packedRadiance = (uint)ceilf(logRadiance * 65534.0f) + 1u;
        if (packedRadiance > 0xffffu) packedRadiance = 0xffffu;
        float unpackedRadiance = UnpackLightRadiance(packedRadiance);
        float3 normalizedRadiance = saturate3(radiance / make_float3(unpackedRadiance));
        b.LogRadiance |= packedRadiance;
        b.ColorTypeAndFlags |= Pack_R8G8B8_UFLOAT(normalizedRadiance);
    }
}

// LightShaping.hlsli:16-42, 76-99
struct LightShaping { float cosConeAngle; float3 primaryAxis; float cosConeSoftness; uint isSpot; float minFalloff; };
static inline LightShaping unpackLightShaping(const PolymorphicLightInfoFull& li) {
    LightShaping s; s.cosConeAngle = 0; s.primaryAxis = make_float3(0.f); s.cosConeSoftness = 0; s.isSpot = 0; s.minFalloff = 0;
    if (li.Base.HasLightShaping()) {
        s.isSpot = 1;
        s.primaryAxis = OctToNDirUnorm32(li.Extended.PrimaryAxis);
        s.cosConeAngle = f16tof32(li.Extended.CosConeAngleAndSoftness);
        s.cosConeSoftness = f16tof32(li.Extended.CosConeAngleAndSoftness >> 16);
        s.minFalloff = (li.Base.ColorTypeAndFlags & kPolymorphicLightShapingUseMinFalloff) ? kMinSpotlightFalloff : 0.0f;
    }
    return s;
}
static inline float smoothstep_(float a, float b, float x) { float t = saturate((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }
static inline float evaluateLightShaping(const LightShaping& s, float3 surfacePosition, float3 lightSamplePosition) {
    if (!s.isSpot) return 1.0f;
    float3 lightToSurface = normalize(surfacePosition - lightSamplePosition);
    float cosTheta = dot(s.primaryAxis, lightToSurface);
    float smoothFalloff = smoothstep_(s.cosConeAngle, s.cosConeAngle + s.cosConeSoftness, cosTheta);
    float softSpotlight = fmaxf_(s.minFalloff, smoothFalloff);
    if (softSpotlight <= 0) return 0.0f;
    return softSpotlight;    // IES disabled in the reference (LightShaping.hlsli:44-74)
}

// PolymorphicLight.hlsli:399-503
struct TriangleLight {
    float3 base, edge1, edge2, radiance, normal; float surfaceArea;
    static TriangleLight Create(const PolymorphicLightInfoFull& li) {
        TriangleLight t;
        t.edge1 = make_float3(f16tof32(li.Base.Direction1 & 0xffffu), f16tof32(li.Base.Direction2 & 0xffffu), f16tof32(li.Base.Scalars & 0xffffu));
        t.edge2 = make_float3(f16tof32(li.Base.Direction1 >> 16), f16tof32(li.Base.Direction2 >> 16), f16tof32(li.Base.Scalars >> 16));
        t.base = li.Base.Center - ((t.edge1 + t.edge2) / 3.0f);
        t.radiance = UnpackLightColor(li.Base);
        float3 lightNormal = cross(t.edge1, t.edge2);
        float lightNormalLength = length(lightNormal);
        if (lightNormalLength > 0.0f) { t.surfaceArea = 0.5f * lightNormalLength; t.normal = lightNormal / lightNormalLength; }
        else { t.surfaceArea = 0.0f; t.normal = make_float3(0.f); }
        return t;
    }
    PolymorphicLightInfoFull Store(uint uniqueID) const {
        PolymorphicLightInfoFull li; memset(&li, 0, sizeof(li));
        PackLightColor(radiance, li.Base);
        li.Base.Center = base + ((edge1 + edge2) / 3.0f);
        // Reference quirk, kept on purpose (PolymorphicLight.hlsli:511-514): the packed words pass through a `float3 edges` temporary, i.e. a numeric
        // uint -> float -> uint round trip that keeps only the 24 leading bits of each word. The fp16 of edge2 (high half) survives, the fp16 of
        // edge1 (low half) keeps its sign, exponent and 2-3 mantissa bits. This is what LightsBaker.hlsl:699 stores and every NEE sample reads back.
        li.Base.Direction1 = (uint)(float)((f32tof16(edge1.x) & 0xffffu) | (f32tof16(edge2.x) << 16));
        li.Base.Direction2 = (uint)(float)((f32tof16(edge1.y) & 0xffffu) | (f32tof16(edge2.y) << 16));
        li.Base.Scalars    = (uint)(float)((f32tof16(edge1.z) & 0xffffu) | (f32tof16(edge2.z) << 16));
        li.Base.ColorTypeAndFlags |= (uint)kTriangle << kPolymorphicLightTypeShift;
        li.Extended.UniqueID = uniqueID;
        return li;
    }
    PolymorphicLightSample CalcSample(float2 random, float3 viewerPosition) const {
        PolymorphicLightSample r; memset(&r, 0, sizeof(r));
        float3 bary = SampleTriangleUniform(random);
        r.Position = (base + edge1 * bary.y) + edge2 * bary.z;
        r.Position = ComputeRayOrigin(r.Position, normal);
        r.Normal = normal;
        float3 toLight = r.Position - viewerPosition;
        float distSqr = fmaxf_(FLT_EPSILON_MINI, dot(toLight, toLight));
        float distance = sqrtf_(distSqr);
        float3 dir = toLight / distance;
        float cosTheta = dot(normal, -dir);
        r.SolidAnglePdf = 0.f; r.Radiance = make_float3(0.f);
        if (cosTheta <= 0.f) return r;
        float areaPdf = fmaxf_(FLT_EPSILON_MINI, 1.0f / surfaceArea);
        r.SolidAnglePdf = fminf_(MAX_SOLID_ANGLE_PDF, pdfAtoW(areaPdf, distance, cosTheta));
        r.Radiance = radiance;
        r.LightSampleableByBSDF = true;
        return r;
    }
    float CalcSolidAnglePdfForMIS(float3 viewerPosition, float3 lightSamplePosition) const {
        float3 toLight = lightSamplePosition - viewerPosition;
        float distSqr = fmaxf_(FLT_EPSILON_MINI, dot(toLight, toLight));
        float distance = sqrtf_(distSqr);
        float3 dir = toLight / distance;
        float cosTheta = dot(normal, -dir);
        float areaPdf = fmaxf_(FLT_EPSILON_MINI, 1.0f / surfaceArea);
        return fminf_(MAX_SOLID_ANGLE_PDF, pdfAtoW(areaPdf, distance, cosTheta));
    }
    float GetPower() const { return surfaceArea * K_PI * Luminance(radiance); }
};

// LightShaping.hlsli:161-174
static inline float getShapingFluxFactor(const LightShaping& shaping) {
    if (!shaping.isSpot) return 1.0f;
    float solidAngleOverTwoPi = (1.0f - shaping.cosConeAngle);
    solidAngleOverTwoPi *= lerpf(1.0f, 0.5f, shaping.cosConeSoftness);
    return solidAngleOverTwoPi * 0.5f;
}

// PolymorphicLight.hlsli:93-259
struct SphereLight {
    float3 position; float radius; float3 radiance; LightShaping shaping;
    static SphereLight Create(const PolymorphicLightInfoFull& li) {
        SphereLight s; s.position = li.Base.Center; s.radius = f16tof32(li.Base.Scalars);
        s.radiance = UnpackLightColor(li.Base); s.shaping = unpackLightShaping(li); return s;
    }
    PolymorphicLightSample CalcSample(float2 random, float3 viewerPosition) const {
        PolymorphicLightSample ls; memset(&ls, 0, sizeof(ls));
        float3 lightVector = position - viewerPosition;
        float lightDistance2 = dot(lightVector, lightVector);
        float radius2 = sq(radius);
        if (lightDistance2 < radius2) {
            ls.Position = position; ls.Normal = make_float3(0.f); ls.Radiance = make_float3(0.f); ls.SolidAnglePdf = 1.0f; ls.LightSampleableByBSDF = false;
            return ls;
        }
        float lightDistance = sqrtf_(lightDistance2);
        float2 u = random;
        float sinThetaMax2 = radius2 / lightDistance2;
        float cosThetaMax = sqrtf_(fmaxf_(0.0f, 1.0f - sinThetaMax2));
        float phi = 2.0f * K_PI * u.x;
        float cosTheta = lerpf(cosThetaMax, 1.0f, u.y);
        float sinTheta = sqrtf_(fmaxf_(0.0f, 1.0f - sq(cosTheta)));
        float sinTheta2 = sinTheta * sinTheta;
        const float cLIGHT_SAMPING_EPSILON = 1e-10f;
        float dc = lightDistance, dc2 = lightDistance2;
        float ds = dc * cosTheta - sqrtf_(fmaxf_(cLIGHT_SAMPING_EPSILON, radius2 - dc2 * sinTheta2));
        float cosAlpha = (dc2 + radius2 - sq(ds)) / (2.0f * dc * radius);
        float sinAlpha = sqrtf_(fmaxf_(0.0f, 1.0f - sq(cosAlpha)));
        float3 n = normalize(lightVector), tg, bt;
        BranchlessONB(n, tg, bt);
        float sinPhi, cosPhi; dm_sincos(phi, sinPhi, cosPhi);
        float3 x = -tg, y = -bt, z = -n;
        float3 radiusVector = (sinAlpha * cosPhi * x + sinAlpha * sinPhi * y) + cosAlpha * z;
        ls.Position = position + radius * radiusVector;
        ls.Normal = normalize(radiusVector);
        ls.Radiance = radiance;
        ls.SolidAnglePdf = 1.0f / (2.0f * K_PI * (1.0f - cosThetaMax));
        ls.LightSampleableByBSDF = false;
        return ls;
    }
    // Geometry.hlsli:85-115 (IntersectRaySphere: analytic, smallest non-negative root; rayDir normalised) + PolymorphicLight.hlsli:190-220
    static bool IntersectRaySphere(float3 rayOrigin, float3 rayDir, float3 sphereCenter, float sphereRadius, float3& outHitPoint) {
        float3 oc = rayOrigin - sphereCenter;
        float a = 1.0f;
        float b = 2.0f * dot(oc, rayDir);
        float c = dot(oc, oc) - sphereRadius * sphereRadius;
        float discriminant = b * b - 4.0f * a * c;
        if (discriminant < 0.0f) { outHitPoint = make_float3(0.f); return false; }
        float sqrtDisc = sqrtf_(discriminant);
        float t1 = (-b - sqrtDisc) / (2.0f * a), t2 = (-b + sqrtDisc) / (2.0f * a);
        float t = (t1 >= 0.0f) ? t1 : ((t2 >= 0.0f) ? t2 : -1.0f);
        if (t < 0.0f) { outHitPoint = make_float3(0.f); return false; }
        outHitPoint = rayOrigin + rayDir * t;
        return true;
    }
    bool Eval(float3 rayPos, float3 rayDir, float3& outRadiance, float3& outLightSamplePosition) const {      // a path that hit the light's proxy mesh (analytic light proxies)
        const float3 lightVector = position - rayPos;
        if (dot(lightVector, lightVector) < sq(radius)) return false;
        if (!IntersectRaySphere(rayPos, rayDir, position, radius, outLightSamplePosition)) return false;
        outRadiance = radiance * evaluateLightShaping(shaping, rayPos, position);
        return true;
    }
    float CalcSolidAnglePdfForMIS(float3 viewerPosition, float3 /*lightSamplePosition*/) const {
        const float3 lightVector = position - viewerPosition;
        const float sinThetaMax2 = sq(radius) / dot(lightVector, lightVector);
        const float cosThetaMax = sqrtf_(fmaxf_(0.0f, 1.0f - sinThetaMax2));
        return 1.0f / (2.0f * K_PI * (1.0f - cosThetaMax));
    }
    float GetPower() const { return 4 * K_PI * sq(radius) * K_PI * Luminance(radiance) * getShapingFluxFactor(shaping); }       // PolymorphicLight.hlsli:224-232
};

// PolymorphicLight.hlsli:562-640. ToWorld uses the env map transform (PathTracerNEE.hlsli:16-33).
struct EnvironmentQuadLight {
    uint NodeX, NodeY, NodeDim; float Weight; float3 Radiance;
    static EnvironmentQuadLight Create(const PolymorphicLightInfoFull& li) {
        EnvironmentQuadLight e;
        e.NodeX = li.Base.Direction1 >> 16; e.NodeY = li.Base.Direction1 & 0xFFFFu; e.NodeDim = li.Base.Direction2 >> 16;
        e.Weight = asfloat(li.Base.Scalars); e.Radiance = UnpackLightColor(li.Base);
        return e;
    }
    PolymorphicLightInfoFull Store(uint uniqueID) const {
        PolymorphicLightInfoFull li; memset(&li, 0, sizeof(li));
        PackLightColor(Radiance, li.Base);
        li.Base.Direction1 = (NodeX << 16) | NodeY;
        li.Base.Direction2 = (NodeDim << 16);
        li.Base.Scalars = asuint(Weight);
        li.Base.ColorTypeAndFlags |= (uint)kEnvironmentQuad << kPolymorphicLightTypeShift;
        li.Extended.UniqueID = uniqueID;
        return li;
    }
    PolymorphicLightSample CalcSample(float2 random, float3 viewerPosition, const float3x4& envToWorld) const {
        PolymorphicLightSample pls;
        float2 subTexelPos = make_float2(((float)NodeX + random.x) / (float)NodeDim, ((float)NodeY + random.y) / (float)NodeDim);
        float3 localDir = oct_to_ndir_equal_area_unorm(subTexelPos);
        float3 worldDir = mul_vec_mat3(localDir, envToWorld);
        pls.Position = viewerPosition + worldDir * DISTANT_LIGHT_DISTANCE;
        pls.Normal = -worldDir;
        pls.Radiance = Radiance;                                      // NEE_AT_SAMPLE_BAKED_ENVIRONMENT == 1
        pls.SolidAnglePdf = (float)(NodeDim * NodeDim) / (4.0f * K_PI);
        pls.LightSampleableByBSDF = true;
        return pls;
    }
    float CalcSolidAnglePdfForMIS() const { return (float)(NodeDim * NodeDim) / (4.0f * K_PI); }
};

static inline uint DecodeLightType(const PolymorphicLightInfo& b) { return (b.ColorTypeAndFlags >> kPolymorphicLightTypeShift) & kPolymorphicLightTypeMask; }

// PolymorphicLight.hlsli:642-676
static inline PolymorphicLightSample PolymorphicLight_CalcSample(const PolymorphicLightInfoFull& li, float2 random, float3 viewerPosition, const float3x4& envToWorld) {
    PolymorphicLightSample s; memset(&s, 0, sizeof(s));
    switch (DecodeLightType(li.Base)) {
    case kSphere: s = SphereLight::Create(li).CalcSample(random, viewerPosition); break;
    case kTriangle: s = TriangleLight::Create(li).CalcSample(random, viewerPosition); break;
    case kEnvironmentQuad: s = EnvironmentQuadLight::Create(li).CalcSample(random, viewerPosition, envToWorld); break;
    default: break;
    }
    if (s.SolidAnglePdf > 0) s.Radiance = s.Radiance * evaluateLightShaping(unpackLightShaping(li), viewerPosition, s.Position);
    return s;
}
// PolymorphicLight.hlsli:700-724
static inline float PolymorphicLight_GetPower(const PolymorphicLightInfoFull& li) {
    switch (DecodeLightType(li.Base)) {
    case kSphere: return SphereLight::Create(li).GetPower();
    case kTriangle: return TriangleLight::Create(li).GetPower();
    case kEnvironmentQuad: return EnvironmentQuadLight::Create(li).Weight;
    default: return 0;
    }
}

// The runtime light table (LightingTypes.hlsli:78-123 LightingControlData subset + the buffers LightSampler binds).
struct LightTable {
    const PolymorphicLightInfo* Lights; const PolymorphicLightInfoEx* LightsEx;
    const uint* ProxyCounters; const uint* ProxyIndices;
    uint TotalLightCount, SamplingProxyCount;
    const uint* EnvLookupMap; uint EnvLookupDim;       // equal-area-octahedral texel -> env quad light index
    float3x4 EnvToWorld, WorldToEnv;
    // NEE-AT (NEEType 2): the screen-tile local samplers and the feedback switch (LightingTypes.hlsli:78-123: LocalSamplingTileJitter, LocalSamplingResolution,
    // LocalToGlobalSampleRatio, ScreenSpaceVsWorldSpaceThreshold, TemporalFeedbackRequired). LocalSamplingBuffer == null: no local layer (ratio 0).
    const uint* LocalSamplingBuffer; uint LocalResX, LocalResY, LocalJitterX, LocalJitterY;
    float LocalToGlobalSampleRatio, ScreenSpaceVsWorldSpaceThreshold; uint TemporalFeedbackRequired;
    // the guide buffer the reference-mode path tracer leaves behind for the baker's disocclusion test (Bridge::ExportSurface / ExportNonSurface, BridgeDonut:1105-1140): clip-space
    // depth of the path's last vertex, one float per pixel; ClipZ / ClipW = columns 2 and 3 of the host's world-to-clip matrix (row vectors). DepthExport == null: nothing is written.
    float* DepthExport; uint DepthWidth; float ClipZ[4], ClipW[4];
};
// `mul(float4(p, 1), matWorldToClip)`: z and w of the row vector times the matrix, each as ((x + y) + z) + w
static inline float LightTable_ClipDepth(const LightTable& t, float3 p) {
    float z = ((p.x * t.ClipZ[0] + p.y * t.ClipZ[1]) + p.z * t.ClipZ[2]) + 1.0f * t.ClipZ[3];
    float w = ((p.x * t.ClipW[0] + p.y * t.ClipW[1]) + p.z * t.ClipW[2]) + 1.0f * t.ClipW[3];
    return z / w;
}
// LightingConfig.h:27-31 (the "default" tier) and LightingTypes.hlsli:148-180
static const uint RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE = 8, RTXPT_LIGHTING_LOCAL_PROXY_COUNT = 128, RTXPT_LIGHTING_LOCAL_PROXY_BINARY_SEARCH_STEPS = 8;
static inline uint ComputeCandidateSampleLocalCount(float localToGlobalRatio, uint totalCandidateSamples) { return (uint)((float)(totalCandidateSamples - 1u) * localToGlobalRatio + 0.75f); }
static inline uint PackMiniListLightAndCount(uint globalLightIndex, uint counter) { return ((globalLightIndex & 0x007FFFFFu) << 9) | ((counter - 1u) & 0x1FFu); }
static inline void UnpackMiniListLightAndCount(uint value, uint& globalLightIndex, uint& counter) { globalLightIndex = value >> 9; counter = (value & 0x1FFu) + 1u; }
static inline uint UnpackMiniListLight(uint value) { return value >> 9; }
static inline uint LLSB_ComputeBaseAddress(uint tileX, uint tileY, uint resX) { return (tileX + tileY * resX) * RTXPT_LIGHTING_LOCAL_PROXY_COUNT; }
// LightingAlgorithms.hlsli:654-682: the tile's entries are sorted by light index; returns the packed entry or RTXPT_INVALID_LIGHT_INDEX. The search may step outside the
// buffer — in the very first tile `indexRight = indexMiddle - 1` wraps below zero when the light sorts before every entry — where a D3D typed-buffer load returns 0
// and the reference carries on; bufferWords restates that (and a light index 0 can then be "found" outside, exactly as there).
static inline uint LocalLightBinarySearch(const uint* storageBuffer, uint bufferWords, uint tileAddress, uint globalLightIndexToFind, uint localLightCount, uint steps) {
    uint indexLeft = tileAddress, indexRight = tileAddress + localLightCount - 1u;
    for (uint i = 0u; i < steps; ++i) {
        uint indexMiddle = (indexLeft + indexRight) >> 1;
        uint value = indexMiddle < bufferWords ? storageBuffer[indexMiddle] : 0u;
        uint keyMiddle = UnpackMiniListLight(value);
        if (keyMiddle < globalLightIndexToFind) indexLeft = indexMiddle + 1u;
        else if (keyMiddle > globalLightIndexToFind) indexRight = indexMiddle - 1u;
        else return value;
    }
    return RTXPT_INVALID_LIGHT_INDEX;
}
// LightingTypes.hlsli:184-320 LightFeedbackReservoir: one slot per pixel, "how much this pixel wanted which light" (the input of next frame's local samplers)
static const uint LFR_SCREEN_SPACE_COHERENT_FLAG = 0x80000000u;
static const float LFR_MAX_WEIGHT = 1e12f;
static inline void LightFeedbackReservoir_Add(float& slotTotalWeight, uint& slotCandidate, float randomValue, uint candidateIndex, float candidateWeight, bool candidateIsScreenSpaceCoherent) {
    candidateWeight = fminf_(LFR_MAX_WEIGHT, candidateWeight);
    float totalWeight = slotTotalWeight;
    totalWeight += candidateWeight;
    slotTotalWeight = fminf_(LFR_MAX_WEIGHT, totalWeight);
    float threshold = saturate(candidateWeight / totalWeight);
    if (candidateIsScreenSpaceCoherent) candidateIndex |= LFR_SCREEN_SPACE_COHERENT_FLAG;
    if (randomValue < threshold) slotCandidate = candidateIndex;
}
// LightSampler.hlsli:29-110 (make), :100-180 (the two samplers and their pdfs), :222-270, :318-345, :400-432. NEEType 0 / 1: the global sampler only
// (LocalToGlobalSampleRatio 0 -> local count 0); NEEType 2 (NEE-AT): candidates [globalCount, total) are drawn from the pixel's tile.
struct LightSampler {
    const LightTable* T;
    uint LocalSamplingTilePos; bool IsScreenSpaceCoherent;
    static bool IsScreenSpaceCoherentHeuristic(const LightTable& t, float rayConeWidth, float totalPathLength) {      // :45-49
        float rayConeWidthOverTotalPathTravel = rayConeWidth / totalPathLength;
        return rayConeWidthOverTotalPathTravel < t.ScreenSpaceVsWorldSpaceThreshold;
    }
    static LightSampler make(const LightTable& t, uint pixelX, uint pixelY, bool isScreenSpaceCoherent) {            // :51-93
        LightSampler ls; ls.T = &t; ls.IsScreenSpaceCoherent = isScreenSpaceCoherent;
        ls.LocalSamplingTilePos = LLSB_ComputeBaseAddress((pixelX + t.LocalJitterX) / RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE, (pixelY + t.LocalJitterY) / RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE, t.LocalResX);
        return ls;
    }
    bool IsTemporalFeedbackRequired() const { return T->TemporalFeedbackRequired != 0; }
    uint SampleLocal(float rnd, float& pdf) const {                                                                  // :120-138
        const uint localProxyCount = RTXPT_LIGHTING_LOCAL_PROXY_COUNT;
        uint indexInIndex = (uint)(rnd * (float)localProxyCount);
        if (indexInIndex > localProxyCount - 1u) indexInIndex = localProxyCount - 1u;
        uint lightIndex, proxyCount;
        UnpackMiniListLightAndCount(T->LocalSamplingBuffer[LocalSamplingTilePos + indexInIndex], lightIndex, proxyCount);
        pdf = (float)proxyCount / (float)localProxyCount;
        return lightIndex;
    }
    float SampleLocalPDF(uint lightIndex) const {                                                                    // :146-180
        uint packedValue = LocalLightBinarySearch(T->LocalSamplingBuffer, T->LocalResX * T->LocalResY * RTXPT_LIGHTING_LOCAL_PROXY_COUNT, LocalSamplingTilePos, lightIndex, RTXPT_LIGHTING_LOCAL_PROXY_COUNT, RTXPT_LIGHTING_LOCAL_PROXY_BINARY_SEARCH_STEPS);
        if (packedValue == RTXPT_INVALID_LIGHT_INDEX) return 0.0f;
        uint lightIndexR, proxyCountR;
        UnpackMiniListLightAndCount(packedValue, lightIndexR, proxyCountR);
        return (float)proxyCountR / (float)RTXPT_LIGHTING_LOCAL_PROXY_COUNT;
    }
    void GetCandidateSampleCounts(uint totalCandidateSamples, uint& localCount, uint& globalCount) const {           // :411-420
        localCount = (IsScreenSpaceCoherent && T->LocalSamplingBuffer) ? ComputeCandidateSampleLocalCount(T->LocalToGlobalSampleRatio, totalCandidateSamples) : 0u;
        globalCount = totalCandidateSamples - localCount;
    }
    // :242-268: the pdf of the sampler the sample was drawn from, the pdf the other sampler gives the same light, and the number of candidates each drew
    void ComputeLightSelectionPdfs(float selectionPdf, uint lightIndex, bool fromLocalDistribution, uint localCandidateCount, uint globalCandidateCount,
                                   float& thisPdf, float& otherPdf, float& thisCount, float& otherCount) const {
        thisPdf = selectionPdf;
        if (fromLocalDistribution) { otherPdf = SampleGlobalPDF(lightIndex); thisCount = (float)localCandidateCount; otherCount = (float)globalCandidateCount; }
        else {
            thisCount = (float)globalCandidateCount;
            if (localCandidateCount != 0) { otherPdf = SampleLocalPDF(lightIndex); otherCount = (float)localCandidateCount; }
            else { otherPdf = 0; otherCount = 0; }
        }
    }
    // :184-200 InsertFeedbackFromNEE: the weight the reservoir receives (RTXPT_LIGHTING_SCREEN_SPACE_COHERENT_FEEDBACK_BIAS is 1.0)
    float FeedbackWeightFromNEE(uint lightIndex, float pixelRadianceContributionAvg) const {
        float feedbackWeight = pixelRadianceContributionAvg;
        feedbackWeight /= dm_pow(SampleGlobalPDF(lightIndex), 0.65f);
        if (IsScreenSpaceCoherent) feedbackWeight *= 1.0f;
        return feedbackWeight;
    }
    bool IsEmpty() const { return T->SamplingProxyCount == 0; }
    uint SampleGlobal(float rnd, float& pdf) const {
        uint total = T->SamplingProxyCount;
        uint idx = (uint)(rnd * (float)total);
        if (idx > total - 1) idx = total - 1;
        uint lightIndex = T->ProxyIndices[idx];
        pdf = (float)T->ProxyCounters[lightIndex] / (float)total;
        return lightIndex;
    }
    float SampleGlobalPDF(uint lightIndex) const { return (float)T->ProxyCounters[lightIndex] / (float)T->SamplingProxyCount; }
    PolymorphicLightInfoFull LoadLight(uint index) const {
        PolymorphicLightInfoFull f; f.Base = T->Lights[index]; memset(&f.Extended, 0, sizeof(f.Extended));
        if (f.Base.HasLightShaping()) f.Extended = T->LightsEx[index];
        return f;
    }
    // :318-332
    float ComputeLightVsBSDF_MIS_ForBSDF(uint lightIndex, float bsdfPdf, float solidAnglePdf, uint candidateSampleCount, uint fullSampleCount) const {
        uint localCount, globalCount;
        GetCandidateSampleCounts(candidateSampleCount, localCount, globalCount);
        float globalPdf = SampleGlobalPDF(lightIndex);
        float localPdf = (localCount > 0) ? SampleLocalPDF(lightIndex) : 0.0f;
        float lightAvgPdf = (localPdf + globalPdf) * (float)fullSampleCount;
        return EvalMIS_Balance(1, bsdfPdf, 1, lightAvgPdf * solidAnglePdf);
    }
    // :334-345
    float ComputeBSDFMISForEmissiveTriangle(uint lightIndex, float bsdfPdf, float3 viewerPosition, float3 lightSamplePosition, uint candidateSampleCount, uint fullSamples) const {
        if (bsdfPdf == 0 || lightIndex == RTXPT_INVALID_LIGHT_INDEX) return 1;
        TriangleLight tl = TriangleLight::Create(LoadLight(lightIndex));
        float solidAnglePdf = tl.CalcSolidAnglePdfForMIS(viewerPosition, lightSamplePosition);
        return ComputeLightVsBSDF_MIS_ForBSDF(lightIndex, bsdfPdf, solidAnglePdf, candidateSampleCount, fullSamples);
    }
    // :363-390 (sphere lights only; exact MIS, RTXPT_USE_APPROXIMATE_MIS 0). Returns false when nothing is added.
    bool ComputeAnalyticLightProxyContribution(uint analyticLightIndex, float bsdfPdf, float3 previousVertex, float3 rayDir, uint candidateSamples, uint fullSamples, float3& contribution) const {
        PolymorphicLightInfoFull lightInfo = LoadLight(analyticLightIndex);
        if (DecodeLightType(lightInfo.Base) != kSphere) return false;
        SphereLight sphereLight = SphereLight::Create(lightInfo);
        float3 radiance, lightSamplePosition;
        if (!sphereLight.Eval(previousVertex, rayDir, radiance, lightSamplePosition)) return false;
        float mis = 1.0f;
        if (bsdfPdf != 0) mis = ComputeLightVsBSDF_MIS_ForBSDF(analyticLightIndex, bsdfPdf, sphereLight.CalcSolidAnglePdfForMIS(previousVertex, lightSamplePosition), candidateSamples, fullSamples);
        contribution = radiance * mis;
        return true;
    }
    // :347-362
    float ComputeBSDFMISForEnvironmentQuad(uint lightIndex, float bsdfPdf, uint candidateSampleCount, uint fullSamples) const {
        if (bsdfPdf == 0 || lightIndex == RTXPT_INVALID_LIGHT_INDEX) return 1;
        EnvironmentQuadLight eq = EnvironmentQuadLight::Create(LoadLight(lightIndex));
        return ComputeLightVsBSDF_MIS_ForBSDF(lightIndex, bsdfPdf, eq.CalcSolidAnglePdfForMIS(), candidateSampleCount, fullSamples);
    }
    // :423-432
    uint LookupEnvLightByDirection(float3 localDir) const {
        if (T->EnvLookupDim == 0) return RTXPT_INVALID_LIGHT_INDEX;
        float2 uv = ndir_to_oct_equal_area_unorm(localDir);
        uint x = (uint)(uv.x * (float)T->EnvLookupDim), y = (uint)(uv.y * (float)T->EnvLookupDim);
        // uv == 1 exactly (a direction on the seam of the octahedral map) gives coord == dim, and an out-of-range Texture2D.Load returns 0 on D3D: light 0 — the reference computes its
        // MIS weight against that light's pdf, and so does this restatement (round 4: found by the 16-sample 4K frame of the reference's text, 2 of 133 M paths; rounds 1-3 clamped)
        if (x > T->EnvLookupDim - 1 || y > T->EnvLookupDim - 1) return 0u;
        return T->EnvLookupMap[y * T->EnvLookupDim + x];
    }
};

} // namespace ptref
