// mi355pt device/host leaf library — StandardBSDF / FalcorBSDF
// Part of the PRODUCT path (libmi355pt.so). Written to the arithmetic contract stated in pt_vec.h so that the HIP kernels
// reproduce the reference estimator bit-for-bit against the independent CPU oracle used by the tests.
// Reference anchors are cited per function (paths relative to /root/reference/Rtxpt/Shaders/PathTracer/ unless noted).
// Follows, in order:
//   Rendering/Materials/Fresnel.hlsli:22-69            evalFresnelSchlick, evalFresnelDielectric
//   Rendering/Materials/Microfacet.hlsli:33-38,105-128,185-207,218-273  GGX NDF, BVNDF pdf/sample, Smith G
//   Rendering/Materials/BxDF.hlsli:157-208 (Frostbite diffuse), :59-96 (Lambert), :212-247 (diffuse transmission),
//     :250-268 (Turquin multi-scatter), :273-381 (specular reflection), :385-607 (specular reflection+transmission),
//     :709-970 (FalcorBSDF init / getLobes / eval / sample / evalPdf)
//   Rendering/Materials/StandardBSDF.hlsli:50-92 (world<->local wrappers)
//   Scene/ShadingData.hlsli:38-127, Scene/Material/MaterialData.hlsli:23-86 (header bit layout)
// (paths relative to /root/reference/Rtxpt/Shaders/PathTracer/)
// Parity knob: lpfloat == float here, i.e. the reference built with RTXPT_LP_TYPES_USE_16BIT_PRECISION == 0.
#pragma once
#include "pt_sampling.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

// Rendering/Materials/LobeType.hlsli:18-42
enum LobeType : uint {
    Lobe_None = 0x00, Lobe_DiffuseReflection = 0x01, Lobe_SpecularReflection = 0x02, Lobe_DeltaReflection = 0x04,
    Lobe_DiffuseTransmission = 0x10, Lobe_SpecularTransmission = 0x20, Lobe_DeltaTransmission = 0x40,
    Lobe_Diffuse = 0x11, Lobe_Specular = 0x22, Lobe_Delta = 0x44, Lobe_NonDelta = 0x33,
    Lobe_Reflection = 0x0f, Lobe_Transmission = 0xf0, Lobe_All = 0xff,
};

static const float kMinCosTheta = 1e-6f;                      // BxDF.hlsli:31
static const float cOneMinusEpsilon = 0.99999994f;           // MathConstants: largest float < 1
static const float kMinGGXAlpha = 0.0064f;                   // BxDF.hlsli:52
enum { DiffuseBrdfLambert = 0, DiffuseBrdfFrostbite = 2 };   // BxDFConfig.hlsli:20-26 (Disney unused)

// Scene/Material/MaterialData.hlsli:23-86
struct MaterialHeader {
    uint packedData;
    static MaterialHeader make() { MaterialHeader h; h.packedData = 0; return h; }
    void setNestedPriority(uint p) { packedData = (packedData & ~0xFu) | (p & 0xFu); }
    uint getNestedPriority() const { return packedData & 0xFu; }
    void setActiveLobes(uint l) { packedData = (packedData & ~(0xFFu << 4)) | ((l & 0xFFu) << 4); }
    uint getActiveLobes() const { return (packedData >> 4) & 0xFFu; }
    void setThinSurface(bool t) { packedData = (packedData & ~(1u << 12)) | ((t ? 1u : 0u) << 12); }
    bool isThinSurface() const { return (packedData & (1u << 12)) != 0; }
};

// Scene/ShadingData.hlsli:38-127
struct ShadingData {
    float3 posW, faceNCorrected, V, N, T, B, vertexN;
    bool frontFacing;
    MaterialHeader mtl;
    uint materialID;
    float IoR;
    float shadowNoLFadeout;
    float3 emission;
    float3 computeNewRayOrigin(bool viewside = true) const { return ComputeRayOrigin(posW, viewside ? faceNCorrected : -faceNCorrected); }
    float3 fromLocal(float3 v) const { return (T * v.x + B * v.y) + N * v.z; }
    float3 toLocal(float3 v) const { return make_float3(dot(v, T), dot(v, B), dot(v, N)); }
};

// Fresnel.hlsli:30-37
static inline float evalFresnelSchlick(float f0, float f90, float cosTheta) {
    return f0 + (f90 - f0) * dm_pow5(fmaxf_(1 - cosTheta, 0));
}
static inline float3 evalFresnelSchlick(float3 f0, float f90, float cosTheta) {
    float w = dm_pow5(fmaxf_(1 - cosTheta, 0));
    return make_float3(f0.x + (f90 - f0.x) * w, f0.y + (f90 - f0.y) * w, f0.z + (f90 - f0.z) * w);
}
// Fresnel.hlsli:52-77
static inline float evalFresnelDielectric(float eta, float cosThetaI, float& cosThetaT) {
    if (cosThetaI < 0) { eta = 1 / eta; cosThetaI = -cosThetaI; }
    float sinThetaTSq = eta * eta * (1 - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1) { cosThetaT = 0; return 1; }
    cosThetaT = sqrtf_(1 - sinThetaTSq);
    float Rs = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    float Rp = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    return 0.5f * (Rs * Rs + Rp * Rp);
}
static inline float evalFresnelDielectric(float eta, float cosThetaI) { float t; return evalFresnelDielectric(eta, cosThetaI, t); }

// Microfacet.hlsli:33-38
static inline float evalNdfGGX(float alpha, float cosTheta) {
    float a2 = alpha * alpha;
    float d = ((cosTheta * a2 - cosTheta) * cosTheta + 1);
    return a2 / (d * d * K_PI);
}
// Microfacet.hlsli:105-128
static inline float evalPdfGGX_BVNDF(float _alpha, float3 i, float3 m) {
    float ndf = evalNdfGGX(_alpha, m.z);
    float2 ai = make_float2(_alpha * i.x, _alpha * i.y);
    float len2 = dot(ai, ai);
    float t = sqrtf_(len2 + i.z * i.z);
    float a = saturate(_alpha);
    float s = 1.0f + length(make_float2(i.x, i.y));
    float a2 = a * a, s2 = s * s;
    float k = (1.0f - a2) * s2 / (s2 + a2 * i.z * i.z);
    return ndf / (2.0f * (k * i.z + t));
}
// Microfacet.hlsli:185-207
static inline float3 sampleGGX_BVNDF(float _alpha, float3 i, float2 rand) {
    float3 i_std = normalize(make_float3(i.x * _alpha, i.y * _alpha, i.z));
    float phi = 2.0f * K_PI * rand.x;
    float a = saturate(_alpha);
    float s = 1.0f + length(make_float2(i.x, i.y));
    float a2 = a * a, s2 = s * s;
    float k = (1.0f - a2) * s2 / (s2 + a2 * i.z * i.z);
    float b = i.z > 0 ? k * i_std.z : i_std.z;
    float z = (1.0f - rand.y) * (1.0f + b) + (-b);          // mad(1-rand.y, 1+b, -b)
    float sinTheta = sqrtf_(saturate(1.0f - z * z));
    float sp, cp; dm_sincos(phi, sp, cp);
    float3 o_std = make_float3(sinTheta * cp, sinTheta * sp, z);
    float3 m_std = i_std + o_std;
    return normalize(make_float3(m_std.x * _alpha, m_std.y * _alpha, m_std.z));
}
// Microfacet.hlsli:232-239, 267-273
static inline float evalLambdaGGX(float alphaSqr, float cosTheta) {
    if (cosTheta <= 0) return 0;
    float cosThetaSqr = cosTheta * cosTheta;
    float tanThetaSqr = fmaxf_(1 - cosThetaSqr, 0) / cosThetaSqr;
    return 0.5f * (-1 + sqrtf_(1 + alphaSqr * tanThetaSqr));
}
static inline float evalMaskingSmithGGXCorrelated(float alpha, float cosThetaI, float cosThetaO) {
    float alphaSqr = alpha * alpha;
    float lambdaI = evalLambdaGGX(alphaSqr, cosThetaI);
    float lambdaO = evalLambdaGGX(alphaSqr, cosThetaO);
    return 1 / (1 + lambdaI + lambdaO);
}

// ---- lobes. All take wi, wo in the local frame (+z = shading normal).
// BxDF.hlsli:59-96 (Lambert) and :157-208 (Frostbite); selected by the DiffuseBrdf macro in the reference
struct DiffuseReflection {
    float3 albedo; float roughness; int model;
    float3 evalWeight(float3 wi, float3 wo) const {
        if (model == DiffuseBrdfLambert) return albedo;
        float3 h = normalize(wi + wo);
        float woDotH = dot(wo, h);
        float energyBias = lerpf(0.f, 0.5f, roughness);
        float energyFactor = lerpf(1.f, 1.f / 1.51f, roughness);
        float fd90 = energyBias + 2.f * woDotH * woDotH * roughness;
        float fd0 = 1.f;
        float wiScatter = evalFresnelSchlick(fd0, fd90, wi.z);
        float woScatter = evalFresnelSchlick(fd0, fd90, wo.z);
        return albedo * wiScatter * woScatter * energyFactor;
    }
    float3 eval(float3 wi, float3 wo) const {
        if (fminf_(wi.z, wo.z) < kMinCosTheta) return make_float3(0.f);
        if (model == DiffuseBrdfLambert) return K_1_PI * albedo * wo.z;
        return evalWeight(wi, wo) * K_1_PI * wo.z;
    }
    bool sample(float3 wi, float3& wo, float& pdf, float3& weight, uint& lobe, float& lobeP, float3 u) const {
        wo = sample_cosine_hemisphere_concentric(make_float2(u.x, u.y), pdf);
        lobe = Lobe_DiffuseReflection;
        if (fminf_(wi.z, wo.z) < kMinCosTheta) { weight = make_float3(0.f); lobeP = 0.0f; return false; }
        weight = evalWeight(wi, wo);
        lobeP = 1.0f;
        return true;
    }
    float evalPdf(float3 wi, float3 wo) const {
        if (fminf_(wi.z, wo.z) < kMinCosTheta) return 0.f;
        return K_1_PI * wo.z;
    }
};
// BxDF.hlsli:212-247
struct DiffuseTransmissionLambert {
    float3 albedo;
    float3 eval(float3 wi, float3 wo) const {
        if (fminf_(wi.z, -wo.z) < kMinCosTheta) return make_float3(0.f);
        return K_1_PI * albedo * -wo.z;
    }
    bool sample(float3 wi, float3& wo, float& pdf, float3& weight, uint& lobe, float& lobeP, float3 u) const {
        wo = sample_cosine_hemisphere_concentric(make_float2(u.x, u.y), pdf);
        wo.z = -wo.z;
        lobe = Lobe_DiffuseTransmission;
        if (fminf_(wi.z, -wo.z) < kMinCosTheta) { weight = make_float3(0.f); lobeP = 0.0f; return false; }
        weight = albedo; lobeP = 1.0f;
        return true;
    }
    float evalPdf(float3 wi, float3 wo) const {
        if (fminf_(wi.z, -wo.z) < kMinCosTheta) return 0.f;
        return K_1_PI * -wo.z;
    }
};
// BxDF.hlsli:250-268
static inline float EmsApprox(float r2, float NdV) {
    float r4 = r2 * r2;
    float nv0 = 0.2f * r2;
    float nv1 = 0.32f * r2 + 1.94f * r4;
    return lerpf(nv0, nv1, NdV);
}
static inline float3 MultiScatterSpecularApprox(float alpha, float NdV, float3 F0) {
    float Ems = EmsApprox(alpha, NdV);
    return make_float3(1 + F0.x * Ems, 1 + F0.y * Ems, 1 + F0.z * Ems);
}
// BxDF.hlsli:273-381
struct SpecularReflectionMicrofacet {
    float3 albedo; float alpha; uint activeLobes;
    bool hasLobe(uint l) const { return (activeLobes & l) != 0; }
    float3 eval(float3 wi, float3 wo) const {
        if (fminf_(wi.z, wo.z) < kMinCosTheta) return make_float3(0.f);
        if (alpha == 0.f) return make_float3(0.f);
        if (!hasLobe(Lobe_SpecularReflection)) return make_float3(0.f);
        float3 h = normalize(wi + wo);
        float wiDotH = dot(wi, h);
        float D = evalNdfGGX(alpha, h.z);
        float G = evalMaskingSmithGGXCorrelated(alpha, wi.z, wo.z);
        float3 F = evalFresnelSchlick(albedo, 1.f, wiDotH);
        float3 ms = MultiScatterSpecularApprox(alpha, wi.z, albedo);
        return ms * F * (D * G * 0.25f / wi.z);
    }
    float evalPdf(float3 wi, float3 wo) const {
        if (fminf_(wi.z, wo.z) < kMinCosTheta) return 0.f;
        if (alpha == 0.f) return 0.f;
        if (!hasLobe(Lobe_SpecularReflection)) return 0.f;
        float3 h = normalize(wi + wo);
        return evalPdfGGX_BVNDF(alpha, wi, h);
    }
    bool sample(float3 wi, float3& wo, float& pdf, float3& weight, uint& lobe, float& lobeP, float3 u) const {
        wo = make_float3(0.f); weight = make_float3(0.f); pdf = 0.f; lobe = Lobe_SpecularReflection; lobeP = 1.0f;
        if (wi.z < kMinCosTheta) return false;
        if (alpha == 0.f) {
            if (!hasLobe(Lobe_DeltaReflection)) return false;
            wo = make_float3(-wi.x, -wi.y, wi.z);
            pdf = 0.f;
            weight = evalFresnelSchlick(albedo, 1.f, wi.z);
            lobe = Lobe_DeltaReflection;
            return true;
        }
        if (!hasLobe(Lobe_SpecularReflection)) return false;
        float3 h = sampleGGX_BVNDF(alpha, wi, make_float2(u.x, u.y));
        float wiDotH = dot(wi, h);
        wo = 2.f * wiDotH * h - wi;
        if (wo.z < kMinCosTheta) return false;
        pdf = evalPdf(wi, wo);
        weight = eval(wi, wo) / pdf;
        lobe = Lobe_SpecularReflection;
        return true;
    }
};
// BxDF.hlsli:385-607
struct SpecularReflectionTransmissionMicrofacet {
    float3 transmissionAlbedo; float alpha; float eta; uint activeLobes; bool isThinSurface;
    bool hasLobe(uint l) const { return (activeLobes & l) != 0; }
    float3 eval(float3 wi, float3 wo) const {
        if (fminf_(wi.z, fabsf(wo.z)) < kMinCosTheta) return make_float3(0.f);
        if (alpha == 0.f) return make_float3(0.f);
        const bool hasReflection = hasLobe(Lobe_SpecularReflection), hasTransmission = hasLobe(Lobe_SpecularTransmission);
        const bool isReflection = wo.z > 0.f;
        if ((isReflection && !hasReflection) || (!isReflection && !hasTransmission)) return make_float3(0.f);
        float actualEta = (isThinSurface && !isReflection) ? 1.0f : eta;
        float3 h = normalize(wo + wi * (isReflection ? 1.f : actualEta));
        h = h * signf_(h.z);
        float wiDotH = dot(wi, h), woDotH = dot(wo, h);
        float D = evalNdfGGX(alpha, h.z);
        float G = evalMaskingSmithGGXCorrelated(alpha, wi.z, fabsf(wo.z));
        float F = evalFresnelDielectric(actualEta, wiDotH);
        if (isReflection) return make_float3(F * D * G * 0.25f / wi.z);
        float sqrtDenom = woDotH + actualEta * wiDotH;
        float t = actualEta * actualEta * wiDotH * woDotH / (wi.z * sqrtDenom * sqrtDenom);
        return transmissionAlbedo * (1.f - F) * D * G * fabsf(t);
    }
    float evalPdf(float3 wi, float3 wo) const {
        if (fminf_(wi.z, fabsf(wo.z)) < kMinCosTheta) return 0.f;
        if (alpha == 0.f) return 0.f;
        bool isReflection = wo.z > 0.f;
        const bool hasReflection = hasLobe(Lobe_SpecularReflection), hasTransmission = hasLobe(Lobe_SpecularTransmission);
        if ((isReflection && !hasReflection) || (!isReflection && !hasTransmission)) return 0.f;
        float actualEta = (isThinSurface && !isReflection) ? 1.0f : eta;
        float3 h = normalize(wo + wi * (isReflection ? 1.f : actualEta));
        h = h * signf_(h.z);
        float wiDotH = dot(wi, h), woDotH = dot(wo, h);
        float F = evalFresnelDielectric(actualEta, wiDotH);
        float pdf = evalPdfGGX_BVNDF(alpha, wi, h);
        if (isReflection) {
            if (woDotH <= 0.f) return 0.f;
            pdf *= wiDotH / woDotH;
        } else {
            if (woDotH > 0.f) return 0.f;
            pdf *= wiDotH * 4.0f;
            float sqrtDenom = woDotH + actualEta * wiDotH;
            float denom = sqrtDenom * sqrtDenom;
            pdf *= fabsf(woDotH) / denom;
        }
        if (hasReflection && hasTransmission) pdf *= isReflection ? F : 1.f - F;
        return clampf(pdf, 0, FLT_MAX_);
    }
    bool sample(float3 wi, float3& wo, float& pdf, float3& weight, uint& lobe, float& lobeP, float3 u) const {
        wo = make_float3(0.f); weight = make_float3(0.f); pdf = 0.f; lobe = Lobe_SpecularReflection; lobeP = 1;
        if (wi.z < kMinCosTheta) return false;
        float lobeSample = u.z;
        if (alpha == 0.f) {
            const bool hasReflection = hasLobe(Lobe_DeltaReflection), hasTransmission = hasLobe(Lobe_DeltaTransmission);
            if (!(hasReflection || hasTransmission)) return false;
            float cosThetaT;
            float F = evalFresnelDielectric(eta, wi.z, cosThetaT);
            bool isReflection = hasReflection;
            if (hasReflection && hasTransmission) { isReflection = lobeSample < F; lobeP = isReflection ? F : (1 - F); }
            else if (hasTransmission && F == 1.f) return false;
            float actualEta = eta;
            if (isThinSurface && !isReflection) { actualEta = 1.0f; F = evalFresnelDielectric(actualEta, wi.z, cosThetaT); }
            pdf = 0.f;
            weight = isReflection ? make_float3(1.f) : transmissionAlbedo;
            if (!(hasReflection && hasTransmission)) weight *= (isReflection ? F : 1.f - F);
            wo = isReflection ? make_float3(-wi.x, -wi.y, wi.z) : make_float3(-wi.x * actualEta, -wi.y * actualEta, -cosThetaT);
            lobe = isReflection ? Lobe_DeltaReflection : Lobe_DeltaTransmission;
            if (fabsf(wo.z) < kMinCosTheta || ((wo.z > 0.f) != isReflection)) return false;
            return true;
        }
        const bool hasReflection = hasLobe(Lobe_SpecularReflection), hasTransmission = hasLobe(Lobe_SpecularTransmission);
        if (!(hasReflection || hasTransmission)) return false;
        float3 h = sampleGGX_BVNDF(alpha, wi, make_float2(u.x, u.y));
        float wiDotH = dot(wi, h);
        float cosThetaT;
        float F = evalFresnelDielectric(eta, wiDotH, cosThetaT);
        bool isReflection = hasReflection;
        if (hasReflection && hasTransmission) isReflection = lobeSample < F;
        else if (hasTransmission && F == 1.f) return false;
        float actualEta = eta;
        if (isThinSurface && !isReflection) { actualEta = 1.0f; F = evalFresnelDielectric(actualEta, wi.z, cosThetaT); }
        wo = isReflection ? (2.f * wiDotH * h - wi) : ((actualEta * wiDotH - cosThetaT) * h - actualEta * wi);
        if (fabsf(wo.z) < kMinCosTheta || ((wo.z > 0.f) != isReflection)) return false;
        lobe = isReflection ? Lobe_SpecularReflection : Lobe_SpecularTransmission;
        pdf = evalPdf(wi, wo);
        weight = pdf > 0.f ? eval(wi, wo) / pdf : make_float3(0.f);
        return true;
    }
};

// BxDF.hlsli:615-702
struct StandardBSDFData {
    float3 diffuse; float roughness; float3 specular; float metallic;
    float3 transmission; float diffuseTransmission; float specularTransmission; float eta;
};
// IBSDF.hlsli:95-116
struct BSDFSample {
    float3 wo; float pdf; float3 weight; uint lobe; float lobeP;
    bool isLobe(uint type) const { return (lobe & type) != 0; }
};

// BxDF.hlsli:709-970
struct FalcorBSDF {
    DiffuseReflection diffuseReflection;
    DiffuseTransmissionLambert diffuseTransmission;
    SpecularReflectionMicrofacet specularReflection;
    SpecularReflectionTransmissionMicrofacet specularReflectionTransmission;
    float diffTrans, specTrans;
    float pDiffuseReflection, pDiffuseTransmission, pSpecularReflection, pSpecularReflectionTransmission;

    // :737-814. NB the reference call passes (V, N) into parameters named (N, V); only dot(V,N) is used.
    void init(const MaterialHeader mtl, float3 N, float3 V, const StandardBSDFData& data, int diffuseModel) {
        bool isThinSurface = mtl.isThinSurface();
        float3 dataTransmission = data.transmission;
        float3 transmissionAlbedo = isThinSurface ? dataTransmission
            : make_float3(sqrtf_(dataTransmission.x), sqrtf_(dataTransmission.y), sqrtf_(dataTransmission.z));
        float dataRoughness = data.roughness;
        diffuseReflection.albedo = data.diffuse;
        diffuseReflection.roughness = dataRoughness;
        diffuseReflection.model = diffuseModel;
        diffuseTransmission.albedo = transmissionAlbedo;
        float alpha = dataRoughness * dataRoughness;
        if (alpha < kMinGGXAlpha) alpha = 0.f;
        const uint activeLobes = mtl.getActiveLobes();
        float3 dataSpecular = data.specular;
        float dataEta = data.eta;
        specularReflection.albedo = dataSpecular;
        specularReflection.alpha = alpha;
        specularReflection.activeLobes = activeLobes;
        specularReflectionTransmission.transmissionAlbedo = transmissionAlbedo;
        specularReflectionTransmission.alpha = (dataEta == 1.f) ? 0.f : alpha;
        specularReflectionTransmission.eta = dataEta;
        specularReflectionTransmission.activeLobes = activeLobes;
        specularReflectionTransmission.isThinSurface = isThinSurface;
        diffTrans = data.diffuseTransmission;
        specTrans = data.specularTransmission;
        float dataMetallic = data.metallic;
        float metallicBRDF = dataMetallic * (1.f - specTrans);
        float dielectricBSDF = (1.f - dataMetallic) * (1.f - specTrans);
        float specularBSDF = specTrans;
        float diffuseWeight = Luminance(data.diffuse);
        float specularWeight = Luminance(evalFresnelSchlick(dataSpecular, 1.f, dot(V, N)));
        pDiffuseReflection = (activeLobes & Lobe_DiffuseReflection) ? diffuseWeight * dielectricBSDF * (1.f - diffTrans) : 0.f;
        pDiffuseTransmission = (activeLobes & Lobe_DiffuseTransmission) ? diffuseWeight * dielectricBSDF * diffTrans : 0.f;
        pSpecularReflection = (activeLobes & (Lobe_SpecularReflection | Lobe_DeltaReflection)) ? specularWeight * (metallicBRDF + dielectricBSDF) : 0.f;
        pSpecularReflectionTransmission = (activeLobes & (Lobe_SpecularReflection | Lobe_DeltaReflection | Lobe_SpecularTransmission | Lobe_DeltaTransmission)) ? specularBSDF : 0.f;
        float normFactor = pDiffuseReflection + pDiffuseTransmission + pSpecularReflection + pSpecularReflectionTransmission;
        if (normFactor > 0.f) {
            normFactor = 1.f / normFactor;
            pDiffuseReflection *= normFactor; pDiffuseTransmission *= normFactor;
            pSpecularReflection *= normFactor; pSpecularReflectionTransmission *= normFactor;
        }
    }
    // :842-863
    static uint getLobes(const StandardBSDFData& data) {
        float alpha = data.roughness * data.roughness;
        bool isDelta = alpha < kMinGGXAlpha;
        float diffTrans = data.diffuseTransmission, specTrans = data.specularTransmission;
        uint lobes = isDelta ? Lobe_DeltaReflection : Lobe_SpecularReflection;
        if (any_gt0(data.diffuse) && specTrans < 1.f) {
            if (diffTrans < 1.f) lobes |= Lobe_DiffuseReflection;
            if (diffTrans > 0.f) lobes |= Lobe_DiffuseTransmission;
        }
        if (specTrans > 0.f) lobes |= (isDelta ? Lobe_DeltaTransmission : Lobe_SpecularTransmission);
        return lobes;
    }
    // :865-874
    float4 eval(float3 wi, float3 wo) const {
        float3 diffuse = make_float3(0.f), specular = make_float3(0.f);
        if (pDiffuseReflection > 0.f) diffuse += (1.f - specTrans) * (1.f - diffTrans) * diffuseReflection.eval(wi, wo);
        if (pDiffuseTransmission > 0.f) diffuse += (1.f - specTrans) * diffTrans * diffuseTransmission.eval(wi, wo);
        if (pSpecularReflection > 0.f) specular += (1.f - specTrans) * specularReflection.eval(wi, wo);
        if (pSpecularReflectionTransmission > 0.f) specular += specTrans * specularReflectionTransmission.eval(wi, wo);
        return make_float4(diffuse + specular, Average(specular));
    }
    // :876-960 (RecycleSelectSamples == 1: three random numbers, .z reused after lobe selection)
    bool sample(float3 wi, float3& wo, float& pdf, float3& weight, uint& lobe, float& lobeP, float3 u) const {
        wo = make_float3(0.f); weight = make_float3(0.f); pdf = 0.f; lobe = Lobe_DiffuseReflection; lobeP = 0.0f;
        bool valid = false;
        float uSelect = u.z;
        if (uSelect < pDiffuseReflection) {
            u.z = clampf(uSelect / pDiffuseReflection, 0, cOneMinusEpsilon);
            valid = diffuseReflection.sample(wi, wo, pdf, weight, lobe, lobeP, u);
            weight = weight / pDiffuseReflection;
            weight *= (1.f - specTrans) * (1.f - diffTrans);
            pdf *= pDiffuseReflection;
            lobeP *= pDiffuseReflection;
            if (pSpecularReflection > 0.f) pdf += pSpecularReflection * specularReflection.evalPdf(wi, wo);
            if (pSpecularReflectionTransmission > 0.f) pdf += pSpecularReflectionTransmission * specularReflectionTransmission.evalPdf(wi, wo);
        } else if (uSelect < pDiffuseReflection + pDiffuseTransmission) {
            valid = diffuseTransmission.sample(wi, wo, pdf, weight, lobe, lobeP, u);
            weight = weight / pDiffuseTransmission;
            weight *= (1.f - specTrans) * diffTrans;
            pdf *= pDiffuseTransmission;
            lobeP *= pDiffuseTransmission;
            if (pSpecularReflectionTransmission > 0.f) pdf += pSpecularReflectionTransmission * specularReflectionTransmission.evalPdf(wi, wo);
        } else if (uSelect < pDiffuseReflection + pDiffuseTransmission + pSpecularReflection) {
            u.z = clampf((uSelect - (pDiffuseReflection + pDiffuseTransmission)) / pSpecularReflection, 0, cOneMinusEpsilon);
            valid = specularReflection.sample(wi, wo, pdf, weight, lobe, lobeP, u);
            weight = weight / pSpecularReflection;
            weight *= (1.f - specTrans);
            pdf *= pSpecularReflection;
            lobeP *= pSpecularReflection;
            if (pDiffuseReflection > 0.f) pdf += pDiffuseReflection * diffuseReflection.evalPdf(wi, wo);
            if (pSpecularReflectionTransmission > 0.f) pdf += pSpecularReflectionTransmission * specularReflectionTransmission.evalPdf(wi, wo);
        } else if (pSpecularReflectionTransmission > 0.f) {
            u.z = clampf((uSelect - (pDiffuseReflection + pDiffuseTransmission + pSpecularReflection)) / pSpecularReflectionTransmission, 0, cOneMinusEpsilon);
            valid = specularReflectionTransmission.sample(wi, wo, pdf, weight, lobe, lobeP, u);
            weight = weight / pSpecularReflectionTransmission;
            weight *= specTrans;
            pdf *= pSpecularReflectionTransmission;
            lobeP *= pSpecularReflectionTransmission;
            if (pDiffuseReflection > 0.f) pdf += pDiffuseReflection * diffuseReflection.evalPdf(wi, wo);
            if (pDiffuseTransmission > 0.f) pdf += pDiffuseTransmission * diffuseTransmission.evalPdf(wi, wo);
            if (pSpecularReflection > 0.f) pdf += pSpecularReflection * specularReflection.evalPdf(wi, wo);
        }
        if (!valid || (lobe & Lobe_Delta) != 0) pdf = 0.0f;
        return valid;
    }
    // :962-970
    float evalPdf(float3 wi, float3 wo) const {
        float pdf = 0.f;
        if (pDiffuseReflection > 0.f) pdf += pDiffuseReflection * diffuseReflection.evalPdf(wi, wo);
        if (pDiffuseTransmission > 0.f) pdf += pDiffuseTransmission * diffuseTransmission.evalPdf(wi, wo);
        if (pSpecularReflection > 0.f) pdf += pSpecularReflection * specularReflection.evalPdf(wi, wo);
        if (pSpecularReflectionTransmission > 0.f) pdf += pSpecularReflectionTransmission * specularReflectionTransmission.evalPdf(wi, wo);
        return pdf;
    }
};

// StandardBSDF.hlsli:34-92 ("ActiveBSDF")
struct StandardBSDF {
    StandardBSDFData data;
    int diffuseModel;     // DiffuseBrdf macro (BxDFConfig.hlsli:24)
    float4 eval(const ShadingData& sd, float3 wo) const {
        float3 wiLocal = sd.toLocal(sd.V), woLocal = sd.toLocal(wo);
        FalcorBSDF b; b.init(sd.mtl, sd.V, sd.N, data, diffuseModel);
        return b.eval(wiLocal, woLocal);
    }
    bool sample(const ShadingData& sd, float4 u, BSDFSample& result) const {
        float3 wiLocal = sd.toLocal(sd.V), woLocal = make_float3(0.f);
        FalcorBSDF b; b.init(sd.mtl, sd.V, sd.N, data, diffuseModel);
        bool valid = b.sample(wiLocal, woLocal, result.pdf, result.weight, result.lobe, result.lobeP, make_float3(u.x, u.y, u.z));
        result.wo = sd.fromLocal(woLocal);
        return valid;
    }
    float evalPdf(const ShadingData& sd, float3 wo) const {
        float3 wiLocal = sd.toLocal(sd.V), woLocal = sd.toLocal(wo);
        FalcorBSDF b; b.init(sd.mtl, sd.V, sd.N, data, diffuseModel);
        return b.evalPdf(wiLocal, woLocal);
    }
    uint getLobes() const { return FalcorBSDF::getLobes(data); }
};

#pragma clang force_cuda_host_device end
} // namespace ptk
