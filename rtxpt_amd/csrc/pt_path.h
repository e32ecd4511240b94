// mi355pt — per-vertex path logic of the reference-mode estimator, shared by the wavefront kernels.
// One PathState is 80 bytes exactly like the reference's PathPayload (PathPayload.hlsli:19-21, PathState.hlsli:83-267);
// in HBM it lives as five uint4 SoA streams (see pt_wavefront.hip) so that a wave reads/writes 1 KiB per stream instruction.
// Function-by-function anchors (paths relative to /root/reference/Rtxpt/Shaders/):
//   PathTracer/PathTracer.hlsli:40-45,47-91,182-208,217-380,382-404,407-503,505-762
//   PathTracer/PathTracerNEE.hlsli:41-161,166-275,277-346          NEE: WRS candidates -> one shadow ray
//   PathTracer/PathTracerNestedDielectrics.hlsli:24-128, PathTracer/Rendering/Materials/InteriorList.hlsli:28-248
//   PathTracer/PathTracerHelpers.hlsli:126-219, PathTracer/Rendering/Materials/TexLODHelpers.hlsli:57-143
//   PathTracerBridgeDonut.hlsli:152-256,280-428,543-564,612-887    Bridge::loadSurface & friends
// Wavefront split of HandleHit: everything up to and including the NEE light sample happens in k_shade; the visibility
// ray + "L += radiance" half of ProcessLightSample (PathTracerNEE.hlsli:199-265) is deferred to the shadow queue. The
// deferred radiance is computed with the same operations, in the same order, as the reference computes it after the ray.
#pragma once
#include "pt_bsdf.h"
#include "pt_rng.h"
#include "pt_scene.h"

#ifndef PT_SHADE_TRI
// 1: loadSurface reads the flat 128-byte ShadeTri record of the hit primitive; 0: the five-hop gather through primInfo / sub-instance / geometry / index /
// vertex streams
#define PT_SHADE_TRI 1
#endif
#ifndef PT_SHADE_NOINLINE
#define PT_SHADE_NOINLINE
#endif

namespace ptk {
#pragma clang force_cuda_host_device begin

static const float kMaxRayTravel = 1e15f;
static const float kSpecularRoughnessThreshold = 0.25f;

struct PathTracerCameraData {
    float3 PosW; float NearZ; float3 DirectionW; float PixelConeSpreadAngle; float3 CameraU; float FarZ;
    float3 CameraV; float FocalDistance; float3 CameraW; float AspectRatio; uint2 ViewportSize; float ApertureRadius; float _padding0;
    float2 Jitter; float _padding1, _padding2;
};
static_assert(sizeof(PathTracerCameraData) == 112, "PathTracerCameraData layout");
struct PtSettings {
    uint bounceCount, diffuseBounceCount; float perPixelJitterAAScale, texLODBias, fireflyFilterThreshold, envMapDiffuseSampleMIPLevel;
    uint NEEEnabled, NEEType, NEECandidateSamples, NEEFullSamples, enableRussianRoulette, nestedDielectricsQuality, enableLDSamplerForBSDF, diffuseBrdf, useFp16Types, _pad;      // useFp16Types: lp types in 16 bits (the reference's default build)
};
static_assert(sizeof(PtSettings) == 64, "PtSettings layout");

// TexLODHelpers.hlsli:57-123
struct RayCone {
    uint widthSpreadAngleFP16;
    float getWidth() const { return f16tof32(widthSpreadAngleFP16 >> 16); }
    float getSpreadAngle() const { return f16tof32(widthSpreadAngleFP16 & 0xffffu); }
    static RayCone make(float width, float angle) { RayCone r; r.widthSpreadAngleFP16 = (f32tof16(width) << 16) | f32tof16(angle); return r; }
    RayCone propagateDistance(float hitT) const { float angle = getSpreadAngle(), width = getWidth(); return make(angle * hitT + width, angle); }
    static float SafeLog2(float x) { return dm_log2(clampf(x, FLT_MIN_, FLT_MAX_)); }
    float computeLOD(float triLODConstant, float3 rayDir, float3 normal, bool moreDetailOnSlopes) const {
        float lambda = triLODConstant;
        float distTerm = fabsf(getWidth());
        float normalTerm = fabsf(dot(rayDir, normal));
        if (moreDetailOnSlopes) normalTerm = sqrtf_(normalTerm);
        lambda += SafeLog2(distTerm / normalTerm);
        return lambda;
    }
};
// InteriorList.hlsli:28-248
struct InteriorList {
    static const uint kNoMaterial = 0xffffffffu, kMaterialMask = (1u << 28) - 1u, kNestedPriorityOffset = 28, kMaxNestedPriority = 15;
    uint slots[2];
    bool isEmpty() const { return slots[0] == 0; }
    uint getTopNestedPriority() const { return slots[0] >> kNestedPriorityOffset; }
    uint getTopMaterialID() const { return slots[0] != 0 ? (slots[0] & kMaterialMask) : kNoMaterial; }
    uint getNextMaterialID() const { return slots[1] != 0 ? (slots[1] & kMaterialMask) : kNoMaterial; }
    bool isTrueIntersection(uint nestedPriority) const { return nestedPriority == 0 || nestedPriority >= getTopNestedPriority(); }
    void handleIntersection(uint materialID, uint nestedPriority, bool entering) {
        if (nestedPriority == 0) nestedPriority = kMaxNestedPriority;
        uint slot = (nestedPriority << kNestedPriorityOffset) | (materialID & kMaterialMask);
        if (entering && slots[0] == 0) slots[0] = slot;
        else if (!entering && slots[0] != 0 && (slots[0] & kMaterialMask) == materialID) slots[0] = 0;
        else if (entering && slots[1] == 0) slots[1] = slot;
        else if (!entering && slots[1] != 0 && (slots[1] & kMaterialMask) == materialID) slots[1] = 0;
        if (slots[0] < slots[1]) { uint t = slots[0]; slots[0] = slots[1]; slots[1] = t; }
    }
};
enum : uint {
    PF_active = 1 << 0, PF_hit = 1 << 1, PF_transmission = 1 << 2, PF_specular = 1 << 3, PF_delta = 1 << 4,
    PF_insideDielectricVolume = 1 << 5, PF_terminateAtNextBounce = 1 << 6, PF_enableThreadReorder = 1 << 9, PF_deltaOnlyPath = 1 << 12,
};
enum { PC_DiffuseBounces = 0, PC_RejectedHits = 1 };
static const uint kVertexIndexBitCount = 10, kVertexIndexBitMask = (1u << 10) - 1u;

// PathState.hlsli:83-267; the stableBranchID word (unused in reference mode) carries the sample index of the path
struct PathState {
    float3 origin; uint id; float3 dir; float sceneLength;
    uint pack23[2]; uint pack45[2];
    InteriorList interiorList; uint packedCounters; RayCone rayCone;
    uint pack0, pack1, flagsAndVertexIndex, sampleIndex;

    void SetFireflyFilterK_BsdfScatterPdf(float k, float pdf) { pack0 = (f32tof16(clampf(k, 0, HLF_MAX)) << 16) | f32tof16(clampf(pdf, 0, HLF_MAX)); }
    float GetFireflyFilterK() const { return f16tof32(pack0 >> 16); }
    float GetBsdfScatterPdf() const { return f16tof32(pack0 & 0xFFFFu); }
    void SetPackedMISInfo_ThpRuRuCorrection(uint mis, float c) { pack1 = (mis << 16) | f32tof16(clampf(c, 0, HLF_MAX)); }
    uint GetPackedMISInfo() const { return pack1 >> 16; }
    float GetThpRuRuCorrection() const { return f16tof32(pack1 & 0xFFFFu); }
    void SetThp(float3 thp) { thp = clamp3(thp, 0.f, HLF_MAX); pack23[0] = Fp32ToFp16NoClamp(make_float2(thp.x, thp.y)); pack23[1] = Fp32ToFp16NoClamp(make_float2(thp.z, 0.f)); }
    float3 GetThp() const { float2 a = Fp16ToFp32(pack23[0]), b = Fp16ToFp32(pack23[1]); return make_float3(a.x, a.y, b.x); }
    void SetL(float4 l) { pack45[0] = Fp32ToFp16NoClamp(make_float2(clampf(l.x, 0, HLF_MAX), clampf(l.y, 0, HLF_MAX))); pack45[1] = Fp32ToFp16NoClamp(make_float2(clampf(l.z, 0, HLF_MAX), clampf(l.w, 0, HLF_MAX))); }
    float4 GetL() const { float2 a = Fp16ToFp32(pack45[0]), b = Fp16ToFp32(pack45[1]); return make_float4(a.x, a.y, b.x, b.y); }
    bool hasFlag(uint f) const { return (flagsAndVertexIndex & (f << kVertexIndexBitCount)) != 0; }
    void setFlag(uint f, bool v = true) { uint bit = f << kVertexIndexBitCount; if (v) flagsAndVertexIndex |= bit; else flagsAndVertexIndex &= ~bit; }
    bool isActive() const { return hasFlag(PF_active); }
    void terminate() { setFlag(PF_active, false); }
    bool isTerminatingAtNextBounce() const { return hasFlag(PF_terminateAtNextBounce); }
    void clearScatterEventFlags() { flagsAndVertexIndex &= ~((PF_transmission | PF_specular | PF_delta) << kVertexIndexBitCount); }
    uint getCounter(uint type) const { return (packedCounters >> (type << 3)) & 0xff; }
    void incrementCounter(uint type) { packedCounters += (1u << (type << 3)); }
    uint getVertexIndex() const { return flagsAndVertexIndex & kVertexIndexBitMask; }
    void incrementVertexIndex() { flagsAndVertexIndex += 1; }
    void decrementVertexIndex() { flagsAndVertexIndex -= 1; }
};
// PathTracerTypes.hlsli:89-160
struct NEEBSDFMISInfo {
    bool LightSamplingEnabled, LightSamplingIsSSC; uint CandidateSamples, FullSamples;
    static NEEBSDFMISInfo empty() { NEEBSDFMISInfo r; r.LightSamplingEnabled = false; r.LightSamplingIsSSC = false; r.CandidateSamples = 0; r.FullSamples = 0; return r; }
    static NEEBSDFMISInfo Unpack16bit(uint p) { NEEBSDFMISInfo r; r.LightSamplingEnabled = (p & (1u << 15)) != 0; r.LightSamplingIsSSC = (p & (1u << 13)) != 0; r.CandidateSamples = (p >> 6) & 0x3F; r.FullSamples = p & 0x3F; return r; }
    uint Pack16bit() const { return ((LightSamplingEnabled ? 1u : 0u) << 15) | ((LightSamplingIsSSC ? 1u : 0u) << 13) | ((CandidateSamples & 0x3F) << 6) | (FullSamples & 0x3F); }
};
struct LightSample {
    float3 Li; float Distance; float3 Direction; uint LightIndex; float SelectionPdf, SolidAnglePdf; bool LightSampleableByBSDF, FromLocalDistribution;
    bool Valid() const { return any_gt0(Li); }
};
struct SurfaceData { ShadingData shadingData; StandardBSDF bsdf; float interiorIoR; uint neeTriangleLightIndex; uint neeAnalyticLightIndex; };
// what k_shade hands to the shadow queue (the deferred half of ProcessLightSample) NEE-AT feedback (TemporalFeedbackRequired, NEEFullSamples 1):
// ProcessLightSample draws one more random number and writes the pixel's feedback reservoir only when the light is VISIBLE (PathTracerNEE.hlsli:266-273), and
// the same generator then serves Russian roulette (PathTracer.hlsli:757-759). k_shade therefore works out both continuations: the path is stored as "not
// visible"; fbLight / fbWeight / fbRandom are the reservoir update and rrFix what the visible case changes on the path (bit 0: the roulette outcomes differ,
// bit 1: terminate-at-next-bounce in the visible case, bits 16-31: the fp16 roulette correction of the visible case).
struct ShadowRequest { bool valid; float3 origin, dir; float tmax; float3 radiance; uint fbLight; float fbWeight, fbRandom; uint rrFix; };
// NEEFullSamples != 1 (HandleNEE_MultipleSamples, PathTracerNEE.hlsli:277-301): every path vertex that applies NEE reserves a group of fullSamples
// consecutive shadow-queue entries (sample s at base + s, samples without a light marked tmax < 0) and k_resolve_nee folds the visible ones in sample order.
struct ShadowSink { float4* q0; float4* q1; float4* q2; uint* count; unsigned long long* valid; uint pathIndex; };

// PathTracerHelpers.hlsli:164-219
static inline float ComputeRayConeSpreadAngleExpansionByScatterPDF(float bsdfScatterPdf, float growthFactor) {
    return growthFactor * 2.0f * FastACos(fmaxf_(-1.0f, 1.0f - (1.0f / bsdfScatterPdf) / (2.0f * K_PI)));
}
// LP = LPOps<false> / LPOps<true> (pt_vec.h): the reference's two builds of its "lp" types (fp32 / binary16, the default)
template <class LP> static inline float ComputeNewScatterFireflyFilterK(float currentK, float bouncePDF, float lobeP) {
    const float minK = 0.00001f;
    float angle = (bouncePDF == 0) ? 0.f : ComputeRayConeSpreadAngleExpansionByScatterPDF(bouncePDF, 1.0f);
    const float k = 32;
    float p = k / (k + angle * angle);
    p *= FastSqrt(lobeP);
    return LP::r(fmaxf_(minK, currentK * p));               // returns lpfloat: the NEE path uses the value before it is ever packed
}
// lpfloat3 (lpfloat3, lpfloat, lpfloat): all three are lp values
template <class LP> static inline float3 FireflyFilter(float3 signalIn, float threshold, float fireflyFilterK) {
    float t = LP::mul(threshold, fireflyFilterK);
    float maxR = LP::average3(signalIn);
    if (maxR > t) signalIn = LP::mul3(LP::div3(signalIn, maxR), t);
    return signalIn;
}
static inline float FireflyFilterShort(float signalAverage, float threshold, float fireflyFilterK) {
    float t = threshold * fireflyFilterK;
    return (signalAverage > t) ? (1.0f / signalAverage * t) : 1.0f;
}
// PathTracerHelpers.hlsli:48-52
static inline float ComputeLowGrazingAngleFalloff(float3 lightDirection, float3 n, float falloffFrom, float falloffRange) {
    return saturate((dot(lightDirection, n) - falloffFrom) / falloffRange);
}
// TexLODHelpers.hlsli:129-143 Donut's ConvertSpecularGlossToMetalRough (donut/shaders/scene_material.hlsli, un-vendored), called by EvaluateSceneMaterialRTXPT
// for PTMaterialFlags_UseSpecularGlossModel materials (PathTracerBridgeDonut.hlsli:318-333; ENABLE_METAL_ROUGH_RECONSTRUCTION 1, :15, :772-774): restated from
// the algorithm it follows, the Khronos KHR_materials_pbrSpecularGlossiness "convert-between-workflows" sample (solveMetallic + the two base-colour estimates
// blended by metallic^2, perceived brightness = sqrt(0.299 r^2 + 0.587 g^2 + 0.114 b^2), dielectric specular 0.04). UNPINNED: the function's own text is
// outside the tree.
static inline float GetPerceivedBrightness(float3 c) { return sqrtf_((0.299f * c.x * c.x + 0.587f * c.y * c.y) + 0.114f * c.z * c.z); }
static inline void ConvertSpecularGlossToMetalRough(float3 diffuseColor, float3 specularColor, float3& baseColor, float& metalness) {
    const float epsilon = 1e-6f, dielectricSpecular = 0.04f;
    float diffuseBrightness = GetPerceivedBrightness(diffuseColor), specularBrightness = GetPerceivedBrightness(specularColor);
    float oneMinusSpecularStrength = 1.0f - fmaxf_(specularColor.x, fmaxf_(specularColor.y, specularColor.z));
    metalness = 0.0f;
    if (!(specularBrightness < dielectricSpecular)) {
        float b = (diffuseBrightness * oneMinusSpecularStrength / (1.0f - dielectricSpecular) + specularBrightness) - 2.0f * dielectricSpecular;
        float c = dielectricSpecular - specularBrightness;
        float D = fmaxf_(b * b - 4.0f * dielectricSpecular * c, 0.0f);
        metalness = saturate((-b + sqrtf_(D)) / (2.0f * dielectricSpecular));
    }
    float3 baseColorFromDiffuse = diffuseColor * (oneMinusSpecularStrength / (1.0f - dielectricSpecular) / fmaxf_(1.0f - metalness, epsilon));
    float3 baseColorFromSpecular = (specularColor - make_float3(dielectricSpecular * (1.0f - metalness))) * (1.0f / fmaxf_(metalness, epsilon));
    baseColor = saturate3(lerp3(baseColorFromDiffuse, baseColorFromSpecular, metalness * metalness));
}
static inline float computeRayConeTriangleLODValue(const float3 v[3], const float2 t[3], const float3x4& M) {
    float2 tx10 = t[1] - t[0], tx20 = t[2] - t[0];
    float Ta = fabsf(tx10.x * tx20.y - tx20.x * tx10.y);
    float3 edge01 = xform_vector(M, v[1] - v[0]);
    float3 edge02 = xform_vector(M, v[2] - v[0]);
    float Pa = length(cross(edge01, edge02));
    return 0.5f * RayCone::SafeLog2(Ta / Pa);
}

// TriangleCurvatureApprox_GradN (PathTracerBridgeDonut.hlsli:92-149): a curvature proxy of one triangle in ~1/length units — the RMS gradient of a linear
// normal field fitted over the triangle in a 2D basis of its (world-space) plane. Feeds the automatic motion-vector block types of Bridge::loadSurface
// (:704-716).
static inline float TriangleCurvatureApprox_GradN(const float3 vertexPositions[3], const float3 vertexNormals[3], const float3x4& transform) {
    const float eps = 1e-8f;
    float3 e10 = xform_vector(transform, vertexPositions[1] - vertexPositions[0]);      // mul((float3x3)transform, p1 - p0)
    float e10Len = length(e10);
    if (e10Len < eps) return 0.0f;
    float3 e1 = e10 / e10Len;
    float3 e20 = xform_vector(transform, vertexPositions[2] - vertexPositions[0]);
    float u2 = dot(e20, e1);
    float3 t = e20 - e1 * u2;
    float tLen = length(t);
    if (tLen < eps) return 0.0f;
    float3 e2 = t / tLen;
    float u1 = e10Len;
    float v2 = dot(e20, e2);
    float3 dn1 = vertexNormals[1] - vertexNormals[0], dn2 = vertexNormals[2] - vertexNormals[0];
    float3 a = dn1 / fmaxf_(u1, eps);
    float denomV = fabsf(v2) < eps ? (v2 >= 0.0f ? eps : -eps) : v2;
    float3 b = (dn2 - a * u2) / denomV;
    return sqrtf_(dot(a, a) + dot(b, b));
}
// what Bridge::loadSurface hands to its motion-vector block decision (BridgeDonut:704-716): donutGS.curvatureWS (0 for a mesh without vertex normals: the
// sample is zero-initialised, :164) and abs(dot(rayDir, -N)) with the shading normal BEFORE adjustShadingNormal. Asked for by the stable-plane passes only
// (null otherwise: nothing is computed).
struct MVBlockInputs { float curvatureWS, projectionTerm; };

// LP16: which build of the reference's lp types this instance restates (PtSettings::useFp16Types selects it at launch; the data members are the same)
template <bool LP16> struct PathKernelContextT {
    typedef LPOps<LP16> LP;
    DeviceScene sc; PtSettings S; PathTracerCameraData cam;

    bool HasFinishedSurfaceBounces(uint vertexIndex, uint diffuseBounces) const {      // PathTracer.hlsli:40-45
        if (S.bounceCount < vertexIndex) return true;
        return diffuseBounces > S.diffuseBounceCount;
    }
    // EmptyPathInitialize + SetupPathPrimaryRay + Bridge::computeCameraRay (PathTracer.hlsli:47-119, BridgeDonut:543-564, PathTracerHelpers.hlsli:126-153)
    // Bridge::computeCameraRay (BridgeDonut:543-564) for a pixel and sample index: origin on the near plane and direction
    void computeCameraRay(uint px, uint py, uint sampleIndex, float3& o, float3& d) const {
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make((px << 16) | py, 0, sampleIndex);
        SampleSequenceGenerator sg = SampleSequenceGenerator::make(vb);
        float2 r0 = sampleNext2D(sg);
        float2 subPixelOffset = make_float2(cam.Jitter.x + (r0.x - 0.5f) * S.perPixelJitterAAScale, cam.Jitter.y + (r0.y - 0.5f) * S.perPixelJitterAAScale);
        float2 dof = sampleNext2D(sg);
        float2 pp = make_float2(((float)px + 0.5f + -subPixelOffset.x) / (float)cam.ViewportSize.x, ((float)py + 0.5f + subPixelOffset.y) / (float)cam.ViewportSize.y);
        float2 ndc = make_float2(2.f * pp.x + -1.f, -2.f * pp.y + 1.f);
        float3 org = cam.PosW;
        float3 dir = (ndc.x * cam.CameraU + ndc.y * cam.CameraV) + cam.CameraW;
        float2 ap = sample_disk(dof);
        float3 rayTarget = org + dir;
        org = org + cam.ApertureRadius * (ap.x * normalize(cam.CameraU) + ap.y * normalize(cam.CameraV));
        dir = normalize(rayTarget - org);
        float invCos = 1.f / dot(normalize(cam.CameraW), dir);
        float tMin = cam.NearZ * invCos;
        o = org + dir * tMin; d = dir;
    }
    // the reference-mode guide-buffer dump (PathTracer.hlsli:487, 684 -> Bridge::ExportNonSurface / ExportSurface): only the depth is kept, for NEE-AT's
    // disocclusion test
    void ExportDepth(const PathState& path, float3 virtualWorldPos) const {
        sc.lights.DepthExport[(path.id & 0xFFFFu) * sc.lights.DepthWidth + (path.id >> 16)] = LightTable_ClipDepth(sc.lights, virtualWorldPos);
    }
    PathState generate(uint px, uint py, uint sampleIndex) const {
        PathState p; __builtin_memset(&p, 0, sizeof(p));
        p.id = (px << 16) | py; p.sampleIndex = sampleIndex;
        p.SetThp(make_float3(1.f));
        p.setFlag(PF_active); p.setFlag(PF_deltaOnlyPath, true);
        p.rayCone = RayCone::make(0, cam.PixelConeSpreadAngle);
        p.SetL(make_float4(0, 0, 0, 0));
        p.SetFireflyFilterK_BsdfScatterPdf(1.0f, 0.0f);
        p.SetPackedMISInfo_ThpRuRuCorrection(NEEBSDFMISInfo::empty().Pack16bit(), 1.0f);
        if (HasFinishedSurfaceBounces(p.getVertexIndex() + 1, p.getCounter(PC_DiffuseBounces))) p.setFlag(PF_terminateAtNextBounce);
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make((px << 16) | py, 0, sampleIndex);
        SampleSequenceGenerator sg = SampleSequenceGenerator::make(vb);
        float2 r0 = sampleNext2D(sg);
        float2 subPixelOffset = make_float2(cam.Jitter.x + (r0.x - 0.5f) * S.perPixelJitterAAScale, cam.Jitter.y + (r0.y - 0.5f) * S.perPixelJitterAAScale);
        float2 dof = sampleNext2D(sg);
        float2 pp = make_float2(((float)px + 0.5f + -subPixelOffset.x) / (float)cam.ViewportSize.x, ((float)py + 0.5f + subPixelOffset.y) / (float)cam.ViewportSize.y);
        float2 ndc = make_float2(2.f * pp.x + -1.f, -2.f * pp.y + 1.f);
        float3 org = cam.PosW;
        float3 dir = (ndc.x * cam.CameraU + ndc.y * cam.CameraV) + cam.CameraW;
        float2 ap = sample_disk(dof);
        float3 rayTarget = org + dir;
        org = org + cam.ApertureRadius * (ap.x * normalize(cam.CameraU) + ap.y * normalize(cam.CameraV));
        dir = normalize(rayTarget - org);
        float invCos = 1.f / dot(normalize(cam.CameraW), dir);
        float tMin = cam.NearZ * invCos;
        p.origin = org + dir * tMin; p.dir = dir;
        return p;
    }

    // BridgeDonut:270-278, TextureSampler.hlsli:126-134
    PT_SHADE_NOINLINE float4 sampleTexture(uint textureIndexAndInfo, float lambdaNoDims, float2 uv) const {
        uint textureIndex = textureIndexAndInfo & 0xFFFFu, baseLOD = textureIndexAndInfo >> 24, mipLevels = (textureIndexAndInfo >> 16) & 0xFFu;
        float lambda = 0.5f * (float)baseLOD + lambdaNoDims;
        lambda = fminf_(lambda, fmaxf_((float)mipLevels - 5.0f, 0.0f));
        return sample_trilinear(sc, sc.textures[textureIndex], uv, lambda);
    }
    static void computeTangentSpace(ShadingData& sd, float4 tangentW, bool ignoreTangent) {     // ShadingUtils.hlsli:110-139
        float3 t3 = xyz(tangentW);
        float NdotT = dot(t3, sd.N);
        bool nonParallel = fabsf(NdotT) < 0.9999f;
        bool nonZero = dot(t3, t3) > 0.f;
        bool valid = tangentW.w != 0.f && nonZero && nonParallel;
        if (!ignoreTangent && valid) { sd.T = normalize(t3 - sd.N * NdotT); sd.B = cross(sd.N, sd.T) * tangentW.w; }
        else { sd.T = perp_stark(sd.N); sd.B = cross(sd.N, sd.T); }
    }
    static void adjustShadingNormal(ShadingData& sd, float4 tangentW, bool recompute, bool ignoreTangent) {   // ShadingUtils.hlsli:146-165
        float3 Ng = sd.faceNCorrected;
        float signN = dot(sd.N, Ng) >= 0.f ? 1.f : -1.f;
        float3 Ns = signN * sd.N;
        const float kCosThetaThreshold = 0.1f;
        float cosTheta = dot(sd.V, Ns);
        if (cosTheta <= kCosThetaThreshold) {
            float t = saturate(cosTheta * (1.f / kCosThetaThreshold));
            sd.N = signN * normalize(lerp3(Ng, Ns, t));
        }
        if (cosTheta <= kCosThetaThreshold || recompute) computeTangentSpace(sd, tangentW, ignoreTangent);
    }
    // prevPosW of Bridge::loadSurface in the stable-plane build pass (BridgeDonut:187-199, 619, 631): the hit point in the previous frame's pose — the previous
    // positions of the triangle's vertices (Donut keeps them for skinned meshes; for the others they equal the current ones, which is what interpolating a copy
    // gives) under the previous transform. Only the base vertices of the planes ask for it: the reference's own five-hop gather is good enough here.
    float3 prevPosW(uint prim, float bu, float bv) const {
        const uint2 pinfo = sc.primInfo[prim];
        const uint2 ig = sc.subInstToInstGeom[pinfo.x];
        const GeometryDesc& g = sc.geometries[ig.y];
        const float3x4& M = (sc.prevInstances ? sc.prevInstances : sc.instances)[ig.x].transform;
        const float* P = sc.prevPositions ? sc.prevPositions : sc.positions;
        const float3 bary = make_float3(1.0f - (bu + bv), bu, bv);
        const uint* idx = sc.indices + g.indexOffset + pinfo.y * 3;
        const uint vi[3] = {g.vertexOffset + idx[0], g.vertexOffset + idx[1], g.vertexOffset + idx[2]};
        float3 vp[3];
        for (int k = 0; k < 3; k++) vp[k] = make_float3(P[3 * vi[k]], P[3 * vi[k] + 1], P[3 * vi[k] + 2]);
        const float3 objPos = (vp[0] * bary.x + vp[1] * bary.y) + vp[2] * bary.z;
        return xform_point(M, objPos);
    }
    // Bridge::loadSurface (BridgeDonut:612-853): the divergent gather of the pipeline. LEAN (round 6): the vertex of a path that terminates right after its
    // emission term (PF_terminateAtNextBounce: a fifth of a bounce's hits, shaded as a class of their own, k_classify) reads of the surface only what HandleHit
    // touches before it returns — position, flat normal and facing, material header, emission (with its texture), the two light links, the interior IoR — so
    // the vertex normals and tangents, the base / normal / metal-rough / transmission fetches, the normal map, the tangent frame and the BSDF inputs are not
    // formed at all. What IS formed is formed by the same expressions: the values HandleHit reads are the same floats.
    template <bool LEAN = false>
    SurfaceData loadSurface(uint prim, float bu, float bv, float3 rayDir, RayCone rayCone, MVBlockInputs* mvBlock = nullptr) const {
#if PT_SHADE_TRI
        // one 128-byte line per primitive (pt_scene.h ShadeTri) instead of primInfo -> subInstToInstGeom -> {instance, subInstance, geometry} -> indices ->
        // vertex streams
        const uint4* rec = reinterpret_cast<const uint4*>(sc.shadeTris + prim);
        // eight independent 16-byte loads of one line
        const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5], r6 = rec[6], r7 = rec[7];
        const uint subInst = r0.y, triangleIndex = r0.z, materialIndex = r0.w & 0xFFFFu;
        struct { uint flags; } g; g.flags = r0.w >> 16;
        const float3x4& M = sc.instances[r0.x].transform;
        float3 bary = make_float3(1.0f - (bu + bv), bu, bv);
        float3 vp[3] = {make_float3(asfloat(r1.x), asfloat(r1.y), asfloat(r1.z)), make_float3(asfloat(r1.w), asfloat(r2.x), asfloat(r2.y)), make_float3(asfloat(r2.z), asfloat(r2.w), asfloat(r3.x))};
        float2 vt[3] = {make_float2(0, 0), make_float2(0, 0), make_float2(0, 0)};
        float3 objPos = (vp[0] * bary.x + vp[1] * bary.y) + vp[2] * bary.z;
        float2 texcoord = make_float2(0, 0);
        if (g.flags & GEOM_HAS_UV) {
            vt[0] = make_float2(asfloat(r3.y), asfloat(r3.z)); vt[1] = make_float2(asfloat(r3.w), asfloat(r4.x)); vt[2] = make_float2(asfloat(r4.y), asfloat(r4.z));
            texcoord = (vt[0] * bary.x + vt[1] * bary.y) + vt[2] * bary.z;
        }
        float3 objFlatN = SafeNormalize(cross(vp[1] - vp[0], vp[2] - vp[0]));
        if (mvBlock) mvBlock->curvatureWS = 0.0f;
        float3 geometryNormal = make_float3(0.f);
        if (!LEAN && (g.flags & GEOM_HAS_NORMAL)) {
            // (the record holds them unpacked, normalised and turned towards the flat normal: k_shade_tris, pt_scene.h ShadeTri)
            const float3 n[3] = {make_float3(asfloat(r4.w), asfloat(r5.x), asfloat(r5.y)), make_float3(asfloat(r5.z), asfloat(r5.w), asfloat(r6.x)), make_float3(asfloat(r6.y), asfloat(r6.z), asfloat(r6.w))};
            if (mvBlock) mvBlock->curvatureWS = TriangleCurvatureApprox_GradN(vp, n, M);
            geometryNormal = (n[0] * bary.x + n[1] * bary.y) + n[2] * bary.z;
            geometryNormal = SafeNormalize(xform_direction4(M, geometryNormal));
        }
        float4 tangent = make_float4(0, 0, 0, 0);
        if (!LEAN && (g.flags & GEOM_HAS_TANGENT)) {
            const uint pg[3] = {r7.x, r7.y, r7.z};
            float4 tg[3];
#else      // the gather as the reference's bridge walks it (developer A/B)
        uint2 pinfo = sc.primInfo[prim];
        uint subInst = pinfo.x, triangleIndex = pinfo.y;
        uint2 ig = sc.subInstToInstGeom[subInst];
        const InstanceDesc& inst = sc.instances[ig.x];
        const uint materialIndex = sc.subInstances[subInst].GlobalGeometryIndex_PTMaterialDataIndex & 0xFFFFu;
        const GeometryDesc& g = sc.geometries[ig.y];
        const float3x4& M = inst.transform;
        float3 bary = make_float3(1.0f - (bu + bv), bu, bv);
        const uint* idx = sc.indices + g.indexOffset + triangleIndex * 3;
        uint vi[3] = {g.vertexOffset + idx[0], g.vertexOffset + idx[1], g.vertexOffset + idx[2]};
        float3 vp[3]; float2 vt[3] = {make_float2(0, 0), make_float2(0, 0), make_float2(0, 0)};
        for (int k = 0; k < 3; k++) vp[k] = make_float3(sc.positions[3 * vi[k]], sc.positions[3 * vi[k] + 1], sc.positions[3 * vi[k] + 2]);
        float3 objPos = (vp[0] * bary.x + vp[1] * bary.y) + vp[2] * bary.z;
        float2 texcoord = make_float2(0, 0);
        if (g.flags & GEOM_HAS_UV) {
            for (int k = 0; k < 3; k++) vt[k] = sc.uvs[vi[k]];
            texcoord = (vt[0] * bary.x + vt[1] * bary.y) + vt[2] * bary.z;
        }
        float3 objFlatN = SafeNormalize(cross(vp[1] - vp[0], vp[2] - vp[0]));
        if (mvBlock) mvBlock->curvatureWS = 0.0f;
        float3 geometryNormal = make_float3(0.f);
        if (!LEAN && (g.flags & GEOM_HAS_NORMAL)) {
            float3 n[3];
            for (int k = 0; k < 3; k++) {
                n[k] = normalize(Unpack_RGB8_SNORM(sc.normals[vi[k]]));
                if (dot(n[k], objFlatN) < 0.f) n[k] = -n[k];
            }
            if (mvBlock) mvBlock->curvatureWS = TriangleCurvatureApprox_GradN(vp, n, M);
            geometryNormal = (n[0] * bary.x + n[1] * bary.y) + n[2] * bary.z;
            geometryNormal = SafeNormalize(xform_direction4(M, geometryNormal));
        }
        float4 tangent = make_float4(0, 0, 0, 0);
        if (!LEAN && (g.flags & GEOM_HAS_TANGENT)) {
            float4 tg[3];
            const uint pg[3] = {sc.tangents[vi[0]], sc.tangents[vi[1]], sc.tangents[vi[2]]};
#endif
            for (int k = 0; k < 3; k++) tg[k] = Unpack_RGBA8_SNORM(pg[k]);
            float3 t3 = (xyz(tg[0]) * bary.x + xyz(tg[1]) * bary.y) + xyz(tg[2]) * bary.z;
            t3 = SafeNormalize(xform_direction4(M, t3));
            tangent = make_float4(t3, tg[0].w);
        }
        float3 flatNormal = SafeNormalize(xform_direction4(M, objFlatN));
        bool frontFacing = dot(-rayDir, flatNormal) >= 0.0f;
        if (LEAN || !(g.flags & GEOM_HAS_NORMAL)) geometryNormal = flatNormal;      // (LEAN: never read)
        float3 posW = xform_point(M, objPos);
        float coneTexLODValue = computeRayConeTriangleLODValue(vp, vt, M);
        float lambda = rayCone.computeLOD(coneTexLODValue, rayDir, flatNormal, true) + S.texLODBias;

        ShadingData sd; __builtin_memset(&sd, 0, sizeof(sd));
        sd.posW = posW; sd.V = -rayDir; sd.N = geometryNormal;
        const PTMaterialData& material = sc.materials[materialIndex];
        const uint mflags = material.Flags;
        float4 texBase = make_float4(1, 1, 1, 1), texEmissive = make_float4(1, 1, 1, 1), texNormal = make_float4(0.5f, 0.5f, 1.0f, 0.f),
               texMR = make_float4(1, 1, 1, 1), texTrans = make_float4(1, 1, 1, 1);
        bool hasUV = (g.flags & GEOM_HAS_UV) != 0;
        if (!LEAN && hasUV && (mflags & PTMaterialFlags_UseBaseOrDiffuseTexture)) texBase = sampleTexture(material.BaseOrDiffuseTextureIndex, lambda, texcoord);
        if (!LEAN && hasUV && (mflags & PTMaterialFlags_UseNormalTexture)) texNormal = sampleTexture(material.NormalTextureIndex, lambda, texcoord);
        if (!LEAN && hasUV && (mflags & PTMaterialFlags_UseMetalRoughOrSpecularTexture)) texMR = sampleTexture(material.MetalRoughOrSpecularTextureIndex, lambda, texcoord);
        if (hasUV && (mflags & PTMaterialFlags_UseEmissiveTexture)) texEmissive = sampleTexture(material.EmissiveTextureIndex, lambda, texcoord);
        if (!LEAN && hasUV && (mflags & PTMaterialFlags_UseTransmissionTexture)) texTrans = sampleTexture(material.TransmissionTextureIndex, lambda, texcoord);

        float3 emissiveColor = LP::r3(material.EmissiveColor);
        if (mflags & PTMaterialFlags_UseEmissiveTexture) emissiveColor = LP::mul3(emissiveColor, LP::r3(xyz(texEmissive)));
        float matIoR = LP::r(material.IoR);
        sd.faceNCorrected = frontFacing ? flatNormal : -flatNormal;
        sd.frontFacing = frontFacing;
        bool thin = (mflags & PTMaterialFlags_ThinSurface) != 0;
        sd.materialID = materialIndex;
        sd.mtl = MaterialHeader::make();
        { uint pr = 1 + (mflags >> PTMaterialFlags_NestedPriorityShift); sd.mtl.setNestedPriority(pr < InteriorList::kMaxNestedPriority ? pr : InteriorList::kMaxNestedPriority); }
        sd.mtl.setThinSurface(thin);
        sd.mtl.setActiveLobes(Lobe_All);
        sd.IoR = 1.f;
        StandardBSDFData bd; __builtin_memset(&bd, 0, sizeof(bd));
        if (!LEAN) {
        float3 mGeometryNormal = normalize(geometryNormal), mShadingNormal = mGeometryNormal;
        // MaterialProperties holds lp values: a conversion lpfloat(x) per assignment, half operations between lp operands
        float3 baseColor = LP::r3(material.BaseOrDiffuseColor * xyz(texBase));
        float roughness = LP::r(material.Roughness * texMR.y);
        float metalness = LP::r((mflags & PTMaterialFlags_MetalnessInRedChannel) ? material.Metalness * texMR.x : material.Metalness * texMR.z);
        // EvaluateSceneMaterialRTXPT, BridgeDonut:318-333: float colours in, lp base colour / metalness out
        if (mflags & PTMaterialFlags_UseSpecularGlossModel) {
            float3 bc; float mt;
            ConvertSpecularGlossToMetalRough(material.BaseOrDiffuseColor * xyz(texBase), material.SpecularColor * xyz(texMR), bc, mt);
            baseColor = LP::r3(bc); metalness = LP::r(mt);
            roughness = LP::r(1.0f - texMR.w * (1.0f - material.Roughness));
        }
        float transmission = LP::r(material.TransmissionFactor), diffuseTransmission = LP::r(material.DiffuseTransmissionFactor);
        if (mflags & PTMaterialFlags_UseTransmissionTexture) { transmission = LP::mul(transmission, LP::r(texTrans.x)); diffuseTransmission = LP::mul(diffuseTransmission, LP::r(texTrans.x)); }
        if (hasUV && (mflags & PTMaterialFlags_UseNormalTexture)) {                                   // ApplyNormalMapRTXPT
            float sqT = dot(xyz(tangent), xyz(tangent));
            if (sqT != 0 && tangent.w != 0) {
                float nx = (texNormal.x * 2.0f - 1.0f) * material.NormalTextureScale, ny = (texNormal.y * 2.0f - 1.0f) * material.NormalTextureScale, nz;
                if (texNormal.z <= 0) nz = sqrtf_(saturate(1.0f - nx * nx - ny * ny)); else nz = fabsf(texNormal.z * 2.0f - 1.0f);
                float sqN = (nx * nx + ny * ny) + nz * nz;
                if (sqN != 0) {
                    float nl = sqrtf_(sqN);
                    float3 localNormal = make_float3(nx / nl, ny / nl, nz / nl);
                    float3 tn = xyz(tangent) * (1.0f / sqrtf_(sqT));
                    float3 bitangent = cross(mGeometryNormal, tn) * tangent.w;
                    mShadingNormal = normalize((tn * localNormal.x + bitangent * localNormal.y) + mGeometryNormal * localNormal.z);
                }
            }
        }
        bool ignoreTangent = (mflags & PTMaterialFlags_IgnoreMeshTangentSpace) != 0;
        computeTangentSpace(sd, tangent, ignoreTangent);
        sd.vertexN = frontFacing ? geometryNormal : -geometryNormal;
        sd.N = frontFacing ? mShadingNormal : -mShadingNormal;
        if (mvBlock) mvBlock->projectionTerm = fabsf(dot(rayDir, -sd.N));
        adjustShadingNormal(sd, tangent, true, ignoreTangent);
        sd.shadowNoLFadeout = LP::r(material.ShadowNoLFadeout);
        // lp * (1 - lp)
        float bsdfSpecTrans = LP::mul(transmission, LP::sub(1, metalness)), bsdfDiffTrans = LP::mul(diffuseTransmission, LP::sub(1, metalness));
        float f = (matIoR - 1.f) / (matIoR + 1.f);
        float F0 = f * f;
        // StandardBSDFData holds lp values (BxDF.hlsli:625-634); FalcorBSDF computes from them in float
        bd.diffuse = LP::lerp3(baseColor, make_float3(0.f), metalness);
        bd.specular = LP::lerp3(make_float3(LP::r(F0)), baseColor, metalness);
        bd.roughness = roughness; bd.metallic = metalness;
        bd.transmission = baseColor; bd.diffuseTransmission = bsdfDiffTrans; bd.specularTransmission = bsdfSpecTrans;
        bd.eta = LP::div(sd.IoR, matIoR);
        if (!sd.mtl.isThinSurface() && !sd.frontFacing) bd.eta = LP::div(matIoR, sd.IoR);
        }
        SurfaceData ret;
        ret.neeTriangleLightIndex = RTXPT_INVALID_LIGHT_INDEX; ret.neeAnalyticLightIndex = RTXPT_INVALID_LIGHT_INDEX;
        // BridgeDonut:828-829
        if (mflags & PTMaterialFlags_EnableAsAnalyticLightProxy) ret.neeAnalyticLightIndex = sc.subInstances[subInst].AnalyticProxyLightIndex;
        if (sd.frontFacing && any_gt0(emissiveColor)) {
            sd.emission = emissiveColor;
            // (the light links are re-baked with the lights: read from the sub-instance, by emissive hits only)
            uint baseIndex = sc.subInstances[subInst].EmissiveLightMappingOffset;
            if (baseIndex != 0xFFFFFFFFu) ret.neeTriangleLightIndex = baseIndex + triangleIndex;
        }
        ret.shadingData = sd; ret.bsdf.data = bd; ret.bsdf.diffuseModel = (int)S.diffuseBrdf; ret.interiorIoR = matIoR;
        return ret;
    }
    float loadIoR(uint materialID) const { return (materialID >= sc.materialCount) ? 1.0f : LP::r(sc.materials[materialID].IoR); }      // (returns lpfloat)
    float3 volumeTransmittance(uint materialID, float t) const {                                        // BridgeDonut:871-887
        if (materialID >= sc.materialCount) return make_float3(1.f);
        const PTMaterialData& m = sc.materials[materialID];
        float3 c = clamp3(m.AttenuationColor, 1e-7f, 1.f);
        float d = fmaxf_(1e-30f, m.AttenuationDistance);
        float3 sigmaA = make_float3(-dm_log(c.x) / d, -dm_log(c.y) / d, -dm_log(c.z) / d);
        return make_float3(dm_exp(-t * sigmaA.x), dm_exp(-t * sigmaA.y), dm_exp(-t * sigmaA.z));
    }
    void UpdatePathTravelled(PathState& path, float rayT) const {                                      // PathTracer.hlsli:382-404
        path.incrementVertexIndex();
        path.rayCone = path.rayCone.propagateDistance(rayT);
        path.sceneLength = fminf_(path.sceneLength + rayT, kMaxRayTravel);
    }
    static void AccumulatePathRadiance(PathState& path, float3 radiance) { float4 L = path.GetL(); path.SetL(make_float4(L.x + radiance.x, L.y + radiance.y, L.z + radiance.z, L.w + 0.f)); }

    // PathTracer.hlsli:407-503 NEEAT == false (NEEType 0 / 1, the kernels every frame without a local table runs): "screen-space coherent" is a compile-time
    // false, the local sampler folds away
    template <bool NEEAT>
    __attribute__((always_inline)) void HandleMiss(PathState& path, float3 rayDir, float rayT) const {
        UpdatePathTravelled(path, rayT);
        float3 environmentEmission = make_float3(0.f);
        NEEBSDFMISInfo misInfo = NEEBSDFMISInfo::Unpack16bit(path.GetPackedMISInfo());
        LightSampler lightSampler = LightSampler::make(sc.lights, path.id >> 16, path.id & 0xFFFFu, NEEAT && misInfo.LightSamplingIsSSC);
        if (sc.envEnabled) {
            float mipLevel = (path.getCounter(PC_DiffuseBounces) > 1) ? S.envMapDiffuseSampleMIPLevel : 0.f;
            float3 localDir = mul_vec_mat3(rayDir, sc.envToLocal);
            float3 Le = env_eval_local(sc, localDir, mipLevel);
            float misWeight = 1.0f;
            float bsdfScatterPdf = path.GetBsdfScatterPdf();
            if (misInfo.LightSamplingEnabled && bsdfScatterPdf != 0) {
                uint envIdx = lightSampler.LookupEnvLightByDirection(localDir);
                misWeight = lightSampler.ComputeBSDFMISForEnvironmentQuad(envIdx, bsdfScatterPdf, misInfo.CandidateSamples, misInfo.FullSamples);
            }
            environmentEmission = LP::r3(misWeight * Le);
        }
        const float baseFFThreshold = LP::r(S.fireflyFilterThreshold);
        if (baseFFThreshold != 0) environmentEmission = FireflyFilter<LP>(environmentEmission, baseFFThreshold, path.GetFireflyFilterK());
        if (NEEAT && sc.lights.DepthExport) ExportDepth(path, path.origin + rayDir * rayT);      // ExportNonSurface(path, rayOrigin + rayDir * rayTCurrent, 0)
        if (any_gt0(environmentEmission)) AccumulatePathRadiance(path, path.GetThp() * environmentEmission);
        path.setFlag(PF_hit, false);
        path.terminate();
    }
    float ComputeOutsideIoR(const InteriorList& il, uint materialID, bool entering) const {             // PathTracerNestedDielectrics.hlsli:24-44
        uint outside = il.getTopMaterialID();
        if (!entering) { if (outside == materialID) outside = il.getNextMaterialID(); }
        if (outside == InteriorList::kNoMaterial) return 1.f;
        return loadIoR(outside);
    }
    bool HandleNestedDielectrics(SurfaceData& sfd, PathState& path) const {                             // :49-113
        if (S.nestedDielectricsQuality == 0) return true;
        const uint kMaxRejected = (S.nestedDielectricsQuality == 1) ? 4u : 16u;
        const bool avoidTermination = (S.nestedDielectricsQuality == 1);
        if (sfd.shadingData.mtl.isThinSurface()) return true;
        uint nestedPriority = sfd.shadingData.mtl.getNestedPriority();
        if ((!avoidTermination || path.getCounter(PC_RejectedHits) < kMaxRejected) && !path.interiorList.isTrueIntersection(nestedPriority)) {
            if (avoidTermination || path.getCounter(PC_RejectedHits) < kMaxRejected) {
                path.incrementCounter(PC_RejectedHits);
                path.interiorList.handleIntersection(sfd.shadingData.materialID, nestedPriority, sfd.shadingData.frontFacing);
                path.origin = ComputeRayOrigin(sfd.shadingData.posW, -sfd.shadingData.faceNCorrected);
                path.decrementVertexIndex();
            } else path.terminate();
            return false;
        }
        float outsideIoR = ComputeOutsideIoR(path.interiorList, sfd.shadingData.materialID, sfd.shadingData.frontFacing);
        sfd.shadingData.IoR = outsideIoR;
        sfd.bsdf.data.eta = sfd.shadingData.frontFacing ? LP::div(sfd.shadingData.IoR, sfd.interiorIoR) : LP::div(sfd.interiorIoR, sfd.shadingData.IoR);
        return true;
    }
    // PathTracer.hlsli:217-380
    bool GenerateScatterRay(const ShadingData& sd, const StandardBSDF& bsdf, PathState& path, const SampleGeneratorVertexBase& sgBase) const {
        float4 u;
        if (S.enableLDSamplerForBSDF && path.getCounter(PC_DiffuseBounces) < kDisableLowDiscrepancySamplingAfterDiffuseBounceCount)
            u = SampleSequenceGenerator::Generate(3, sgBase, SGES_ScatterBSDF);
        else
            u = UniformSampleSequenceGenerator::Generate(3, sgBase, SGES_ScatterBSDF);
        BSDFSample bs;
        bool valid = bsdf.sample(sd, u, bs);
        if (!valid) return false;
        path.dir = bs.wo;
        path.SetThp(path.GetThp() * bs.weight);
        path.clearScatterEventFlags();
        path.origin = sd.computeNewRayOrigin(bs.isLobe(Lobe_Reflection));
        float roughness = bsdf.data.roughness;
        bool isDiffuse = bs.isLobe(Lobe_DiffuseReflection) || bs.isLobe(Lobe_DiffuseTransmission) || roughness > kSpecularRoughnessThreshold;
        if (isDiffuse) {
            if (!(bs.isLobe(Lobe_DiffuseTransmission) && ((path.getVertexIndex() % 2) == 1))) path.incrementCounter(PC_DiffuseBounces);
        } else path.setFlag(PF_specular);
        if (bs.isLobe(Lobe_Transmission)) {
            path.setFlag(PF_transmission);
            if (S.nestedDielectricsQuality > 0 && !sd.mtl.isThinSurface()) {
                path.interiorList.handleIntersection(sd.materialID, sd.mtl.getNestedPriority(), sd.frontFacing);
                path.setFlag(PF_insideDielectricVolume, !path.interiorList.isEmpty());
            }
        }
        if (bs.isLobe(Lobe_Delta)) path.setFlag(PF_delta);
        else {
            path.setFlag(PF_deltaOnlyPath, false);
            path.rayCone = RayCone::make(path.rayCone.getWidth(), fminf_(path.rayCone.getSpreadAngle() + ComputeRayConeSpreadAngleExpansionByScatterPDF(bs.pdf, 0.3f), 2.0f * K_PI));
        }
        float fireflyFilterK = ComputeNewScatterFireflyFilterK<LP>(path.GetFireflyFilterK(), bs.pdf, bs.lobeP);
        path.SetFireflyFilterK_BsdfScatterPdf(fireflyFilterK, bs.pdf);
        path.setFlag(PF_enableThreadReorder, true);
        return true;
    }
    // PathTracerNEE.hlsli:88-161
    LightSample GenerateLightSample(const LightSampler& lightSampler, const ShadingData& sd, const StandardBSDF& bsdf, uint candidateSampleCount, UniformSampleSequenceGenerator& sg) const {
        LightSample cand; __builtin_memset(&cand, 0, sizeof(cand));
        float weightSum = 0, candWeight = 0;
        uint localCount, globalCount;
        lightSampler.GetCandidateSampleCounts(candidateSampleCount, localCount, globalCount);
        for (uint i = 0; i < candidateSampleCount; i++) {
            const bool sampleIsLocal = i >= globalCount;
            float selectionPdf = 0;
            float rnd = sampleNext1D(sg);
            uint lightIndex = sampleIsLocal ? lightSampler.SampleLocal(rnd, selectionPdf) : lightSampler.SampleGlobal(rnd, selectionPdf);
            PolymorphicLightInfoFull li = lightSampler.LoadLight(lightIndex);
            float2 interior = sampleNext2D(sg);
            PolymorphicLightSample ls = PolymorphicLight_CalcSample(li, interior, sd.posW, sc.envToWorld);
            LightSample c;
            float pdf = ls.SolidAnglePdf * selectionPdf;
            c.Li = pdf > 0.f ? (ls.Radiance / pdf) : make_float3(0.f);
            c.SolidAnglePdf = ls.SolidAnglePdf;
            float3 surfToLight = ls.Position - sd.posW;
            c.Distance = length(surfToLight);
            c.Direction = surfToLight / fmaxf_(c.Distance, 1e-7f);
            c.LightIndex = lightIndex; c.SelectionPdf = selectionPdf; c.LightSampleableByBSDF = ls.LightSampleableByBSDF; c.FromLocalDistribution = sampleIsLocal;
            float wrsWeight = max3(c.Li) * bsdf.evalPdf(sd, c.Direction);
            float r = sampleNext1D(sg);
            weightSum += wrsWeight;
            float thr = saturate(wrsWeight / weightSum);
            if (r < thr) { cand = c; candWeight = wrsWeight; }
        }
        cand.Li = cand.Li * (1.0f / (candWeight / weightSum));
        return cand;
    }
    // HandleNEE (PathTracerNEE.hlsli:303-346) / HandleNEE_MultipleSamples (:277-301) with ProcessLightSample (:185-275) split at the visibility ray.
    // MULTI == false is NEEFullSamples == 1: the one request goes back to k_shade through `req`. MULTI: the requests are written to the sink's group.
    template <bool MULTI, bool NEEAT>
    uint HandleNEE(const PathState& pre, const ShadingData& sd, const StandardBSDF& bsdf, UniformSampleSequenceGenerator& sg, ShadowRequest& req, const ShadowSink* sink) const {
        req.valid = false; req.fbLight = RTXPT_INVALID_LIGHT_INDEX; req.rrFix = 0u;
        const LightSampler lightSampler = LightSampler::make(sc.lights, pre.id >> 16, pre.id & 0xFFFFu, NEEAT && LightSampler::IsScreenSpaceCoherentHeuristic(sc.lights, pre.rayCone.getWidth(), pre.sceneLength));
        const uint fullSamples = MULTI ? (S.NEEFullSamples < 63u ? S.NEEFullSamples : 63u) : 1u;      // min(RTXPT_LIGHTING_MAX_SAMPLE_COUNT, NEEFullSamples)
        bool hasNonDeltaLobes = (bsdf.getLobes() & Lobe_NonDelta) != 0;
        bool applyNEE = hasNonDeltaLobes && !lightSampler.IsEmpty() && fullSamples > 0;
        if (!applyNEE) return NEEBSDFMISInfo::empty().Pack16bit();
        uint candidateSampleCount = S.NEECandidateSamples;
        NEEBSDFMISInfo info; info.LightSamplingEnabled = true; info.LightSamplingIsSSC = lightSampler.IsScreenSpaceCoherent; info.CandidateSamples = candidateSampleCount; info.FullSamples = fullSamples;
        uint base = 0, numValid = 0;
        if (MULTI) base = __hip_atomic_fetch_add(sink->count, fullSamples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint s = 0; s < fullSamples; s++) {
            LightSample ls = GenerateLightSample(lightSampler, sd, bsdf, candidateSampleCount, sg);
            bool valid = ls.Valid();
            if (valid) {
                float faceSide = dot(sd.N, ls.Direction) >= 0 ? 1.f : -1.f;
                float3 o = ComputeRayOrigin(sd.posW, sd.faceNCorrected * faceSide);
                float fadeOut = (sd.shadowNoLFadeout > 0) ? ComputeLowGrazingAngleFalloff(ls.Direction, sd.vertexN, sd.shadowNoLFadeout, 2.0f * sd.shadowNoLFadeout) : 1.0f;
                uint localCount, globalCount;
                lightSampler.GetCandidateSampleCounts(candidateSampleCount, localCount, globalCount);
                float thisPdf, otherPdf, thisCount, otherCount;
                lightSampler.ComputeLightSelectionPdfs(ls.SelectionPdf, ls.LightIndex, ls.FromLocalDistribution, localCount, globalCount, thisPdf, otherPdf, thisCount, otherCount);
                float wrsMIS = EvalMIS_Balance(1, thisPdf, 1, otherPdf);
                wrsMIS = wrsMIS / thisCount;
                float scatterPdfForDir = bsdf.evalPdf(sd, ls.Direction);
                float lightAvgPdf = (thisPdf + otherPdf) * (float)fullSamples;
                float pathMIS = EvalMIS_Balance(1, lightAvgPdf * ls.SolidAnglePdf, 1, ls.LightSampleableByBSDF ? scatterPdfForDir : 0.f);
                float3 Li = ls.Li * (fadeOut * wrsMIS * pathMIS / (float)fullSamples);
                float4 bsdfThp = bsdf.eval(sd, ls.Direction);
                float3 radiance = xyz(bsdfThp) * Li;
                float radianceAvg = Average(radiance);
                if (S.fireflyFilterThreshold != 0) {
                    float pdf = ls.SelectionPdf * ls.SolidAnglePdf;
                    float k = ComputeNewScatterFireflyFilterK<LP>(pre.GetFireflyFilterK(), pdf, 1.0f);
                    radiance = radiance * FireflyFilterShort(radianceAvg, S.fireflyFilterThreshold, k);
                }
                radiance = radiance * pre.GetThp();
                // the reservoir update of the visible case (:246-273; radianceAvg is the value before the firefly filter)
                if (NEEAT && !MULTI && lightSampler.IsTemporalFeedbackRequired()) {
                    req.fbLight = ls.LightIndex | (lightSampler.IsScreenSpaceCoherent ? LFR_SCREEN_SPACE_COHERENT_FLAG : 0u);
                    req.fbWeight = lightSampler.FeedbackWeightFromNEE(ls.LightIndex, radianceAvg * Average(pre.GetThp()));
                    UniformSampleSequenceGenerator after = sg; req.fbRandom = sampleNext1D(after);
                }
                if (MULTI) {
                    sink->q0[base + s] = make_float4(o.x, o.y, o.z, ls.Distance * 0.9985f);
                    sink->q1[base + s] = make_float4(ls.Direction.x, ls.Direction.y, ls.Direction.z, asfloat(sink->pathIndex));
                    sink->q2[base + s] = make_float4(radiance.x, radiance.y, radiance.z, 0.f);
                    numValid++;
                } else { req.valid = true; req.origin = o; req.dir = ls.Direction; req.tmax = ls.Distance * 0.9985f; req.radiance = radiance; }
            } else if (MULTI) {                 // no light sample: an entry that no traversal step accepts and that contributes nothing
                sink->q0[base + s] = make_float4(0.f, 0.f, 0.f, -1.0f);
                sink->q1[base + s] = make_float4(0.57735026f, 0.57735026f, 0.57735026f, asfloat(sink->pathIndex));
                sink->q2[base + s] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (MULTI && numValid) (void)__hip_atomic_fetch_add(sink->valid, (unsigned long long)numValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return info.Pack16bit();
    }
    bool HandleRussianRoulette(PathState& path, UniformSampleSequenceGenerator& sg) const {             // PathTracer.hlsli:182-208
        if (!S.enableRussianRoulette) return false;
        float rrVal = sqrtf_(Luminance(path.GetThp()));
        float prob = saturate(0.85f - rrVal); prob = prob * prob;
        prob = saturate(prob + fmaxf_(0.f, ((float)path.getVertexIndex() / (float)S.bounceCount - 0.4f)));
        if (sampleNext1D(sg) < prob) return true;
        path.SetPackedMISInfo_ThpRuRuCorrection(path.GetPackedMISInfo(), 1.0f / (1.0f - prob));
        return false;
    }
    // Where the path's state lives while HandleHit runs. PathInRegisters: the caller loaded all of it and stores it afterwards (the tail kernel, the probes). A
    // kernel that streams paths through the pool (k_shade, pt_wavefront.hip) passes an IO that (i) loads the words the surface does not need only after
    // loadSurface, (ii) stores the scattered path's first four word groups as soon as GenerateScatterRay has made them — before the light sampling, the
    // vertex's register peak, which then carries 5 words of the path instead of 21 — and
    // (iii) stores the last group ({firefly K | pdf, MIS info | roulette correction, flags | vertex index, sample index}) at the end. Same values either way:
    //       order of loads and stores only.
    struct PathInRegisters { static constexpr bool streams = false; void mark(int) const {} void load_rest(PathState&) const {} void store_front(const PathState&) const {} void store_back(const PathState&) const {} void store_all(const PathState&) const {} };
    // PathTracer.hlsli:505-762 (reference mode), shadow test deferred through `req` (a split of this vertex at NEE — a light-sample kernel and a scatter kernel
    // — was built and measured in round 4: 21.5 -> 27.5 ms, profiles/r04p_shade_split_ab.txt; history: af4c2b2)
    template <bool MULTI, bool NEEAT, class IO = PathInRegisters>
    __attribute__((always_inline)) void HandleHit(PathState& path, const HitInfo& hit, ShadowRequest& req, const ShadowSink* sink, const IO& io = IO()) const {
        req.valid = false; io.mark(0);      // (mark: cycle stamps of the phases in PT_SHADE_PHASE_PROBE builds, nothing otherwise)
        const float3 rayDir = path.dir;
        // UpdatePathTravelled, the two updates the surface needs ...
        path.rayCone = path.rayCone.propagateDistance(hit.t); path.sceneLength = fminf_(path.sceneLength + hit.t, kMaxRayTravel);
        // a vertex that ends right after its emission term needs a fraction of the surface (loadSurface<LEAN>); the flag is in the word group the IO loads
        // first
        SurfaceData sfd = path.isTerminatingAtNextBounce() ? loadSurface<true>(hit.prim, hit.u, hit.v, rayDir, path.rayCone) : loadSurface<false>(hit.prim, hit.u, hit.v, rayDir, path.rayCone);
        io.mark(1);
        io.load_rest(path);
        // ... and the third, once the flags word is there
        path.incrementVertexIndex();
        const float3 rayOrigin = path.origin;
        if (S.nestedDielectricsQuality > 0 && !path.interiorList.isEmpty()) {
            float3 tr = volumeTransmittance(path.interiorList.getTopMaterialID(), hit.t);
            path.SetThp(path.GetThp() * tr);
        }
        if (!HandleNestedDielectrics(sfd, path)) { io.store_all(path); return; }
        const ShadingData& sd = sfd.shadingData; const StandardBSDF& bsdf = sfd.bsdf;
        float3 surfaceEmission = make_float3(0.f);
        NEEBSDFMISInfo misInfo = NEEBSDFMISInfo::Unpack16bit(path.GetPackedMISInfo());
        if (any_gt0(sd.emission)) {
            float misWeight = 1.0f;
            float bsdfScatterPdf = path.GetBsdfScatterPdf();
            if (misInfo.LightSamplingEnabled && bsdfScatterPdf != 0) {
                LightSampler lightSampler = LightSampler::make(sc.lights, path.id >> 16, path.id & 0xFFFFu, NEEAT && misInfo.LightSamplingIsSSC);
                misWeight = lightSampler.ComputeBSDFMISForEmissiveTriangle(sfd.neeTriangleLightIndex, bsdfScatterPdf, rayOrigin, sd.posW, misInfo.CandidateSamples, misInfo.FullSamples);
            }
            surfaceEmission = LP::r3(sd.emission * misWeight);
        }
        // PathTracer.hlsli:636-648: the mesh stands in for an analytic (sphere) light
        if (sfd.neeAnalyticLightIndex != RTXPT_INVALID_LIGHT_INDEX) {
            LightSampler lightSampler = LightSampler::make(sc.lights, path.id >> 16, path.id & 0xFFFFu, NEEAT && misInfo.LightSamplingIsSSC);
            const float bsdfPdf = misInfo.LightSamplingEnabled ? LP::r(path.GetBsdfScatterPdf()) : 0.0f; float3 add;
            if (lightSampler.ComputeAnalyticLightProxyContribution(sfd.neeAnalyticLightIndex, bsdfPdf, rayOrigin, rayDir, misInfo.CandidateSamples, misInfo.FullSamples, add)) {
                add = LP::r3(add); surfaceEmission = make_float3(LP::add(surfaceEmission.x, add.x), LP::add(surfaceEmission.y, add.y), LP::add(surfaceEmission.z, add.z));
            }
        }
        if (any_gt0(surfaceEmission)) {
            const float baseFFThreshold = LP::r(S.fireflyFilterThreshold);
            if (baseFFThreshold != 0) surfaceEmission = FireflyFilter<LP>(surfaceEmission, baseFFThreshold, path.GetFireflyFilterK());
            if (any_gt0(surfaceEmission)) AccumulatePathRadiance(path, path.GetThp() * surfaceEmission);
        }
        // ExportSurface(path, surfaceData, path.GetSceneLength(), 0): the camera ray of this pixel and sample, at the path's length
        if (NEEAT && sc.lights.DepthExport) {
            float3 co, cd; computeCameraRay(path.id >> 16, path.id & 0xFFFFu, path.sampleIndex, co, cd);
            ExportDepth(path, co + cd * path.sceneLength);
        }
        if (path.isTerminatingAtNextBounce()) { path.terminate(); io.store_all(path); return; }
        float rr = path.GetThpRuRuCorrection();
        path.SetThp(path.GetThp() * make_float3(rr));
        io.mark(2);
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make(path.id, path.getVertexIndex(), path.sampleIndex);
        UniformSampleSequenceGenerator uniformSG = UniformSampleSequenceGenerator::make(vb, SGES_Base);
        const PathState preScatterPath = path;
        bool scatterValid = GenerateScatterRay(sd, bsdf, path, vb);
        io.mark(3);
        io.store_front(path);      // origin | id, direction | length, throughput | radiance, interior list | counters | ray cone: final from here on
        const uint misPacked = S.NEEEnabled ? HandleNEE<MULTI, NEEAT>(preScatterPath, sd, bsdf, uniformSG, req, sink) : NEEBSDFMISInfo::empty().Pack16bit();
        io.mark(4);
        path.SetPackedMISInfo_ThpRuRuCorrection(misPacked, path.GetThpRuRuCorrection());
        if (!scatterValid) path.terminate();
        bool shouldTerminate = HasFinishedSurfaceBounces(path.getVertexIndex() + 1, path.getCounter(PC_DiffuseBounces));
        // feedback pending on the visibility test: the visible case has drawn one more number before the roulette
        if (NEEAT && req.fbLight != RTXPT_INVALID_LIGHT_INDEX) {
            UniformSampleSequenceGenerator sgVisible = uniformSG; (void)sampleNext1D(sgVisible);
            PathState visiblePath = path;
            const bool terminateVisible = shouldTerminate | HandleRussianRoulette(visiblePath, sgVisible);
            shouldTerminate |= HandleRussianRoulette(path, uniformSG);
            req.rrFix = (terminateVisible != shouldTerminate ? 1u : 0u) | (terminateVisible ? 2u : 0u) | ((visiblePath.pack1 & 0xFFFFu) << 16);
        } else shouldTerminate |= HandleRussianRoulette(path, uniformSG);
        if (shouldTerminate) path.setFlag(PF_terminateAtNextBounce);
        io.store_back(path);
        io.mark(5);
    }
    // the deferred half: NEEResult::AccumulateRadiance (fp16, PathTracerTypes.hlsli:170-207) then AccumulatePathRadiance (PathTracer.hlsli:722-746)
    // NEEResult::AccumulateRadiance (the spec-average lane is not used in reference mode)
    static void NeeAccumulate(uint nee[2], float3 radiance) {
        float2 a = Fp16ToFp32(nee[0]), b = Fp16ToFp32(nee[1]);
        nee[0] = Fp32ToFp16(make_float2(a.x + radiance.x, a.y + radiance.y)); nee[1] = Fp32ToFp16(make_float2(b.x + radiance.z, b.y + 0.f));
    }
    static void NeeCommit(uint pack45[2], const uint nee[2]) {                      // HandleHit: `if any(neeResult > 0) AccumulatePathRadiance`
        float2 a = Fp16ToFp32(nee[0]), b = Fp16ToFp32(nee[1]);
        if (!(a.x > 0 || a.y > 0 || b.x > 0)) return;
        float2 l0 = Fp16ToFp32(pack45[0]), l1 = Fp16ToFp32(pack45[1]);
        pack45[0] = Fp32ToFp16NoClamp(make_float2(clampf(l0.x + a.x, 0, HLF_MAX), clampf(l0.y + a.y, 0, HLF_MAX)));
        pack45[1] = Fp32ToFp16NoClamp(make_float2(clampf(l1.x + b.x, 0, HLF_MAX), clampf(l1.y + 0.f, 0, HLF_MAX)));
    }
    static void ResolveShadow(uint pack45[2], float3 radiance) { uint nee[2] = {0u, 0u}; NeeAccumulate(nee, radiance); NeeCommit(pack45, nee); }
};
typedef PathKernelContextT<false> PathKernelContext;      // the fp32 build of the lp types; PathKernelContextT<true> has the same data members

#pragma clang force_cuda_host_device end
} // namespace ptk
