// ORACLE (test infrastructure only) — the reference-mode path tracing mega-pass, restated per pixel
// (SURVEY.md §8a rows a1-a4, a8-a10, a12, a14-a17, a21, a22). Follows, in order:
//   Rtxpt/Shaders/PathTracerSample.hlsl:200-250                      RAYGEN_ENTRY loop
//   Rtxpt/Shaders/PathTracer/PathState.hlsli:83-267                   PathState (80 B, fp16-packed thp/L/pdf/RR)
//   Rtxpt/Shaders/PathTracer/PathTracer.hlsli:40-45,47-91,139-175,182-208,217-380,382-404,407-503,505-762
//   Rtxpt/Shaders/PathTracer/PathTracerNEE.hlsli:41-50,52-86,88-161,166-182,185-275,277-346
//   Rtxpt/Shaders/PathTracer/PathTracerNestedDielectrics.hlsli:24-128, Rendering/Materials/InteriorList.hlsli:28-248
//   Rtxpt/Shaders/PathTracer/PathTracerHelpers.hlsli:126-153 (thin lens), :164-219 (ray-cone growth, firefly filter)
//   Rtxpt/Shaders/PathTracer/Rendering/Materials/TexLODHelpers.hlsli:57-161 (RayCone, triangle LOD)
//   Rtxpt/Shaders/PathTracerBridgeDonut.hlsli:152-256,280-428,543-564,612-853,871-887 (Bridge::*)
// Pinning: this integrator is compared, frame for frame and bit for bit, with the reference's own integrator text compiled over the same scene
// services (oracle/refpin/hlsl_tu.py --integrator, tests/test_oracle_refpin_integrator.py); leaf functions, BSDF, lights and RNG separately
// (tests/test_oracle_refpin_hlsl.py, tests/test_oracle_kat.py). Unpinned: the Donut side of loadSurface and DXR traversal.
// Fixed parity knobs (SURVEY.md §8a "parity knobs"): PATH_TRACER_MODE_REFERENCE, NEEType=1 (power, no local sampler,
// no temporal feedback), full MIS (RTXPT_USE_APPROXIMATE_MIS=0), no ReSTIR, no stable planes, no STF.
// lp types: both builds of the reference are restated. PT_LP16 = 0 is RTXPT_LP_TYPES_USE_16BIT_PRECISION 0; PT_LP16 = 1 (libptref_lp16.so) is the
// reference's default (SampleUI.h:182, Sample.cpp:1035): every value the reference declares lpfloat / lpfloat3 is rounded to binary16 where it is
// converted, and the operations the reference performs between lp values are half operations (LPOps, vec.h) — MaterialProperties
// (BridgeDonut:311-380), the BSDF inputs of Bridge::loadSurface (:732-790), ShadingData::IoR / emission / shadowNoLFadeout, SurfaceData::interiorIoR,
// updateOutsideIoR / loadIoR (:855-869), the emission terms and FireflyFilter (PathTracer.hlsli:438-480, 593-660, PathTracerHelpers.hlsli:206-213).
#pragma once
#include "bsdf.h"
#include "rng.h"
#include "scene.h"

namespace ptref {

#ifndef PT_LP16
#define PT_LP16 0          // 1: the reference's default build, lp types in 16 bits (libptref_lp16.so)
#endif
typedef LPOps<PT_LP16 != 0> LP;
static const float kMaxRayTravel = 1e15f;                 // Config.h
static const float kSpecularRoughnessThreshold = 0.25f;   // PathTracer.hlsli:23
static const float c_DielectricSpecular = 0.04f;          // Donut material_cb.h (absent; value per glTF spec)

// PathTracerShared.h:24-42 (exact layout, 112 bytes)
struct PathTracerCameraData {
    float3 PosW; float NearZ; float3 DirectionW; float PixelConeSpreadAngle; float3 CameraU; float FarZ;
    float3 CameraV; float FocalDistance; float3 CameraW; float AspectRatio; uint2 ViewportSize; float ApertureRadius; float _padding0;
    float2 Jitter; float _padding1, _padding2;
};
static_assert(sizeof(PathTracerCameraData) == 112, "PathTracerCameraData layout");

// subset of PathTracerConstants (PathTracerShared.h:45-103) + the shader macros of Sample.cpp:988-1042 that matter here
struct PtSettings {
    uint  bounceCount, diffuseBounceCount;
    float perPixelJitterAAScale;          // 1 in reference AA (Sample.cpp:1501)
    float texLODBias;
    float fireflyFilterThreshold;         // 0 = disabled
    float envMapDiffuseSampleMIPLevel;    // EnvironmentMapDiffuseSampleMIPLevel
    uint  NEEEnabled, NEEType, NEECandidateSamples, NEEFullSamples;
    uint  enableRussianRoulette;          // PT_ENABLE_RUSSIAN_ROULETTE
    uint  nestedDielectricsQuality;       // RTXPT_NESTED_DIELECTRICS_QUALITY (0,1,2)
    uint  enableLDSamplerForBSDF;         // RTXPT_ENABLE_LOW_DISCREPANCY_SAMPLER_FOR_BSDF
    uint  diffuseBrdf;                    // DiffuseBrdf: 0 Lambert, 2 Frostbite
    uint  useFp16Types;                   // which build of the lp types the caller asked for; THIS library restates the one PT_LP16 names (libptref.so / libptref_lp16.so)
    uint  _pad;
};
static_assert(sizeof(PtSettings) == 64, "PtSettings layout");

struct RayCounters { uint64_t extendRays, shadowRays, hits, nodeVisitsExt, triTestsExt, nodeVisitsSh, triTestsSh; };

// TexLODHelpers.hlsli:57-123 (USE_RAYCONES_WITH_FP16_IN_RAYPAYLOAD)
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
// TexLODHelpers.hlsli:129-143
// Donut's ConvertSpecularGlossToMetalRough (donut/shaders/scene_material.hlsli, un-vendored), called by EvaluateSceneMaterialRTXPT for
// PTMaterialFlags_UseSpecularGlossModel materials (PathTracerBridgeDonut.hlsli:318-333; ENABLE_METAL_ROUGH_RECONSTRUCTION 1, :15, :772-774): restated from the
// algorithm it follows, the Khronos KHR_materials_pbrSpecularGlossiness "convert-between-workflows" sample (solveMetallic + the two base-colour estimates blended
// by metallic^2, perceived brightness = sqrt(0.299 r^2 + 0.587 g^2 + 0.114 b^2), dielectric specular 0.04). UNPINNED: the function's own text is outside the tree.
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
    float3 edge01 = xform_vector(M, v[1] - v[0]);       // mul(v, transpose(M3x3)) == M3x3 * v
    float3 edge02 = xform_vector(M, v[2] - v[0]);
    float Pa = length(cross(edge01, edge02));
    return 0.5f * RayCone::SafeLog2(Ta / Pa);
}

// TriangleCurvatureApprox_GradN (PathTracerBridgeDonut.hlsli:92-149): a curvature proxy of one triangle in ~1/length units — the RMS gradient of a linear normal field fitted
// over the triangle in a 2D basis of its (world-space) plane. Feeds the automatic motion-vector block types of Bridge::loadSurface (:704-716).
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
// what Bridge::loadSurface hands to its motion-vector block decision (BridgeDonut:704-716): donutGS.curvatureWS (0 for a mesh without vertex normals: the sample is zero-initialised, :164)
// and abs(dot(rayDir, -N)) with the shading normal BEFORE adjustShadingNormal. Asked for by the stable-plane passes only (null otherwise: nothing is computed).
struct MVBlockInputs { float curvatureWS, projectionTerm; };


// InteriorList.hlsli:28-248 (2 slots)
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

// PathState.hlsli:28-81 flag bits (shifted left by 10 in flagsAndVertexIndex)
enum : uint {
    PF_active = 1 << 0, PF_hit = 1 << 1, PF_transmission = 1 << 2, PF_specular = 1 << 3, PF_delta = 1 << 4,
    PF_insideDielectricVolume = 1 << 5, PF_terminateAtNextBounce = 1 << 6, PF_enableThreadReorder = 1 << 9, PF_deltaOnlyPath = 1 << 12,
};
enum { PC_DiffuseBounces = 0, PC_RejectedHits = 1, PC_BouncesFromStablePlane = 2 };
static const uint kVertexIndexBitCount = 10, kVertexIndexBitMask = (1u << 10) - 1u;

// PathState.hlsli:83-267
struct PathState {
    float3 origin; uint id; float3 dir; float sceneLength;
    uint pack23[2];   // thp fp16x4
    uint pack45[2];   // L fp16x4
    InteriorList interiorList; uint packedCounters; uint stableBranchID;
    RayCone rayCone; uint pack0, pack1, flagsAndVertexIndex;

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
    bool wasScatterTransmission() const { return hasFlag(PF_transmission); }
    bool isTerminatingAtNextBounce() const { return hasFlag(PF_terminateAtNextBounce); }
    void clearScatterEventFlags() { flagsAndVertexIndex &= ~((PF_transmission | PF_specular | PF_delta) << kVertexIndexBitCount); }
    uint getCounter(uint type) const { return (packedCounters >> (type << 3)) & 0xff; }
    void incrementCounter(uint type) { packedCounters += (1u << (type << 3)); }
    uint getVertexIndex() const { return flagsAndVertexIndex & kVertexIndexBitMask; }
    void incrementVertexIndex() { flagsAndVertexIndex += 1; }
    void decrementVertexIndex() { flagsAndVertexIndex -= 1; }
    uint2 GetPixelPos() const { uint2 p = {id >> 16, id & 0xFFFFu}; return p; }
};

// PathTracerTypes.hlsli:89-160 (PT_USE_RESTIR_DI == 0)
struct NEEBSDFMISInfo {
    bool LightSamplingEnabled, LightSamplingIsSSC; uint CandidateSamples, FullSamples;
    static NEEBSDFMISInfo empty() { NEEBSDFMISInfo r; r.LightSamplingEnabled = false; r.LightSamplingIsSSC = false; r.CandidateSamples = 0; r.FullSamples = 0; return r; }
    static NEEBSDFMISInfo Unpack16bit(uint p) { NEEBSDFMISInfo r; r.LightSamplingEnabled = (p & (1u << 15)) != 0; r.LightSamplingIsSSC = (p & (1u << 13)) != 0; r.CandidateSamples = (p >> 6) & 0x3F; r.FullSamples = p & 0x3F; return r; }
    uint Pack16bit() const { return ((LightSamplingEnabled ? 1u : 0u) << 15) | ((LightSamplingIsSSC ? 1u : 0u) << 13) | ((CandidateSamples & 0x3F) << 6) | (FullSamples & 0x3F); }
};
// PathTracerTypes.hlsli:164-207 (RTXPT_NEE_RESULT_MANUAL_PACK: fp16 accumulation)
struct NEEResult {
    uint RadianceAndSpecAvgPkg[2]; NEEBSDFMISInfo BSDFMISInfo;
    static NEEResult empty() { NEEResult r; r.RadianceAndSpecAvgPkg[0] = r.RadianceAndSpecAvgPkg[1] = 0; r.BSDFMISInfo = NEEBSDFMISInfo::empty(); return r; }
    float4 Get() const { float2 a = Fp16ToFp32(RadianceAndSpecAvgPkg[0]), b = Fp16ToFp32(RadianceAndSpecAvgPkg[1]); return make_float4(a.x, a.y, b.x, b.y); }
    void AccumulateRadiance(float3 radiance, float specAvg) {
        float4 c = Get();
        RadianceAndSpecAvgPkg[0] = Fp32ToFp16(make_float2(c.x + radiance.x, c.y + radiance.y));
        RadianceAndSpecAvgPkg[1] = Fp32ToFp16(make_float2(c.z + radiance.z, c.w + specAvg));
    }
};
// LightingTypes.hlsli:327-358
struct LightSample {
    float3 Li; float Distance; float3 Direction; uint LightIndex; float SelectionPdf, SolidAnglePdf; bool LightSampleableByBSDF, FromLocalDistribution;
    bool Valid() const { return any_gt0(Li); }
};

struct SurfaceData { ShadingData shadingData; StandardBSDF bsdf; float interiorIoR; uint neeTriangleLightIndex; uint neeAnalyticLightIndex; };

// PathTracerHelpers.hlsli:164-219
static inline float ComputeRayConeSpreadAngleExpansionByScatterPDF(float bsdfScatterPdf, float growthFactor = 0.3f) {
    return growthFactor * 2.0f * FastACos(fmaxf_(-1.0f, 1.0f - (1.0f / bsdfScatterPdf) / (2.0f * K_PI)));
}
static inline float ComputeNewScatterFireflyFilterK(float currentK, float bouncePDF, float lobeP) {
    const float minK = 0.00001f;
    float angle = (bouncePDF == 0) ? 0.f : ComputeRayConeSpreadAngleExpansionByScatterPDF(bouncePDF, 1.0f);
    const float k = 32;
    float p = k / (k + angle * angle);
    p *= FastSqrt(lobeP);
    return LP::r(fmaxf_(minK, currentK * p));               // returns lpfloat: the NEE path uses the value before it is ever packed
}
static inline float3 FireflyFilter(float3 signalIn, float threshold, float fireflyFilterK) {      // lpfloat3 (lpfloat3, lpfloat, lpfloat): all three are lp values
    float t = LP::mul(threshold, fireflyFilterK);
    float maxR = LP::average3(signalIn);
    if (maxR > t) signalIn = LP::mul3(LP::div3(signalIn, maxR), t);
    return signalIn;
}
static inline float FireflyFilterShort(float signalAverage, float threshold, float fireflyFilterK) {
    float t = threshold * fireflyFilterK;
    return (signalAverage > t) ? (1.0f / signalAverage * t) : 1.0f;
}
static inline float ComputeLowGrazingAngleFalloff(float3 lightDirection, float3 n, float falloffFrom, float falloffRange) {
    return saturate((dot(lightDirection, n) - falloffFrom) / falloffRange);
}

struct PathTracer {
    const Scene& sc; PtSettings S; PathTracerCameraData cam; uint sampleIndex;   // Bridge::getSampleIndex() = sampleBaseIndex + subSampleIndex
    RayCounters* counters;
    float* fbTotalWeight = nullptr; uint* fbCandidates = nullptr; uint fbWidth = 0;      // NEE-AT feedback reservoirs of this sample (one slot per pixel), or null

    PathTracer(const Scene& scene, const PtSettings& s, const PathTracerCameraData& c, uint sidx, RayCounters* ctr)
        : sc(scene), S(s), cam(c), sampleIndex(sidx), counters(ctr) {}
    // the reference-mode guide-buffer dump, of which only the depth is kept (NEE-AT's disocclusion test reads it)
    void ExportDepth(const PathState& path, float3 virtualWorldPos) const {
        sc.lightTable.DepthExport[(path.id & 0xFFFFu) * sc.lightTable.DepthWidth + (path.id >> 16)] = LightTable_ClipDepth(sc.lightTable, virtualWorldPos);
    }
    LightSampler CreateLightSampler(uint pathId, bool isScreenSpaceCoherent) const { return LightSampler::make(sc.lightTable, pathId >> 16, pathId & 0xFFFFu, isScreenSpaceCoherent); }      // BridgeDonut:1075-1084

    // PathTracer.hlsli:40-45
    bool HasFinishedSurfaceBounces(uint vertexIndex, uint diffuseBounces) const {
        if (S.bounceCount < vertexIndex) return true;
        return diffuseBounces > S.diffuseBounceCount;
    }
    // PathTracer.hlsli:47-91
    PathState EmptyPathInitialize(uint px, uint py) const {
        PathState p; memset(&p, 0, sizeof(p));
        p.id = (px << 16) | py;
        p.SetThp(make_float3(1.f));
        p.setFlag(PF_active); p.setFlag(PF_deltaOnlyPath, true);
        p.rayCone = RayCone::make(0, cam.PixelConeSpreadAngle);
        p.SetL(make_float4(0, 0, 0, 0));
        p.SetFireflyFilterK_BsdfScatterPdf(1.0f, 0.0f);
        p.SetPackedMISInfo_ThpRuRuCorrection(NEEBSDFMISInfo::empty().Pack16bit(), 1.0f);
        p.stableBranchID = 1;
        if (HasFinishedSurfaceBounces(p.getVertexIndex() + 1, p.getCounter(PC_DiffuseBounces))) p.setFlag(PF_terminateAtNextBounce);
        return p;
    }
    // BridgeDonut:543-564 + PathTracerHelpers.hlsli:126-153
    void computeCameraRay(uint px, uint py, float3& o, float3& d) const {
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make((px << 16) | py, 0, sampleIndex);
        SampleSequenceGenerator sg = SampleSequenceGenerator::make(vb);     // defaults: Base seed, lowDiscrepancy=false
        float2 r0 = sampleNext2D(sg);
        float2 subPixelOffset = make_float2(cam.Jitter.x + (r0.x - 0.5f) * S.perPixelJitterAAScale, cam.Jitter.y + (r0.y - 0.5f) * S.perPixelJitterAAScale);
        float2 dof = sampleNext2D(sg);
        float2 p = make_float2(((float)px + 0.5f + -subPixelOffset.x) / (float)cam.ViewportSize.x, ((float)py + 0.5f + subPixelOffset.y) / (float)cam.ViewportSize.y);
        float2 ndc = make_float2(2.f * p.x + -1.f, -2.f * p.y + 1.f);
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

    // ---- Bridge::loadSurface (BridgeDonut:612-853) and helpers
    float4 sampleTexture(uint textureIndexAndInfo, float lambdaNoDims, float2 uv) const {   // BridgeDonut:270-278 + TextureSampler.hlsli:126-134
        uint textureIndex = textureIndexAndInfo & 0xFFFFu, baseLOD = textureIndexAndInfo >> 24, mipLevels = (textureIndexAndInfo >> 16) & 0xFFu;
        float lambda = 0.5f * (float)baseLOD + lambdaNoDims;
        lambda = fminf_(lambda, fmaxf_((float)mipLevels - 5.0f, 0.0f));
        return sample_trilinear(sc.textures[textureIndex], uv, lambda);
    }
    // ShadingUtils.hlsli:110-165
    static void computeTangentSpace(ShadingData& sd, float4 tangentW, bool ignoreTangent) {
        float3 t3 = xyz(tangentW);
        float NdotT = dot(t3, sd.N);
        bool nonParallel = fabsf(NdotT) < 0.9999f;
        bool nonZero = dot(t3, t3) > 0.f;
        bool valid = tangentW.w != 0.f && nonZero && nonParallel;
        if (!ignoreTangent && valid) { sd.T = normalize(t3 - sd.N * NdotT); sd.B = cross(sd.N, sd.T) * tangentW.w; }
        else { sd.T = perp_stark(sd.N); sd.B = cross(sd.N, sd.T); }
    }
    static void adjustShadingNormal(ShadingData& sd, float4 tangentW, bool recompute, bool ignoreTangent) {
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
    // prevPosW of Bridge::loadSurface in the stable-plane build pass (BridgeDonut:187-199, 619, 631): the hit point in the previous frame's pose — the previous positions of the
    // triangle's vertices (Donut keeps them for skinned meshes; for the others they equal the current ones, which is what interpolating a copy gives) under the previous transform
    float3 prevPosW(uint prim, float bu, float bv) const {
        const Triangle& tr = sc.tris[prim];
        const uint instanceIndex = sc.subInstToInstGeom[tr.subInstance].x;
        const GeometryDesc& g = sc.geometries[sc.subInstances[tr.subInstance].GlobalGeometryIndex_PTMaterialDataIndex >> 16];
        const float3x4& M = (sc.prevInstances.size() == sc.instances.size() ? sc.prevInstances : sc.instances)[instanceIndex].transform;
        const std::vector<float3>& P = sc.prevPositions.size() == sc.positions.size() ? sc.prevPositions : sc.positions;
        const float3 bary = make_float3(1.0f - (bu + bv), bu, bv);
        const uint* idx = &sc.indices[g.indexOffset + tr.triIndex * 3];
        const float3 objPos = (P[g.vertexOffset + idx[0]] * bary.x + P[g.vertexOffset + idx[1]] * bary.y) + P[g.vertexOffset + idx[2]] * bary.z;
        return xform_point(M, objPos);
    }
    SurfaceData loadSurface(uint prim, float bu, float bv, float3 rayDir, RayCone rayCone, MVBlockInputs* mvBlock = nullptr) const {
        const Triangle& tr = sc.tris[prim];
        uint subInst = tr.subInstance, triangleIndex = tr.triIndex;
        uint instanceIndex = sc.subInstToInstGeom[subInst].x;
        const InstanceDesc& inst = sc.instances[instanceIndex];
        const SubInstanceData& si = sc.subInstances[subInst];
        const GeometryDesc& g = sc.geometries[si.GlobalGeometryIndex_PTMaterialDataIndex >> 16];
        const float3x4& M = inst.transform;
        // getGeometryFromHit (BridgeDonut:152-256)
        float3 bary = make_float3(1.0f - (bu + bv), bu, bv);
        const uint* idx = &sc.indices[g.indexOffset + triangleIndex * 3];
        float3 vp[3]; float2 vt[3] = {make_float2(0, 0), make_float2(0, 0), make_float2(0, 0)};
        for (int k = 0; k < 3; k++) vp[k] = sc.positions[g.vertexOffset + idx[k]];
        float3 objPos = (vp[0] * bary.x + vp[1] * bary.y) + vp[2] * bary.z;
        float2 texcoord = make_float2(0, 0);
        if (g.flags & GEOM_HAS_UV) {
            for (int k = 0; k < 3; k++) vt[k] = sc.uvs[g.vertexOffset + idx[k]];
            texcoord = (vt[0] * bary.x + vt[1] * bary.y) + vt[2] * bary.z;
        }
        float3 objFlatN = SafeNormalize(cross(vp[1] - vp[0], vp[2] - vp[0]));
        if (mvBlock) mvBlock->curvatureWS = 0.0f;
        float3 geometryNormal = make_float3(0.f);
        if (g.flags & GEOM_HAS_NORMAL) {
            float3 n[3];
            for (int k = 0; k < 3; k++) {
                n[k] = normalize(Unpack_RGB8_SNORM(sc.normals[g.vertexOffset + idx[k]]));
                if (dot(n[k], objFlatN) < 0.f) n[k] = -n[k];                       // FlipIfOpposite
            }
            if (mvBlock) mvBlock->curvatureWS = TriangleCurvatureApprox_GradN(vp, n, M);
            geometryNormal = (n[0] * bary.x + n[1] * bary.y) + n[2] * bary.z;
            geometryNormal = SafeNormalize(xform_direction4(M, geometryNormal));
        }
        float4 tangent = make_float4(0, 0, 0, 0);
        if (g.flags & GEOM_HAS_TANGENT) {
            float4 tg[3];
            for (int k = 0; k < 3; k++) tg[k] = Unpack_RGBA8_SNORM(sc.tangents[g.vertexOffset + idx[k]]);
            float3 t3 = (xyz(tg[0]) * bary.x + xyz(tg[1]) * bary.y) + xyz(tg[2]) * bary.z;
            t3 = SafeNormalize(xform_direction4(M, t3));
            tangent = make_float4(t3, tg[0].w);
        }
        float3 flatNormal = SafeNormalize(xform_direction4(M, objFlatN));
        bool frontFacing = dot(-rayDir, flatNormal) >= 0.0f;
        if (!(g.flags & GEOM_HAS_NORMAL)) geometryNormal = flatNormal;   // Donut meshes always carry normals; generated ones fall back to the face normal

        float3 posW = xform_point(M, objPos);
        float coneTexLODValue = computeRayConeTriangleLODValue(vp, vt, M);
        float lambda = rayCone.computeLOD(coneTexLODValue, rayDir, flatNormal, true) + S.texLODBias;     // createTextureSampler (BridgeDonut:580-606)

        ShadingData sd; memset(&sd, 0, sizeof(sd));
        sd.posW = posW; sd.V = -rayDir; sd.N = geometryNormal;
        uint materialIndex = si.GlobalGeometryIndex_PTMaterialDataIndex & 0xFFFFu;
        const PTMaterialData& material = sc.materials[materialIndex];

        // sampleGeometryMaterialRTXPT + EvaluateSceneMaterialRTXPT (BridgeDonut:311-428), metal-rough model
        float4 texBase = make_float4(1, 1, 1, 1), texEmissive = make_float4(1, 1, 1, 1), texNormal = make_float4(0.5f, 0.5f, 1.0f, 0.f),
               texMR = make_float4(1, 1, 1, 1), texTrans = make_float4(1, 1, 1, 1);
        bool hasUV = (g.flags & GEOM_HAS_UV) != 0;
        if (hasUV && (material.Flags & PTMaterialFlags_UseBaseOrDiffuseTexture)) texBase = sampleTexture(material.BaseOrDiffuseTextureIndex, lambda, texcoord);
        if (hasUV && (material.Flags & PTMaterialFlags_UseEmissiveTexture)) texEmissive = sampleTexture(material.EmissiveTextureIndex, lambda, texcoord);
        if (hasUV && (material.Flags & PTMaterialFlags_UseNormalTexture)) texNormal = sampleTexture(material.NormalTextureIndex, lambda, texcoord);
        if (hasUV && (material.Flags & PTMaterialFlags_UseMetalRoughOrSpecularTexture)) texMR = sampleTexture(material.MetalRoughOrSpecularTextureIndex, lambda, texcoord);
        if (hasUV && (material.Flags & PTMaterialFlags_UseTransmissionTexture)) texTrans = sampleTexture(material.TransmissionTextureIndex, lambda, texcoord);

        float3 mGeometryNormal = normalize(geometryNormal), mShadingNormal = mGeometryNormal;
        // MaterialProperties holds lp values: a conversion lpfloat(x) per assignment, half operations between lp operands
        float3 baseColor = LP::r3(material.BaseOrDiffuseColor * xyz(texBase));
        float roughness = LP::r(material.Roughness * texMR.y);
        float metalness = LP::r((material.Flags & PTMaterialFlags_MetalnessInRedChannel) ? material.Metalness * texMR.x : material.Metalness * texMR.z);
        if (material.Flags & PTMaterialFlags_UseSpecularGlossModel) {                                  // EvaluateSceneMaterialRTXPT, BridgeDonut:318-333: float colours in, lp base colour / metalness out
            float3 bc; float mt;
            ConvertSpecularGlossToMetalRough(material.BaseOrDiffuseColor * xyz(texBase), material.SpecularColor * xyz(texMR), bc, mt);
            baseColor = LP::r3(bc); metalness = LP::r(mt);
            roughness = LP::r(1.0f - texMR.w * (1.0f - material.Roughness));
        }
        float transmission = LP::r(material.TransmissionFactor), diffuseTransmission = LP::r(material.DiffuseTransmissionFactor);
        if (material.Flags & PTMaterialFlags_UseTransmissionTexture) { transmission = LP::mul(transmission, LP::r(texTrans.x)); diffuseTransmission = LP::mul(diffuseTransmission, LP::r(texTrans.x)); }
        float3 emissiveColor = LP::r3(material.EmissiveColor);
        if (material.Flags & PTMaterialFlags_UseEmissiveTexture) emissiveColor = LP::mul3(emissiveColor, LP::r3(xyz(texEmissive)));
        float matIoR = LP::r(material.IoR);
        if (hasUV && (material.Flags & PTMaterialFlags_UseNormalTexture)) {          // ApplyNormalMapRTXPT (BridgeDonut:280-309)
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
        bool ignoreTangent = (material.Flags & PTMaterialFlags_IgnoreMeshTangentSpace) != 0;
        computeTangentSpace(sd, tangent, ignoreTangent);
        sd.faceNCorrected = frontFacing ? flatNormal : -flatNormal;
        sd.vertexN = frontFacing ? geometryNormal : -geometryNormal;
        sd.frontFacing = frontFacing;
        sd.N = frontFacing ? mShadingNormal : -mShadingNormal;
        if (mvBlock) mvBlock->projectionTerm = fabsf(dot(rayDir, -sd.N));
        bool thin = (material.Flags & PTMaterialFlags_ThinSurface) != 0;
        sd.materialID = materialIndex;
        sd.mtl = MaterialHeader::make();
        { uint pr = 1 + (material.Flags >> PTMaterialFlags_NestedPriorityShift); sd.mtl.setNestedPriority(pr < InteriorList::kMaxNestedPriority ? pr : InteriorList::kMaxNestedPriority); }
        sd.mtl.setThinSurface(thin);
        adjustShadingNormal(sd, tangent, true, ignoreTangent);
        sd.shadowNoLFadeout = LP::r(material.ShadowNoLFadeout);

        float bsdfSpecTrans = LP::mul(transmission, LP::sub(1, metalness)), bsdfDiffTrans = LP::mul(diffuseTransmission, LP::sub(1, metalness));      // lp * (1 - lp)
        sd.mtl.setActiveLobes(Lobe_All);
        float f = (matIoR - 1.f) / (matIoR + 1.f);                        // lp op float literal: float arithmetic
        float F0 = f * f;
        StandardBSDFData bd;                                              // the fields are lp values (BxDF.hlsli:625-634); FalcorBSDF computes from them in float
        bd.diffuse = LP::lerp3(baseColor, make_float3(0.f), metalness);
        bd.specular = LP::lerp3(make_float3(LP::r(F0)), baseColor, metalness);
        bd.roughness = roughness; bd.metallic = metalness;
        bd.transmission = baseColor; bd.diffuseTransmission = bsdfDiffTrans; bd.specularTransmission = bsdfSpecTrans;
        sd.IoR = 1.f;
        bd.eta = LP::div(sd.IoR, matIoR);
        if (!sd.mtl.isThinSurface() && !sd.frontFacing) bd.eta = LP::div(matIoR, sd.IoR);

        SurfaceData ret;
        ret.neeTriangleLightIndex = RTXPT_INVALID_LIGHT_INDEX; ret.neeAnalyticLightIndex = RTXPT_INVALID_LIGHT_INDEX;
        if (material.Flags & PTMaterialFlags_EnableAsAnalyticLightProxy) ret.neeAnalyticLightIndex = si.AnalyticProxyLightIndex;      // BridgeDonut:828-829
        if (sd.frontFacing && any_gt0(emissiveColor)) {
            sd.emission = emissiveColor;
            uint baseIndex = si.EmissiveLightMappingOffset;
            if (baseIndex != 0xFFFFFFFFu) ret.neeTriangleLightIndex = baseIndex + triangleIndex;
        }
        ret.shadingData = sd; ret.bsdf.data = bd; ret.bsdf.diffuseModel = (int)S.diffuseBrdf; ret.interiorIoR = matIoR;
        return ret;
    }
    float loadIoR(uint materialID) const { return (materialID >= sc.materials.size()) ? 1.0f : LP::r(sc.materials[materialID].IoR); }   // BridgeDonut:863-869 (returns lpfloat)
    float3 volumeTransmittance(uint materialID, float t) const {                                                                   // BridgeDonut:871-887
        if (materialID >= sc.materials.size()) return make_float3(1.f);
        const PTMaterialData& m = sc.materials[materialID];
        float3 c = clamp3(m.AttenuationColor, 1e-7f, 1.f);
        float d = fmaxf_(1e-30f, m.AttenuationDistance);
        float3 sigmaA = make_float3(-dm_log(c.x) / d, -dm_log(c.y) / d, -dm_log(c.z) / d);
        return make_float3(dm_exp(-t * sigmaA.x), dm_exp(-t * sigmaA.y), dm_exp(-t * sigmaA.z));
    }

    // PathTracer.hlsli:382-404
    void UpdatePathTravelled(PathState& path, float rayT) const {
        path.incrementVertexIndex();
        path.rayCone = path.rayCone.propagateDistance(rayT);
        path.sceneLength = fminf_(path.sceneLength + rayT, kMaxRayTravel);
    }
    void AccumulatePathRadiance(PathState& path, float3 radiance) const { float4 L = path.GetL(); path.SetL(make_float4(L.x + radiance.x, L.y + radiance.y, L.z + radiance.z, L.w + 0.f)); }

    // PathTracer.hlsli:407-503
    void HandleMiss(PathState& path, float3 rayDir, float rayT) const {
        UpdatePathTravelled(path, rayT);
        float3 environmentEmission = make_float3(0.f);
        NEEBSDFMISInfo misInfo = NEEBSDFMISInfo::Unpack16bit(path.GetPackedMISInfo());
        if (sc.env.enabled) {
            float mipLevel = (path.getCounter(PC_DiffuseBounces) > 1) ? S.envMapDiffuseSampleMIPLevel : 0.f;
            float3 localDir = sc.env.ToLocal(rayDir);
            float3 Le = sc.env.EvalLocal(localDir, mipLevel);
            float misWeight = 1.0f;
            float bsdfScatterPdf = path.GetBsdfScatterPdf();
            if (misInfo.LightSamplingEnabled && bsdfScatterPdf != 0) {
                LightSampler lightSampler = CreateLightSampler(path.id, misInfo.LightSamplingIsSSC);
                uint envIdx = lightSampler.LookupEnvLightByDirection(localDir);
                misWeight = lightSampler.ComputeBSDFMISForEnvironmentQuad(envIdx, bsdfScatterPdf, misInfo.CandidateSamples, misInfo.FullSamples);
            }
            environmentEmission = LP::r3(misWeight * Le);
        }
        const float baseFFThreshold = LP::r(S.fireflyFilterThreshold);
        if (baseFFThreshold != 0) environmentEmission = FireflyFilter(environmentEmission, baseFFThreshold, path.GetFireflyFilterK());
        if (sc.lightTable.DepthExport) ExportDepth(path, path.origin + rayDir * rayT);      // Bridge::ExportNonSurface(path, rayOrigin + rayDir * rayTCurrent, 0) (PathTracer.hlsli:487)
        if (any_gt0(environmentEmission)) AccumulatePathRadiance(path, path.GetThp() * environmentEmission);
        path.setFlag(PF_hit, false);
        path.terminate();
    }

    // PathTracerNestedDielectrics.hlsli:24-128
    float ComputeOutsideIoR(const InteriorList& il, uint materialID, bool entering) const {
        uint outside = il.getTopMaterialID();
        if (!entering) { if (outside == materialID) outside = il.getNextMaterialID(); }
        if (outside == InteriorList::kNoMaterial) return 1.f;
        return loadIoR(outside);
    }
    bool HandleNestedDielectrics(SurfaceData& sfd, PathState& path) const {
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
        sfd.shadingData.IoR = outsideIoR;                                          // Bridge::updateOutsideIoR (BridgeDonut:855-861)
        sfd.bsdf.data.eta = sfd.shadingData.frontFacing ? LP::div(sfd.shadingData.IoR, sfd.interiorIoR) : LP::div(sfd.interiorIoR, sfd.shadingData.IoR);
        return true;
    }

    // PathTracer.hlsli:217-345 (reference mode) + :353-380
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
            if (S.nestedDielectricsQuality > 0 && !sd.mtl.isThinSurface()) {       // UpdateNestedDielectricsOnScatterTransmission
                path.interiorList.handleIntersection(sd.materialID, sd.mtl.getNestedPriority(), sd.frontFacing);
                path.setFlag(PF_insideDielectricVolume, !path.interiorList.isEmpty());
            }
        }
        if (bs.isLobe(Lobe_Delta)) path.setFlag(PF_delta);
        else {
            path.setFlag(PF_deltaOnlyPath, false);
            path.rayCone = RayCone::make(path.rayCone.getWidth(), fminf_(path.rayCone.getSpreadAngle() + ComputeRayConeSpreadAngleExpansionByScatterPDF(bs.pdf), 2.0f * K_PI));
        }
        float fireflyFilterK = ComputeNewScatterFireflyFilterK(path.GetFireflyFilterK(), bs.pdf, bs.lobeP);   // RTXPT_FIREFLY_FILTER == 1
        path.SetFireflyFilterK_BsdfScatterPdf(fireflyFilterK, bs.pdf);
        path.setFlag(PF_enableThreadReorder, true);
        return true;
    }

    // PathTracerNEE.hlsli:88-161
    LightSample GenerateLightSample(const LightSampler& lightSampler, const ShadingData& sd, const StandardBSDF& bsdf, uint candidateSampleCount, UniformSampleSequenceGenerator& sg) const {
        LightSample cand; memset(&cand, 0, sizeof(cand));
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
            PolymorphicLightSample ls = PolymorphicLight_CalcSample(li, interior, sd.posW, sc.env.toWorld);
            LightSample c;
            float pdf = ls.SolidAnglePdf * selectionPdf;
            c.Li = pdf > 0.f ? (ls.Radiance / pdf) : make_float3(0.f);
            c.SolidAnglePdf = ls.SolidAnglePdf;
            float3 surfToLight = ls.Position - sd.posW;
            c.Distance = length(surfToLight);
            c.Direction = surfToLight / fmaxf_(c.Distance, 1e-7f);
            c.LightIndex = lightIndex; c.SelectionPdf = selectionPdf; c.LightSampleableByBSDF = ls.LightSampleableByBSDF; c.FromLocalDistribution = sampleIsLocal;
            float wrsWeight = max3(c.Li) * bsdf.evalPdf(sd, c.Direction);          // EvalSampleWeight (:41-50)
            float r = sampleNext1D(sg);
            weightSum += wrsWeight;                                                 // NEEWeightedReservoirSampler::Add (:70-80)
            float thr = saturate(wrsWeight / weightSum);
            if (r < thr) { cand = c; candWeight = wrsWeight; }
        }
        cand.Li = cand.Li * (1.0f / (candWeight / weightSum));                      // LATE_WRS_MIS: wrsMIS == 1 here
        return cand;
    }
    // PathTracerNEE.hlsli:166-275 (ProcessLightSample) + :277-346
    NEEResult HandleNEE(const PathState& pre, const ShadingData& sd, const StandardBSDF& bsdf, UniformSampleSequenceGenerator& sg) const {
        const LightSampler lightSampler = CreateLightSampler(pre.id, LightSampler::IsScreenSpaceCoherentHeuristic(sc.lightTable, pre.rayCone.getWidth(), pre.sceneLength));      // :306
        uint fullSamples = S.NEEFullSamples < 63u ? S.NEEFullSamples : 63u;
        bool hasNonDeltaLobes = (bsdf.getLobes() & Lobe_NonDelta) != 0;
        bool applyNEE = hasNonDeltaLobes && !lightSampler.IsEmpty() && fullSamples > 0;
        if (!applyNEE) return NEEResult::empty();
        uint candidateSampleCount = S.NEECandidateSamples;
        NEEResult result = NEEResult::empty();
        result.BSDFMISInfo.LightSamplingEnabled = true; result.BSDFMISInfo.LightSamplingIsSSC = lightSampler.IsScreenSpaceCoherent;
        result.BSDFMISInfo.CandidateSamples = candidateSampleCount; result.BSDFMISInfo.FullSamples = fullSamples;
        for (uint s = 0; s < fullSamples; s++) {
            LightSample ls = GenerateLightSample(lightSampler, sd, bsdf, candidateSampleCount, sg);
            bool visible = false;
            if (ls.Valid()) {
                float faceSide = dot(sd.N, ls.Direction) >= 0 ? 1.f : -1.f;                        // ComputeVisibilityRay (:166-182)
                float3 o = ComputeRayOrigin(sd.posW, sd.faceNCorrected * faceSide);
                if (counters) counters->shadowRays++;
                visible = trace_visibility(sc, o, ls.Direction, 0.0f, ls.Distance * 0.9985f, counters ? &counters->nodeVisitsSh : 0, counters ? &counters->triTestsSh : 0);
            }
            if (!visible) continue;
            float fadeOut = (sd.shadowNoLFadeout > 0) ? ComputeLowGrazingAngleFalloff(ls.Direction, sd.vertexN, sd.shadowNoLFadeout, 2.0f * sd.shadowNoLFadeout) : 1.0f;
            uint localCount, globalCount;
            lightSampler.GetCandidateSampleCounts(candidateSampleCount, localCount, globalCount);
            float thisPdf, otherPdf, thisCount, otherCount;                                          // the inner (WRS) MIS between the two samplers (:217-224)
            lightSampler.ComputeLightSelectionPdfs(ls.SelectionPdf, ls.LightIndex, ls.FromLocalDistribution, localCount, globalCount, thisPdf, otherPdf, thisCount, otherCount);
            float wrsMIS = EvalMIS_Balance(1, thisPdf, 1, otherPdf);
            wrsMIS = wrsMIS / thisCount;
            float scatterPdfForDir = bsdf.evalPdf(sd, ls.Direction);
            float lightAvgPdf = (thisPdf + otherPdf) * (float)fullSamples;                          // ComputeLightVsBSDF_MIS_ForLight (LightSampler.hlsli:282-316)
            float pathMIS = EvalMIS_Balance(1, lightAvgPdf * ls.SolidAnglePdf, 1, ls.LightSampleableByBSDF ? scatterPdfForDir : 0.f);
            float3 Li = ls.Li * (fadeOut * wrsMIS * pathMIS / (float)fullSamples);
            float4 bsdfThp = bsdf.eval(sd, ls.Direction);
            float3 radiance = xyz(bsdfThp) * Li;
            float radianceAvg = Average(radiance);
            float specAvg = bsdfThp.w * Average(Li);
            if (S.fireflyFilterThreshold != 0) {
                float pdf = ls.SelectionPdf * ls.SolidAnglePdf;
                float k = ComputeNewScatterFireflyFilterK(pre.GetFireflyFilterK(), pdf, 1.0f);
                radiance = radiance * FireflyFilterShort(radianceAvg, S.fireflyFilterThreshold, k);
            }
            float3 preThp = pre.GetThp();
            radiance = radiance * preThp;
            specAvg *= Average(preThp);
            result.AccumulateRadiance(radiance, specAvg);
            if (ls.LightIndex != RTXPT_INVALID_LIGHT_INDEX && lightSampler.IsTemporalFeedbackRequired()) {      // :266-273, LightSampler.hlsli:184-200
                float feedbackWeight = lightSampler.FeedbackWeightFromNEE(ls.LightIndex, radianceAvg * Average(preThp));
                float rnd = sampleNext1D(sg);
                if (fbTotalWeight) { const uint slot = (pre.id & 0xFFFFu) * fbWidth + (pre.id >> 16);
                    LightFeedbackReservoir_Add(fbTotalWeight[slot], fbCandidates[slot], rnd, ls.LightIndex, feedbackWeight, lightSampler.IsScreenSpaceCoherent); }
            }
        }
        return result;
    }
    // PathTracer.hlsli:182-208
    bool HandleRussianRoulette(PathState& path, UniformSampleSequenceGenerator& sg) const {
        if (!S.enableRussianRoulette) return false;
        float rrVal = sqrtf_(Luminance(path.GetThp()));
        float prob = saturate(0.85f - rrVal); prob = prob * prob;
        prob = saturate(prob + fmaxf_(0.f, ((float)path.getVertexIndex() / (float)S.bounceCount - 0.4f)));
        if (sampleNext1D(sg) < prob) return true;
        path.SetPackedMISInfo_ThpRuRuCorrection(path.GetPackedMISInfo(), 1.0f / (1.0f - prob));
        return false;
    }

    // PathTracer.hlsli:505-762 (reference mode)
    void HandleHit(PathState& path, float3 rayOrigin, float3 rayDir, const HitInfo& hit) const {
        UpdatePathTravelled(path, hit.t);
        SurfaceData sfd = loadSurface(hit.prim, hit.u, hit.v, rayDir, path.rayCone);
        if (S.nestedDielectricsQuality > 0 && !path.interiorList.isEmpty()) {
            float3 tr = volumeTransmittance(path.interiorList.getTopMaterialID(), hit.t);
            path.SetThp(path.GetThp() * tr);
        }
        bool rejectedFalseHit = !HandleNestedDielectrics(sfd, path);
        if (rejectedFalseHit) return;
        const ShadingData& sd = sfd.shadingData; const StandardBSDF& bsdf = sfd.bsdf;
        float3 surfaceEmission = make_float3(0.f);
        NEEBSDFMISInfo misInfo = NEEBSDFMISInfo::Unpack16bit(path.GetPackedMISInfo());
        const LightSampler lightSampler = CreateLightSampler(path.id, misInfo.LightSamplingIsSSC);      // "configured same as it was at previous vertex" (PathTracer.hlsli:617,639)
        if (any_gt0(sd.emission)) {
            float misWeight = 1.0f;
            float bsdfScatterPdf = path.GetBsdfScatterPdf();
            if (misInfo.LightSamplingEnabled && bsdfScatterPdf != 0)
                misWeight = lightSampler.ComputeBSDFMISForEmissiveTriangle(sfd.neeTriangleLightIndex, bsdfScatterPdf, rayOrigin, sd.posW, misInfo.CandidateSamples, misInfo.FullSamples);
            surfaceEmission = LP::r3(sd.emission * misWeight);
        }
        if (sfd.neeAnalyticLightIndex != RTXPT_INVALID_LIGHT_INDEX) {                  // PathTracer.hlsli:636-648: the mesh stands in for an analytic (sphere) light
            const float bsdfPdf = misInfo.LightSamplingEnabled ? LP::r(path.GetBsdfScatterPdf()) : 0.0f; float3 add;
            if (lightSampler.ComputeAnalyticLightProxyContribution(sfd.neeAnalyticLightIndex, bsdfPdf, rayOrigin, rayDir, misInfo.CandidateSamples, misInfo.FullSamples, add)) {
                add = LP::r3(add); surfaceEmission = make_float3(LP::add(surfaceEmission.x, add.x), LP::add(surfaceEmission.y, add.y), LP::add(surfaceEmission.z, add.z));
            }
        }
        if (any_gt0(surfaceEmission)) {
            const float baseFFThreshold = LP::r(S.fireflyFilterThreshold);
            if (baseFFThreshold != 0) surfaceEmission = FireflyFilter(surfaceEmission, baseFFThreshold, path.GetFireflyFilterK());
            if (any_gt0(surfaceEmission)) AccumulatePathRadiance(path, path.GetThp() * surfaceEmission);
        }
        if (sc.lightTable.DepthExport) {                            // Bridge::ExportSurface(path, surfaceData, path.GetSceneLength(), 0) (PathTracer.hlsli:684, BridgeDonut:1105-1118)
            float3 co, cd; computeCameraRay(path.id >> 16, path.id & 0xFFFFu, co, cd);
            ExportDepth(path, co + cd * path.sceneLength);
        }
        if (path.isTerminatingAtNextBounce()) { path.terminate(); return; }
        float rr = path.GetThpRuRuCorrection();
        path.SetThp(path.GetThp() * make_float3(rr));
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make(path.id, path.getVertexIndex(), sampleIndex);
        UniformSampleSequenceGenerator uniformSG = UniformSampleSequenceGenerator::make(vb, SGES_Base);
        const PathState preScatterPath = path;
        bool scatterValid = GenerateScatterRay(sd, bsdf, path, vb);
        NEEResult nee = S.NEEEnabled ? HandleNEE(preScatterPath, sd, bsdf, uniformSG) : NEEResult::empty();
        path.SetPackedMISInfo_ThpRuRuCorrection(nee.BSDFMISInfo.Pack16bit(), path.GetThpRuRuCorrection());
        float4 neeR = nee.Get();
        if (neeR.x > 0 || neeR.y > 0 || neeR.z > 0 || neeR.w > 0) AccumulatePathRadiance(path, xyz(neeR));
        if (!scatterValid) path.terminate();
        bool shouldTerminate = HasFinishedSurfaceBounces(path.getVertexIndex() + 1, path.getCounter(PC_DiffuseBounces));
        shouldTerminate |= HandleRussianRoulette(path, uniformSG);
        if (shouldTerminate) path.setFlag(PF_terminateAtNextBounce);
    }

    // PathTracerSample.hlsl:200-250: returns float4(L.rgb, 1) as written to u_OutputColor
    float4 tracePixel(uint px, uint py) const {
        PathState path = EmptyPathInitialize(px, py);
        computeCameraRay(px, py, path.origin, path.dir);
        while (path.isActive()) {
            float3 o = path.origin, d = path.dir;
            if (counters) counters->extendRays++;
            HitInfo h = trace_closest(sc, o, d, 0.0f, kMaxRayTravel, counters ? &counters->nodeVisitsExt : 0, counters ? &counters->triTestsExt : 0);
            if (h.prim == 0xFFFFFFFFu) HandleMiss(path, d, kMaxRayTravel);
            else { if (counters) counters->hits++; HandleHit(path, o, d, h); }
        }
        float4 L = path.GetL();
        return make_float4(L.x, L.y, L.z, 1.0f);
    }
};

} // namespace ptref
