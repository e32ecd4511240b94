// mi355pt — stable planes (second half): the realtime mode's delta-path decomposition and the guide buffers a denoiser reads.
// Part of the PRODUCT path (libmi355pt.so). Written to the arithmetic contract stated in pt_vec.h.
// Reference anchors (paths relative to /root/reference/Rtxpt/Shaders/):
//   PathTracer/StablePlanes.hlsli:28-371                 StablePlane (80 B), StablePlanesContext (header / plane buffer / stable radiance), branch ids
//   PathTracer/PathTracerStablePlanes.hlsli:24-414       SplitDeltaPath, StablePlanesHandleHit (build), StablePlanesOnScatter (fill), StablePlanesHandleMiss
//   PathTracer/PathTracer.hlsli:47-91,407-503,505-762    the BUILD / FILL branches of EmptyPathInitialize, HandleMiss, HandleHit
//   PathTracerSample.hlsl:33-112,200-250                 FirstHitFromVBuffer, postProcessHit, the raygen loop of the two passes
//   PathTracer/PathState.hlsli:98-160                    the BUILD-mode view of the 80-byte path state (imageXform in place of L, motion-vector scene length in pack0)
//   PathTracer/Rendering/Materials/BxDF.hlsli:972-1053, StandardBSDF.hlsli:227-238, IBSDF.hlsli:96-110, Microfacet.hlsli:282-352    evalDeltaLobes, estimateSpecDiffBSDF
//   PathTracer/Utils/Utils.hlsli:85-90,154-187,272-356   ReinhardMax, NDirToOctUnorm30, PackOrthoMatrix, Morton16BitEncode, GenericTSPixelToAddress
//   PathTracer/PathTracerHelpers.hlsli:227-268           MatrixRotateFromTo
//   PathTracerBridgeDonut.hlsli:890-909,1098-1177        computeMotionVector, ExportSurfaceInit / ExportSurface / ExportNonSurface / ExportSpecHitT*
// This header is written once for both sides of the parity fence: SP_BRANCH_FIELD names the PathState word that carries stableBranchID
// (the wavefront pool keeps the sample index there in reference mode, pt_path.h:86).
#pragma once
#ifndef SP_BRANCH_FIELD
#define SP_BRANCH_FIELD sampleIndex
#endif

static const uint cStablePlaneCount = 3u, cStablePlaneMaxVertexIndex = 15u, cMaxDeltaLobes = 3u;
static const uint cStablePlaneInvalidBranchID = 0xFFFFFFFFu, cStablePlaneEnqueuedBranchID = 0xFFFFFFFEu, cStablePlaneJustStartedID = 0u;
static const float kEnvironmentMapSceneDistance = 50000.0f * 100.0f;      // Config.h:84-85 (kMaxSceneDistance * 100)
enum : uint { PF_stablePlaneOnPlane = 1u << 16, PF_stablePlaneOnBranch = 1u << 17, PF_stablePlaneBaseScatterDiff = 1u << 18, PF_exportSpecHitTQueued = 1u << 19, PF_stablePlaneOnDominantBranch = 1u << 20 };
static const uint SP_PC_BouncesFromStablePlane = 2;      // PackedCounters::BouncesFromStablePlane
static const uint kStablePlaneIndexBitOffset = 14u + kVertexIndexBitCount, kStablePlaneIndexBitMask = 3u << kStablePlaneIndexBitOffset;
enum : uint { PTMaterialFlags_PSDExcludeBit = 0x400u, PTMaterialFlags_PSDBlockMVsAtSurfaceTypeB0 = 1u << 13, PTMaterialFlags_PSDBlockMVsAtSurfaceTypeB1 = 1u << 14,
              PTMaterialFlags_PSDDominantDeltaLobeP1Mask = 0x0F000000u, PTMaterialFlags_PSDDominantDeltaLobeP1Shift = 24 };

// ---- PathState accessors the reference mode has no use for (PathState.hlsli:100-160, 214-262)
static inline void SP_setCounter(PathState& p, uint type, uint v) { const uint shift = type << 3; p.packedCounters = (p.packedCounters & ~(0xffu << shift)) | ((v & 0xffu) << shift); }
static inline void SP_setVertexIndex(PathState& p, uint index) { p.flagsAndVertexIndex &= ~kVertexIndexBitMask; p.flagsAndVertexIndex |= index; }
static inline uint SP_getStablePlaneIndex(const PathState& p) { return (p.flagsAndVertexIndex & kStablePlaneIndexBitMask) >> kStablePlaneIndexBitOffset; }
static inline void SP_setStablePlaneIndex(PathState& p, uint index) { p.flagsAndVertexIndex &= ~kStablePlaneIndexBitMask; p.flagsAndVertexIndex |= index << kStablePlaneIndexBitOffset; }
static inline void SP_SetMotionVectorSceneLength(PathState& p, float l) { p.pack0 = asuint(l); }
static inline float SP_GetMotionVectorSceneLength(const PathState& p) { return asfloat(p.pack0); }

// ---- Utils.hlsli
static inline float3 ReinhardMax(float3 color) {
    float luminance = fmaxf_(1e-7f, fmaxf_(fmaxf_(color.x, color.y), color.z));
    float reinhard = luminance / (luminance + 1.0f);
    return color * (reinhard / luminance);
}
static inline uint NDirToOctUnorm30(float3 n) {
    float2 p = Encode_Oct(n);
    p = make_float2(saturate(p.x * 0.5f + 0.5f), saturate(p.y * 0.5f + 0.5f));
    return ((uint)(p.x * 32767.0f + 0.5f) & 0x7fffu) | (((uint)(p.y * 32767.0f + 0.5f) & 0x7fffu) << 15);
}
static inline float3 OctToNDirUnorm30(uint pUnorm) {
    float2 p;
    p.x = saturate((float)(pUnorm & 0x7fffu) / 32767.0f);
    p.y = saturate((float)(pUnorm >> 15) / 32767.0f);
    p = make_float2(p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f);
    return Decode_Oct(p);
}
struct float3x3 { float3 r[3]; };
static inline float3x3 make_float3x3(float3 a, float3 b, float3 c) { float3x3 m; m.r[0] = a; m.r[1] = b; m.r[2] = c; return m; }
static inline float3 mul(float3 v, const float3x3& M) {          // row vector times matrix, every sum as (x + y) + z
    return make_float3((v.x * M.r[0].x + v.y * M.r[1].x) + v.z * M.r[2].x, (v.x * M.r[0].y + v.y * M.r[1].y) + v.z * M.r[2].y, (v.x * M.r[0].z + v.y * M.r[1].z) + v.z * M.r[2].z);
}
static inline float3 mul(const float3x3& M, float3 v) { return make_float3(dot(M.r[0], v), dot(M.r[1], v), dot(M.r[2], v)); }
static inline float3x3 mul(const float3x3& A, const float3x3& B) { float3x3 R; for (int i = 0; i < 3; i++) R.r[i] = mul(A.r[i], B); return R; }
static inline float3x3 transpose(const float3x3& M) {
    return make_float3x3(make_float3(M.r[0].x, M.r[1].x, M.r[2].x), make_float3(M.r[0].y, M.r[1].y, M.r[2].y), make_float3(M.r[0].z, M.r[1].z, M.r[2].z));
}
static inline uint2 PackOrthoMatrix(const float3x3& xform) {
    uint2 packed;
    uint handedness = dot(cross(xform.r[0], xform.r[1]), xform.r[2]) > 0 ? 1u : 0u;
    packed.x = NDirToOctUnorm30(xform.r[0]);
    packed.y = NDirToOctUnorm30(xform.r[1]);
    packed.y |= handedness << 31;
    return packed;
}
static inline float3x3 UnpackOrthoMatrix(uint2 packed) {
    float3x3 xform;
    uint handedness = packed.y >> 31;
    packed.y &= 0x7FFFFFFFu;
    xform.r[0] = OctToNDirUnorm30(packed.x);
    xform.r[1] = OctToNDirUnorm30(packed.y);
    xform.r[2] = handedness ? cross(xform.r[0], xform.r[1]) : cross(xform.r[1], xform.r[0]);
    return xform;
}
static inline float3x3 SP_GetImageXform(const PathState& p) { return UnpackOrthoMatrix(make_uint2(p.pack45[0], p.pack45[1])); }
static inline void SP_SetImageXform(PathState& p, const float3x3& m) { uint2 k = PackOrthoMatrix(m); p.pack45[0] = k.x; p.pack45[1] = k.y; }
static inline uint Morton16BitEncode(uint x, uint y) {
    uint temp = (x & 0xffu) | ((y & 0xffu) << 16);
    temp = (temp ^ (temp << 4)) & 0x0f0f0f0fu;
    temp = (temp ^ (temp << 2)) & 0x33333333u;
    temp = (temp ^ (temp << 1)) & 0x55555555u;
    return ((temp >> 15) | temp) & 0xffffu;
}
// Generic tiled swizzled addressing (8 x 8 tiles, Morton order inside a tile)
static inline uint GenericTSComputeLineStride(uint imageWidth, uint) { return ((imageWidth + 7u) / 8u) * 8u; }
static inline uint GenericTSComputePlaneStride(uint imageWidth, uint imageHeight) { return GenericTSComputeLineStride(imageWidth, imageHeight) * ((imageHeight + 7u) / 8u) * 8u; }
static inline uint GenericTSPixelToAddress(uint px, uint py, uint planeIndex, uint lineStride, uint planeStride) {
    uint xInTile = px % 8u, yInTile = py % 8u;
    uint tilePixelIndex = Morton16BitEncode(xInTile, yInTile);
    uint tileBaseX = px - xInTile, tileBaseY = py - yInTile;
    return tileBaseX * 8u + tileBaseY * lineStride + tilePixelIndex + planeIndex * planeStride;
}
// Packing.hlsli:175-184, 196-197
static inline uint Pack_R11G11B10_FLOAT(float3 rgb) {
    const float top = asfloat(0x477C0000u);
    rgb = make_float3(fminf_(rgb.x, top), fminf_(rgb.y, top), fminf_(rgb.z, top));
    uint r = ((f32tof16(rgb.x) + 8u) >> 4) & 0x000007FFu;
    uint g = ((f32tof16(rgb.y) + 8u) << 7) & 0x003FF800u;
    uint b = ((f32tof16(rgb.z) + 16u) << 17) & 0xFFC00000u;
    return r | g | b;
}
static inline uint PackTwoFp32ToFp16(float a, float b) { return (f32tof16(clampf(a, -HLF_MAX, HLF_MAX)) << 16) | f32tof16(clampf(b, -HLF_MAX, HLF_MAX)); }
// PathTracerHelpers.hlsli:227-268 (columnMajor = true)
static inline float3x3 MatrixRotateFromTo(float3 from, float3 to) {
    const float e = dot(from, to);
    const float f = fabsf(e);
    if (f > 1.0f)       // `float(1.0f - 1e-10f)` is 1.0f
        return make_float3x3(make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, 0, 1));
    const float3 v = cross(from, to);
    const float h = 1.0f / (1.0f + e);
    const float hvx = h * v.x, hvz = h * v.z, hvxy = hvx * v.y, hvxz = hvx * v.z, hvyz = hvz * v.y;
    float3x3 mtx;
    mtx.r[0] = make_float3(e + hvx * v.x, hvxy - v.z, hvxz + v.y);
    mtx.r[1] = make_float3(hvxy + v.z, e + h * v.y * v.y, hvyz - v.x);
    mtx.r[2] = make_float3(hvxz - v.y, hvyz + v.x, e + hvz * v.z);
    return mtx;
}

// ---- delta lobes (IBSDF.hlsli:24-35, BxDF.hlsli:972-1053, StandardBSDF.hlsli:227-238)
struct DeltaLobe { float3 thp; float probability; float3 dir; int transmission; };
static inline DeltaLobe DeltaLobe_make() { DeltaLobe r; r.thp = make_float3(0.f); r.dir = make_float3(0.f); r.transmission = 0; r.probability = 0; return r; }
static inline void FalcorBSDF_evalDeltaLobes(const FalcorBSDF& b, bool psdExclude, float3 wi, DeltaLobe deltaLobes[cMaxDeltaLobes], uint& deltaLobeCount, float& nonDeltaPart) {
    deltaLobeCount = 2;
    for (uint i = 0; i < cMaxDeltaLobes; i++) deltaLobes[i] = DeltaLobe_make();      // (the reference initialises the first two; the third is never read)
    nonDeltaPart = b.pDiffuseReflection + b.pDiffuseTransmission;
    if (b.specularReflection.alpha > 0) nonDeltaPart += b.pSpecularReflection;
    if (b.specularReflectionTransmission.alpha > 0) nonDeltaPart += b.pSpecularReflectionTransmission;
    if ((b.pSpecularReflection + b.pSpecularReflectionTransmission) == 0 || psdExclude) return;
    DeltaLobe deltaReflection = DeltaLobe_make(), deltaTransmission = DeltaLobe_make();
    deltaReflection.transmission = 0; deltaTransmission.transmission = 1;
    deltaReflection.dir = make_float3(-wi.x, -wi.y, wi.z);
    if (b.specularReflection.alpha == 0 && b.specularReflection.hasLobe(Lobe_DeltaReflection)) {
        deltaReflection.probability = b.pSpecularReflection;
        deltaReflection.thp = (1.0f - b.pSpecularReflectionTransmission) * evalFresnelSchlick(b.specularReflection.albedo, 1.f, wi.z);
    }
    if (b.specularReflectionTransmission.alpha == 0.f) {
        const bool hasReflection = b.specularReflectionTransmission.hasLobe(Lobe_DeltaReflection);
        const bool hasTransmission = b.specularReflectionTransmission.hasLobe(Lobe_DeltaTransmission);
        if (hasReflection || hasTransmission) {
            float cosThetaT;
            float F = evalFresnelDielectric(b.specularReflectionTransmission.eta, wi.z, cosThetaT);
            if (hasReflection) {
                float localProbability = b.pSpecularReflectionTransmission * F;
                float3 weight = make_float3(1.f) * localProbability;
                deltaReflection.thp += weight;
                deltaReflection.probability += localProbability;
            }
            if (hasTransmission) {
                float actualEta = b.specularReflectionTransmission.eta;
                if (b.specularReflectionTransmission.isThinSurface) { actualEta = 1.0f; F = evalFresnelDielectric(actualEta, wi.z, cosThetaT); }
                float localProbability = b.pSpecularReflectionTransmission * (1.0f - F);
                float3 weight = b.specularReflectionTransmission.transmissionAlbedo * localProbability;
                deltaTransmission.dir = make_float3(-wi.x * actualEta, -wi.y * actualEta, -cosThetaT);
                deltaTransmission.thp = weight;
                deltaTransmission.probability = localProbability;
            }
        }
    }
    deltaLobes[0] = deltaTransmission;
    deltaLobes[1] = deltaReflection;
}
static inline void StandardBSDF_evalDeltaLobes(const StandardBSDF& bsdf, const ShadingData& sd, bool psdExclude, DeltaLobe deltaLobes[cMaxDeltaLobes], uint& deltaLobeCount, float& nonDeltaPart) {
    float3 wiLocal = sd.toLocal(sd.V);
    FalcorBSDF b; b.init(sd.mtl, sd.V, sd.N, bsdf.data, bsdf.diffuseModel);
    FalcorBSDF_evalDeltaLobes(b, psdExclude, wiLocal, deltaLobes, deltaLobeCount, nonDeltaPart);
    for (uint i = 0; i < deltaLobeCount; i++) deltaLobes[i].dir = sd.fromLocal(deltaLobes[i].dir);
}
// Microfacet.hlsli:282-352, the coefficients of the correlated G term (SpecularMaskingFunction = SmithGGXCorrelated, BxDFConfig.hlsli)
static inline float3 approxSpecularIntegralGGX(float3 specularReflectance, float alpha, float cosTheta) {
    cosTheta = fabsf(cosTheta);
    float4 X; X.x = 1.f; X.y = cosTheta; X.z = cosTheta * cosTheta; X.w = cosTheta * X.z;
    float4 Y; Y.x = 1.f; Y.y = alpha; Y.z = alpha * alpha; Y.w = alpha * Y.z;
    // mul(M, v) of a row-major matrix: one dot product per row; the dots of two / three terms as x + y and (x + y) + z
    float2 m1 = make_float2(0.995367f * X.x + -1.38839f * X.y, -0.24751f * X.x + 1.97442f * X.y);
    float3 m2 = make_float3((1.0f * X.x + 2.68132f * X.y) + 52.366f * X.w, (16.0932f * X.x + -3.98452f * X.y) + 59.3013f * X.w, (-5.18731f * X.x + 255.259f * X.y) + 2544.07f * X.w);
    float2 m3 = make_float2(-0.0564526f * X.x + 3.82901f * X.y, 16.91f * X.x + -11.0303f * X.y);
    float3 m4 = make_float3((1.0f * X.x + 4.11118f * X.z) + -1.37886f * X.w, (19.3254f * X.x + -28.9947f * X.z) + 16.9514f * X.w, (0.545386f * X.x + 96.0994f * X.z) + -79.4492f * X.w);
    float bias = (m1.x * Y.x + m1.y * Y.y) * (1.0f / ((m2.x * Y.x + m2.y * Y.y) + m2.z * Y.w));
    float scale = (m3.x * Y.x + m3.y * Y.y) * (1.0f / ((m4.x * Y.x + m4.y * Y.y) + m4.z * Y.w));
    const float third = 1.f / 3.f;
    float specularReflectanceLuma = dot(specularReflectance, make_float3(third, third, third));
    bias *= saturate(specularReflectanceLuma * 50.0f);
    const float s = fmaxf_(0.0f, scale), bb = fmaxf_(0.0f, bias);
    return make_float3(specularReflectance.x * s + bb, specularReflectance.y * s + bb, specularReflectance.z * s + bb);      // mad(): unfused under the arithmetic contract
}
// IBSDF.hlsli:96-110; LP: the lp build of the data fields (lpfloat products are rounded to the lp type)
template <class LP> static inline void estimateSpecDiffBSDF(const StandardBSDFData& data, float3& outDiffEstimate, float3& outSpecEstimate, float3 normal, float3 viewVector) {
    float dataRoughness = data.roughness;
    float alpha = LP::mul(dataRoughness, dataRoughness);
    float roughness = alpha < kMinGGXAlpha ? 0.f : dataRoughness;
    float dataDiffuseTransmission = data.diffuseTransmission, dataSpecularTransmission = data.specularTransmission;
    float3 dataTransmission = data.transmission, dataSpecular = data.specular;
    const float omdt = LP::sub(1.f, dataDiffuseTransmission), omst = LP::sub(1.f, dataSpecularTransmission);
    float3 diffuseReflectionAlbedo = LP::mul3(data.diffuse, LP::mul(omdt, omst));
    float3 diffuseTransmissionAlbedo = LP::mul3(LP::mul3(dataTransmission, dataDiffuseTransmission), omst);
    float3 specularReflectionAlbedo = LP::mul3(dataSpecular, omst);
    float3 specularTransmissionAlbedo = LP::mul3(dataTransmission, dataSpecularTransmission);
    outDiffEstimate = make_float3(LP::add(diffuseReflectionAlbedo.x, diffuseTransmissionAlbedo.x), LP::add(diffuseReflectionAlbedo.y, diffuseTransmissionAlbedo.y), LP::add(diffuseReflectionAlbedo.z, diffuseTransmissionAlbedo.z));
    const float NdotV = saturate(dot(normal, viewVector));
    const float ggxAlpha = roughness * roughness;
    float3 specularReflectance = approxSpecularIntegralGGX(specularReflectionAlbedo, ggxAlpha, NdotV);
    specularReflectance += specularTransmissionAlbedo;
    outSpecEstimate = specularReflectance;
}

// ---- branch ids (StablePlanes.hlsli:260-300)
static inline uint StablePlanesAdvanceBranchID(uint prevStableBranchID, uint deltaLobeID) { return (prevStableBranchID << 2) | deltaLobeID; }
static inline uint SP_firstbithigh(uint v) { uint r = 0xFFFFFFFFu; for (uint i = 0; i < 32u; i++) if (v & (1u << i)) r = i; return r; }      // HLSL firstbithigh: -1 for 0
static inline uint StablePlanesVertexIndexFromBranchID(uint stableBranchID) { return SP_firstbithigh(stableBranchID) / 2u + 1u; }
static inline bool StablePlaneIsOnPlane(uint planeBranchID, uint vertexBranchID) { return planeBranchID == vertexBranchID; }
static inline bool StablePlaneIsOnStablePath(uint planeBranchID, uint planeVertexIndex, uint vertexBranchID, uint vertexIndex) {
    if (vertexIndex > planeVertexIndex) return false;
    const uint sh = (planeVertexIndex - vertexIndex) * 2u;
    return ((sh >= 32u) ? 0u : (planeBranchID >> sh)) == vertexBranchID;
}

// ---- the buffers (StablePlanes.hlsli:41-76, RenderTargets.cpp:60-141, 340-352)
struct StablePlane {
    float3 RayOrigin; float LastRayTCurrent; float3 RayDir; float SceneLength;
    uint PackedThpAndMVs[3]; uint VertexIndexAndRoughness; uint DenoiserPackedBSDFEstimate[3]; uint PackedNormal;
    uint PackedNoisyRadianceAndSpecAvg[2]; uint FlagsAndVertexIndex; uint PackedCounters;
};
static_assert(sizeof(StablePlane) == 80, "StablePlane layout");
// the host's matrices are row-major float4x4 for row vectors (donut PlanarViewConstants): clip = mul(float4(p, 1), M)
struct StablePlanesConsts {
    uint imageWidth, imageHeight, genericTSLineStride, genericTSPlaneStride;
    uint activeStablePlaneCount, maxStablePlaneVertexDepth, allowPrimarySurfaceReplacement; float invSubSampleCount;
    float matWorldToClip[16], matWorldToClipNoOffset[16], prevMatWorldToClipNoOffset[16]; float clipToWindowScale[2]; float _pad[2];
};
// what the host sets per frame (Sample.cpp:1509-1540): the record both sides of the fence take, turned into the shader constants by SP_make_consts
struct StablePlanesParams {
    uint activeStablePlaneCount, maxStablePlaneVertexDepth, allowPrimarySurfaceReplacement, subSampleCount;
    float matWorldToClip[16], matWorldToClipNoOffset[16], prevMatWorldToClipNoOffset[16]; float clipToWindowScale[2]; float _pad[2];
};
static inline StablePlanesConsts SP_make_consts(const StablePlanesParams& p, uint width, uint height, uint bounceCount) {
    StablePlanesConsts c; __builtin_memset(&c, 0, sizeof(c));
    c.imageWidth = width; c.imageHeight = height; c.genericTSLineStride = GenericTSComputeLineStride(width, height); c.genericTSPlaneStride = GenericTSComputePlaneStride(width, height);
    c.activeStablePlaneCount = p.activeStablePlaneCount < 1u ? 1u : (p.activeStablePlaneCount > cStablePlaneCount ? cStablePlaneCount : p.activeStablePlaneCount);
    uint d = p.maxStablePlaneVertexDepth < cStablePlaneMaxVertexIndex ? p.maxStablePlaneVertexDepth : cStablePlaneMaxVertexIndex;      // min(min(StablePlanesMaxVertexDepth, cStablePlaneMaxVertexIndex), BounceCount)
    c.maxStablePlaneVertexDepth = d < bounceCount ? d : bounceCount;
    c.allowPrimarySurfaceReplacement = p.allowPrimarySurfaceReplacement;
    c.invSubSampleCount = 1.0f / (float)(p.subSampleCount ? p.subSampleCount : 1u);
    for (int i = 0; i < 16; i++) { c.matWorldToClip[i] = p.matWorldToClip[i]; c.matWorldToClipNoOffset[i] = p.matWorldToClipNoOffset[i]; c.prevMatWorldToClipNoOffset[i] = p.prevMatWorldToClipNoOffset[i]; }
    c.clipToWindowScale[0] = p.clipToWindowScale[0]; c.clipToWindowScale[1] = p.clipToWindowScale[1];
    return c;
}
// Header: [plane 0..2] branch ids, [3] first-hit ray length | dominant plane index, each imageWidth x imageHeight (a texture array in the reference: scan-line order);
// StableRadiance / MotionVectors are RGBA16F render targets (four binary16 values per pixel: every store rounds), Depth / SpecularHitT R32F, Throughput R11G11B10.
struct StablePlanesBuffers { uint* Header; StablePlane* Planes; uint2* StableRadiance; float* Depth; float* SpecularHitT; uint2* MotionVectors; uint* Throughput; };

static inline float4 SP_mul_row(float3 p, const float* M) {      // mul(float4(p, 1), M): ((x + y) + z) + w per column
    return make_float4(((p.x * M[0] + p.y * M[4]) + p.z * M[8]) + 1.0f * M[12], ((p.x * M[1] + p.y * M[5]) + p.z * M[9]) + 1.0f * M[13],
                       ((p.x * M[2] + p.y * M[6]) + p.z * M[10]) + 1.0f * M[14], ((p.x * M[3] + p.y * M[7]) + p.z * M[11]) + 1.0f * M[15]);
}
static inline uint2 SP_PackHalf4(float4 v) { return make_uint2((f32tof16(v.y) << 16) | f32tof16(v.x), (f32tof16(v.w) << 16) | f32tof16(v.z)); }      // an RGBA16F store (round to nearest even, no clamp)
static inline float4 SP_UnpackHalf4(uint2 v) { return make_float4(f16tof32(v.x & 0xffffu), f16tof32(v.x >> 16), f16tof32(v.y & 0xffffu), f16tof32(v.y >> 16)); }

struct StablePlanesContext {
    StablePlanesBuffers B; StablePlanesConsts C;
    uint PixelToAddress(uint px, uint py, uint planeIndex) const { return GenericTSPixelToAddress(px, py, planeIndex, C.genericTSLineStride, C.genericTSPlaneStride); }
    uint& Hdr(uint px, uint py, uint plane) const { return B.Header[((size_t)plane * C.imageHeight + py) * C.imageWidth + px]; }
    uint GetBranchID(uint px, uint py, uint planeIndex) const { return Hdr(px, py, planeIndex); }
    void SetBranchID(uint px, uint py, uint planeIndex, uint id) const { Hdr(px, py, planeIndex) = id; }
    void StoreStableRadiance(uint px, uint py, float3 radiance) const { B.StableRadiance[(size_t)py * C.imageWidth + px] = SP_PackHalf4(make_float4(clampf(radiance.x, 0, HLF_MAX), clampf(radiance.y, 0, HLF_MAX), clampf(radiance.z, 0, HLF_MAX), 0.f)); }
    void AccumulateStableRadiance(uint px, uint py, float3 radiance) const {
        uint2& t = B.StableRadiance[(size_t)py * C.imageWidth + px]; float4 c = SP_UnpackHalf4(t);
        t = SP_PackHalf4(make_float4(c.x + radiance.x, c.y + radiance.y, c.z + radiance.z, c.w));
    }
    float3 LoadStableRadiance(uint px, uint py) const { return xyz(SP_UnpackHalf4(B.StableRadiance[(size_t)py * C.imageWidth + px])); }
    void StoreFirstHitRayLengthAndClearDominantToZero(uint px, uint py, float length) const { Hdr(px, py, 3) = asuint(fminf_(kMaxRayTravel, length)) & 0xFFFFFFFCu; }
    float LoadFirstHitRayLength(uint px, uint py) const { return asfloat(Hdr(px, py, 3) & 0xFFFFFFFCu); }
    void StoreDominantIndex(uint px, uint py, uint index) const { Hdr(px, py, 3) = (Hdr(px, py, 3) & 0xFFFFFFFCu) | (0x3u & index); }
    uint LoadDominantIndex(uint px, uint py) const { return Hdr(px, py, 3) & 0x3u; }
    void StartPixelBuild(uint px, uint py) const {
        StoreStableRadiance(px, py, make_float3(0.f));
        Hdr(px, py, 0) = cStablePlaneInvalidBranchID; Hdr(px, py, 1) = cStablePlaneInvalidBranchID; Hdr(px, py, 2) = cStablePlaneInvalidBranchID;
    }
    void StoreStablePlane(uint px, uint py, uint planeIndex, uint vertexIndex, float3 rayOrigin, float3 rayDir, uint stableBranchID, float sceneLength, float rayTCurrent, float3 thp, float3 motionVectors,
                          float roughness, float3 worldNormal, float3 diffBSDFEstimate, float3 specBSDFEstimate, bool dominantSP, uint flagsAndVertexIndex, uint packedCounters) const {
        StablePlane sp;
        sp.RayOrigin = rayOrigin; sp.RayDir = rayDir; sp.SceneLength = sceneLength;
        sp.VertexIndexAndRoughness = (vertexIndex << 16) | f32tof16(roughness);
        sp.PackedThpAndMVs[0] = PackTwoFp32ToFp16(thp.x, motionVectors.x); sp.PackedThpAndMVs[1] = PackTwoFp32ToFp16(thp.y, motionVectors.y); sp.PackedThpAndMVs[2] = PackTwoFp32ToFp16(thp.z, motionVectors.z);
        const float kNRDMinReflectance = 0.04f, kNRDMaxReflectance = 6.5504e+4F;
        float3 d = clamp3(diffBSDFEstimate, kNRDMinReflectance, kNRDMaxReflectance), s = clamp3(specBSDFEstimate, kNRDMinReflectance, kNRDMaxReflectance);
        sp.DenoiserPackedBSDFEstimate[0] = PackTwoFp32ToFp16(d.x, s.x); sp.DenoiserPackedBSDFEstimate[1] = PackTwoFp32ToFp16(d.y, s.y); sp.DenoiserPackedBSDFEstimate[2] = PackTwoFp32ToFp16(d.z, s.z);
        sp.PackedNormal = NDirToOctUnorm32(worldNormal);
        sp.PackedNoisyRadianceAndSpecAvg[0] = 0; sp.PackedNoisyRadianceAndSpecAvg[1] = 0;      // Fp32ToFp16(float4(0,0,0,0))
        sp.LastRayTCurrent = rayTCurrent; sp.FlagsAndVertexIndex = flagsAndVertexIndex; sp.PackedCounters = packedCounters;
        B.Planes[PixelToAddress(px, py, planeIndex)] = sp;
        SetBranchID(px, py, planeIndex, stableBranchID);
        if (dominantSP && planeIndex != 0) StoreDominantIndex(px, py, planeIndex);
    }
    // the enqueued delta paths wait in the plane's own 80 bytes (PackCustomPayload / UnpackCustomPayload, PathPayload::pack / unpack in the BUILD layout)
    void StoreExplorationStart(uint px, uint py, uint planeIndex, const PathState& p) const {
        StablePlane sp;
        sp.RayOrigin = p.origin; sp.LastRayTCurrent = asfloat(p.id); sp.RayDir = p.dir; sp.SceneLength = p.sceneLength;
        sp.PackedThpAndMVs[0] = p.pack23[0]; sp.PackedThpAndMVs[1] = p.pack23[1]; sp.PackedThpAndMVs[2] = p.pack45[0]; sp.VertexIndexAndRoughness = p.pack45[1];
        sp.DenoiserPackedBSDFEstimate[0] = p.interiorList.slots[0]; sp.DenoiserPackedBSDFEstimate[1] = p.interiorList.slots[1]; sp.DenoiserPackedBSDFEstimate[2] = p.packedCounters; sp.PackedNormal = p.SP_BRANCH_FIELD;
        sp.PackedNoisyRadianceAndSpecAvg[0] = p.rayCone.widthSpreadAngleFP16; sp.PackedNoisyRadianceAndSpecAvg[1] = p.pack0; sp.FlagsAndVertexIndex = p.pack1; sp.PackedCounters = p.flagsAndVertexIndex;
        B.Planes[PixelToAddress(px, py, planeIndex)] = sp;
        SetBranchID(px, py, planeIndex, cStablePlaneEnqueuedBranchID);
    }
    void ExplorationStart(uint px, uint py, uint planeIndex, PathState& p) const {
        const StablePlane sp = B.Planes[PixelToAddress(px, py, planeIndex)];
        p.origin = sp.RayOrigin; p.id = asuint(sp.LastRayTCurrent); p.dir = sp.RayDir; p.sceneLength = sp.SceneLength;
        p.pack23[0] = sp.PackedThpAndMVs[0]; p.pack23[1] = sp.PackedThpAndMVs[1]; p.pack45[0] = sp.PackedThpAndMVs[2]; p.pack45[1] = sp.VertexIndexAndRoughness;
        p.interiorList.slots[0] = sp.DenoiserPackedBSDFEstimate[0]; p.interiorList.slots[1] = sp.DenoiserPackedBSDFEstimate[1]; p.packedCounters = sp.DenoiserPackedBSDFEstimate[2]; p.SP_BRANCH_FIELD = sp.PackedNormal;
        p.rayCone.widthSpreadAngleFP16 = sp.PackedNoisyRadianceAndSpecAvg[0]; p.pack0 = sp.PackedNoisyRadianceAndSpecAvg[1]; p.pack1 = sp.FlagsAndVertexIndex; p.flagsAndVertexIndex = sp.PackedCounters;
        SetBranchID(px, py, planeIndex, cStablePlaneJustStartedID);
    }
    int FindNextToExplore(uint px, uint py, uint fromPlane) const {
        for (uint i = fromPlane; i < cStablePlaneCount; i++) if (GetBranchID(px, py, i) == cStablePlaneEnqueuedBranchID) return (int)i;
        return -1;
    }
    void GetAvailableEmptyPlanes(uint px, uint py, int& availableCount, int availablePlanes[cStablePlaneCount]) const {
        availableCount = 0;
        const uint n = C.activeStablePlaneCount < cStablePlaneCount ? C.activeStablePlaneCount : cStablePlaneCount;
        for (uint i = 1; i < n; i++) if (GetBranchID(px, py, i) == cStablePlaneInvalidBranchID) availablePlanes[availableCount++] = (int)i;
    }
    // StablePlanesContext::GetAllRadiance (StablePlanes.hlsli:262-275): what PostProcess.hlsl's NO_DENOISER_FINAL_MERGE writes to the output colour — the stable radiance plus every existing plane's noisy radiance
    float3 GetAllRadiance(uint px, uint py) const {
        float3 pathL = LoadStableRadiance(px, py);
        for (uint i = 0; i < cStablePlaneCount; i++) {
            if (GetBranchID(px, py, i) == cStablePlaneInvalidBranchID) continue;
            const StablePlane& rec = B.Planes[PixelToAddress(px, py, i)];
            const float2 a = Fp16ToFp32(rec.PackedNoisyRadianceAndSpecAvg[0]), b = Fp16ToFp32(rec.PackedNoisyRadianceAndSpecAvg[1]);
            pathL = pathL + make_float3(a.x, a.y, b.x);
        }
        return pathL;
    }
    // Bridge::computeMotionVector (BridgeDonut:890-909)
    float3 computeMotionVector(float3 posW, float3 prevPosW) const {
        float4 clipPos = SP_mul_row(posW, C.matWorldToClipNoOffset);
        clipPos.x = clipPos.x / clipPos.w; clipPos.y = clipPos.y / clipPos.w; clipPos.z = clipPos.z / clipPos.w;
        float4 prevClipPos = SP_mul_row(prevPosW, C.prevMatWorldToClipNoOffset);
        prevClipPos.x = prevClipPos.x / prevClipPos.w; prevClipPos.y = prevClipPos.y / prevClipPos.w; prevClipPos.z = prevClipPos.z / prevClipPos.w;
        if (clipPos.w <= 0 || prevClipPos.w <= 0) return make_float3(0, 0, 0);
        float3 motion;
        motion.x = (prevClipPos.x - clipPos.x) * C.clipToWindowScale[0]; motion.y = (prevClipPos.y - clipPos.y) * C.clipToWindowScale[1];
        motion.z = prevClipPos.w - clipPos.w;
        return motion;
    }
    // Bridge::ExportSurfaceInit / ExportSurface / ExportNonSurface (BridgeDonut:1098-1152)
    void ExportSurfaceInit(uint px, uint py) const { const size_t i = (size_t)py * C.imageWidth + px; B.Depth[i] = 0; B.SpecularHitT[i] = 0; }
    void ExportGuides(uint px, uint py, float3 virtualWorldPos, float3 motionVectors, uint throughput) const {
        const size_t i = (size_t)py * C.imageWidth + px;
        B.MotionVectors[i] = SP_PackHalf4(make_float4(motionVectors, 0.f));
        float4 clipPos = SP_mul_row(virtualWorldPos, C.matWorldToClip);
        B.Depth[i] = clipPos.z / clipPos.w;
        B.Throughput[i] = throughput;
    }
};
// what Bridge::loadSurface writes into the material header for the decomposition (BridgeDonut:699-718). Motion-vector block types: 0 Off, 3 Full, and the two automatic ones —
// 1 AutoLow, 2 AutoHigh — which block stochastically where the triangle's curvature (TriangleCurvatureApprox_GradN, pt_path.h / pathtracer.h), seen through the ray cone's width at
// the hit and the angle of incidence, exceeds a threshold jittered per (pixel, vertex, sample) by a MicroRng of its own (BridgeDonut:704-716): no other random stream moves.
struct SPMaterialInfo { bool psdExclude, blockMVs; uint dominantDeltaLobeP1; };
static inline SPMaterialInfo SP_material_info(uint flags, const MVBlockInputs& geo, const RayCone& rayCone, uint pathId, uint pathVertexIndex, uint sampleIndex) {
    SPMaterialInfo m; m.psdExclude = (flags & PTMaterialFlags_PSDExcludeBit) != 0;
    m.dominantDeltaLobeP1 = (flags & PTMaterialFlags_PSDDominantDeltaLobeP1Mask) >> PTMaterialFlags_PSDDominantDeltaLobeP1Shift;
    const int blockType = ((flags & PTMaterialFlags_PSDBlockMVsAtSurfaceTypeB0) != 0 ? 1 : 0) + ((flags & PTMaterialFlags_PSDBlockMVsAtSurfaceTypeB1) != 0 ? 2 : 0);
    m.blockMVs = blockType == 3;
    if (blockType == 1 || blockType == 2) {
        const float pixelCurvature = (geo.curvatureWS * rayCone.getWidth()) / fmaxf_(geo.projectionTerm, 1e-6f);
        const float threshold = (blockType == 1) ? 0.03f : 0.0005f;
        MicroRng rng = MicroRng::make(pathId >> 16, pathId & 0xFFFFu, pathVertexIndex, sampleIndex);
        m.blockMVs |= pixelCurvature > ((rng.NextFloat() * 0.9f + 0.3f) * threshold);
    }
    return m;
}

// ---- the BUILD pass over one path vertex. PT is the path tracer of the side that includes this header (the wavefront kernels' PathKernelContextT<LP16>, or the CPU restatement's PathTracer):
// loadSurface, HandleNestedDielectrics, volumeTransmittance, UpdatePathTravelled, HasFinishedSurfaceBounces and the scene's materials come from it.
template <class PT> struct StablePlanesBuilder {
    typedef typename SPTraits<PT>::LP LP;
    const PT& pt; StablePlanesContext sp; uint sampleIndex;

    void cameraRay(uint px, uint py, float3& o, float3& d) const { SP_camera_ray(pt, px, py, sampleIndex, o, d); }
    // EmptyPathInitialize + SetupPathPrimaryRay + StartPixel (PathTracer.hlsli:47-108, PathTracerSample.hlsl:200-212) in the BUILD layout
    PathState generate(uint px, uint py) const {
        PathState p; __builtin_memset(&p, 0, sizeof(p));
        p.id = (px << 16) | py;
        p.SetThp(make_float3(1.f));
        p.setFlag(PF_active); p.setFlag(PF_deltaOnlyPath, true);
        p.rayCone = RayCone::make(0, pt.cam.PixelConeSpreadAngle);
        SP_SetImageXform(p, make_float3x3(make_float3(1.f, 0.f, 0.f), make_float3(0.f, 1.f, 0.f), make_float3(0.f, 0.f, 1.f)));
        p.setFlag(PF_stablePlaneOnDominantBranch, true);
        SP_SetMotionVectorSceneLength(p, 0);
        SP_setStablePlaneIndex(p, 0);
        p.SP_BRANCH_FIELD = 1;
        if (pt.HasFinishedSurfaceBounces(p.getVertexIndex() + 1, p.getCounter(PC_DiffuseBounces))) p.setFlag(PF_terminateAtNextBounce);
        cameraRay(px, py, p.origin, p.dir);
        sp.StartPixelBuild(px, py);
        sp.ExportSurfaceInit(px, py);
        return p;
    }
    // PathTracerStablePlanes.hlsli:24-101
    PathState SplitDeltaPath(const PathState& oldPath, float3 rayDir, const SurfaceData& surfaceData, const SPMaterialInfo& mi, const DeltaLobe& lobe, uint deltaLobeIndex, bool verifyDominantFlag) const {
        const ShadingData& shadingData = surfaceData.shadingData;
        PathState newPath = oldPath;
        newPath.dir = lobe.dir;
        newPath.SetThp(newPath.GetThp() * lobe.thp);
        newPath.origin = shadingData.computeNewRayOrigin(lobe.transmission == 0);
        newPath.SP_BRANCH_FIELD = StablePlanesAdvanceBranchID(oldPath.SP_BRANCH_FIELD, deltaLobeIndex);
        newPath.setFlag(PF_delta);
        if (!lobe.transmission) newPath.setFlag(PF_specular);
        else {
            newPath.setFlag(PF_transmission);
            if (pt.S.nestedDielectricsQuality > 0 && !shadingData.mtl.isThinSurface()) {
                uint nestedPriority = shadingData.mtl.getNestedPriority();
                newPath.interiorList.handleIntersection(shadingData.materialID, nestedPriority, shadingData.frontFacing);
                newPath.setFlag(PF_insideDielectricVolume, !newPath.interiorList.isEmpty());
            }
        }
        // `if (!newPath.GetMotionVectorSceneLength() != 0)`: (!x) != 0, true while the stored length is still zero
        if (SP_GetMotionVectorSceneLength(newPath) == 0) {
            float3x3 localT;      // lpfloat3x3: kept in float on both sides of the fence (the pin's shim does the same; a half matrix would be rounded into 15-bit octahedral codes right away)
            if (lobe.transmission) localT = MatrixRotateFromTo(lobe.dir, rayDir);
            else {
                const float3x3 toTangent = make_float3x3(shadingData.T, shadingData.B, shadingData.N);
                const float3x3 mirror = make_float3x3(make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, 0, -1));
                localT = mul(mirror, toTangent);
                localT = mul(transpose(toTangent), localT);
            }
            SP_SetImageXform(newPath, mul(SP_GetImageXform(newPath), localT));
        }
        if (verifyDominantFlag && newPath.hasFlag(PF_stablePlaneOnDominantBranch)) {
            int psdDominantDeltaLobeIndex = (int)mi.dominantDeltaLobeP1 - 1;
            if ((int)deltaLobeIndex != psdDominantDeltaLobeIndex) newPath.setFlag(PF_stablePlaneOnDominantBranch, false);
        }
        return newPath;
    }
    // PathTracerStablePlanes.hlsli:104-330 (BUILD)
    void StablePlanesHandleHit(PathState& path, float3 rayOrigin, float3 rayDir, float rayTCurrent, const SurfaceData& surfaceData, const SPMaterialInfo& mi, bool pathStopping, float3 prevPosW) const {      // prevPosW: SurfaceData's member in this mode (PathTracerTypes.hlsli:58)
        const uint vertexIndex = path.getVertexIndex();
        const uint currentSPIndex = SP_getStablePlaneIndex(path);
        const uint px = path.id >> 16, py = path.id & 0xFFFFu;
        if (mi.blockMVs && SP_GetMotionVectorSceneLength(path) == 0) SP_SetMotionVectorSceneLength(path, path.sceneLength);
        if (vertexIndex == 1) sp.StoreFirstHitRayLengthAndClearDominantToZero(px, py, path.sceneLength);
        bool setAsBase = true;
        if ((vertexIndex < sp.C.maxStablePlaneVertexDepth) && !pathStopping) {
            DeltaLobe deltaLobes[cMaxDeltaLobes]; uint deltaLobeCount; float nonDeltaPart;
            StandardBSDF_evalDeltaLobes(surfaceData.bsdf, surfaceData.shadingData, mi.psdExclude, deltaLobes, deltaLobeCount, nonDeltaPart);
            deltaLobeCount = (cMaxDeltaLobes - 1u > deltaLobeCount) ? cMaxDeltaLobes - 1u : deltaLobeCount;      // max(cMaxDeltaLobes-1, deltaLobeCount)
            bool potentiallyVolumeTransmission = false;
            const float nonDeltaIgnoreThreshold = 1e-5f, deltaIgnoreThreshold = 0.001f;
            bool hasNonDeltaLobes = nonDeltaPart > nonDeltaIgnoreThreshold;
            int nonZeroDeltaLobes[cMaxDeltaLobes]; for (uint i = 0; i < cMaxDeltaLobes; i++) nonZeroDeltaLobes[i] = 0;
            int nonZeroDeltaLobeCount = 0;
            for (uint k = 0; k < deltaLobeCount; k++) {
                const DeltaLobe& lobe = deltaLobes[k];
                const float thp = Average(lobe.thp);
                if (thp > deltaIgnoreThreshold) { nonZeroDeltaLobes[nonZeroDeltaLobeCount] = (int)k; nonZeroDeltaLobeCount++; potentiallyVolumeTransmission |= lobe.transmission != 0; }
            }
            if (nonZeroDeltaLobeCount > 0) {
                bool allowPSR = sp.C.allowPrimarySurfaceReplacement && (nonZeroDeltaLobeCount == 1) && (currentSPIndex == 0) && !potentiallyVolumeTransmission;
                allowPSR &= !mi.blockMVs;
                bool canReuseExisting = (currentSPIndex != 0) && (nonZeroDeltaLobeCount > 0);
                canReuseExisting |= allowPSR;
                canReuseExisting &= !hasNonDeltaLobes;
                int availablePlaneCount = 0; int availablePlanes[cStablePlaneCount];
                sp.GetAvailableEmptyPlanes(px, py, availablePlaneCount, availablePlanes);
                canReuseExisting &= (currentSPIndex == 0) || (mi.dominantDeltaLobeP1 > 0);
                const int room = availablePlaneCount + (canReuseExisting ? 1 : 0);
                nonZeroDeltaLobeCount = nonZeroDeltaLobeCount < room ? nonZeroDeltaLobeCount : room;
                int lobeForReuse = -1;
                if (canReuseExisting) { lobeForReuse = nonZeroDeltaLobes[nonZeroDeltaLobeCount - 1]; nonZeroDeltaLobeCount--; }
                for (int i = 0; i < nonZeroDeltaLobeCount; i++) {
                    const int lobeToExplore = nonZeroDeltaLobes[i];
                    PathState splitPath = SplitDeltaPath(path, rayDir, surfaceData, mi, deltaLobes[lobeToExplore], (uint)lobeToExplore, true);
                    SP_setStablePlaneIndex(splitPath, (uint)availablePlanes[i]);
                    sp.StoreExplorationStart(px, py, (uint)availablePlanes[i], splitPath);
                }
                if (lobeForReuse != -1) {
                    setAsBase = false;
                    path = SplitDeltaPath(path, rayDir, surfaceData, mi, deltaLobes[lobeForReuse], (uint)lobeForReuse, nonZeroDeltaLobeCount > 0);
                }
            }
        }
        if (setAsBase) {
            float3 camO, camD; cameraRay(px, py, camO, camD);
            const float3x3 imageXform = SP_GetImageXform(path);
            const bool blockedAtSurface = SP_GetMotionVectorSceneLength(path) != 0;
            float sceneLengthForMVs = blockedAtSurface ? SP_GetMotionVectorSceneLength(path) : path.sceneLength;
            float3 virtualWorldPos = camO + camD * sceneLengthForMVs;
            float3 worldMotion = prevPosW - surfaceData.shadingData.posW;      // PathTracerStablePlanes.hlsli:286; exactly zero where the scene did not move (the same arithmetic on the same operands)
            float3 virtualWorldMotion = mul(imageXform, worldMotion);
            float3 motionVectors = sp.computeMotionVector(virtualWorldPos, virtualWorldPos + virtualWorldMotion);
            float roughness = saturate(surfaceData.bsdf.data.roughness);
            float3 worldNormal = surfaceData.shadingData.N;
            worldNormal = normalize(mul(imageXform, worldNormal));
            float3 diffBSDFEstimate, specBSDFEstimate;
            estimateSpecDiffBSDF<LP>(surfaceData.bsdf.data, diffBSDFEstimate, specBSDFEstimate, surfaceData.shadingData.N, surfaceData.shadingData.V);
            if (blockedAtSurface) roughness *= kSpecularRoughnessThreshold * 0.95f;
            bool isDominant = path.hasFlag(PF_stablePlaneOnDominantBranch);
            sp.StoreStablePlane(px, py, currentSPIndex, vertexIndex, rayOrigin, rayDir, path.SP_BRANCH_FIELD, path.sceneLength, rayTCurrent, path.GetThp(), motionVectors, roughness, worldNormal,
                                diffBSDFEstimate, specBSDFEstimate, isDominant, 0, 0);
            if (isDominant) {       // Bridge::ExportSurface(path, surfaceData, sceneLengthForMVs, motionVectors)
                float3 eo, ed; cameraRay(px, py, eo, ed);
                sp.ExportGuides(px, py, eo + ed * sceneLengthForMVs, motionVectors, Pack_R11G11B10_FLOAT(saturate3(path.GetThp())));
            }
            path.terminate();
        }
    }
    // PathTracerStablePlanes.hlsli:378-414 (BUILD)
    void StablePlanesHandleMiss(PathState& path, float3 emission, float3 rayOrigin, float3 rayDir) const {
        const uint px = path.id >> 16, py = path.id & 0xFFFFu;
        const uint vertexIndex = path.getVertexIndex();
        if (vertexIndex == 1) sp.StoreFirstHitRayLengthAndClearDominantToZero(px, py, kMaxRayTravel);
        float3 camO, camD; cameraRay(px, py, camO, camD);
        const bool blockedAtSurface = SP_GetMotionVectorSceneLength(path) != 0;
        const float sceneLengthForMVs = blockedAtSurface ? SP_GetMotionVectorSceneLength(path) : kEnvironmentMapSceneDistance;
        const float3 virtualWorldPos = camO + camD * sceneLengthForMVs;
        float3 motionVectors = sp.computeMotionVector(virtualWorldPos, virtualWorldPos);
        bool isDominant = path.hasFlag(PF_stablePlaneOnDominantBranch);
        float3 rm = ReinhardMax(emission);
        float3 skyAlbedo = make_float3(sqrtf_(rm.x), sqrtf_(rm.y), sqrtf_(rm.z));
        sp.StoreStablePlane(px, py, SP_getStablePlaneIndex(path), vertexIndex, rayOrigin, rayDir, path.SP_BRANCH_FIELD, blockedAtSurface ? sceneLengthForMVs : __builtin_inff(), 0, path.GetThp(), motionVectors,
                            blockedAtSurface ? 0.1f : 1.f, -rayDir, skyAlbedo, blockedAtSurface ? make_float3(0.5f) : make_float3(0.f), isDominant, 0, 0);
        if (isDominant) sp.ExportGuides(px, py, virtualWorldPos, motionVectors, 0u);      // Bridge::ExportNonSurface
    }
    // HandleMiss in the BUILD configuration (PathTracer.hlsli:407-503: no MIS state, no firefly filter)
    void HandleMiss(PathState& path, float3 rayOrigin, float3 rayDir, float rayTCurrent) const {
        pt.UpdatePathTravelled(path, rayTCurrent);
        float3 environmentEmission = make_float3(0.f);
        if (SP_env_enabled(pt)) {      // `!(misInfo.GetSkipEmissiveBRDF() && ...)`: the packed MIS info reads as empty in this pass
            float mipLevel = (path.getCounter(PC_DiffuseBounces) > 1) ? pt.S.envMapDiffuseSampleMIPLevel : 0.f;
            float3 Le = SP_env_eval(pt, rayDir, mipLevel);
            environmentEmission = LP::r3(1.0f * Le);
        }
        StablePlanesHandleMiss(path, environmentEmission, rayOrigin, rayDir);
        if (any_gt0(environmentEmission)) sp.AccumulateStableRadiance(path.id >> 16, path.id & 0xFFFFu, path.GetThp() * environmentEmission);
        path.setFlag(PF_hit, false);
        path.terminate();
    }
    // HandleHit in the BUILD configuration (PathTracer.hlsli:505-700)
    void HandleHit(PathState& path, float3 rayOrigin, float3 rayDir, uint prim, float rayTCurrent, float bu, float bv) const {
        pt.UpdatePathTravelled(path, rayTCurrent);
        MVBlockInputs mvGeo;
        SurfaceData surfaceData = pt.loadSurface(prim, bu, bv, rayDir, path.rayCone, &mvGeo);
        const SPMaterialInfo mi = SP_material_info(SP_material_flags(pt, surfaceData.shadingData.materialID), mvGeo, path.rayCone, path.id, path.getVertexIndex(), sampleIndex);
        if (pt.S.nestedDielectricsQuality > 0 && !path.interiorList.isEmpty()) {
            const float3 transmittance = pt.volumeTransmittance(path.interiorList.getTopMaterialID(), rayTCurrent);
            path.SetThp(path.GetThp() * transmittance);
        }
        bool rejectedFalseHit = !pt.HandleNestedDielectrics(surfaceData, path);
        if (rejectedFalseHit) {
            // `#if BUILD && !NESTED_DIELECTRICS_AVOID_TERMINATION`: quality 2 can terminate the path in a loop of false hits; the plane is then closed as a miss with no emission
            if (pt.S.nestedDielectricsQuality != 1 && !path.isActive()) StablePlanesHandleMiss(path, make_float3(0.f), rayOrigin, rayDir);
            return;
        }
        const ShadingData& shadingData = surfaceData.shadingData;
        float3 surfaceEmission = make_float3(0.f);
        if (any_gt0(shadingData.emission)) surfaceEmission = LP::r3(shadingData.emission * 1.0f);
        if (surfaceData.neeAnalyticLightIndex != RTXPT_INVALID_LIGHT_INDEX) {      // the mesh that stands in for an analytic light, seen without MIS (bsdfPdf 0)
            float3 add;
            if (SP_analytic_proxy(pt, path.id, surfaceData.neeAnalyticLightIndex, rayOrigin, rayDir, add)) {
                add = LP::r3(add); surfaceEmission = make_float3(LP::add(surfaceEmission.x, add.x), LP::add(surfaceEmission.y, add.y), LP::add(surfaceEmission.z, add.z));
            }
        }
        if (any_gt0(surfaceEmission)) sp.AccumulateStableRadiance(path.id >> 16, path.id & 0xFFFFu, path.GetThp() * surfaceEmission);
        bool pathStopping = path.isTerminatingAtNextBounce();
        StablePlanesHandleHit(path, rayOrigin, rayDir, rayTCurrent, surfaceData, mi, pathStopping, pt.prevPosW(prim, bu, bv));
        if (pathStopping) { path.terminate(); return; }
        path.SetThp(path.GetThp() * make_float3(1.0f));      // UpdatePathThroughput(path, GetThpRuRuCorrection()): 1 in this pass
    }
    // postProcessHit (PathTracerSample.hlsl:96-112): a finished plane hands over to the next enqueued one of its pixel
    void postProcessHit(PathState& path) const {
        const uint px = path.id >> 16, py = path.id & 0xFFFFu; int next;
        if (!path.isActive() && (next = sp.FindNextToExplore(px, py, SP_getStablePlaneIndex(path) + 1u)) != -1) sp.ExplorationStart(px, py, (uint)next, path);
    }
};

// ---- DenoisingGuidesBaker.hlsl:50-113: the 5 x 5 depth-aware fill-in of the specular hit distance (Sample.cpp:2544 runs one ping (main -> scratch) and one pong (scratch -> main) after the noisy passes)
static inline float SpecHitTNeighbourhood(const float* texSrc, const float* depthTex, uint width, uint height, int px, int py) {
    const float centerD = depthTex[(size_t)py * width + px];
    float prevHitT = fmaxf_(0.f, texSrc[(size_t)py * width + px]);
    const float minSpecHitT = 5e-2f;
    if (prevHitT < minSpecHitT) prevHitT = 0;
    const int neigS = 2;
    float vAvg = prevHitT;
    float sumW = prevHitT > 0 ? 1.f : 0.f;
    for (int x = -neigS; x <= +neigS; x++)
        for (int y = -neigS; y <= +neigS; y++) {
            if (x == 0 && y == 0) continue;
            const int nx = px + x, ny = py + y;
            if (nx >= 0 && ny >= 0 && nx < (int)width && ny < (int)height) {
                float v = fminf_(texSrc[(size_t)ny * width + nx], HLF_MAX);
                float d = fmaxf_(0.f, depthTex[(size_t)ny * width + nx]);
                const float depthThreshold = 0.025f;
                float weight = v > 0 ? 1.f : 0.f;
                weight *= (fabsf(d - centerD) <= ((d + centerD) + 1e-5f) * depthThreshold) ? 1.f : 0.f;
                if (weight > 0) { vAvg += v * weight; sumW += weight; }
            }
        }
    if (sumW == 0) return prevHitT;
    vAvg /= sumW;
    return (prevHitT <= 0) ? vAvg : fminf_(prevHitT * 1.5f + 0.5f, vAvg);
}

// ================================================================================================================================================================================
// ---- the FILL passes (PATH_TRACER_MODE_FILL_STABLE_PLANES): the noisy path tracer of the realtime mode. A path starts on plane 0 as the build pass left it (FirstHitFromVBuffer),
// follows the recorded delta tree while its branch id matches (StablePlanesOnScatter), and deposits its radiance — split into a total and a specular average — on the plane it last
// touched (CommitDenoiserRadiance); emission on the stable branches was already collected by the build pass. One sub-sample per call; the planes' noisy radiance accumulates over the
// sub-samples of a frame. NEE with one full sample per vertex (the reference's default); the visibility ray is the caller's: HandleHit returns what a visible light adds (SPNeeRequest).
struct SPNeeRequest { bool valid; float3 origin, dir; float tmax; float4 newL;       // newL: AccumulatePathRadiance's increment of the path's L, noisy-radiance attenuation applied
    // NEE-AT feedback of the visible case (PathTracerNEE.hlsli:266-273; as pt_path.h ShadowRequest): the light (| LFR_SCREEN_SPACE_COHERENT_FLAG) or RTXPT_INVALID_LIGHT_INDEX, its weight, the
    // random number the reservoir draws, and the Russian-roulette outcome of a path whose light was visible where it differs (bit 0: differs, bit 1: terminates, high half: its RR correction)
    uint fbLight; float fbWeight, fbRandom; uint rrFix; };
static inline uint2 SP_Fp32ToFp16(float4 v) { return make_uint2(Fp32ToFp16(make_float2(v.x, v.y)), Fp32ToFp16(make_float2(v.z, v.w))); }      // Packing.hlsli Fp32ToFp16(float4): clamped to +-HLF_MAX
static inline float4 SP_Fp16ToFp32(uint2 v) { float2 a = Fp16ToFp32(v.x), b = Fp16ToFp32(v.y); return make_float4(a.x, a.y, b.x, b.y); }
static inline uint BSDFSample_getDeltaLobeIndex(uint lobe) { if ((lobe & Lobe_Delta) == 0u) return 0xFFFFFFFFu; return (lobe & Lobe_Transmission) == 0u ? 1u : 0u; }      // IBSDF.hlsli:55-60

template <class PT> struct StablePlanesFiller {
    typedef typename SPTraits<PT>::LP LP;
    const PT& pt; StablePlanesContext sp; uint sampleIndex;

    // StablePlanes.hlsli:205-226
    void CommitDenoiserRadiance(PathState& path) const {
        const uint px = path.id >> 16, py = path.id & 0xFFFFu, planeIndex = SP_getStablePlaneIndex(path);
        StablePlane& rec = sp.B.Planes[sp.PixelToAddress(px, py, planeIndex)];
        float4 accumRadiance = path.GetL();
        const uint2 existing = make_uint2(rec.PackedNoisyRadianceAndSpecAvg[0], rec.PackedNoisyRadianceAndSpecAvg[1]);
        if (existing.x != 0 && existing.y != 0) accumRadiance = accumRadiance + SP_Fp16ToFp32(existing);
        const uint2 pk = SP_Fp32ToFp16(accumRadiance);
        rec.PackedNoisyRadianceAndSpecAvg[0] = pk.x; rec.PackedNoisyRadianceAndSpecAvg[1] = pk.y;
        path.SetL(make_float4(0, 0, 0, 0));
    }
    // PathTracer.hlsli:139-166 (FILL)
    void AccumulatePathRadiance(PathState& path, float3 radiance, float specularRadianceAvg, bool stablePlaneOnBranch) const {
        if (stablePlaneOnBranch) return;      // stable radiance: the build pass has it
        float4 newL = make_float4(radiance, specularRadianceAvg) * sp.C.invSubSampleCount;
        path.SetL(path.GetL() + newL);
    }
    // the light sample of `req` turned out visible: its radiance lands and — with temporal feedback on — it is offered to the pixel's feedback reservoir (fbTotalWeight / fbCandidates: one slot per
    // pixel, may be null) and the path continues as the reference's does after drawing that one more random number before the roulette
    static void ApplyVisibleLight(PathState& path, const SPNeeRequest& req, float* fbTotalWeight = nullptr, uint* fbCandidates = nullptr, uint fbWidth = 0) {
        path.SetL(path.GetL() + req.newL);
        if (req.fbLight == RTXPT_INVALID_LIGHT_INDEX) return;
        if (fbTotalWeight) {
            const uint slot = (path.id & 0xFFFFu) * fbWidth + (path.id >> 16);
            LightFeedbackReservoir_Add(fbTotalWeight[slot], fbCandidates[slot], req.fbRandom, req.fbLight & ~LFR_SCREEN_SPACE_COHERENT_FLAG, req.fbWeight, (req.fbLight & LFR_SCREEN_SPACE_COHERENT_FLAG) != 0u);
        }
        if (req.rrFix & 1u) {
            const uint bit = (uint)PF_terminateAtNextBounce << kVertexIndexBitCount;
            if (req.rrFix & 2u) path.flagsAndVertexIndex |= bit; else { path.flagsAndVertexIndex &= ~bit; path.pack1 = (path.pack1 & 0xFFFF0000u) | (req.rrFix >> 16); }
        }
    }
    void ExportSpecHitTStart(const PathState& path) const { sp.B.SpecularHitT[(size_t)(path.id & 0xFFFFu) * sp.C.imageWidth + (path.id >> 16)] = -path.sceneLength; }
    void ExportSpecHitTStop(const PathState& path) const {
        float& t = sp.B.SpecularHitT[(size_t)(path.id & 0xFFFFu) * sp.C.imageWidth + (path.id >> 16)];
        const float denoisingSceneLength = t;
        if (denoisingSceneLength < 0) t = fmaxf_(0.f, path.sceneLength + denoisingSceneLength);
    }
    // PathTracerStablePlanes.hlsli:332-376
    void StablePlanesOnScatter(PathState& path, uint bsLobe) const {
        const uint px = path.id >> 16, py = path.id & 0xFFFFu;
        const bool wasOnStablePlane = path.hasFlag(PF_stablePlaneOnPlane);
        if (wasOnStablePlane) path.setFlag(PF_stablePlaneBaseScatterDiff, (bsLobe & Lobe_Diffuse) != 0);
        path.setFlag(PF_stablePlaneOnPlane, false);
        const uint nextVertexIndex = path.getVertexIndex() + 1u;
        if (path.hasFlag(PF_stablePlaneOnBranch) && nextVertexIndex <= cStablePlaneMaxVertexIndex) {
            path.SP_BRANCH_FIELD = StablePlanesAdvanceBranchID(path.SP_BRANCH_FIELD, BSDFSample_getDeltaLobeIndex(bsLobe));
            bool onStablePath = false;
            for (uint spi = 0; spi < cStablePlaneCount; spi++) {
                const uint planeBranchID = sp.GetBranchID(px, py, spi);
                if (planeBranchID == cStablePlaneInvalidBranchID) continue;
                if (StablePlaneIsOnPlane(planeBranchID, path.SP_BRANCH_FIELD)) {
                    CommitDenoiserRadiance(path);
                    SP_setStablePlaneIndex(path, spi);
                    path.setFlag(PF_stablePlaneOnDominantBranch, spi == sp.LoadDominantIndex(px, py));
                    path.setFlag(PF_stablePlaneOnPlane, true);
                    SP_setCounter(path, SP_PC_BouncesFromStablePlane, 0);
                    onStablePath = true;
                    break;
                }
                onStablePath |= StablePlaneIsOnStablePath(planeBranchID, StablePlanesVertexIndexFromBranchID(planeBranchID), path.SP_BRANCH_FIELD, nextVertexIndex);
            }
            path.setFlag(PF_stablePlaneOnBranch, onStablePath);
        } else {
            path.SP_BRANCH_FIELD = cStablePlaneInvalidBranchID;
            path.setFlag(PF_stablePlaneOnBranch, false);
            path.incrementCounter(SP_PC_BouncesFromStablePlane);
        }
        if (!path.hasFlag(PF_stablePlaneOnPlane)) path.incrementCounter(SP_PC_BouncesFromStablePlane);
    }
    // HandleMiss (PathTracer.hlsli:407-503, FILL)
    void HandleMiss(PathState& path, float3 rayDir, float rayTCurrent) const {
        pt.UpdatePathTravelled(path, rayTCurrent);
        if (path.hasFlag(PF_exportSpecHitTQueued)) { ExportSpecHitTStop(path); path.setFlag(PF_exportSpecHitTQueued, false); }
        float3 environmentEmission = make_float3(0.f);
        NEEBSDFMISInfo misInfo = NEEBSDFMISInfo::Unpack16bit(path.GetPackedMISInfo());
        if (SP_env_enabled(pt)) {
            float mipLevel = (path.getCounter(PC_DiffuseBounces) > 1) ? pt.S.envMapDiffuseSampleMIPLevel : 0.f;
            float3 localDir = SP_env_to_local(pt, rayDir);
            float3 Le = SP_env_eval_local(pt, localDir, mipLevel);
            float misWeight = 1.0f;
            float bsdfScatterPdf = path.GetBsdfScatterPdf();
            if (misInfo.LightSamplingEnabled && bsdfScatterPdf != 0) {
                const LightSampler lightSampler = SP_light_sampler(pt, path.id, misInfo.LightSamplingIsSSC);
                uint environmentQuadLightIndex = lightSampler.LookupEnvLightByDirection(localDir);
                misWeight = lightSampler.ComputeBSDFMISForEnvironmentQuad(environmentQuadLightIndex, bsdfScatterPdf, misInfo.CandidateSamples, misInfo.FullSamples);
            }
            environmentEmission = LP::r3(misWeight * Le);
        }
        const float baseFFThreshold = LP::r(pt.S.fireflyFilterThreshold);
        if (baseFFThreshold != 0) environmentEmission = SP_firefly_filter(pt, environmentEmission, baseFFThreshold, path.GetFireflyFilterK());
        if (any_gt0(environmentEmission)) {
            float3 radiance = path.GetThp() * environmentEmission;
            float specRadianceAvg = path.hasFlag(PF_stablePlaneBaseScatterDiff) ? 0.0f : Average(radiance);
            AccumulatePathRadiance(path, radiance, specRadianceAvg, path.hasFlag(PF_stablePlaneOnBranch));
        }
        path.setFlag(PF_hit, false);
        path.terminate();
    }
    // EmptyPathInitialize + SetupPathPrimaryRay + FirstHitFromVBuffer(path, 0) (PathTracer.hlsli:47-91, PathTracerSample.hlsl:33-94, 206-218); `active` afterwards: the first ray to trace
    PathState generate(uint px, uint py) const {
        PathState path; __builtin_memset(&path, 0, sizeof(path));
        path.id = (px << 16) | py;
        path.SetThp(make_float3(1.f));
        path.setFlag(PF_active); path.setFlag(PF_deltaOnlyPath, true);
        path.rayCone = RayCone::make(0, pt.cam.PixelConeSpreadAngle);
        path.SetL(make_float4(0, 0, 0, 0));
        path.SetFireflyFilterK_BsdfScatterPdf(1.0f, 0.0f);
        path.SetPackedMISInfo_ThpRuRuCorrection(NEEBSDFMISInfo::empty().Pack16bit(), 1.0f);
        SP_setStablePlaneIndex(path, 0);
        path.SP_BRANCH_FIELD = 1;
        if (pt.HasFinishedSurfaceBounces(path.getVertexIndex() + 1, path.getCounter(PC_DiffuseBounces))) path.setFlag(PF_terminateAtNextBounce);
        SP_camera_ray(pt, px, py, sampleIndex, path.origin, path.dir);
        // FirstHitFromVBuffer: the narrowed ray interval [0.99, 1.01] x LastRayTCurrent is a performance hint — the same ray finds the same closest hit over the full interval
        const StablePlane& rec = sp.B.Planes[sp.PixelToAddress(px, py, 0)];
        const uint stableBranchID = sp.GetBranchID(px, py, 0);
        float sceneLength = rec.SceneLength; const float lastRayTCurrent = rec.LastRayTCurrent;
        const uint vertexIndex = rec.VertexIndexAndRoughness >> 16;
        const float3 thp = make_float3(f16tof32(rec.PackedThpAndMVs[0] >> 16), f16tof32(rec.PackedThpAndMVs[1] >> 16), f16tof32(rec.PackedThpAndMVs[2] >> 16));
        bool isMiss = false;
        if (!SP_isfinite(sceneLength)) { sceneLength = kMaxRayTravel; isMiss = true; }
        else sceneLength -= lastRayTCurrent;
        SP_setVertexIndex(path, vertexIndex - 1u);
        path.dir = rec.RayDir; path.origin = rec.RayOrigin;
        path.setFlag(PF_stablePlaneOnPlane, true); path.setFlag(PF_stablePlaneOnBranch, true);
        SP_setStablePlaneIndex(path, 0);
        path.SP_BRANCH_FIELD = stableBranchID;
        path.SetThp(thp);
        path.SetL(make_float4(0, 0, 0, 0));
        path.setFlag(PF_stablePlaneOnDominantBranch, sp.LoadDominantIndex(px, py) == 0u);
        SP_setCounter(path, SP_PC_BouncesFromStablePlane, 0);
        if (pt.HasFinishedSurfaceBounces(path.getVertexIndex() + 1, path.getCounter(PC_DiffuseBounces))) path.setFlag(PF_terminateAtNextBounce);
        path.rayCone = path.rayCone.propagateDistance(sceneLength);                   // UpdatePathTravelledLengthOnly
        path.sceneLength = fminf_(path.sceneLength + sceneLength, kMaxRayTravel);
        if (isMiss) HandleMiss(path, path.dir, sceneLength);
        return path;
    }
    // FirstHitFromVBuffer's ray interval (PathTracerSample.hlsl:38, 56-58): [0.99, 1.01] x LastRayTCurrent when plane 0 holds a surface, the whole ray otherwise. A performance hint, as
    // the reference says: the build pass found the closest hit of this very ray over the whole interval at LastRayTCurrent, so the narrowed ray finds the same one. The device's first
    // traversal launch of the pass uses it (k_extend<., RANGED>); the oracle traces the whole interval.
    void firstHitInterval(uint px, uint py, float& tmin, float& tmax) const {
        const StablePlane& rec = sp.B.Planes[sp.PixelToAddress(px, py, 0)];
        tmin = 0.f; tmax = kMaxRayTravel;
        if (SP_isfinite(rec.SceneLength)) { tmin = rec.LastRayTCurrent * 0.99f; tmax = rec.LastRayTCurrent * 1.01f; }
    }
    // GenerateScatterRay (PathTracer.hlsli:217-380, FILL)
    bool GenerateScatterRay(const ShadingData& sd, const StandardBSDF& bsdf, bool blockMVs, PathState& path, const SampleGeneratorVertexBase& sgBase) const {
        float4 u;
        if (pt.S.enableLDSamplerForBSDF && path.getCounter(PC_DiffuseBounces) < kDisableLowDiscrepancySamplingAfterDiffuseBounceCount) u = SampleSequenceGenerator::Generate(3, sgBase, SGES_ScatterBSDF);
        else u = UniformSampleSequenceGenerator::Generate(3, sgBase, SGES_ScatterBSDF);
        BSDFSample bs;
        bool valid = bsdf.sample(sd, u, bs);
        if (!valid) return false;
        path.dir = bs.wo;
        const bool onDominantDenoisingLayer = path.hasFlag(PF_stablePlaneOnPlane) && path.hasFlag(PF_stablePlaneOnDominantBranch);
        path.SetThp(path.GetThp() * bs.weight);
        path.clearScatterEventFlags();
        path.origin = sd.computeNewRayOrigin(bs.isLobe(Lobe_Reflection));
        const float roughness = bsdf.data.roughness;
        bool isDiffuse = bs.isLobe(Lobe_DiffuseReflection) || bs.isLobe(Lobe_DiffuseTransmission) || roughness > kSpecularRoughnessThreshold;
        if (isDiffuse) { if (!(bs.isLobe(Lobe_DiffuseTransmission) && ((path.getVertexIndex() % 2) == 1))) path.incrementCounter(PC_DiffuseBounces); }
        else path.setFlag(PF_specular);
        if (bs.isLobe(Lobe_Transmission)) {
            path.setFlag(PF_transmission);
            if (pt.S.nestedDielectricsQuality > 0 && !sd.mtl.isThinSurface()) {
                path.interiorList.handleIntersection(sd.materialID, sd.mtl.getNestedPriority(), sd.frontFacing);
                path.setFlag(PF_insideDielectricVolume, !path.interiorList.isEmpty());
            }
        }
        if (bs.isLobe(Lobe_Delta)) path.setFlag(PF_delta);
        else {
            path.setFlag(PF_deltaOnlyPath, false);
            path.rayCone = RayCone::make(path.rayCone.getWidth(), fminf_(path.rayCone.getSpreadAngle() + SP_ray_cone_expansion(pt, bs.pdf), 2.0f * K_PI));
        }
        const float kSpecularRoughnessThresholdForHitT = 0.35f;
        const bool isDiffuseForSpecHitT = bs.isLobe(Lobe_DiffuseReflection) || bs.isLobe(Lobe_DiffuseTransmission) || roughness > kSpecularRoughnessThresholdForHitT;
        if (onDominantDenoisingLayer && !isDiffuseForSpecHitT) {
            if (!blockMVs) { path.setFlag(PF_exportSpecHitTQueued, true); ExportSpecHitTStart(path); }
        } else if (path.hasFlag(PF_exportSpecHitTQueued)) {
            const bool hasNonDeltaLobes = (bsdf.getLobes() & Lobe_NonDelta) != 0;
            const uint maxHitTSpecBounces = 4;
            if (hasNonDeltaLobes || path.getCounter(SP_PC_BouncesFromStablePlane) > maxHitTSpecBounces) { ExportSpecHitTStop(path); path.setFlag(PF_exportSpecHitTQueued, false); }
        }
        float fireflyFilterK = SP_new_scatter_ffk(pt, path.GetFireflyFilterK(), bs.pdf, bs.lobeP);
        path.SetFireflyFilterK_BsdfScatterPdf(fireflyFilterK, bs.pdf);
        StablePlanesOnScatter(path, bs.lobe);
        path.setFlag(PF_enableThreadReorder, true);
        return true;
    }
    // HandleNEE + ProcessLightSample up to the visibility ray (PathTracerNEE.hlsli:185-275, 303-346), one full sample; returns the packed NEEBSDFMISInfo of the vertex
    uint HandleNEE(const PathState& pre, const ShadingData& sd, const StandardBSDF& bsdf, UniformSampleSequenceGenerator& sg, SPNeeRequest& req, float4& neeRadianceAndSpecAvg) const {
        req.valid = false; req.fbLight = RTXPT_INVALID_LIGHT_INDEX; req.fbWeight = req.fbRandom = 0.f; req.rrFix = 0u; neeRadianceAndSpecAvg = make_float4(0, 0, 0, 0);
        const LightSampler lightSampler = SP_light_sampler(pt, pre.id, SP_ssc_heuristic(pt, pre.rayCone.getWidth(), pre.sceneLength));
        const uint fullSamples = pt.S.NEEFullSamples < 63u ? pt.S.NEEFullSamples : 63u;
        const bool hasNonDeltaLobes = (bsdf.getLobes() & Lobe_NonDelta) != 0;
        const bool applyNEE = hasNonDeltaLobes && !lightSampler.IsEmpty() && fullSamples > 0;
        if (!applyNEE) return NEEBSDFMISInfo::empty().Pack16bit();
        const uint candidateSampleCount = pt.S.NEECandidateSamples;
        NEEBSDFMISInfo info; info.LightSamplingEnabled = true; info.LightSamplingIsSSC = lightSampler.IsScreenSpaceCoherent; info.CandidateSamples = candidateSampleCount; info.FullSamples = fullSamples;
        LightSample ls = pt.GenerateLightSample(lightSampler, sd, bsdf, candidateSampleCount, sg);
        if (ls.Valid()) {
            float faceSide = dot(sd.N, ls.Direction) >= 0 ? 1.f : -1.f;
            req.origin = ComputeRayOrigin(sd.posW, sd.faceNCorrected * faceSide); req.dir = ls.Direction; req.tmax = ls.Distance * 0.9985f; req.valid = true;
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
            float specAvg = bsdfThp.w * Average(Li);
            if (pt.S.fireflyFilterThreshold != 0) {
                float pdf = ls.SelectionPdf * ls.SolidAnglePdf;
                float k = SP_new_scatter_ffk(pt, pre.GetFireflyFilterK(), pdf, 1.0f);
                radiance = radiance * FireflyFilterShort(radianceAvg, pt.S.fireflyFilterThreshold, k);
            }
            float3 preThp = pre.GetThp();
            radiance = radiance * preThp;
            specAvg *= Average(preThp);
            if (ls.LightIndex != RTXPT_INVALID_LIGHT_INDEX && lightSampler.IsTemporalFeedbackRequired()) {      // the reservoir update of the visible case (:266-273; radianceAvg is the value before the firefly filter)
                req.fbLight = ls.LightIndex | (lightSampler.IsScreenSpaceCoherent ? LFR_SCREEN_SPACE_COHERENT_FLAG : 0u);
                req.fbWeight = lightSampler.FeedbackWeightFromNEE(ls.LightIndex, radianceAvg * Average(preThp));
                UniformSampleSequenceGenerator after = sg; req.fbRandom = sampleNext1D(after);
            }
            neeRadianceAndSpecAvg = SP_Fp16ToFp32(SP_Fp32ToFp16(make_float4(0, 0, 0, 0) + make_float4(radiance, specAvg)));      // NEEResult::AccumulateRadiance on the empty result, then GetRadianceAndSpecAvg
        }
        return info.Pack16bit();
    }
    // HandleHit (PathTracer.hlsli:505-762, FILL). req.valid afterwards: trace req's visibility ray and call ApplyVisibleLight if nothing is hit.
    void HandleHit(PathState& path, float3 rayOrigin, float3 rayDir, uint prim, float rayTCurrent, float bu, float bv, SPNeeRequest& req) const {
        req.valid = false;
        pt.UpdatePathTravelled(path, rayTCurrent);
        MVBlockInputs mvGeo;
        SurfaceData surfaceData = pt.loadSurface(prim, bu, bv, rayDir, path.rayCone, &mvGeo);
        const SPMaterialInfo mi = SP_material_info(SP_material_flags(pt, surfaceData.shadingData.materialID), mvGeo, path.rayCone, path.id, path.getVertexIndex(), sampleIndex);
        if (pt.S.nestedDielectricsQuality > 0 && !path.interiorList.isEmpty()) {
            const float3 transmittance = pt.volumeTransmittance(path.interiorList.getTopMaterialID(), rayTCurrent);
            path.SetThp(path.GetThp() * transmittance);
        }
        if (!pt.HandleNestedDielectrics(surfaceData, path)) return;
        const ShadingData& sd = surfaceData.shadingData; const StandardBSDF& bsdf = surfaceData.bsdf;
        float3 surfaceEmission = make_float3(0.f);
        NEEBSDFMISInfo misInfo = NEEBSDFMISInfo::Unpack16bit(path.GetPackedMISInfo());
        if (any_gt0(sd.emission)) {
            float misWeight = 1.0f;
            float bsdfScatterPdf = path.GetBsdfScatterPdf();
            if (misInfo.LightSamplingEnabled && bsdfScatterPdf != 0) {
                const LightSampler lightSampler = SP_light_sampler(pt, path.id, misInfo.LightSamplingIsSSC);
                misWeight = lightSampler.ComputeBSDFMISForEmissiveTriangle(surfaceData.neeTriangleLightIndex, bsdfScatterPdf, rayOrigin, sd.posW, misInfo.CandidateSamples, misInfo.FullSamples);
            }
            surfaceEmission = LP::r3(sd.emission * misWeight);
        }
        if (surfaceData.neeAnalyticLightIndex != RTXPT_INVALID_LIGHT_INDEX) {
            const LightSampler lightSampler = SP_light_sampler(pt, path.id, misInfo.LightSamplingIsSSC);
            const float bsdfPdf = misInfo.LightSamplingEnabled ? LP::r(path.GetBsdfScatterPdf()) : 0.0f; float3 add;
            if (lightSampler.ComputeAnalyticLightProxyContribution(surfaceData.neeAnalyticLightIndex, bsdfPdf, rayOrigin, rayDir, misInfo.CandidateSamples, misInfo.FullSamples, add)) {
                add = LP::r3(add); surfaceEmission = make_float3(LP::add(surfaceEmission.x, add.x), LP::add(surfaceEmission.y, add.y), LP::add(surfaceEmission.z, add.z));
            }
        }
        if (any_gt0(surfaceEmission)) {
            const float baseFFThreshold = LP::r(pt.S.fireflyFilterThreshold);
            if (baseFFThreshold != 0) surfaceEmission = SP_firefly_filter(pt, surfaceEmission, baseFFThreshold, path.GetFireflyFilterK());
            if (any_gt0(surfaceEmission)) {
                float3 radiance = path.GetThp() * surfaceEmission;
                float specRadianceAvg = path.hasFlag(PF_stablePlaneBaseScatterDiff) ? 0.0f : Average(radiance);
                AccumulatePathRadiance(path, radiance, specRadianceAvg, path.hasFlag(PF_stablePlaneOnBranch));
            }
        }
        if (path.isTerminatingAtNextBounce()) { path.terminate(); return; }      // (StablePlanesHandleHit has no body in this pass)
        const float rr = path.GetThpRuRuCorrection();
        path.SetThp(path.GetThp() * make_float3(rr));
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make(path.id, path.getVertexIndex(), sampleIndex);
        UniformSampleSequenceGenerator uniformSG = UniformSampleSequenceGenerator::make(vb, SGES_Base);
        const PathState preScatterPath = path;
        bool scatterValid = GenerateScatterRay(sd, bsdf, mi.blockMVs, path, vb);
        float4 neeRadianceAndSpecAvg = make_float4(0, 0, 0, 0);
        uint packedMIS = NEEBSDFMISInfo::empty().Pack16bit();
        if (pt.S.NEEEnabled) packedMIS = HandleNEE(preScatterPath, sd, bsdf, uniformSG, req, neeRadianceAndSpecAvg);
        path.SetPackedMISInfo_ThpRuRuCorrection(packedMIS, path.GetThpRuRuCorrection());
        if (neeRadianceAndSpecAvg.x > 0 || neeRadianceAndSpecAvg.y > 0 || neeRadianceAndSpecAvg.z > 0 || neeRadianceAndSpecAvg.w > 0) {
            const int bouncesFromStablePlane = (int)preScatterPath.getCounter(SP_PC_BouncesFromStablePlane) + 1;
            float3 radiance = xyz(neeRadianceAndSpecAvg);
            float specRadianceAvg = 0;
            if (!preScatterPath.hasFlag(PF_stablePlaneBaseScatterDiff)) {
                bool pathIsDeltaOnlyPath = preScatterPath.hasFlag(PF_deltaOnlyPath);
                bool specialCondition = (bouncesFromStablePlane == 1) || (pathIsDeltaOnlyPath && bouncesFromStablePlane <= 3);
                specRadianceAvg = specialCondition ? neeRadianceAndSpecAvg.w : Average(xyz(neeRadianceAndSpecAvg));
            }
            req.newL = make_float4(radiance, specRadianceAvg) * sp.C.invSubSampleCount;      // AccumulatePathRadiance(path, radiance, specRadianceAvg, false) once the light is known to be visible
        } else req.newL = make_float4(0, 0, 0, 0);      // (the visibility ray is traced whatever the sample is worth: the ray counts stay the reference's)
        if (!scatterValid) path.terminate();
        bool shouldTerminate = pt.HasFinishedSurfaceBounces(path.getVertexIndex() + 1, path.getCounter(PC_DiffuseBounces));
        if (req.fbLight != RTXPT_INVALID_LIGHT_INDEX) {       // feedback pending on the visibility test: the visible case has drawn one more number before the roulette (as pt_path.h HandleHit)
            UniformSampleSequenceGenerator sgVisible = uniformSG; (void)sampleNext1D(sgVisible);
            PathState visiblePath = path;
            const bool terminateVisible = shouldTerminate | pt.HandleRussianRoulette(visiblePath, sgVisible);
            shouldTerminate |= pt.HandleRussianRoulette(path, uniformSG);
            req.rrFix = (terminateVisible != shouldTerminate ? 1u : 0u) | (terminateVisible ? 2u : 0u) | ((visiblePath.pack1 & 0xFFFFu) << 16);
        } else shouldTerminate |= pt.HandleRussianRoulette(path, uniformSG);
        if (shouldTerminate) path.setFlag(PF_terminateAtNextBounce);
    }
};
