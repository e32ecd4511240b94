// ORACLE pin (test infrastructure only): what PathTracerBridgeDonut.hlsli takes from outside the reference tree, so that ITS OWN text
// (getGeometryFromHit, sampleGeometryMaterialRTXPT, EvaluateSceneMaterialRTXPT, ApplyNormalMapRTXPT, Bridge::loadSurface, createTextureSampler,
// computeCameraRay, loadIoR, loadHomogeneousVolumeData, CreateLightSampler ...) can be compiled and run:
//   * the Donut side (NVIDIA-RTX/Donut, an un-vendored submodule: donut/shaders/bindless.h, utils.hlsli, scene_material.hlsli) — data layouts and
//     helpers restated from their use in the reference; these few lines are NOT pinned, everything that calls them is;
//   * the resource bindings (Bindings/*.hlsli) as globals the driver fills from the oracle's scene.
// Included inside namespace hl, after the integrator text.
#define ENABLE_METAL_ROUGH_RECONSTRUCTION 1      // PathTracerBridgeDonut.hlsli:15
struct InstanceData { float3x4 transform; float3x4 prevTransform; uint firstGeometryIndex, firstGeometryInstanceIndex, numGeometries, flags; };   // BridgeDonut:166-168, 628-631, 674
struct GeometryData { uint indexBufferIndex, vertexBufferIndex, indexOffset, positionOffset, prevPositionOffset, texCoord1Offset, normalOffset, tangentOffset, materialIndex; };   // BridgeDonut:170-243
struct GeometryDebugData { int ommIndexBufferIndex; uint ommIndexBufferOffset; };
static const uint c_SizeOfTriangleIndices = 12, c_SizeOfPosition = 12, c_SizeOfTexcoord = 8, c_SizeOfNormal = 4;
static const float3 c_DielectricSpecular = float3(0.04f, 0.04f, 0.04f);
static const uint MaterialFlags_UseBaseOrDiffuseTexture = PTMaterialFlags_UseBaseOrDiffuseTexture;
struct MaterialTextureSample { float4 baseOrDiffuse, metalRoughOrSpecular, normal, emissive, occlusion, transmission; };
static inline MaterialTextureSample DefaultMaterialTextures() { MaterialTextureSample t; t.baseOrDiffuse = float4(1, 1, 1, 1); t.metalRoughOrSpecular = float4(1, 1, 1, 1); t.normal = float4(0.5f, 0.5f, 1.0f, 0.f);
    t.emissive = float4(1, 1, 1, 1); t.occlusion = float4(1, 1, 1, 1); t.transmission = float4(1, 1, 1, 1); return t; }
// Donut's ConvertSpecularGlossToMetalRough: NOT pinned (its text is outside the tree) — the Khronos KHR_materials_pbrSpecularGlossiness conversion sample it follows, restated;
// what IS pinned through it is the reference's own call site (EvaluateSceneMaterialRTXPT's spec-gloss branch, the lp conversions of its out arguments, loadSurface downstream).
static inline float GetPerceivedBrightness(float3 c) { return sqrt((0.299f * c.x * c.x + 0.587f * c.y * c.y) + 0.114f * c.z * c.z); }
template <class C3, class C1> static inline void ConvertSpecularGlossToMetalRough(float3 diffuseColor, float3 specularColor, C3& baseColor, C1& metalness) {
    const float epsilon = 1e-6f, dielectricSpecular = 0.04f;
    float diffuseBrightness = GetPerceivedBrightness(diffuseColor), specularBrightness = GetPerceivedBrightness(specularColor);
    float oneMinusSpecularStrength = 1.0f - max(specularColor.x, max(specularColor.y, specularColor.z));
    float m = 0.0f;
    if (!(specularBrightness < dielectricSpecular)) {
        float b = (diffuseBrightness * oneMinusSpecularStrength / (1.0f - dielectricSpecular) + specularBrightness) - 2.0f * dielectricSpecular;
        float c = dielectricSpecular - specularBrightness;
        float D = max(b * b - 4.0f * dielectricSpecular * c, 0.0f);
        m = saturate((-b + sqrt(D)) / (2.0f * dielectricSpecular));
    }
    float3 fromDiffuse = diffuseColor * (oneMinusSpecularStrength / (1.0f - dielectricSpecular) / max(1.0f - m, epsilon));
    float3 fromSpecular = (specularColor - float3(dielectricSpecular * (1.0f - m))) * (1.0f / max(m, epsilon));
    baseColor = C3(saturate(lerp(fromDiffuse, fromSpecular, m * m))); metalness = C1(m);
}
static inline float3 interpolate(const float3 v[3], float3 b) { return (v[0] * b.x + v[1] * b.y) + v[2] * b.z; }
static inline float2 interpolate(const float2 v[3], float3 b) { return (v[0] * b.x + v[1] * b.y) + v[2] * b.z; }
static inline float4 interpolate(const float4 v[3], float3 b) { return (v[0] * b.x + v[1] * b.y) + v[2] * b.z; }
static inline float square(float v) { return v * v; }
static inline uint NonUniformResourceIndex(uint i) { return i; }
static inline float4 mul(float3x4 M, float4 v) { return float4(dot(M.r[0], v), dot(M.r[1], v), dot(M.r[2], v), 0.f); }      // `mul(transform, float4(p, w)).xyz`; dot4 = ((x+y)+z)+w below
struct OpacityMicroMapDebugInfo { bool hasOmmAttachment; float3 opacityStateDebugColor; static OpacityMicroMapDebugInfo initDefault() { OpacityMicroMapDebugInfo d; d.hasOmmAttachment = false; d.opacityStateDebugColor = float3(0, 0, 0); return d; } };
struct DonutGeometrySample;
static OpacityMicroMapDebugInfo loadOmmDebugInfo(const DonutGeometrySample&, uint, float2) { return OpacityMicroMapDebugInfo::initDefault(); }
static void surfaceDebugViz(uint2, PathTracer::SurfaceData, float2, float3, RayCone, int, OpacityMicroMapDebugInfo, uint, DebugContext) {}
// ---- bindings (Bindings/SceneBindings.hlsli, LightingBindings.hlsli, SamplerBindings.hlsli, ShaderResourceBindings.hlsli), filled by the driver per frame
struct BindlessBuffers { const ByteAddressBuffer* b = nullptr; ByteAddressBuffer operator[](uint i) const { return b[i]; } };
struct BindlessTextures { const Texture2D<float4>* t = nullptr; Texture2D<float4> operator[](uint i) const { return t[i]; } };
struct PlanarViewPin { float4x4 matWorldToClip, matWorldToClipNoOffset; float2 clipToWindowScale; };      // donut PlanarViewConstants: the members the guide-buffer dump and Bridge::computeMotionVector read
typedef PlanarViewPin SimpleViewConstants;
static inline float4 mul(float4 v, float4x4 M) { return float4(((v.x * M.r[0].x + v.y * M.r[1].x) + v.z * M.r[2].x) + v.w * M.r[3].x, ((v.x * M.r[0].y + v.y * M.r[1].y) + v.z * M.r[2].y) + v.w * M.r[3].y,
    ((v.x * M.r[0].z + v.y * M.r[1].z) + v.z * M.r[2].z) + v.w * M.r[3].z, ((v.x * M.r[0].w + v.y * M.r[1].w) + v.z * M.r[2].w) + v.w * M.r[3].w); }
struct SampleConstantsPin { PlanarViewPin view, previousView; PathTracerConstants ptConsts; EnvMapSceneParams envMapSceneParams; EnvMapImportanceSamplingParams envMapImportanceSamplingParams; uint MaterialCount; };
struct SampleMiniConstantsPin { uint4 params; };
static SampleConstantsPin g_Const; static SampleMiniConstantsPin g_MiniConst;
static StructuredBuffer<InstanceData> t_InstanceData; static StructuredBuffer<GeometryData> t_GeometryData; static StructuredBuffer<GeometryDebugData> t_GeometryDebugData;
static StructuredBuffer<SubInstanceData> t_SubInstanceData; static StructuredBuffer<PTMaterialData> t_PTMaterialData;
static BindlessBuffers t_BindlessBuffers; static BindlessTextures t_BindlessTextures; static SamplerState s_MaterialSampler, s_EnvironmentMapSampler, s_EnvironmentMapImportanceSampler;
static TextureCube<float4> t_EnvironmentMap; static Texture2D<float> t_EnvironmentMapImportanceMap;
static StructuredBuffer<LightingControlData> t_LightsCB; static StructuredBuffer<PolymorphicLightInfo> t_Lights; static StructuredBuffer<PolymorphicLightInfoEx> t_LightsEx;
static Buffer<uint> t_LightProxyCounters, t_LightProxyIndices, t_LightLocalSamplingBuffer; static Texture2D<uint> t_EnvLookupMap;
static RWTexture2D<float> u_LightFeedbackTotalWeight; static RWTexture2D<uint> u_LightFeedbackCandidates;
static RWTexture2D<float> u_Depth, u_SpecularHitT; static RWTexture2D<float4> u_MotionVectors; static RWTexture2D<uint> u_Throughput;      // the guide buffers (only u_Depth is backed by memory, when a world-to-clip matrix was set)
