/* mi355pt — C-ABI of the MI355X-native wavefront path tracer (libmi355pt.so).
 *
 * Drop-in boundary for ONE hot path of NVIDIA-RTX/RTXPT v1.8.1: the reference-mode ray-tracing mega-pass
 * (Rtxpt/Shaders/PathTracerSample.hlsl and the Rtxpt/Shaders/PathTracer/ directory) together with the driver-side BVH build/refit it
 * depends on. The entry points mirror the seam `Sample` exposes to Donut's render loop:
 * every function returns an int32 status (PT_OK == 0), never throws; the last error text is available through
 * pt_get_last_error(). A pt_context is single-thread-affine, like Donut's render thread (Rtxpt/SampleCommon/SampleBaseApp.cpp:93,183).
 *
 * Plain C, plain pointers and sizes; no torch / HIP types in any signature. All pointers are HOST pointers unless a
 * parameter is explicitly called "device pointer".
 */
#ifndef MI355PT_H
#define MI355PT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_context pt_context;

enum {
    PT_OK = 0,
    PT_ERROR_INVALID_ARGUMENT = 1,
    PT_ERROR_NO_DEVICE = 2,          /* no HIP device / HIP runtime failure: the library has NO CPU fallback */
    PT_ERROR_HIP = 3,
    PT_ERROR_IO = 4,                 /* glTF / file problems (Sample::LoadScene returning false) */
    PT_ERROR_UNSUPPORTED = 5,        /* a setting outside the supported parity-knob set */
    PT_ERROR_NOT_READY = 6,
};

/* --- data contract ------------------------------------------------------------------------------------------------ */

/* Rtxpt/Shaders/PathTracer/Materials/MaterialPT.h:45-77 (128 bytes, identical field order); flag bits :24-42.
 * Texture words: baseLOD<<24 | mipLevels<<16 | textureIndex (Rtxpt/Materials/MaterialsBaker.cpp:497-508), 0xFFFFFFFF = none. */
typedef struct PTMaterialData {
    float    BaseOrDiffuseColor[3]; uint32_t Flags;
    float    SpecularColor[3];      int32_t  _padding0;
    float    EmissiveColor[3];      float    ShadowNoLFadeout;
    float    Opacity, Roughness, Metalness, NormalTextureScale;
    float    _padding1, AlphaCutoff, TransmissionFactor; uint32_t BaseOrDiffuseTextureIndex;
    uint32_t MetalRoughOrSpecularTextureIndex, EmissiveTextureIndex, NormalTextureIndex, OcclusionTextureIndex;
    uint32_t TransmissionTextureIndex; float IoR, ThicknessFactor, DiffuseTransmissionFactor;
    float    AttenuationColor[3];   float    AttenuationDistance;          /* VolumePTConstants */
} PTMaterialData;

/* Rtxpt/Shaders/PathTracer/PathTracerShared.h:24-42 (112 bytes, identical field order) */
typedef struct PathTracerCameraData {
    float PosW[3];       float NearZ;
    float DirectionW[3]; float PixelConeSpreadAngle;
    float CameraU[3];    float FarZ;
    float CameraV[3];    float FocalDistance;
    float CameraW[3];    float AspectRatio;
    uint32_t ViewportSize[2]; float ApertureRadius; float _padding0;
    float Jitter[2];     float _padding1, _padding2;
} PathTracerCameraData;

/* Rtxpt/Shaders/PathTracer/Lighting/PolymorphicLight.h:47-72 */
typedef struct PolymorphicLightInfo { float Center[3]; uint32_t ColorTypeAndFlags; uint32_t Direction1, Direction2, Scalars, LogRadiance; } PolymorphicLightInfo;
typedef struct PolymorphicLightInfoEx { uint32_t IesProfileIndex, PrimaryAxis, CosConeAngleAndSoftness, UniqueID; } PolymorphicLightInfoEx;

/* One glTF primitive ("geometry" in Donut). The vertex streams of all geometries live in shared arrays; offsets are in elements.
 * Stream formats follow Donut's GeometryData use at PathTracerBridgeDonut.hlsli:166-243: indices u32, positions float3,
 * uv float2, normal / tangent RGBA8_SNORM (Packing.hlsli:127-167). */
typedef struct PtGeometryDesc {
    uint32_t indexOffset, numIndices;
    uint32_t vertexOffset, numVertices;
    uint32_t flags;            /* PT_GEOM_HAS_* */
    uint32_t materialIndex;
    uint32_t geomFlags;        /* PT_GEOMF_* (Rtxpt/SampleCommon/AccelerationStructureUtil.h:35-104: opaque unless alpha tested / excluded from NEE) */
    uint32_t _pad;
} PtGeometryDesc;
enum { PT_GEOM_HAS_UV = 1, PT_GEOM_HAS_NORMAL = 2, PT_GEOM_HAS_TANGENT = 4 };
enum { PT_GEOMF_ALPHA_TESTED = 1, PT_GEOMF_EXCLUDE_FROM_NEE = 2 };
typedef struct PtMeshDesc { uint32_t firstGeometry, numGeometries; } PtMeshDesc;            /* one BLAS in the reference */
typedef struct PtInstanceDesc { float transform[12]; uint32_t meshIndex; uint32_t analyticProxyLight; uint32_t _pad[2]; } PtInstanceDesc;   /* row-major 3x4, InstanceData.transform */
/* analyticProxyLight: 0 = none; k + 1 = the instance's geometries whose material has EnableAsAnalyticLightProxy stand in for record k of pt_set_lights — a path that
   hits them adds that sphere light's radiance, MIS-weighted against its NEE samples (SubInstanceData.AnalyticProxyLightIndex, LightsBaker.cpp:718-753: the light
   leaf above the mesh node, or the light whose proxyMeshNodes names it; PathTracer.hlsli:636-648; LightSampler.hlsli:363-390). Sphere lights only, as in the reference. */

typedef struct PtGeometryBuffers {
    const uint32_t* indices;   uint32_t numIndices;
    const float*    positions;                      /* numVertices x 3 */
    const float*    uvs;                            /* numVertices x 2 or NULL */
    const uint32_t* normals;                        /* numVertices or NULL */
    const uint32_t* tangents;                       /* numVertices or NULL */
    uint32_t        numVertices;
} PtGeometryBuffers;

enum { PT_TEX_RGBA8_UNORM = 0, PT_TEX_RGBA8_SRGB = 1, PT_TEX_RGBA32F = 2 };
typedef struct PtTextureDesc { uint32_t width, height, format; const void* pixels; } PtTextureDesc;

/* EnvMapSceneParams (Rtxpt/Shaders/PathTracer/Lighting/EnvMap.hlsli:22-30): local->world rotation and colour multiplier. */
typedef struct PtEnvMapSceneParams { float Transform[12]; float ColorMultiplier[3]; float Enabled; } PtEnvMapSceneParams;

/* EMB_DirectionalLight (Rtxpt/Lighting/Distant/EnvMapBaker.hlsl:20-26): ColorIntensity.rgb = colour, .a = irradiance-like intensity; Direction = the direction
 * the light travels (the disc is drawn at -Direction); AngularSize in radians. */
typedef struct PtEnvDirectionalLight { float ColorIntensity[4]; float Direction[3]; float AngularSize; } PtEnvDirectionalLight;

/* Subset of PathTracerConstants (PathTracerShared.h:45-103) plus the shader macros Sample::FillPTPipelineGlobalMacros pushes
 * (Rtxpt/Sample.cpp:988-1042) that affect the reference-mode estimator. Defaults: pt_default_settings(). */
typedef struct PtSettings {
    uint32_t bounceCount, diffuseBounceCount;
    float    perPixelJitterAAScale;
    float    texLODBias;
    float    fireflyFilterThreshold;        /* 0 = disabled */
    float    envMapDiffuseSampleMIPLevel;
    uint32_t NEEEnabled, NEEType, NEECandidateSamples, NEEFullSamples;   /* NEEFullSamples: 0..63 (clamped like RTXPT_LIGHTING_MAX_SAMPLE_COUNT); 1 is the fast path */
    uint32_t enableRussianRoulette;
    uint32_t nestedDielectricsQuality;      /* RTXPT_NESTED_DIELECTRICS_QUALITY 0/1/2 */
    uint32_t enableLDSamplerForBSDF;
    uint32_t diffuseBrdf;                   /* 0 Lambert, 2 Frostbite */
    uint32_t useFp16Types;                  /* the reference's "Use explicit fp16 types" (SampleUI.h:182 UseFp16Types, Sample.cpp:1035 RTXPT_LP_TYPES_USE_16BIT_PRECISION):
                                               1 = lpfloat is binary16 — the reference's DEFAULT build, and what pt_default_settings returns; 0 = lpfloat is fp32 */
    uint32_t _pad;
} PtSettings;

typedef struct PtDeviceDesc {
    int32_t  deviceOrdinal;                 /* HIP device of this context (one context per GPU / process) */
    uint32_t shardRank, shardCount;         /* pixel-tile shard of this context: tile t belongs to rank morton(t) % shardCount */
    uint32_t flags;                         /* PT_DEVICE_* */
} PtDeviceDesc;
#define PT_DEVICE_SERIAL_KERNELS 1u         /* one batch, one stream: kernels of a pt_render call never overlap (profiling / per-kernel timing) */
#define PT_DEVICE_PREFER_FAST_BUILD 2u      /* scene builds (pt_set_geometry / pt_set_instances / pt_load_scene_gltf) use the plain device-side PLOC builder (15 ms at
                                               2.8 M triangles) instead of the default: AccelStructBuildFlags::PreferFastTrace as the reference sets it
                                               (Rtxpt/Sample.cpp:1093) = PLOC + 12 passes of parallel re-insertion + cost-driven wide nodes, all on the device
                                               (81 ms at 2.8 M triangles, 24 % more rays per second than plain PLOC). pt_animate(rebuild = 1) always builds fast;
                                               refits keep the topology they find. The image does not depend on the tree. */
#define PT_DEVICE_HOST_SAH_BUILDER 4u       /* the fast-trace tree of round 2 instead: binned-SAH topology + insertion-based optimisation on the host's cores
                                               (1.8 s at 2.8 M triangles, the same trace speed within 0.2 %); bounds, collapse and refit on the device as always */

typedef struct PtFrameStats {
    uint64_t extendRays, shadowRays, hits;                  /* "rays" of the Mrays/s metric = extendRays + shadowRays */
    uint64_t nodeVisitsExtend, triTestsExtend, nodeVisitsShadow, triTestsShadow;   /* in-kernel BVH counters (when enabled) */
    uint64_t leafVisitsExtend, waveItersExtend, leafVisitsShadow, waveItersShadow; /* leaf steps; traversal-loop iterations summed over waves */
    uint64_t extendPhaseCycles[4];          /* s_memtime cycles summed over waves: refill, inner block, leaf block, slot bookkeeping */
    uint64_t leafBlocksExtend;              /* iterations in which the wave executed the leaf block */
    uint64_t waveItersMaxExtend;            /* longest traversal loop of any single wave in any extend launch (tail indicator) */
    uint64_t extendRayIterHist[16];         /* extend rays (or their sub-tree tasks) that needed >= 128 loop iterations: bin k (7..15) counts 2^k <= iterations < 2^(k+1) (counters build) */
    uint32_t longRayCount, _padLong;        /* (counters build) sample of extend rays that needed more than 2048 iterations: origin, dir, iterations, tag */
    float    longRays[32][8];
    uint64_t extendEvents[8];               /* wave-level block executions: refill, chunk load, inner, leaf, alpha test, hit reduction, pop loop, pop trips */
    double   gpuMilliseconds;                               /* whole pt_render call, HIP events */
    double   extendKernelMs, shadeKernelMs, shadowKernelMs; /* summed per-launch HIP-event time: filled for serial-kernel frames (pt_set_serial_kernels), counter builds and under
                                                               MI355PT_PASS_LOG; zero for pipelined frames, whose launches carry no events (ten API calls per pass and batch) */
    uint32_t extendLaunches, iterations;
    uint32_t pathsTraced, tailLaunches;     /* tailLaunches: launches of the tail kernel (pt_set_tail_paths), summed over the batches */
} PtFrameStats;

/* Tone mapping constants (Rtxpt/ToneMapper/ToneMapping_cb.h:30-45) with the colour transform as the 3x3 that `mul(color, M)` uses
   (Rtxpt/ToneMapper/ToneMapping.ps.hlsli:164); 80 bytes. pt_default_tonemap fills the reference defaults (ToneMappingPasses.h:36-53). */
typedef struct PtToneMapParams {
    float    whiteScale, whiteMaxLuminance;
    uint32_t toneMapOperator;               /* 0 Linear, 1 Reinhard, 2 ReinhardModified, 3 HejiHableAlu, 4 HableUc2, 5 Aces (ToneMapping_cb.h:17-25) */
    uint32_t clamped;
    uint32_t autoExposure; float avgLuminance, autoExposureLumValueMin, autoExposureLumValueMax;   /* TONEMAPPING_AUTOEXPOSURE_CPU: the host supplies avgLuminance */
    float    colorTransform[9];             /* row-major, result_j = sum_i color_i * M[i][j] */
    uint32_t enabled, _pad0, _pad1;
} PtToneMapParams;

/* --- entry points (reference seam each one replaces) -------------------------------------------------------------- */

/* DeviceManager creation + Sample::Init (Rtxpt/SampleCommon/SampleBaseApp.cpp:63-140, Rtxpt/Sample.cpp:136) */
int32_t pt_create(const PtDeviceDesc* desc, pt_context** out);
/* SceneUnloading + destructor (Rtxpt/Sample.cpp:523-560) */
int32_t pt_destroy(pt_context* ctx);
const char* pt_get_last_error(pt_context* ctx);

/* Sample::LoadScene + SceneLoaded + MaterialsBaker::ImportFromDonut (Rtxpt/Sample.cpp:447-560, Rtxpt/Materials/MaterialsBaker.cpp:660-705): a `.gltf` / `.glb` file, or —
   a path ending in `.json` — an RTXPT `.scene.json` asset folder (the media folder is the file's own): pt_scene_json_import + pt_scene_import_apply, the graph's directional
   lights (pt_set_scene_directional_lights) and the EnvironmentLight's image, lat-long (.exr / .hdr / float .dds) or cube map (.dds), with the environment UI block at identity
   as after a scene load (Sample.cpp:554); an unreadable image leaves the scene without one, as there. Camera and SampleSettings: pt_scene_json_import's getters. */
int32_t pt_load_scene_gltf(pt_context* ctx, const char* path);
/* glTF 2.0 animations of the file pt_load_scene_gltf read (Sample::Animate -> Scene::Animate, Rtxpt/Sample.cpp:785-811; Donut's SceneGraphAnimation is not
   vendored: samplers and channels are evaluated as the glTF specification defines them — LINEAR (spherical for rotations) / STEP / CUBICSPLINE over node
   translation / rotation / scale / weights, time clamped to the key range; skins and morph targets: pt_gltf_animation_positions). pt_gltf_animation_instances returns the number of instances and
   writes up to `capacity` transforms in the order pt_load_scene_gltf created them: the `instances` argument of pt_animate. Host only, no device. */
typedef struct pt_gltf_animation pt_gltf_animation;
int32_t pt_gltf_animation_load(const char* path, pt_gltf_animation** out, uint32_t* numAnimations, float* durationSeconds);
int32_t pt_gltf_animation_instances(pt_gltf_animation* anim, uint32_t animation, float timeSeconds, PtInstanceDesc* out, uint32_t capacity);
/* glTF skins and morph targets at the same time (Donut's SkinnedMeshInstance: the reference rewrites a skinned instance's vertices every frame and updates its BLAS, Sample.cpp:1065,
   1170-1198): the object-space positions of the file's whole vertex stream with every skinned primitive posed, SUM_k w_k (inverse(meshNode) * joint_k * inverseBind_k) p —
   the `positions` argument of pt_animate (refit, or rebuild). Returns the vertex count; with capacityVertices below it nothing is written. Normals / tangents keep the
   bind pose; a mesh shared by several nodes takes the last node's pose. Morph targets (primitive.targets, POSITION displacements) are applied before the skin:
   p = base + SUM_i w_i target_i, the weights from the animation's "weights" channel, else the node's, else the mesh's. Host only. */
int32_t pt_gltf_animation_positions(pt_gltf_animation* anim, uint32_t animation, float timeSeconds, float* positionsXYZ, uint32_t capacityVertices);
/* the posed NORMAL / TANGENT streams of the same pose (SNORM8 x 3 / x 4, the packing of PtGeometryBuffers; either pointer may be NULL): a skinned vertex's normal is
   normalize(SUM_k w_k inverse-transpose(J_k) n), its tangent normalize(SUM_k w_k J_k t.xyz) with the handedness kept; morph targets' NORMAL / TANGENT displacements are added first (renormalised); other vertices keep theirs.
   Returns the number of vertices (capacity 0: a query). -> pt_animate_normals */
int32_t pt_gltf_animation_normals(pt_gltf_animation* anim, uint32_t animation, float timeSeconds, uint32_t* normalsSnorm8, uint32_t* tangentsSnorm8, uint32_t capacityVertices);
void    pt_gltf_animation_free(pt_gltf_animation* anim);
/* raw-buffer path: the same data the bakers upload (GeometryData/InstanceData/PTMaterialData/SubInstanceData, Rtxpt/Sample.cpp:2319-2384) */
int32_t pt_set_geometry(pt_context* ctx, const PtGeometryBuffers* buffers, const PtGeometryDesc* geometries, uint32_t numGeometries,
                        const PtMeshDesc* meshes, uint32_t numMeshes);
int32_t pt_set_instances(pt_context* ctx, const PtInstanceDesc* instances, uint32_t numInstances);
int32_t pt_set_materials(pt_context* ctx, const PTMaterialData* materials, uint32_t numMaterials, const PtTextureDesc* textures, uint32_t numTextures);
/* EnvMapBaker source + EnvMapSceneParams (Rtxpt/Sample.cpp:1364-1388,1939): lat-long float RGB image, row 0 at +Y. width==0 disables.
 * params->ColorMultiplier is the reference's own: tint * intensity / c_envMapRadianceScale (Sample.cpp:1939-1940; the baked cube holds radiance x 1/4, so a
 * multiplier of 4 renders the image's radiance as supplied). params == NULL means identity orientation and exactly that: ColorMultiplier = 1 / c_envMapRadianceScale. */
int32_t pt_set_environment(pt_context* ctx, const float* rgbLatLong, uint32_t width, uint32_t height, const PtEnvMapSceneParams* params);
/* The same with a CUBE map as the source image (EnvMapBaker.cpp:399-411: a loaded texture with six array slices becomes m_loadedSourceBackgroundTextureCubemap;
 * EnvMapBaker.hlsl:98-110 BackgroundSourceType 2: t_SrcCubemapEnvMap.SampleLevel(s_Linear, direction, 0) — e.g. a cube the reference itself saved as .dds): six faces of
 * dim x dim RGBA float texels in D3D's face order +X -X +Y -Y +Z -Z, top row first (what pt_image_read_dds_cube returns). Kept as the RGBA16F texels a BC6H / RGBA16F file
 * decodes to; sampled bilinearly within the face the direction looks at. One source at a time: this call replaces a lat-long image and vice versa; dim == 0 disables. */
int32_t pt_set_environment_cube(pt_context* ctx, const float* rgbaFaces, uint32_t dim, const PtEnvMapSceneParams* params);
/* EnvMapBaker::Update (Rtxpt/Lighting/Distant/EnvMapBaker.cpp:298-343, 425-620; EnvMapBaker.hlsl:194-246, 268-371): the path tracer and the light baker do
 * not sample the lat-long source but the RGBA16F cube the baker makes of it: cubeDim^2 x 6 texels (2048 for an image source, 0 = keep), solid-angle weighted
 * mips down to 8x8, radiance x 1/4 (c_envMapRadianceScale, Sample.cpp:88 - the host compensates in ColorMultiplier: Sample.cpp:1939-1940) clamped to the
 * fp16 range, and the scene's directional lights drawn into it as anti-aliased discs (Sample::UpdateLighting, Sample.cpp:1361-1388). At most 16 lights
 * (EMB_MAXDIRLIGHTS). The bake runs on the device at the next pt_render / pt_prepare. */
int32_t pt_set_environment_bake(pt_context* ctx, uint32_t cubeDim, const PtEnvDirectionalLight* directionalLights, uint32_t numDirectionalLights);
/* EnvMapBaker's BC6U compression of the cube (EnvMapBaker.cpp:593-633, BC6UCompress.hlsl; m_compressionQuality = 1 and enabled by default on D3D12, off on Vulkan): with
 * quality 1 ("Fast": one-region mode 11) every level goes through the reference's encoder and the BC6H_UF16 decode the texture unit applies, and the path tracer samples the
 * result (≈ 0.5 % per texel, 2e-3 relative L2 on an environment-lit frame); the light baker's importance map keeps reading the uncompressed cube, as there. 0 = off (the
 * library's default, the reference on Vulkan); 2 = "Quality" (QUALITY 1: the best of the 32 two-region partitions in modes 7.6 / 9.5 replaces the one-region block where its error
 * estimate is lower). */
int32_t pt_set_environment_compression(pt_context* ctx, uint32_t quality);
/* Sample::UpdateLighting (Rtxpt/Sample.cpp:1361-1388), the host step in front of EnvMapBaker::Update: world-space directional lights -> the records
 * pt_set_environment_bake takes. AngularSize is raised to pi / (cubeDim / 2) (smaller discs cannot be drawn into the cube), Direction is taken into the
 * environment's local frame with params->Transform (NULL: identity) so the disc keeps its world direction under an environment rotation. No device needed. */
int32_t pt_env_bake_lights(const PtEnvDirectionalLight* worldLights, uint32_t numLights, const PtEnvMapSceneParams* params, uint32_t cubeDim, PtEnvDirectionalLight* out);
/* The same step inside the library, for directional lights that belong to the LOADED SCENE (pt_load_scene_gltf hands a file's KHR_lights_punctual directional lights over this way):
 * world-space records, converted with pt_env_bake_lights' arithmetic at every cube bake — with that bake's cube size and the environment's current orientation — and drawn into the
 * cube after the lights of pt_set_environment_bake (16 in all). n = 0 removes them. */
int32_t pt_set_scene_directional_lights(pt_context* ctx, const PtEnvDirectionalLight* worldLights, uint32_t numLights);
/* The procedural sky as the cube's source (EnvMapBaker.cpp:372-375, 422, 454-470, 516-533; EnvMapBaker.hlsl:228-236, 247-265; SampleProceduralSky.hlsli,
 * precomputed_sky.hlsli): with it the base layer adds ProceduralSky() — Bruneton's precomputed atmosphere, the sun disc and ray-marched clouds from a half-resolution
 * pre-pass cube — to whatever pt_set_environment's image (none: black) and the directional lights give. The constants are the shader's own constant block; the
 * look-up textures are the host's (the reference loads transmittance_earth / inscatter_earth / irradiance_earth / clouds .dds; its noise texture is bound but
 * never sampled): RGBA float texels, d = 1 for the 2-D ones. The reference bakes a sky at cubeDim 1024 (pt_set_environment_bake).
 *   consts == NULL: the sky is switched off;  textures == NULL: the textures of the previous call stay (per-frame constants: time of day, clouds).
 * Without a pt_set_environment call the environment is enabled with the identity orientation and ColorMultiplier = 1 / c_envMapRadianceScale. */
typedef struct PtAtmosphereParameters {                /* precomputed_sky.hlsli:23-35 */
    float StarIrradiance[3], StarAngularDiameter, RayleightScatteringRGB[3], PlanetSurfaceRadius, MieScatteringRGB[3], PlanetAtmosphereRadius;
    float MieHenyeyGreensteinG, SqDistanceToHorizontalBoundary, AtmosphereHeight, reserved;
} PtAtmosphereParameters;
typedef struct PtProceduralSkyConstants {              /* SampleProceduralSky.hlsli:18-46 */
    PtAtmosphereParameters SkyParams;
    float FinalRadianceMultiplier[3], _padding3, SunDir[3], CloudsTime, GroundAlbedo[3], SunAngularDiameter;
    float _padding0, _padding1, sun_solid_angle, _padding2, physical_sky_ground_radiance[3], cloud_density_offset;
    float sky_transmittance, sky_phase_g, sky_amb_phase_g, sky_scattering;
} PtProceduralSkyConstants;
typedef struct PtSkyTexture { const float* rgba; uint32_t width, height, depth, _pad; } PtSkyTexture;
typedef struct PtProceduralSkyTextures { PtSkyTexture transmittance, scattering, irradiance, clouds; } PtProceduralSkyTextures;
int32_t pt_set_procedural_sky(pt_context* ctx, const PtProceduralSkyConstants* consts, const PtProceduralSkyTextures* textures);
/* SampleProceduralSky::Update (Rtxpt/Lighting/Distant/SampleProceduralSky.cpp:67-153) and the members it reads (SampleProceduralSky.h:70-86): scene time and the
 * preset name of the environment path ("==PROCEDURAL_SKY==", "..._MORNING==", "..._MIDDAY==", "..._EVENING==", "..._DAWN==", "..._PITCHBLACK==", SampleCommon.h:62-67)
 * -> the constant block. `state` carries the two low-pass filtered times of day and the last scene time between calls (zero it once); params NULL = the
 * reference's defaults. Returns 1 when the constants differ from the previous call's (the cube must be re-baked), 0 when not, < 0 on error. Host only. */
typedef struct PtProceduralSkyParams { float colorTint[3], brightness, sunBrightness, cloudsMovementSpeed, timeOfDayMovementSpeed, sunTimeOfDayOffset, sunEastWestRotation,
                                       sunAngularDiameterDeg, cloudDensityOffset, cloudTransmittance, cloudScattering; } PtProceduralSkyParams;
typedef struct PtProceduralSkyState { double lastSceneTime; float timeOfDayL1, timeOfDayL2; PtProceduralSkyConstants lastConstants; } PtProceduralSkyState;
void pt_procedural_sky_default_params(PtProceduralSkyParams* out);
int32_t pt_procedural_sky_update(PtProceduralSkyState* state, const PtProceduralSkyParams* params, double sceneTime, const char* preset, int32_t forceInstantUpdate, PtProceduralSkyConstants* out);
/* analytic lights already converted by the host (LightsBaker.cpp:456-556 ConvertLight); emissive triangles are baked automatically */
int32_t pt_set_lights(pt_context* ctx, const PolymorphicLightInfo* lights, const PolymorphicLightInfoEx* lightsEx, uint32_t numLights);
/* The frustum term of LightsBaker's ImportanceBooster (Rtxpt/Lighting/LightsBaker.hlsl:108-136, LightsBaker.cpp:884-925 UpdateFrustumConsts; on by default in the reference for every
 * NEEType: LightsBaker.h:247-249, multiplier 8, fade distance 5): a light inside the camera frustum weighs 1 + mul times as much in the global proxy table, one within the fade distance
 * of it proportionally less, environment lights half the boost. viewProjRowMajor16 is the matrix the reference hands its baker — BakeSettings::ViewProjMatrix =
 * IView::GetViewProjectionMatrix() of Donut (row vectors: clip = p * M); NULL or mul 0 turns the term off (the state after pt_create). Call it whenever the camera moves:
 * only the weights and the proxy table are rebuilt (0.2 ms), not the lights. */
int32_t pt_set_light_importance_boost(pt_context* ctx, const float* viewProjRowMajor16, float frustumMul, float frustumFadeDistance);
/* NEE-AT, the path tracer's side (NEEType 2; Rtxpt/Shaders/PathTracer/Lighting/LightSampler.hlsli:51-93,120-180,184-200,242-268,318-332,411-420, PathTracerNEE.hlsli:88-161,
 * 199-273). Replaces the bindings t_LightLocalSamplingBuffer (t17), u_LightFeedbackTotalWeight (u20), u_LightFeedbackCandidates (u21) of Sample.cpp:2338-2342 and the
 * LightingControlData fields LocalSamplingTileJitter / LocalSamplingResolution / LocalToGlobalSampleRatio / ScreenSpaceVsWorldSpaceThreshold / TemporalFeedbackRequired
 * (LightsBaker.cpp:1055-1075). `table`: resX * resY tiles of 8 x 8 pixels (LightingConfig.h:27-31), 128 packed entries each — light index << 9 | (proxies of that light in the
 * tile - 1), sorted by light index (LightingTypes.hlsli:172-175) — what LightsBaker's ProcessFeedbackHistory passes write; NULL removes the local layer. With a table, the
 * candidates [globalCount, NEECandidateSamples) of a screen-space-coherent vertex (ray-cone width / path length < threshold; LightsBaker.h:240 uses 0.3) are drawn from the
 * pixel's tile, and every MIS weight uses both samplers' pdfs. temporalFeedback != 0: every visible NEE sample is offered to its pixel's feedback reservoir (weight =
 * contribution / globalPdf^0.65); needs NEEFullSamples 1. The baker that turns feedback into the next frame's tables is the host's (remainder). */
int32_t pt_set_local_light_sampling(pt_context* ctx, const uint32_t* table, uint32_t resX, uint32_t resY, uint32_t jitterX, uint32_t jitterY, float localToGlobalSampleRatio,
                                    float screenSpaceVsWorldSpaceThreshold, int32_t temporalFeedback);
/* the feedback reservoirs the last pt_render call filled: one plane of width x height slots per sample of the call (sample = 0 .. sampleCount-1), cleared at the start of the
 * call (total weight 0, candidate 0xFFFFFFFF); candidate = light index | 0x80000000 when the vertex was screen-space coherent (LightingTypes.hlsli:184-320) */
int32_t pt_get_light_feedback(pt_context* ctx, uint32_t sample, float* totalWeight, uint32_t* candidates);
/* NEE-AT with the light baker in the loop (the reference's default sampler, CommandLine.h:42; LightsBaker::UpdateBegin / UpdateEnd, Rtxpt/Lighting/LightsBaker.cpp:964-1420, and the
 * ProcessFeedbackHistory* / ClearFeedbackHistory / ComputeProxyCounts passes of LightsBaker.hlsl:753-830, 880-948, 1062-1855): while enabled, every sample of pt_render is one
 * frame — last frame's feedback re-weights the global proxy table (globalTemporalFeedbackWeight, SampleUI.h:158) and becomes this frame's tile tables (sampled with
 * localToGlobalSampleRatio, :159, once feedback exists), then the frame is traced and fills the reservoirs again. History is read at the same pixel (no motion vectors on this
 * path: exact for the still camera of an accumulation run; call pt_neeat_reset after a camera cut). Needs NEEFullSamples 1, NEEType 2 is the matching setting (the global table
 * is built as for type 1). preFilter: LightsBaker.h:251 m_importanceBoost_PreFilter (default on). The frustum importance boost (LightsBaker.h:247-249) is not applied.
 * pt_get_light_feedback(0, ...) returns the run's reservoirs after the last frame; pt_get_neeat_tables the tile tables and jitter that frame was traced with. */
int32_t pt_set_neeat(pt_context* ctx, int32_t enable, float globalTemporalFeedbackWeight, float localToGlobalSampleRatio, float screenSpaceVsWorldSpaceThreshold, int32_t preFilter);
/* PlanarViewConstants::matWorldToClip of the frame (Donut; row vectors: clip = p * M, 16 floats row-major) — what Bridge::ExportSurface / ExportNonSurface project the path's last
 * vertex with when the reference-mode path tracer "dumps guide buffers" (PathTracer.hlsli:487, 684; PathTracerBridgeDonut.hlsli:1105-1140). Of those buffers this path keeps the depth,
 * because NEE-AT's Reproject (LightsBaker.hlsl:1348-1375; motion vectors are zero in reference mode) tests it: a pixel whose last two frames exported depths more than 1.5 x apart counts
 * as disoccluded and its feedback is not blended. NULL (the state after pt_create): nothing is exported, every pixel counts as valid. */
int32_t pt_set_view_projection(pt_context* ctx, const float* worldToClipRowMajor16);
/* ---- Stable planes: the realtime mode's pre-pass (Sample.cpp:2456-2473 dispatches RayGen with PATH_TRACER_MODE_BUILD_STABLE_PLANES once per frame).
 * From every pixel the pass follows the delta (perfectly specular) lobes only — a Whitted-style tree of at most three branches, PathTracerStablePlanes.hlsli:104-330 — and stops each
 * branch at the first surface a denoiser can work on: that vertex becomes the branch's "stable plane". Output, as RenderTargets.cpp:60-141, 340-352 declares it:
 *   header         4 x height x width words: [0..2] the planes' stable branch ids (0xFFFFFFFF: no such plane), [3] asuint(first-hit ray length) & ~3 | dominant plane index
 *   planes         3 x plane stride records of 80 bytes (PtStablePlane = StablePlanes.hlsli:41-58) at GenericTSPixelToAddress(pixel, plane) (8 x 8 tiles, Morton order inside)
 *   stableRadiance RGBA16F: emission and sky reached along the delta paths (noise-free; the fill pass does not count it again)
 *   depth R32F, motionVectors RGBA16F, throughput R11G11B10F: Bridge::ExportSurface / ExportNonSurface for the dominant plane; specularHitT R32F is cleared here and filled by the noisy pass
 * Object motion enters the motion vectors through pt_set_motion_history / pt_set_previous_pose (below); without them it reads as zero. Not carried over: the two automatic motion-vector block types of a
 * material (PTMaterialFlags_PSDBlockMVsAtSurfaceType 1 / 2 need Donut's per-triangle curvature; "Off" and "Full" are honoured). */
typedef struct PtStablePlanesParams {
    uint32_t activeStablePlaneCount;            /* m_ui.StablePlanesActiveCount, 1..3 */
    uint32_t maxStablePlaneVertexDepth;         /* m_ui.StablePlanesMaxVertexDepth; used as min(min(it, 15), bounceCount) (Sample.cpp:1532) */
    uint32_t allowPrimarySurfaceReplacement;    /* m_ui.AllowPrimarySurfaceReplacement */
    uint32_t subSampleCount;                    /* m_ui.ActualSamplesPerPixel(): invSubSampleCount of the noisy passes */
    float matWorldToClip[16];                   /* PlanarViewConstants of the frame (row vectors, row-major): with the jitter offset, ... */
    float matWorldToClipNoOffset[16];           /* ... without it, ... */
    float prevMatWorldToClipNoOffset[16];       /* ... and last frame's */
    float clipToWindowScale[2];                 /* (width / 2, -height / 2) */
    float _pad[2];
} PtStablePlanesParams;
typedef struct PtStablePlane {                  /* StablePlanes.hlsli:41-58 */
    float RayOrigin[3]; float LastRayTCurrent; float RayDir[3]; float SceneLength;
    uint32_t PackedThpAndMVs[3]; uint32_t VertexIndexAndRoughness; uint32_t DenoiserPackedBSDFEstimate[3]; uint32_t PackedNormal;
    uint32_t PackedNoisyRadianceAndSpecAvg[2]; uint32_t FlagsAndVertexIndex; uint32_t PackedCounters;
} PtStablePlane;
int32_t pt_stable_planes_plane_stride(uint32_t width, uint32_t height, uint32_t* stride);     /* GenericTSComputePlaneStride (Utils.hlsli:328-332) */
/* traces the pass for the context's pixels with the camera ray of sampleIndex (the sub-samples of a realtime frame share it); stats: rays, passes, GPU time */
int32_t pt_build_stable_planes(pt_context* ctx, uint32_t sampleIndex, const PtStablePlanesParams* params, PtFrameStats* stats);
/* One sub-sample of the realtime mode's noisy passes (Sample.cpp:2497-2516: RayGen with PATH_TRACER_MODE_FILL_STABLE_PLANES once per sub-sample, sample index sampleBaseIndex + subSampleIndex)
 * over the buffers pt_build_stable_planes left: every path starts on plane 0 (FirstHitFromVBuffer), follows the recorded delta tree while its branch id matches, and deposits its radiance —
 * total and specular average, attenuated by 1 / subSampleCount — on the plane it last touched (PtStablePlane.PackedNoisyRadianceAndSpecAvg, four binary16 values); emission along the stable
 * branches was collected by the build pass and is not counted again. specularHitT is filled for the dominant plane. NEE with one full sample per vertex (NEEFullSamples 0 or 1); NEE-AT's local
 * sampling tables are honoured; with pt_set_neeat on (after the frame's baker passes: pt_realtime_frame, or pt_neeat_update_begin / _end) the visible light samples feed the temporal
 * feedback reservoirs the next frame's UpdateBegin reads. ReSTIR DI / GI hand-offs do not exist here. */
int32_t pt_fill_stable_planes(pt_context* ctx, uint32_t sampleIndex, const PtStablePlanesParams* params, PtFrameStats* stats);
/* DenoisingGuidesBaker::DenoiseSpecHitT (Sample.cpp:2544, after the noisy passes of a frame): the 5 x 5 depth-aware fill-in of specularHitT, one ping and one pong (DenoisingGuidesBaker.hlsl:50-113) */
int32_t pt_denoise_spec_hit_t(pt_context* ctx);
/* PostProcess.hlsl NO_DENOISER_FINAL_MERGE (Sample.cpp:2764-2765: the realtime frame when no denoiser runs): output colour = stable radiance + every existing plane's noisy radiance
 * (StablePlanesContext::GetAllRadiance), alpha 1 — written into the context's radiance buffer (pt_map_radiance, pt_tonemap, pt_gather read it; it counts as one accumulated sample). */
int32_t pt_stable_planes_merge(pt_context* ctx);
/* Tile-sharded realtime frames (PtDeviceDesc.shardCount > 1; no reference analogue): every rank builds and fills the planes of its own tiles, the rank that denoises or shows the frame needs
 * them all. Per pixel 284 bytes travel: the four header words, the three 80-byte plane records, stable radiance, depth, specular hit distance, motion vectors, throughput.
 * pt_gather_stable_planes: with a communicator (pt_comm_init) every rank sends its records to rank 0 — RCCL point-to-point in one group, un-padded, on the library's stream, like pt_gather (a
 * world of one runs the protocol as a loop-back). Without one the host moves pt_pack_stable_planes' buffer (device memory, pt_stable_planes_shard_bytes(rank) bytes, the rank's pixels in
 * pt_pack_shard's order) and hands it to pt_unpack_stable_planes(buffer, rank) on the receiving context, which must have run a build pass of that size. pt_denoise_spec_hit_t is then allowed
 * on a sharded context. (When only the picture is needed: pt_stable_planes_merge on every rank, then pt_gather, moves 16 bytes per pixel instead.) */
int32_t pt_stable_planes_shard_bytes(pt_context* ctx, uint32_t rank, size_t* bytes);
int32_t pt_pack_stable_planes(pt_context* ctx, void* dstDevice, size_t bytes);
int32_t pt_unpack_stable_planes(pt_context* ctx, const void* srcDevice, size_t bytes, uint32_t rank);
int32_t pt_gather_stable_planes(pt_context* ctx);
/* One frame of the realtime mode with its passes coupled as Sample::PathTrace runs them (Rtxpt/Sample.cpp:2438-2516) and pt_set_neeat on: LightsBaker::UpdateBegin (usage counts, global proxy
 * table) -> pt_build_stable_planes(sampleIndex) -> LightsBaker::UpdateEnd on THIS frame's depth and screen-space motion vectors — Reproject (LightsBaker.hlsl:1348-1375) finds every pixel's
 * history where the build pass says it was, and drops it where the depths disagree — -> params->subSampleCount fill passes (sample indices sampleIndex, sampleIndex + 1, ...), whose light samples
 * come from the tile tables just built and whose visible samples fill the reservoirs the next frame's UpdateBegin reads. Without pt_set_neeat: build + fill passes with the global sampler.
 * The host calls pt_denoise_spec_hit_t / pt_stable_planes_merge / pt_get_stable_planes afterwards as it needs them. On tile shards: next comment. */
int32_t pt_realtime_frame(pt_context* ctx, uint32_t sampleIndex, const PtStablePlanesParams* params, PtFrameStats* buildStats, PtFrameStats* fillStats);
/* The realtime frame on TILE SHARDS with NEE-AT (no reference analogue). The baker's passes read whole neighbourhoods of three things a rank has for its own tiles only: last
 * frame's reservoirs (UpdateBegin), this frame's depth and motion vectors (UpdateEnd). With a communicator (pt_comm_init) pt_realtime_frame exchanges both itself — RCCL
 * point-to-point in one group, un-padded: 12 bytes per pixel before UpdateBegin, 16 after the build pass — and every rank runs the same baker passes on the same planes: tables,
 * proxy counts and, after pt_gather_stable_planes, the frame equal the unsharded run's. Without a communicator the host drives the parts and moves the packed buffers:
 *   pt_neeat_pack_feedback / pt_neeat_unpack_feedback (from the second frame on) -> pt_neeat_update_begin -> pt_build_stable_planes ->
 *   pt_pack_stable_plane_guides / pt_unpack_stable_plane_guides -> pt_neeat_update_end -> pt_fill_stable_planes.
 * pt_neeat_update_begin / pt_neeat_update_end are LightsBaker::UpdateBegin / UpdateEnd (Rtxpt/Sample.cpp:1380-1412, 2491-2494) as calls of their own; UpdateEnd reads the depth and
 * motion vectors the last pt_build_stable_planes left. The guides of a rank: depth, specular hit distance, motion vectors — 16 bytes per pixel in the rank's pixel order
 * (pt_shard_layout). */
int32_t pt_neeat_update_begin(pt_context* ctx);
int32_t pt_neeat_update_end(pt_context* ctx);
int32_t pt_pack_stable_plane_guides(pt_context* ctx, void* deviceDst, size_t bytes);
int32_t pt_unpack_stable_plane_guides(pt_context* ctx, const void* deviceSrc, size_t bytes, uint32_t rank);
/* copies the last pass's buffers to the host; any pointer may be NULL. planeCapacity in records (>= 3 x plane stride); the two RGBA16F targets as 4 binary16 bit patterns per pixel */
int32_t pt_get_stable_planes(pt_context* ctx, uint32_t* header, PtStablePlane* planes, size_t planeCapacity, uint16_t* stableRadiance, float* depth, float* specularHitT,
                             uint16_t* motionVectors, uint32_t* throughput);
int32_t pt_neeat_reset(pt_context* ctx);                                                      /* LightsBaker::BakeSettings::ResetFeedback */
int32_t pt_get_neeat_tables(pt_context* ctx, uint32_t tilesXY[2], uint32_t jitterXY[2], uint32_t* table, uint32_t tableCapacityWords);
/* Tile-sharded frames (PtDeviceDesc.shardCount > 1; no reference analogue): a rank traces and feeds back for its own pixels, the baker's passes read whole neighbourhoods, so
 * between two frames every rank needs the other ranks' reservoirs and exported depth — (weight, candidate, depth): 12 bytes per pixel; the depth is what the baker's
 * reprojection compares, also across shard borders — and then runs the same deterministic passes as everybody else: same tables and proxy counts on every rank, the same as
 * the unsharded run, with or without pt_set_view_projection. With a communicator (pt_comm_init) pt_render does the exchange itself (RCCL point-to-point in one group, un-padded, on
 * the library's stream). Without one the host moves the buffers after every frame: pt_neeat_pack_feedback (the rank's own pixels, pt_pack_shard's order, device memory) ->
 * the host's transport -> pt_neeat_unpack_feedback(rank r's buffer, r) on every other rank, before their next pt_render. */
int32_t pt_neeat_pack_feedback(pt_context* ctx, void* dstDevice, size_t bytes);
int32_t pt_neeat_unpack_feedback(pt_context* ctx, const void* srcDevice, size_t bytes, uint32_t rank);

/* BridgeCamera (PathTracerShared.h:109-141) */
int32_t pt_bridge_camera(uint32_t viewportWidth, uint32_t viewportHeight, const float camPos[3], const float camDir[3], const float camUp[3], float fovY,
                         float nearZ, float farZ, float focalDistance, float apertureRadius, const float jitter[2], PathTracerCameraData* out);
int32_t pt_set_camera(pt_context* ctx, const PathTracerCameraData* camera);                 /* Sample.cpp:2052 */
int32_t pt_default_settings(PtSettings* out);                                               /* SampleUI.h:152-183 with the §8a parity knobs pinned */
int32_t pt_set_settings(pt_context* ctx, const PtSettings* settings);                       /* UpdatePathTracerConstants, Sample.cpp:1464-1556 */

/* Animate + Scene::Refresh + UpdateSkinnedBLASs/BuildTLAS (Rtxpt/Sample.cpp:785-811,1170-1240): new instance transforms and/or
 * new vertex positions (same topology) -> LBVH refit (or rebuild when rebuild != 0) + emissive light re-bake. Either pointer may be NULL. */
int32_t pt_animate(pt_context* ctx, const PtInstanceDesc* instances, uint32_t numInstances, const float* positions, uint32_t numVertices, int32_t rebuild);
/* pt_animate for a host that knows which meshes it deformed — UpdateSkinnedBLASs walks the skinned mesh instances only (Rtxpt/Sample.cpp:1170-1198): `positions` is still the whole
 * array, but only the vertices [first, first + count) of each (first, count) pair of vertexRanges are read and uploaded, and only the shading records of the geometries those
 * vertices belong to are rewritten (C5: 1 of 80 geometries; the upload shrinks from 14 MB to 0.3 MB). vertexRanges == NULL: every vertex, i.e. pt_animate. */
int32_t pt_animate_ranges(pt_context* ctx, const PtInstanceDesc* instances, uint32_t numInstances, const float* positions, uint32_t numVertices,
                          const uint32_t* vertexRanges, uint32_t numRanges, int32_t rebuild);
/* Motion history for the realtime passes. Donut keeps, per scene refresh, every node's previous global transform (SceneGraph::Refresh -> InstanceData.prevTransform) and the previous
 * positions of the meshes its skinning pass rewrites (GeometryData.prevPositionOffset); Bridge::loadSurface makes prevPosW from them in the stable-plane build pass
 * (Rtxpt/Shaders/PathTracerBridgeDonut.hlsli:187-199, 619, 631) and PathTracerStablePlanes.hlsli:286 turns prevPosW - posW into the motion vectors' object term.
 * pt_set_motion_history(ctx, 1): from now on every pt_animate / pt_animate_ranges call is one scene refresh — the pose it finds becomes the previous pose (device-to-device copies
 * of the instance table and of the vertex ranges that differ), the pose it brings the current one; a call with neither instances nor positions only advances the history (a frame
 * in which nothing moved: previous = current; no refit, no re-bake). Off (the default, and what reference mode needs): object motion reads as zero, nothing is kept.
 * pt_set_previous_pose hands the previous pose over directly (a host with its own history; arrays shaped like the scene's, either may be NULL = that part did not move) and
 * turns the history on. Neither call touches the accumulation or the BVH. */
int32_t pt_set_motion_history(pt_context* ctx, int32_t enable);
int32_t pt_set_previous_pose(pt_context* ctx, const PtInstanceDesc* instances, uint32_t numInstances, const float* positions, uint32_t numVertices);
/* deformed meshes' vertex normals / tangents (Donut's skinning rewrites them with the positions, Sample.cpp:1170-1198): replaces the packed streams of pt_set_geometry (either may be NULL)
   and rewrites the shading records; the BVH does not depend on them. Resets the accumulation like pt_animate. */
int32_t pt_animate_normals(pt_context* ctx, const uint32_t* normalsSnorm8, const uint32_t* tangentsSnorm8, uint32_t numVertices);
/* BackBufferResizing / RenderTargets::Init (Rtxpt/SampleCommon/RenderTargets.cpp:35-240) */
int32_t pt_resize(pt_context* ctx, uint32_t width, uint32_t height);
/* Render -> SampleRenderCode -> PathTrace -> AccumulationPass for samples [first, first+count) (Rtxpt/Sample.cpp:1891-2313, 2438-2559, 2770-2778).
 * All `count` samples are traced as one wavefront pool and folded into the accumulation buffer in sample order. */
int32_t pt_render(pt_context* ctx, uint32_t sampleIndexFirst, uint32_t sampleCount, PtFrameStats* stats);
int32_t pt_reset_accumulation(pt_context* ctx);                                             /* m_ui.ResetAccumulation */
/* AccumulatedRadiance read-back (SaveTextureToFile seam): full-frame RGBA32F on the host; pixels of other shards are zero. */
int32_t pt_map_radiance(pt_context* ctx, const float** rgba32f, size_t* rowPitchBytes);
int32_t pt_unmap_radiance(pt_context* ctx);

/* RTXPT `.material.json` (the part the reference tree defines completely): PTMaterial::Read + PTMaterial::FillData
   (Rtxpt/Materials/MaterialsBaker.cpp:150-259, 516-591; defaults Rtxpt/Materials/MaterialsBaker.h:126-193). Host only, no context needed.
   textureWords[5]: packed PTMaterialData texture words (baseLOD << 24 | mipLevels << 16 | texture index, GetBindlessTextureIndex :487-509) of the
   Base / OcclusionRoughnessMetallic / Normal / Emissive / Transmission textures the caller has loaded for the document's "path" entries, or
   0xFFFFFFFF for a texture that is absent (its flag is then cleared, as in the reference). info (optional) receives what does not live in
   PTMaterialData: the five texture paths / sRGB / NormalMap flags and EnableAlphaTesting, ExcludeFromNEE, SkipRender, UseDonutEmissiveIntensity. */
typedef struct PtMaterialJsonInfo {
    char     texturePath[5][256];
    uint32_t textureSRGB[5], textureNormalMap[5];
    uint32_t enableAlphaTesting, excludeFromNEE, skipRender, useDonutEmissiveIntensity;
} PtMaterialJsonInfo;
int32_t pt_material_from_json(const char* jsonText, const uint32_t textureWords[5], PTMaterialData* out, PtMaterialJsonInfo* info);

/* Analytic lights as the host of the reference describes them: a Donut PointLight / SpotLight with RTXPT's LightExtension. pt_convert_light is
   LightsBaker::ConvertLight (Rtxpt/Lighting/LightsBaker.cpp:456-556): radius > 0 gives a sphere light (radiance = color * intensity / (pi r^2)); a spot
   adds cone shaping (cos outer angle, softness = 1 - inner / |outer|; outerAngle < 0 selects the minimum-falloff variant). radius == 0 gives a
   point-type record; the path tracer's light set has that type compiled out (PolymorphicLightPTConfig.h:17-22), so pt_set_lights carries it as the
   reference does: a slot in the light buffer with no power and empty samples (uniform light selection still lands on it). Host only. */
typedef struct PtAnalyticLightDesc {
    uint32_t type;                 /* 0 point, 1 spot */
    float    position[3], direction[3];
    float    color[3], intensity, radius;
    float    innerAngle, outerAngle;          /* degrees, spot only */
} PtAnalyticLightDesc;
int32_t pt_convert_light(const PtAnalyticLightDesc* light, PolymorphicLightInfo* base, PolymorphicLightInfoEx* ex);

/* RTXPT `.scene.json` asset folders. pt_scene_json_import reads what ExtendedScene::LoadWithThreadPool + Sample::SceneLoaded +
   MaterialsBaker::CreateFromScene consume (Rtxpt/SampleCommon/ExtendedScene.cpp:104-143, 203-375; Rtxpt/Sample.cpp:457-479, 520-640;
   Rtxpt/Materials/MaterialsBaker.cpp:707-748, 857-864) into a HOST-side import object — no device needed — that pt_scene_import_apply hands to a
   context through pt_set_materials / pt_set_geometry / pt_set_instances / pt_set_lights:
     "models"  glTF / GLB files, relative to the scene file; "graph" nodes with name / translation / rotation (xyzw) / scaling / model / type /
     children; leaves EnvironmentLight {radianceScale, textureIndex, rotation, path}, PointLight / SpotLight (Donut keys color, intensity, radius,
     range, innerAngle, outerAngle + RTXPT's proxyMeshNodes) -> pt_convert_light records in scene-graph order with the invisible ones dropped
     (|color * intensity| <= 1e-7, Sample.cpp:567-573), PerspectiveCamera(Ex) {verticalFov, zNear + the exposure keys}, SampleSettings (seven keys);
     `<media>/Materials/[<scene>/][<model>.]<material>.material.json` overrides in the reference's candidate order, through pt_material_from_json
     (a document replaces the glTF material as a whole; SkipRender removes the geometries, EnableAlphaTesting / ExcludeFromNEE set their flags).
   The file format of the graph itself and the light / camera keys belong to Donut (donut/engine/Scene.cpp, SceneGraph.cpp), which the reference
   tree does not vendor: they are restated from Donut's published sources. DirectionalLight leaves are returned by pt_scene_import_directional_lights. A model file's own
   KHR_lights_punctual lights and perspective cameras hang below its model node like its meshes (lights: as the graph's leaves of the same kind; cameras: plain PerspectiveCamera
   leaves, exposureMask bit 31, which leave the tone-mapping block alone; orthographic cameras are not listed). Not imported: animations,
   textures other than PNG, JPEG and .dds files (counted in texturesNotLoaded, the
   material then renders untextured as when the reference fails to load one). A point / spot light's "proxyMeshNodes" (ExtendedScene.cpp:46, 246-263) are resolved
   as Donut's SceneGraph::FindNode resolves them — '/'-separated node names from the root, a model's own root node named after its file — and the mesh instances at those
   nodes come out with analyticProxyLight set (LightsBaker.cpp:718-753). The environment map is reported (envPath), not loaded by the import itself (pt_load_scene_gltf on the `.scene.json` does load it): pt_image_read_float reads .exr / .hdr / float .dds files for pt_set_environment, pt_image_read_dds_cube cube maps for pt_set_environment_cube. NOTE: of
   an EnvironmentLight the reference application consumes only `path` (Sample.cpp:552-553); radianceScale / rotation / textureIndex are read by
   EnvironmentLight::Load but never used — tint, intensity and rotation of the environment come from the UI block (EnvironmentMapRuntimeParameters,
   reset to identity on every scene load, Sample.cpp:554, 1936-1948). They are reported for completeness; to match the reference do not apply them. */
typedef struct pt_scene_import pt_scene_import;
typedef struct PtToneMappingParameters PtToneMappingParameters;      /* defined with the display path below */
typedef struct PtSceneCameraDesc {          /* Sample::UpdateCameraFromScene inputs: LookAt(position, position + direction, up) */
    float    position[3], direction[3], up[3];
    float    verticalFov, zNear;            /* radians; Donut defaults 1.0 / 1.0 */
    uint32_t exposureMask;                  /* bit 0 enableAutoExposure, 1 exposureCompensation, 2 exposureValue, 3 exposureValueMin, 4 exposureValueMax present; bit 31: a glTF file's own camera */
    uint32_t enableAutoExposure; float exposureCompensation, exposureValue, exposureValueMin, exposureValueMax;
    char     name[64];
} PtSceneCameraDesc;
typedef struct PtSceneJsonInfo {
    uint32_t numModels, numGeometries, numMeshes, numInstances, numMaterials, numTextures, numLights, numCameras;
    uint32_t materialOverrides, texturesNotLoaded, lightsDropped, lightProxies, skippedGeometries, directionalLights;
    uint32_t hasEnvironment; float envRadianceScale[3]; float envRotation; int32_t envTextureIndex; char envPath[260];
    uint32_t settingsMask;                  /* bit i: key i of SampleSettings::Load present (realtimeMode, enableAnimations, startingCamera, realtimeFireflyFilter, maxBounces, maxDiffuseBounces, textureMIPBias) */
    uint32_t realtimeMode, enableAnimations; int32_t startingCamera; float realtimeFireflyFilter; int32_t maxBounces, maxDiffuseBounces; float textureMIPBias;
    int32_t  selectedCamera;                /* startingCamera when present and valid, else the last camera (Sample.cpp:590-603, 618-619); -1 without cameras */
    uint32_t lightProxiesResolved;          /* mesh instances linked to the point / spot light that names their node in "proxyMeshNodes" (PtInstanceDesc.analyticProxyLight) */
} PtSceneJsonInfo;
/* mediaPath: the folder that holds "Materials/" (NULL: the scene file's folder). Errors: PT_ERROR_IO (unreadable / malformed file or
   model, unknown model reference). A node's "euler" rotation follows Donut's rotationQuat: about the fixed x axis first, then y, then z (unpinned: Donut is not vendored). */
int32_t pt_scene_json_import(const char* scenePath, const char* mediaPath, pt_scene_import** out, PtSceneJsonInfo* info);
void    pt_scene_import_free(pt_scene_import* scene);
/* copies of the imported arrays, up to `capacity` records; return the number available (negative: error) */
int32_t pt_scene_import_cameras(const pt_scene_import* scene, PtSceneCameraDesc* out, uint32_t capacity);
int32_t pt_scene_import_lights(const pt_scene_import* scene, PolymorphicLightInfo* base, PolymorphicLightInfoEx* ex, uint32_t capacity);
/* DirectionalLight leaves (Donut keys color, irradiance, angularSize [deg]) as EMB_DirectionalLight records in WORLD space: ColorIntensity = (color,
   irradiance), Direction = the node's -Z, AngularSize = radians(clamp(angularSize, 0, 90)) (DirectionalLight::FillLightConstants, restated from Donut's
   published sources; the reference reads the same fields: Rtxpt/RTXDI/PrepareLightsPass.cpp:246-249). LightsBaker does not take them (LightsBaker.cpp:600):
   they go through pt_env_bake_lights into pt_set_environment_bake. */
int32_t pt_scene_import_directional_lights(const pt_scene_import* scene, PtEnvDirectionalLight* out, uint32_t capacity);
int32_t pt_scene_import_instances(const pt_scene_import* scene, PtInstanceDesc* out, uint32_t capacity);
int32_t pt_scene_import_geometries(const pt_scene_import* scene, PtGeometryDesc* out, uint32_t capacity);
int32_t pt_scene_import_materials(const pt_scene_import* scene, PTMaterialData* out, uint32_t capacity);
/* the imported vertex positions (object space, three floats per vertex, the order PtGeometryDesc.vertexOffset indexes); returns the vertex count (capacity 0: a query) */
int32_t pt_scene_import_vertices(const pt_scene_import* scene, float* positions, uint32_t capacityVertices);
/* texture `index` (0 .. PtSceneJsonInfo.numTextures - 1; the low 16 bits of a material's texture word) as decoded: RGBA8 texels of the top level; out->pixels points
   into the import object and lives as long as it does */
int32_t pt_scene_import_texture(const pt_scene_import* scene, uint32_t index, PtTextureDesc* out);
/* SampleSettings -> PtSettings as Sample::SceneLoaded applies them (Sample.cpp:613-629): maxBounces, maxDiffuseBounces, textureMIPBias overwrite
   bounceCount, diffuseBounceCount, texLODBias when the scene names them; everything else in *settings is left alone. */
int32_t pt_scene_import_settings(const pt_scene_import* scene, PtSettings* settings);
/* The tone-mapping block after a scene load: Sample::SceneLoaded sets exposureCompensation = 2, exposureValue = 0 ("sensible defaults",
   Sample.cpp:547-549), then — when the scene has a camera — Sample::UpdateCameraFromScene (:467-478) overwrites autoExposure, exposureCompensation,
   exposureValue, exposureValueMin / Max with the camera node's keys or, for a missing key, the ToneMappingParameters DEFAULT (so a camera without
   exposure keys resets the compensation to 0). cameraIndex < 0: the import's selected camera. The other members of *ui are left alone. */
int32_t pt_scene_import_tone_mapping(const pt_scene_import* scene, int32_t cameraIndex, PtToneMappingParameters* ui);
/* pt_set_materials + pt_set_geometry + pt_set_instances + pt_set_lights on ctx (camera, environment and settings stay with the caller) */
int32_t pt_scene_import_apply(pt_context* ctx, const pt_scene_import* scene);

/* Display path. pt_default_tonemap: ToneMappingParameters defaults + UpdateColorTransform with manual exposure
   (Rtxpt/ToneMapper/ToneMappingPasses.h:36-53, ToneMappingPasses.cpp:428-441): exposureCompensation in stops, filmSpeed/shutter/fNumber as in the UI.
   pt_tonemap_color_transform: the same UpdateColorTransform preceded by UpdateWhiteBalanceTransform (ToneMappingPasses.cpp:392-401) — with
   whiteBalance != 0 the von Kries / CAT02 matrix of calculateWhiteBalanceTransformRGB_Rec709(whitePoint in kelvin, 1667..25000;
   Rtxpt/ToneMapper/ColorUtils.h:128-197) is folded into params->colorTransform; with params->autoExposure set the manual exposure factor is 1
   as in the reference. Only colorTransform is written. Host only.
   pt_tonemap: ToneMappingPass::Render into the SRGBA8_UNORM LdrColor target (ToneMapping.ps.hlsli:136-174, RenderTargets.cpp:241) of THIS
   context's accumulation buffer; rgba8 receives width*height*4 bytes (R,G,B,A; rows top to bottom).
   pt_write_png / pt_write_bmp: the screenshot writers behind --captureSimple/--capturePath (Rtxpt/SampleCommon/CaptureScriptManager.cpp:29-60,
   Rtxpt/Sample.cpp:2295); host only, no context needed. */
int32_t pt_default_tonemap(PtToneMapParams* out, float exposureCompensation, float filmSpeed, float shutter, float fNumber);
int32_t pt_tonemap_color_transform(PtToneMapParams* params, uint32_t whiteBalance, float whitePoint, float exposureCompensation, float filmSpeed,
                                   float shutter, float fNumber);
int32_t pt_tonemap(pt_context* ctx, const PtToneMapParams* params, uint8_t* rgba8, size_t bytes);
/* The tone mapper's UI block (ToneMappingParameters, ToneMappingPasses.h:36-53; the reference keeps it in m_ui.ToneMappingParams) and
   ToneMappingPass::PreRender + the constant fill of ::Render on it (ToneMappingPasses.cpp:186-193, 316-348, 373-441): exposureValue is clamped to the
   range shutter x fNumber^2 allows; in aperture priority (exposureMode 0, the default) the shutter is DERIVED as 2^EV / fNumber^2 and `shutter` is
   ignored, in shutter priority fNumber = sqrt(2^EV / shutter); then white balance and exposure scales as in pt_tonemap_color_transform; the
   auto-exposure luminance limits are exp2(exposureValueMin/Max) with auto exposure, exp2(-/+16) without. avgLuminance: pt_average_luminance (only
   read when autoExposure is set); enabled: m_ui.EnableToneMapping. Host only. */
typedef struct PtToneMappingParameters {
    uint32_t exposureMode;                  /* 0 AperturePriority, 1 ShutterPriority */
    uint32_t toneMapOperator;               /* ToneMapperOperator, default Aces (5) */
    uint32_t autoExposure;
    float    exposureCompensation, exposureValue, filmSpeed, fNumber, shutter;
    uint32_t whiteBalance; float whitePoint, whiteMaxLuminance, whiteScale;
    uint32_t clamped;
    float    exposureValueMin, exposureValueMax;
} PtToneMappingParameters;
int32_t pt_default_tone_mapping_parameters(PtToneMappingParameters* out);
int32_t pt_tonemap_from_parameters(const PtToneMappingParameters* ui, float avgLuminance, uint32_t enabled, PtToneMapParams* out);
/* Auto exposure, the luminance capture of ToneMappingPass::Render (ToneMappingPasses.cpp:78-97, 225-288; luminance_ps.hlsl:10-26; capture_cs,
   ToneMapping.hlsl:25-34): log2(max(1e-4, luminance)) of THIS context's accumulation buffer drawn through the linear sampler into a target of
   power-of-two-lowered size, averaged down its mip chain; avgLuminance = exp2(last mip) is what TONEMAPPING_AUTOEXPOSURE_CPU puts into
   ToneMappingConstants::avgLuminance (the reference reads it back with a lag of a few frames; here it is the current image). */
int32_t pt_average_luminance(pt_context* ctx, float* avgLuminance);
/* Float images for the environment source. The reference takes .exr / .hdr / .dds environment maps (Rtxpt/Sample.cpp:116) through Donut's TextureCache
   (EnvMapBaker.cpp:392-415; Donut is not vendored: the formats are read from their published specifications). OpenEXR: single-part scan-line and tiled files (of a mip- / rip-mapped tiled file: level 0) with
   half or float R G B (or Y) channels, compression NONE / RLE / ZIPS / ZIP / PIZ; Radiance .hdr: 32-bit_rle_rgbe, "-Y h +X w". *rgb: width x height x 3 floats, top
   row first (what pt_set_environment takes), allocated by the library, released with pt_image_free. PT_ERROR_IO: unreadable or malformed;
   PT_ERROR_UNSUPPORTED: multi-part / deep EXR, PXR24 / B44 / DWA compression (NONE / RLE / ZIPS / ZIP / PIZ are read), sub-sampled or integer channels, other orientations.
   A 2D .dds with RGBA16F / RGBA32F / BC6H (UF16, SF16) pixels is read too (alpha dropped) — the reference takes BC6H lat-long .dds environments (EnvMapBaker.cpp:399-411);
   8-bit and BC1-5 / BC7 .dds files are textures, not environment sources: pt_image_read_dds. Cube-map .dds files: pt_image_read_dds_cube. */
int32_t pt_image_read_float(const char* path, uint32_t* width, uint32_t* height, float** rgb);
void    pt_image_free(float* rgb);
/* .dds textures: the reference's material pipeline prefers `x.dds` next to `x.png` (MaterialsBaker.cpp:178-191; its compression script writes BC7, SampleCommon.cpp:
   700-730) and Donut's TextureCache decodes them. Top mip level of a 2D .dds as RGBA8 (*format = PT_TEX_RGBA8_UNORM / PT_TEX_RGBA8_SRGB per the file's DXGI format)
   or, for R16G16B16A16_FLOAT / R32G32B32A32_FLOAT files, RGBA32F (*format = PT_TEX_RGBA32F): what PtTextureDesc takes. Block formats BC1 / BC2 / BC3 / BC4 / BC5 /
   BC7, uncompressed RGBA8 / BGRA8 / BGRX8; legacy FourCC and DX10 headers. *pixels is allocated by the library: pt_image_free((float*)pixels).
   PT_ERROR_IO: unreadable / truncated; PT_ERROR_UNSUPPORTED: cube maps (pt_image_read_dds_cube reads those), volumes, arrays, other formats. BC6H (UF16 / SF16, all fourteen modes) decodes to
   PT_TEX_RGBA32F texels. */
int32_t pt_image_read_dds(const char* path, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels);
int32_t pt_image_read_dds_memory(const void* bytes, size_t size, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels);   /* the same from memory (glTF images: MSFT_texture_dds) */
/* a cube-map .dds as an environment source (Sample.cpp:116 lists .dds files of the environment-map folder; EnvMapBaker.cpp:399-411): the top level of all six faces of a
   R16G16B16A16_FLOAT / R32G32B32A32_FLOAT / BC6H (UF16, SF16) cube — legacy DDSCAPS2_CUBEMAP with all faces or a DX10 header with the TEXTURECUBE flag — as
   6 x dim x dim RGBA floats in the file's (= D3D's) face order: what pt_set_environment_cube takes. Released with pt_image_free. PT_ERROR_UNSUPPORTED: a 2D file, a partial
   cube, cube arrays, other formats. */
int32_t pt_image_read_dds_cube(const char* path, uint32_t* dim, float** rgbaFaces);
/* JPEG images of glTF files (Donut's TextureCache gives them to stb_image): a baseline / extended / progressive Huffman stream of 8-bit greyscale or YCbCr (or
   Adobe-RGB) samples in memory -> RGBA8, top row first, alpha 255; *rgba8 is allocated by the library: pt_image_free((float*)rgba8). The samples are the IJG
   reference decoder's (islow IDCT, fancy up-sampling): any conforming decoder, stb_image included, may differ from another by a unit per sample.
   PT_ERROR_IO: not a JPEG this reader handles (arithmetic coding, 12 bits, CMYK, damaged). pt_load_scene_gltf / pt_scene_json_import use it for image/jpeg. */
int32_t pt_image_read_jpeg(const void* bytes, size_t size, uint32_t* width, uint32_t* height, void** rgba8);
int32_t pt_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
int32_t pt_write_bmp(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* --- multi-GPU tile sharding (new; no reference analogue) ------------------------------------------- */
/* number of pixels this shard owns and the packed RGBA32F byte size */
int32_t pt_shard_info(pt_context* ctx, uint32_t* numOwnedPixels, size_t* packedBytes);
/* pack this shard's pixels contiguously into a DEVICE buffer (e.g. a torch tensor) — the send buffer of the RCCL gather */
int32_t pt_pack_shard(pt_context* ctx, void* devicePtrDst, size_t bytes);
/* rank 0: scatter the packed pixels of shard `rank` (DEVICE pointer, as received from the gather) into the full frame */
int32_t pt_unpack_shard(pt_context* ctx, const void* devicePtrSrc, size_t bytes, uint32_t rank);
/* the accumulation buffer as a device pointer (RGBA32F, width*height) for zero-copy consumers */
int32_t pt_device_radiance(pt_context* ctx, void** devicePtr);

/* --- the frame gather itself (north_star: "a single RCCL gather of the radiance buffer over xGMI"; no reference analogue: RTXPT is single-GPU,
 *     Rtxpt/SampleCommon/CommandLine.cpp:41 only selects an adapter). One process per GPU, each with a context created with its shardRank / shardCount.
 *     The collective lives behind the C ABI so that a C++ host needs neither torch nor its own RCCL code:
 *       rank 0:  pt_comm_unique_id(id)  -> the host hands the 128 bytes to the other ranks by its own means (MPI_Bcast, a file, torch.distributed)
 *       all:     pt_comm_init(ctx, id, rank, world)        ncclCommInitRank on the context's device (collective)
 *       frame:   pt_render(...) ; pt_gather(ctx)           every rank's tiles -> rank 0's accumulation buffer
 *     pt_gather packs the rank's tiles and issues UN-PADDED point-to-point transfers (rank r: one ncclSend of its own byte count; rank 0: the
 *     matching ncclRecv's inside one ncclGroupStart/End, then one unpack kernel) on the library's own stream, asynchronously: the next call that
 *     reads the frame (pt_map_radiance, pt_tonemap, ...) is ordered behind it. RCCL is bound at run time (dlopen of the librccl.so already in the
 *     process, else the system one; MI355PT_RCCL_LIB overrides), so the library has no link-time dependency on it. */
#define PT_COMM_ID_BYTES 128
int32_t pt_comm_unique_id(void* id128);
int32_t pt_comm_init(pt_context* ctx, const void* id128, uint32_t rank, uint32_t world);
int32_t pt_comm_destroy(pt_context* ctx);
int32_t pt_gather(pt_context* ctx);
/* host only, no device: the pixel ids (x<<16 | y) `rank` of `world` owns in a width x height frame, in pack order (32x32 tiles in Morton order dealt
 * round-robin, 8x8 blocks inside a tile). count receives the number of owned pixels (also when pixels is NULL or capacity too small). */
int32_t pt_shard_layout(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t* pixels, uint32_t capacity, uint32_t* count);
/* the same gather protocol over HOST memory and a caller-supplied point-to-point transport (MPI, sockets, gloo in the CPU tests): rgba is the
 * full width x height RGBA32F frame of this rank (only its own tiles need to be valid); on return rank 0's frame holds every rank's tiles. */
typedef struct PtTransport {
    void* user;
    int32_t (*send)(void* user, const void* buf, size_t bytes, uint32_t peer);      /* blocking or queued until group_end; 0 = ok */
    int32_t (*recv)(void* user, void* buf, size_t bytes, uint32_t peer);
    int32_t (*group_begin)(void* user);                                             /* may be NULL */
    int32_t (*group_end)(void* user);                                               /* may be NULL */
} PtTransport;
int32_t pt_gather_host(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, float* rgba, const PtTransport* transport);
/* the NEE-AT feedback exchange of tile-sharded frames over HOST memory and the same transport: totalWeight / candidates / depth are this rank's full width x height planes
 * (only its own tiles need to be valid); on return every rank holds every rank's reservoirs and exported depth. Pairs of ranks meet in rank order, the lower one sends first. */
int32_t pt_neeat_exchange_host(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, float* totalWeight, uint32_t* candidates, float* depth, const PtTransport* transport);
/* The same protocol for any set of per-pixel planes in HOST memory (what the CPU tests drive, and what a host with its own fabric can model its transfers on): numPlanes row-major
 * width x height planes of bytesPerPixel[k] bytes per pixel. toRoot = 0: all-to-all — every rank sends the records of its own pixels (the planes' bytes of a pixel back to back, in
 * pt_shard_layout order) to every other rank: the depth / motion-vector exchange of a tile-sharded realtime frame, with three 4-byte planes pt_neeat_exchange_host. toRoot = 1: to
 * rank 0 only — the plane-buffer gather. Un-padded; a rank sends exactly its own bytes. */
int32_t pt_exchange_planes_host(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, void* const* planes, const uint32_t* bytesPerPixel, uint32_t numPlanes, int32_t toRoot, const PtTransport* transport);

/* --- probes used by the parity tests and bench.py (not part of the reference seam) --------------------------------- */
/* closest-hit / any-hit queries through the same BVH + kernels the renderer uses. rays: n x 8 floats (o.xyz,tmin,d.xyz,tmax);
 * closest out: n x 4 (t, prim bits, u, v) with prim 0xFFFFFFFF on miss; visibility out: n x u32 (1 = visible) */
int32_t pt_trace_closest(pt_context* ctx, const float* rays, uint32_t n, float* out, double* kernelMs);
int32_t pt_trace_visibility(pt_context* ctx, const float* rays, uint32_t n, uint32_t* out, double* kernelMs);
/* light table / sub-instance read-back; pass NULL pointers to query sizes */
int32_t pt_get_lights(pt_context* ctx, uint32_t* numLights, uint32_t* numProxies, void* lights32B, void* lightsEx16B, uint32_t* proxyCounters,
                      uint32_t* proxyIndices, uint32_t* envLookup, uint32_t* envLookupDim);
/* the baked environment cube as the kernels sample it (EnvMapBaker::GetEnvMapCube: m_cubemap, or — with pt_set_environment_compression — the decoded texels of m_cubemapBC6H;
 * EnvMapBaker.cpp:298-343, 593-633): 8 bytes per RGBA16F texel, mips one after the
 * other (dim, dim/2 .. 8), face-major (+X -X +Y -Y +Z -Z) within a mip. texels8B may be NULL to query the sizes. */
int32_t pt_get_env_cube(pt_context* ctx, uint32_t* texelCount, uint32_t* dim, uint32_t* mipLevels, void* texels8B, uint32_t capacityTexels);
int32_t pt_get_subinstances(pt_context* ctx, uint32_t* count, void* out32B);
int32_t pt_get_scene_info(pt_context* ctx, uint32_t* numTriangles, uint32_t* numBvhNodes, uint32_t* numInstances, uint32_t* numMaterials);
/* BVH build/refit timing of the last geometry update, milliseconds */
int32_t pt_get_build_stats(pt_context* ctx, double* buildMs, double* refitMs, double* lightBakeMs);
/* which builder made the tree the kernels traverse, and where it ran. builder: 0 = PLOC (device, "prefer fast build"), 1 = Karras radix tree (device, developer A/B),
 * 2 = binned SAH + insertion-based optimisation on the host's cores ("prefer fast trace", topology only; bounds / collapse / refit on the device),
 * 3 = PLOC + parallel re-insertion + cost-driven wide nodes, all on the device ("prefer fast trace"). hostMs: the host part of the last build (0 for device builders). */
typedef struct PtBvhInfo { uint32_t builder; uint32_t builtOnDevice; uint32_t numTriangles; uint32_t numWideNodes; uint32_t collapseLevels; uint32_t optimiserPasses; float hostMs; float buildMs; } PtBvhInfo;
int32_t pt_get_bvh_info(pt_context* ctx, PtBvhInfo* out);
/* enable in-kernel BVH node/triangle counters (slower); default off */
int32_t pt_set_counters(pt_context* ctx, int32_t enable);
/* runtime form of PT_DEVICE_SERIAL_KERNELS: 1 = pt_render uses one batch on one stream (kernels never overlap: clean per-kernel HIP-event /
   rocprofv3 durations), 0 = two pipelined half-frame batches (default) */
int32_t pt_set_serial_kernels(pt_context* ctx, int32_t enable);
/* The tail kernel: once a batch of pt_render holds at most `maxPaths` live paths, ONE launch runs them to their end — every wave loops trace -> shade -> visibility ->
   next bounce over 32 paths, the shape of the reference's raygen loop (Rtxpt/Shaders/PathTracerSample.hlsl:200-250) where it fits: few paths, bound by the length of
   the launch chain of a wavefront pass, not by throughput. 0 = never (every pass is a wavefront pass); default 4096 since round 6 (32768 before: fused traversal launches made passes of tens of thousands of paths cheaper than the tail kernel's under-filled GPU; the chains of tiny passes that nested dielectrics leave are what it is for; environment MI355PT_TAIL_PATHS overrides it at
   pt_create). The image does not depend on the value (paths do not interact; tests render whole frames through the tail kernel). Ignored for NEEFullSamples > 1,
   serial-kernel and counter frames. */
int32_t pt_set_tail_paths(pt_context* ctx, uint32_t maxPaths);
/* Fused traversal launches: the visibility rays of path vertex k (Bridge::traceVisibilityRay, PathTracerNEE.hlsli:185-275) are traced in the same launch as the closest-hit rays of
   vertex k + 1 (Bridge::traceScatterRay) — blocks of either kind side by side, straggler rounds and resolve passes shared — instead of in a launch of their own; their contributions
   land before vertex k + 1 is shaded, as before, so the image does not depend on the mode (tests/test_gpu_fused_traversal.py). Launch composition only: it halves the traversal
   launches of a bounce, which is what a small frame (one rank of a tile-sharded frame) is bound by. mode 0 = off, 1 = on, 2 = by the size of the pt_render call (default; environment
   MI355PT_FUSED_TRAVERSAL overrides it at pt_create). Ignored for NEEFullSamples > 1, serial-kernel and counter frames. */
int32_t pt_set_fused_traversal(pt_context* ctx, uint32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
