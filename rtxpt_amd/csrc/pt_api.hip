// mi355pt — C-ABI implementation (include/mi355pt.h): context, scene upload, BVH build/refit orchestration, light baking,
// the wavefront frame loop and read-back. Host side of the seam `Sample` implements in the reference
// (Rtxpt/Sample.cpp:1891-2313 Render, :2438-2559 PathTrace, :1464-1556 UpdatePathTracerConstants, :2770-2778 accumulation).
// There is NO CPU fallback: every entry point that needs the device fails with PT_ERROR_NO_DEVICE / PT_ERROR_HIP.
#include "../../include/mi355pt.h"
#ifdef MI355PT_TEST_HOOKS
#include "../../include/mi355pt_testhooks.h"
#endif
#include "pt_wavefront.h"
#include "pt_stableplanes_launch.h"
#include "pt_build.h"
#include <rocprim/rocprim.hpp>
#include <rccl/rccl.h>      // types only: the functions are bound at run time (dlopen), see pt_comm_init
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

using namespace ptk;

static_assert(sizeof(::PTMaterialData) == sizeof(ptk::PTMaterialData), "material ABI");
static_assert(sizeof(::PathTracerCameraData) == sizeof(ptk::PathTracerCameraData), "camera ABI");
static_assert(sizeof(::PtSettings) == sizeof(ptk::PtSettings), "settings ABI");
static_assert(sizeof(::PtGeometryDesc) == sizeof(ptk::GeometryDesc), "geometry ABI");
static_assert(sizeof(::PtInstanceDesc) == sizeof(ptk::InstanceDesc), "instance ABI");
static_assert(sizeof(::PolymorphicLightInfo) == sizeof(ptk::PolymorphicLightInfo), "light ABI");

namespace {

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    hipError_t resize(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc(&p, sizeof(T) * (count ? count : 1));
        if (e == hipSuccess) n = count ? count : 1;
        return e;
    }
    hipError_t upload(const T* src, size_t count, hipStream_t st) {
        hipError_t e = resize(count); if (e != hipSuccess) return e;
        if (!count) return hipSuccess;
        return hipMemcpyAsync(p, src, sizeof(T) * count, hipMemcpyHostToDevice, st);
    }
    hipError_t upload(const std::vector<T>& v, hipStream_t st) { return upload(v.data(), v.size(), st); }
    void free() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct HostTexture { uint w, h, mipLevels; std::vector<std::vector<ptk::float4>> mips; };

static const uint TILE = 32;
static const uint TASK_QUEUE_CAPACITY = 1u << 22;      // sub-tree tasks per queue (2 queues per pipelined batch, 16 B each)
#ifndef PT_SHARD_TILE_GROUP
#define PT_SHARD_TILE_GROUP 1      // consecutive Morton-ordered 32x32 tiles dealt to the same rank (locality vs load balance)
#endif
#ifndef PT_PIPELINE_FULL_AT
// paths per pt_render call from which all PT_PIPELINE_BATCHES are used (one rank of an 8-way sharded 4K frame has 4.1 M)
#define PT_PIPELINE_FULL_AT (1u << 21)
#endif
#ifndef PT_PIPELINE_MID_BATCHES
#define PT_PIPELINE_MID_BATCHES 2      // batches between 1 M paths and PT_PIPELINE_FULL_AT
#endif
#ifndef PT_CLASSIFY_FROM
// passes with fewer paths skip k_classify (class-ordered shading pays through coherence, which a handful of waves do not have)
#define PT_CLASSIFY_FROM 65536u
#endif
#ifndef PT_SP_FILL_CLASSES
#define PT_SP_FILL_CLASSES 1     // the stable-plane fill pass shades in class order (k_classify), like reference mode; 0: queue order (A/B)
#endif
#ifndef PT_SP_FILL_RANGED
// the fill pass's first traversal launch uses FirstHitFromVBuffer's narrowed ray interval (pt_stableplanes.h firstHitInterval); 0: the whole ray (A/B) — same
// hits
#define PT_SP_FILL_RANGED 1
#endif
#ifndef PT_TAIL_PATHS
// a batch with at most this many live paths is finished by the tail kernel (pt_tail.hip, pt_set_tail_paths); 0: never. 32768 in rounds 4-5; with fused
// traversal
#define PT_TAIL_PATHS 4096u
                                  // launches and free-running small passes (round 6) a pass of tens of thousands of paths is cheaper as a wavefront pass than
                                  // in the tail kernel's under-filled GPU (rank of eight 12.56 -> 12.11 ms without it), while the chains of passes that hold a
                                  // few hundred paths each — nested-dielectric re-traces: C5 runs 19 passes, twelve of them below 10 k paths at ~0.25 ms each —
                                  // are what the kernel is for: 4096 takes C5's rank of eight 13.9 -> 13.2 ms, C3's 12.0 -> 11.8
                                  // (profiles/r06o_tail_small_ab.txt)
#endif
#ifndef PT_FUSED_TRAVERSAL
// pt_set_fused_traversal (default: on — it pays at every size, profiles/r06b_fused_traversal_ab.txt): 0 = every bounce traces its visibility rays in a launch
// of their own, 1 = together with the closest-hit rays of the next bounce
#define PT_FUSED_TRAVERSAL 1u
#endif                              // (k_trace_pair, pt_wavefront.hip), 2 = by the size of the call (PT_FUSED_BELOW)
#ifndef PT_FUSED_BELOW
// mode 2: calls of fewer paths than this fuse (one rank of a 4- or 8-way sharded 4K frame, 1080p frames); a full 4K x 4 spp frame (33 M) keeps its own launches
#define PT_FUSED_BELOW (12u << 20)
#endif
#ifndef PT_COMPACT_POOL
#define PT_COMPACT_POOL 1      // pt_render keeps the live paths' state compacted by queue position (ptk::PathPool::home; environment MI355PT_COMPACT_POOL overrides)
#endif
#ifndef PT_FREE_RUN_BELOW
// pt_render: once every live batch holds fewer paths than this, the batches stop advancing in lockstep (0: lockstep to the end)
#define PT_FREE_RUN_BELOW (1u << 22)
#endif
#ifndef PT_PIPELINE_BATCHES
#define PT_PIPELINE_BATCHES 4      // independent sub-frame batches pt_render keeps in flight on separate streams (A/B on C3 in DESIGN.md)
#endif
// a full task queue is reported as an error (pt_render), it does not silently disable splitting
} // namespace

struct pt_context {
    int device = 0; hipStream_t stream = nullptr; uint shardRank = 0, shardCount = 1;
    hipStream_t streams[PT_PIPELINE_BATCHES] = {}; WaveCounters* hostCounters = nullptr; bool serialKernels = false; uint tailBelow = PT_TAIL_PATHS, tailDefer = 0, fusedTraversal = PT_FUSED_TRAVERSAL; bool compactPool = PT_COMPACT_POOL != 0;   // second half-frame batch (pt_render pipelines two batches)
    std::string lastError;
    // host copies of the scene (kept for re-bake / animation)
    std::vector<uint> indices; std::vector<float> positions; std::vector<ptk::float2> uvs; std::vector<uint> normals, tangents;
    std::vector<GeometryDesc> geometries; std::vector<MeshDesc> meshes; std::vector<InstanceDesc> instances;
    std::vector<ptk::PTMaterialData> materials; std::vector<HostTexture> textures; HostTexture envTex; bool envEnabled = false;
    float3x4 envToWorld, envToLocal; ptk::float3 envColorMul;
    // sceneDirLights: world-space lights of the loaded scene (pt_set_scene_directional_lights), converted at bake time
    uint envCubeDim = 2048; std::vector<ptk::EnvDirectionalLight> envDirLights, sceneDirLights; bool envCubeDirty = true;
    ptk::EnvCube envCube;      // EnvMapBaker state (pt_set_environment_bake)
    std::vector<PolymorphicLightInfoFull> analyticLights;
    std::vector<SubInstanceData> subInstances; std::vector<ptk::uint2> subInstToInstGeom; std::vector<ptk::uint2> primInfo; std::vector<uint> subInstFirstPrim;
    std::vector<ptk::PolymorphicLightInfo> lights; std::vector<ptk::PolymorphicLightInfoEx> lightsEx; std::vector<uint> envLookup; uint envLookupDim = 0; uint numProxies = 0, envLightsBaked = 0;      // (light weights / proxy table live on the device only)
    DevBuf<float> dLightW; DevBuf<uint> dProxyOffsets; void* dScanTemp = nullptr; size_t scanTempBytes = 0;
    // device
    DevBuf<uint> dIndices, dNormals, dTangents, dProxyCounters, dProxyIndices, dEnvLookup, dOwned, dQueue[2], dEmissiveList, dEmissiveOffsets;
    // pt_set_motion_history: the previous frame's pose; the (first, count) vertex ranges in which it differs from the current one
    DevBuf<float> dPrevPositions; DevBuf<InstanceDesc> dPrevInstances; bool motionHistory = false, prevAllStale = false; std::vector<uint32_t> prevStaleRanges;
    DevBuf<float> dPositions; DevBuf<ptk::float2> dUvs; DevBuf<GeometryDesc> dGeometries; DevBuf<InstanceDesc> dInstances; DevBuf<SubInstanceData> dSubInstances;
    DevBuf<ptk::AlphaPlane> dAlphaPlanes; DevBuf<unsigned char> dAlphaPool; DevBuf<ptk::ShadeTri> dShadeTris; DevBuf<ptk::uint2> dSubInstToInstGeom, dPrimInfo; DevBuf<ptk::PTMaterialData> dMaterials; DevBuf<TexInfo> dTexInfos; DevBuf<ptk::float4> dTexels;
    bool skyEnabled = false; ptk::ProceduralSkyContext sky; DevBuf<ptk::float4> dSkyTex[4]; DevBuf<ptk::ProceduralSkyContext> dSky; DevBuf<ptk::uint2> dSkyLowRes;      // pt_set_procedural_sky
    // pt_set_environment_cube: the environment image as a cube map (RGBA16F), uploaded by the setter
    DevBuf<ptk::uint2> dEnvImageCube; uint envImageCubeDim = 0;
    // envCompression: EnvMapBaker's BC6U compression (0 off, 1 fast)
    DevBuf<ptk::uint2> dEnvCube, dEnvCubeSource; DevBuf<ptk::EnvDirectionalLight> dEnvDirLights; uint envCompression = 0;
    DevBuf<ptk::PolymorphicLightInfo> dLights; DevBuf<ptk::PolymorphicLightInfoEx> dLightsEx;
    // NEE-AT (pt_set_local_light_sampling): the screen-tile local samplers as the host hands them in, and the feedback reservoirs of the last pt_render call
    // (one plane per sample)
    DevBuf<uint> dLocalTable; uint localResX = 0, localResY = 0, localJitterX = 0, localJitterY = 0, localMaxLight = 0; float localRatio = 0.f, sscThreshold = 0.f; bool feedbackRequired = false;
    DevBuf<float> dFbWeight; DevBuf<uint> dFbCand; DevBuf<ptk::float4> dSq3; uint fbSamples = 0;
    ptk::LightFrustumBoost lightBoost = {}; bool weightsDirty = false;      // pt_set_light_importance_boost: ImportanceBooster's frustum term (mul 0: off)
    // NEE-AT with the baker in the loop (pt_set_neeat): what LightsBaker keeps between frames (LightsBaker.h:225-260) and the textures / buffers its feedback
    // passes bind
    struct NeeAt {
        bool enabled = false; float globalFeedbackWeight = 0.75f, localRatio = 0.65f, sscThreshold = 0.3f, dropoff = 0.005f, intensityDeltaMul = 64.0f; bool preFilter = true;
        uint updateCounter = 0; float jitterF[2] = {0, 0}; uint jitter[2] = {0, 0}, prevJitter[2] = {0, 0};
        bool feedbackFilled = false, lastFeedbackAvailable = false; uint historicTotalLightCount = 0, W = 0, H = 0, nHist = 0;
        // between UpdateBegin and UpdateEnd of a frame (realtime mode: the build pass runs in between)
        bool frameOpen = false, frameFeedbackAvailable = false, frameLocalAvailable = false, exportDepth = true; uint framePrevLightCount = 0;
        DevBuf<float> fbW, scW, blW, snapW, curW, histW; DevBuf<uint> fbC, scC, blC, snapC, local, counters;
        // the exported depth of the last traced frame / of the one before; columns 2 and 3 of pt_set_view_projection's matrix
        DevBuf<float> depth, histDepth; bool haveClip = false; float clipZ[4] = {0, 0, 0, 0}, clipW[4] = {0, 0, 0, 0};
        // tile-sharded frames: the exchange of the owned pixels' reservoirs between frames
        DevBuf<uint> xSend, xRecv; DevBuf<uint> xPixels; uint xW = 0, xH = 0;
        void reset() { W = H = 0; updateCounter = 0; jitterF[0] = jitterF[1] = 0; jitter[0] = jitter[1] = prevJitter[0] = prevJitter[1] = 0; feedbackFilled = lastFeedbackAvailable = false; frameOpen = false; exportDepth = true; historicTotalLightCount = 0; W = H = 0; nHist = 0; }
        void free() { fbW.free(); scW.free(); blW.free(); snapW.free(); curW.free(); histW.free(); fbC.free(); scC.free(); blC.free(); snapC.free(); local.free(); counters.free(); xSend.free(); xRecv.free(); xPixels.free(); depth.free(); histDepth.free(); }
    } neeat;
    DevBuf<ptk::uint4> dS0, dS1, dS2, dS3, dS4, dHit, dS0b, dS1b, dS3b, dS4b, dHitb /* ...b: the second array set of a compacted pool (pt_render) */; DevBuf<ptk::float4> dSq0, dSq1, dSq2, dAccum, dScratch4; DevBuf<WaveCounters> dCounters; DevBuf<ptk::uint2> dTravSpill; DevBuf<ptk::TravTask> dTaskQ; DevBuf<uint> dTravCounts, dResolveList, dResolveListSh; DevBuf<unsigned long long> dBestKey, dBestKeySh; DevBuf<ptk::TravTask> dTaskQSh;      // ...Sh: the visibility rays' own straggler state in a frame of fused traversal launches
    std::vector<TexInfo> texInfos; TexInfo envTexInfo;
    BvhBuildBuffers bvh; bool bvhAllocated = false; uint numTris = 0; uint bvhBuilder = BVH_BUILDER_SAH;
    DeviceScene dsc;
    // frame state
    ptk::PtSettings S; ptk::PathTracerCameraData cam; uint width = 0, height = 0, accumCount = 0; std::vector<uint> owned; std::vector<std::vector<uint>> shardPixels;
    std::vector<float> hostRadiance; bool countersEnabled = false;
    bool geomDirty = true, lightsDirty = true, texDirty = true;
    double buildMs = 0, refitMs = 0, lightBakeMs = 0;
    uint poolCapacity = 0; size_t shadowCapacity = 0;
    // stable planes (pt_build_stable_planes): the realtime mode's per-frame buffers (RenderTargets.cpp:60-141, 340-352) of the last pre-pass
    DevBuf<uint> dSpHeader, dSpThroughput; DevBuf<ptk::StablePlane> dSpPlanes; DevBuf<ptk::uint2> dSpRadiance, dSpMotion; DevBuf<float> dSpDepth, dSpHitT; uint spW = 0, spH = 0; DevBuf<ptk::uint4> dSpMark; DevBuf<ptk::float4> dSpNewL; DevBuf<float> dSpScratch; DevBuf<uint> dSpGatherSend, dSpGatherRecv, dSpGatherPixels; uint spGatherW = 0, spGatherH = 0; bool spGathered = false;      // (the last two: scratch of the fill passes)
    // frame gather (pt_comm_init / pt_gather)
    ncclComm_t comm = nullptr; uint commRank = 0, commWorld = 0; DevBuf<ptk::float4> dGatherSend, dGatherRecv; DevBuf<uint> dGatherPixels; std::vector<size_t> gatherCounts; uint gatherW = 0, gatherH = 0;
};

namespace {

int fail(pt_context* c, int code, const std::string& msg) { if (c) c->lastError = msg; return code; }
#define PT_CHECK_HIP(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(c, PT_ERROR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

void build_mips(HostTexture& t) {
    uint lv = 1; { uint m = std::max(t.w, t.h); while (m > 1) { m >>= 1; lv++; } }
    t.mipLevels = lv; t.mips.resize(lv);
    for (uint l = 1; l < lv; l++) {
        uint pw = std::max(1u, t.w >> (l - 1)), ph = std::max(1u, t.h >> (l - 1));
        uint mw = std::max(1u, t.w >> l), mh = std::max(1u, t.h >> l);
        t.mips[l].resize((size_t)mw * mh);
        const std::vector<ptk::float4>& p = t.mips[l - 1];
        for (uint y = 0; y < mh; y++) for (uint x = 0; x < mw; x++) {
            uint x0 = std::min(2 * x, pw - 1), x1 = std::min(2 * x + 1, pw - 1), y0 = std::min(2 * y, ph - 1), y1 = std::min(2 * y + 1, ph - 1);
            ptk::float4 s = (p[(size_t)y0 * pw + x0] + p[(size_t)y0 * pw + x1]) + (p[(size_t)y1 * pw + x0] + p[(size_t)y1 * pw + x1]);
            t.mips[l][(size_t)y * mw + x] = s * 0.25f;
        }
    }
}
float srgb_to_linear(float c) { return (c <= 0.04045f) ? c / 12.92f : dm_pow((c + 0.055f) / 1.055f, 2.4f); }

uint morton2(uint x, uint y) {
    auto part = [](uint v) { v &= 0xFFFF; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v; };
    return part(x) | (part(y) << 1);
}
// pixel ownership: 32x32 tiles, tile t -> rank morton(t) % shardCount; inside a tile pixels are listed in 8x8 blocks so that the
// 64 lanes of a wave start as an 8x8 screen block (coherent primary rays)
void shard_pixel_lists(uint width, uint height, uint world, std::vector<std::vector<uint>>& out) {
    out.assign(world, std::vector<uint>());
    uint tx = (width + TILE - 1) / TILE, ty = (height + TILE - 1) / TILE;
    std::vector<std::pair<uint, uint>> tiles;
    for (uint y = 0; y < ty; y++) for (uint x = 0; x < tx; x++) tiles.push_back({morton2(x, y), y * tx + x});
    std::sort(tiles.begin(), tiles.end());
    uint order = 0;
    for (auto& t : tiles) {
        uint tX = t.second % tx, tY = t.second / tx;
        std::vector<uint>& dst = out[(order / PT_SHARD_TILE_GROUP) % world];
        order++;
        for (uint by = 0; by < TILE; by += 8) for (uint bx = 0; bx < TILE; bx += 8)
            for (uint y = 0; y < 8; y++) for (uint x = 0; x < 8; x++) {
                uint px = tX * TILE + bx + x, py = tY * TILE + by + y;
                if (px < width && py < height) dst.push_back((px << 16) | py);
            }
    }
}
void build_shards(pt_context* c) {
    shard_pixel_lists(c->width, c->height, c->shardCount, c->shardPixels);
    c->owned = c->shardPixels[c->shardRank];
    // pt_gather's (and the stable-plane guide exchange's) per-rank counts and pixel lists follow the shard lists, not only the frame size
    c->gatherW = c->gatherH = 0; c->spGatherW = c->spGatherH = 0;
}

// ---- RCCL, bound at run time. One process may already hold a librccl.so (PyTorch ships its own): RTLD_NOLOAD finds that copy first, so that a single
// RCCL runs on the single HIP runtime of the process; a plain C++ host gets the system library.
struct RcclApi {
    void* lib = nullptr; bool tried = false; std::string error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (tried) return lib != nullptr;
        tried = true;
        const char* env = getenv("MI355PT_RCCL_LIB");
        if (env && *env) lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { const char* e = dlerror(); error = std::string("librccl.so not found: ") + (e ? e : ""); return false; }
#define PT_RCCL_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { error = std::string("librccl.so lacks ") + name; lib = nullptr; return false; }
        PT_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") PT_RCCL_SYM(CommInitRank, "ncclCommInitRank") PT_RCCL_SYM(CommDestroy, "ncclCommDestroy") PT_RCCL_SYM(Send, "ncclSend")
        PT_RCCL_SYM(Recv, "ncclRecv") PT_RCCL_SYM(GroupStart, "ncclGroupStart") PT_RCCL_SYM(GroupEnd, "ncclGroupEnd") PT_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PT_RCCL_SYM
        return true;
    }
};
RcclApi g_rccl;
#define PT_CHECK_NCCL(c, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(c, PT_ERROR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)

int upload_textures(pt_context* c) {
    std::vector<ptk::float4> pool; c->texInfos.clear();
    auto add = [&](const HostTexture& t) {
        TexInfo ti; memset(&ti, 0, sizeof(ti)); ti.w = t.w; ti.h = t.h; ti.mipLevels = t.mipLevels; ti.base = pool.size();
        size_t off = 0;
        for (uint l = 0; l < t.mipLevels && l < 16; l++) { ti.mipOffset[l] = (uint)off; off += t.mips[l].size(); pool.insert(pool.end(), t.mips[l].begin(), t.mips[l].end()); }
        return ti;
    };
    for (auto& t : c->textures) c->texInfos.push_back(add(t));
    memset(&c->envTexInfo, 0, sizeof(TexInfo));
    if (c->envEnabled && c->envTex.w) c->envTexInfo = add(c->envTex);      // (a procedural sky alone has no image)
    // alpha planes (pt_scene.h AlphaPlane): the opacity channel of every texture's mip 0 on its own, a byte per texel where every value is k / 255
    std::vector<ptk::AlphaPlane> planes(std::max<size_t>(1, c->textures.size())); std::vector<unsigned char> apool;
    memset(planes.data(), 0, sizeof(ptk::AlphaPlane) * planes.size());
    for (size_t ti = 0; ti < c->textures.size(); ti++) {
        const HostTexture& t = c->textures[ti]; const std::vector<ptk::float4>& m0 = t.mips[0];
        bool bytes = true;
        for (size_t k = 0; k < m0.size() && bytes; k++) { const float a = m0[k].w; const int q = (a >= 0.f && a <= 1.f) ? (int)(a * 255.0f + 0.5f) : -1; bytes = q >= 0 && (float)q / 255.0f == a; }
        while (apool.size() % 16u) apool.push_back(0);
        if (apool.size() + m0.size() * (bytes ? 1u : 4u) > 0xFFFFFFF0ull) return fail(c, PT_ERROR_UNSUPPORTED, "alpha planes above 4 GB are not supported");
        if (t.w > 0xFFFFu || t.h > 0xFFFFu) return fail(c, PT_ERROR_UNSUPPORTED, "textures larger than 65535 texels on a side are not supported (the traversal's alpha test packs the size into 16 + 16 bits)");
        ptk::AlphaPlane& ap = planes[ti]; ap.wh = (t.w & 0xFFFFu) | (t.h << 16); ap.offset = (uint)apool.size(); ap.fmt = bytes ? 0u : 1u;
        if (bytes) for (size_t k = 0; k < m0.size(); k++) apool.push_back((unsigned char)(int)(m0[k].w * 255.0f + 0.5f));
        else { const size_t at = apool.size(); apool.resize(at + 4u * m0.size()); for (size_t k = 0; k < m0.size(); k++) memcpy(&apool[at + 4u * k], &m0[k].w, 4); }
    }
    while (apool.size() % 16u || apool.empty()) apool.push_back(0);
    PT_CHECK_HIP(c, c->dAlphaPlanes.upload(planes, c->stream)); PT_CHECK_HIP(c, c->dAlphaPool.upload(apool, c->stream));
    PT_CHECK_HIP(c, c->dTexels.upload(pool, c->stream));
    PT_CHECK_HIP(c, c->dTexInfos.upload(c->texInfos, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    c->texDirty = false;
    return PT_OK;
}

void refresh_scene_view(pt_context* c) {
    DeviceScene& d = c->dsc;
    d.travSpill = c->dTravSpill.p;                           // allocated (and checked) by finalize_geometry
    d.indices = c->dIndices.p; d.positions = c->dPositions.p; d.uvs = c->dUvs.p; d.normals = c->dNormals.p; d.tangents = c->dTangents.p;
    d.geometries = c->dGeometries.p; d.instances = c->dInstances.p; d.subInstances = c->dSubInstances.p; d.subInstToInstGeom = c->dSubInstToInstGeom.p;
    const bool prevPose = c->motionHistory && c->dPrevPositions.p && c->dPrevInstances.p && c->dPrevPositions.n >= c->positions.size() && c->dPrevInstances.n >= c->instances.size();
    d.prevPositions = prevPose ? c->dPrevPositions.p : nullptr; d.prevInstances = prevPose ? c->dPrevInstances.p : nullptr;
    d.materials = c->dMaterials.p; d.materialCount = (uint)c->materials.size(); d.textures = c->dTexInfos.p; d.texels = c->dTexels.p;
    d.envCube = c->envCube; d.envCube.texels = c->dEnvCube.p;
    d.sky = c->skyEnabled ? c->dSky.p : nullptr; memset(&d.skyLowRes, 0, sizeof(d.skyLowRes)); d.skyLowRes.texels = c->dSkyLowRes.p; d.skyLowRes.dim = c->envCubeDim / 2u; d.skyLowRes.mipLevels = 1u;
    d.envCubeSource = d.envCube; if (c->envCompression && c->dEnvCubeSource.p) d.envCubeSource.texels = c->dEnvCubeSource.p;
    d.envImageCube = c->envImageCubeDim ? c->dEnvImageCube.p : nullptr; d.envImageCubeDim = c->envImageCubeDim; d._padEnvImageCube = 0u;
    d.envTex = c->envTexInfo; d.envEnabled = c->envEnabled ? 1u : 0u; d.envToWorld = c->envToWorld; d.envToLocal = c->envToLocal; d.envColorMultiplier = c->envColorMul;
    d.lights.Lights = c->dLights.p; d.lights.LightsEx = c->dLightsEx.p; d.lights.ProxyCounters = c->dProxyCounters.p; d.lights.ProxyIndices = c->dProxyIndices.p;
    d.lights.TotalLightCount = (uint)c->lights.size(); d.lights.SamplingProxyCount = c->numProxies;
    d.lights.EnvLookupMap = c->dEnvLookup.p; d.lights.EnvLookupDim = c->envLookupDim; d.lights.EnvToWorld = c->envToWorld; d.lights.WorldToEnv = c->envToLocal;
    d.lights.LocalSamplingBuffer = c->localResX ? (c->neeat.enabled ? c->neeat.local.p : c->dLocalTable.p) : nullptr; d.lights.LocalResX = c->localResX; d.lights.LocalResY = c->localResY; d.lights.LocalJitterX = c->localJitterX; d.lights.LocalJitterY = c->localJitterY;
    d.lights.DepthExport = (c->neeat.enabled && c->neeat.exportDepth && c->neeat.haveClip && c->neeat.W == c->width && c->neeat.H == c->height) ? c->neeat.depth.p : nullptr; d.lights.DepthWidth = c->width;
    memcpy(d.lights.ClipZ, c->neeat.clipZ, 16); memcpy(d.lights.ClipW, c->neeat.clipW, 16);
    d.lights.LocalToGlobalSampleRatio = c->localResX ? c->localRatio : 0.f; d.lights.ScreenSpaceVsWorldSpaceThreshold = c->sscThreshold; d.lights.TemporalFeedbackRequired = c->feedbackRequired ? 1u : 0u;
    d.nodes = c->bvh.bvh2Stale ? nullptr : c->bvh.nodes; d.nodes8 = c->bvh.nodes8; d.tris = c->bvh.triSorted; d.alphaRecs = c->bvh.alphaRecs; d.alphaPlanes = c->dAlphaPlanes.p; d.alphaPool = c->dAlphaPool.p; d.shadeTris = c->dShadeTris.p; d.primToSlot = c->bvh.primToSlot; d.primInfo = c->dPrimInfo.p; d.numTris = c->numTris; d.rootIsValid = c->numTris ? 1u : 0u;
}

// SubInstanceData fill (Rtxpt/Materials/MaterialsBaker.cpp:960-1017) + primitive table; then GPU LBVH build
// (pt_set_motion_history, below) previous pose <- current pose, on the device
static int32_t motion_history_sync(pt_context* c) {
    if (!c->motionHistory) return PT_OK;
    const bool fresh = c->dPrevPositions.n < c->positions.size() || c->dPrevInstances.n < c->instances.size() || !c->dPrevPositions.p || !c->dPrevInstances.p;
    PT_CHECK_HIP(c, c->dPrevPositions.resize(c->positions.size())); PT_CHECK_HIP(c, c->dPrevInstances.resize(c->instances.size()));
    if (c->instances.size()) PT_CHECK_HIP(c, hipMemcpyAsync(c->dPrevInstances.p, c->dInstances.p, sizeof(InstanceDesc) * c->instances.size(), hipMemcpyDeviceToDevice, c->stream));
    if (fresh || c->prevAllStale) { if (c->positions.size()) PT_CHECK_HIP(c, hipMemcpyAsync(c->dPrevPositions.p, c->dPositions.p, 4 * c->positions.size(), hipMemcpyDeviceToDevice, c->stream)); }
    else for (size_t r = 0; r + 1 < c->prevStaleRanges.size(); r += 2) {
        const size_t first = 3 * (size_t)c->prevStaleRanges[r], count = 3 * (size_t)c->prevStaleRanges[r + 1];
        if (count) PT_CHECK_HIP(c, hipMemcpyAsync(c->dPrevPositions.p + first, c->dPositions.p + first, 4 * count, hipMemcpyDeviceToDevice, c->stream));
    }
    c->prevAllStale = false; c->prevStaleRanges.clear();
    return PT_OK;
}
int finalize_geometry(pt_context* c) {
    c->subInstances.clear(); c->subInstToInstGeom.clear(); c->primInfo.clear(); c->subInstFirstPrim.clear();
    for (size_t i = 0; i < c->instances.size(); i++) {
        if (c->instances[i].meshIndex >= c->meshes.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "instance references a missing mesh");
        const MeshDesc& m = c->meshes[c->instances[i].meshIndex];
        for (uint g = 0; g < m.numGeometries; g++) {
            uint gi = m.firstGeometry + g;
            if (gi >= c->geometries.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "mesh references a missing geometry");
            const GeometryDesc& gd = c->geometries[gi];
            if (gd.materialIndex >= c->materials.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "geometry references a missing material");
            const ptk::PTMaterialData& mat = c->materials[gd.materialIndex];
            SubInstanceData si; memset(&si, 0, sizeof(si));
            bool alphaTested = (gd.geomFlags & GEOMF_ALPHA_TESTED) && (mat.Flags & PTMaterialFlags_UseBaseOrDiffuseTexture) && (gd.flags & GEOM_HAS_UV);
            if (alphaTested) {
                si.FlagsAndAlphaInfo |= SubInstanceData::Flags_AlphaTested;
                uint cutoff = (uint)(saturate(mat.AlphaCutoff) * 255.0f);
                si.FlagsAndAlphaInfo |= (cutoff & 0xFFu) << SubInstanceData::Flags_AlphaOffsetOffset;
                si.FlagsAndAlphaInfo |= (mat.BaseOrDiffuseTextureIndex & 0xFFFFu);
            }
            if (gd.geomFlags & GEOMF_EXCLUDE_FROM_NEE) si.FlagsAndAlphaInfo |= SubInstanceData::Flags_ExcludeFromNEE;
            si.GlobalGeometryIndex_PTMaterialDataIndex = (gi << 16) | (gd.materialIndex & 0xFFFFu);
            si.EmissiveLightMappingOffset = 0xFFFFFFFFu; si.AnalyticProxyLightIndex = 0xFFFFFFFFu;
            si.IndexOffset = gd.indexOffset; si.TexCoord1Offset = gd.vertexOffset;
            uint subInst = (uint)c->subInstances.size();
            c->subInstToInstGeom.push_back(ptk::make_uint2((uint)i, gi));
            c->subInstances.push_back(si); c->subInstFirstPrim.push_back((uint)c->primInfo.size());
            for (uint t = 0; t < gd.numIndices / 3; t++) c->primInfo.push_back(ptk::make_uint2(subInst, t));
        }
    }
    // traversal addresses triangles with 32-bit byte offsets (48 B records) and 28-bit leaf references
    if (c->primInfo.size() > 0x7FFFFFFull / 3ull * 2ull)
        return fail(c, PT_ERROR_UNSUPPORTED, "more than 89 M triangles per scene are not supported by the traversal kernels");
    c->numTris = (uint)c->primInfo.size();
    hipStream_t st = c->stream;
    PT_CHECK_HIP(c, c->dIndices.upload(c->indices, st)); PT_CHECK_HIP(c, c->dPositions.upload(c->positions, st)); PT_CHECK_HIP(c, c->dUvs.upload(c->uvs, st));
    PT_CHECK_HIP(c, c->dNormals.upload(c->normals, st)); PT_CHECK_HIP(c, c->dTangents.upload(c->tangents, st));
    PT_CHECK_HIP(c, c->dGeometries.upload(c->geometries, st)); PT_CHECK_HIP(c, c->dInstances.upload(c->instances, st));
    PT_CHECK_HIP(c, c->dSubInstances.upload(c->subInstances, st)); PT_CHECK_HIP(c, c->dSubInstToInstGeom.upload(c->subInstToInstGeom, st));
    PT_CHECK_HIP(c, c->dPrimInfo.upload(c->primInfo, st)); PT_CHECK_HIP(c, c->dMaterials.upload(c->materials, st));
    // a new scene: its history starts here (previous = current)
    if (c->motionHistory) { c->prevAllStale = true; c->prevStaleRanges.clear(); int r = motion_history_sync(c); if (r != PT_OK) return r; }
    if (c->bvhAllocated && c->bvh.capacity < c->numTris) { bvh_free(c->bvh); c->bvhAllocated = false; }
    if (!c->bvhAllocated) { PT_CHECK_HIP(c, bvh_alloc(c->bvh, c->numTris)); c->bvhAllocated = true; }
    c->bvh.builder = c->bvhBuilder;
    // BVH_BUILDER_PLOC_OPT: parallel re-insertion passes (MI355X, C3: 8 passes = 1463 Mrays/s in 63 ms, 12 = 1477 in 81 ms, 16 = 1476 in 98 ms; host SAH +
    // re-insertion 1479 in 1824 ms — profiles/r03k_device_reinsertion_sweep.txt)
    { const char* e = getenv("MI355PT_REINSERT_PASSES"); c->bvh.riPasses = e ? (uint)atoi(e) : 12u; }
    PT_CHECK_HIP(c, c->dShadeTris.resize(c->numTris));
    // traversal stack tails, one region per pipelined batch (4 x 302 MB of 288 GB)
    if (!c->dTravSpill.p) PT_CHECK_HIP(c, c->dTravSpill.resize(PT_PIPELINE_BATCHES * (size_t)T8_MAX_BLOCKS * T8_GROUPS_PER_BLOCK * T8_SPILL_DEPTH));
    refresh_scene_view(c);
    hipEvent_t e0, e1; PT_CHECK_HIP(c, hipEventCreate(&e0)); PT_CHECK_HIP(c, hipEventCreate(&e1));
    PT_CHECK_HIP(c, hipEventRecord(e0, st));
    PT_CHECK_HIP(c, bvh_build(c->bvh, c->dsc, c->numTris, st));
    PT_CHECK_HIP(c, hipEventRecord(e1, st));
    launch_shade_tris(c->dsc, 0u, c->numTris, c->dShadeTris.p, st);
    PT_CHECK_HIP(c, hipStreamSynchronize(st));
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); c->buildMs = ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    c->geomDirty = false; c->lightsDirty = true;
    return PT_OK;
}

// ---- light baking. Order (Rtxpt/Lighting/LightsBaker.cpp:663-827): env quads, analytic lights, emissive triangles per sub-instance.
const uint RTXPT_LIGHTING_MAX_LIGHTS = 512 * 1024, RTXPT_LIGHTING_SAMPLING_PROXY_RATIO = 12, RTXPT_LIGHTING_MAX_SAMPLING_PROXIES_PER_LIGHT = 256 * 1024;
const float RTXPT_LIGHTING_MIN_WEIGHT_THRESHOLD = 1e-8f;
const uint QT_BASE_RES = 4, QT_SUBDIV = 24, QT_UNBOOSTED = QT_BASE_RES * QT_BASE_RES + 3 * QT_SUBDIV, QT_BOOST_DPT = 3, QT_BOOST_SUBDIV = 20,
           QT_BOOST_MULT = QT_BOOST_SUBDIV * 3 + 1, QT_TOTAL = QT_UNBOOSTED * QT_BOOST_MULT, EMISB_DIM = 1024;
struct EnvImportance { uint dim, mipCount; std::vector<std::vector<ptk::float4>> mips; };
uint firstbithigh(uint v) { uint r = 0; while (v >>= 1) r++; return r; }
// EnvironmentComputeWeightForQTBuild (LightsBaker.hlsl:181-198)
uint qt_weight(const EnvImportance& im, uint dim, uint x, uint y, uint lightIndex, uint depthLimit) {
    uint mipLevel = im.mipCount - firstbithigh(dim) - 1;
    float areaMul = (float)(1u << (mipLevel * 2));
    float radiance = im.mips[mipLevel][(size_t)y * dim + x].w;
    float ret = areaMul * radiance;
    ret = fmaxf_(sq(1.0f / 100.0f) * (float)mipLevel, ret);
    ret *= (mipLevel > depthLimit) ? 1.0f : 0.0f;
    uint v = (uint)(FastSqrt(ret) * 100 + 0.5f); if (v > 0x000FFFFFu) v = 0x000FFFFFu;
    return (v << 12) | lightIndex;
}
struct QTNode { uint dim, x, y; };
// EnvLightsSubdivideBase / EnvLightsSubdivideBoost (LightsBaker.hlsl:262-467): greedy split of the heaviest node
void qt_subdivide(const EnvImportance& im, std::vector<QTNode>& nodes, std::vector<uint>& packed, uint subdivisions, uint depthLimit) {
    for (uint si = 0; si < subdivisions; si++) {
        uint best = 0; for (size_t i = 0; i < packed.size(); i++) best = std::max(best, packed[i]);
        uint gi = best & 0xFFFu;
        QTNode n = nodes[gi];
        for (uint k = 0; k < 4; k++) {
            QTNode ch; ch.dim = n.dim * 2; ch.x = n.x * 2 + (k % 2); ch.y = n.y * 2 + (k / 2);
            uint ni = (k == 0) ? gi : (uint)nodes.size();
            if (k == 0) { nodes[gi] = ch; packed[gi] = qt_weight(im, ch.dim, ch.x, ch.y, ni, depthLimit); }
            else { nodes.push_back(ch); packed.push_back(qt_weight(im, ch.dim, ch.x, ch.y, ni, depthLimit)); }
        }
    }
}
int bake_env_quads(pt_context* c) {
    EnvImportance im; im.dim = EMISB_DIM; im.mipCount = 11; im.mips.resize(im.mipCount);
    PT_CHECK_HIP(c, c->dScratch4.resize((size_t)im.dim * im.dim));
    launch_env_importance(c->dsc, im.dim, 4, 4, c->dScratch4.p, c->stream);
    im.mips[0].resize((size_t)im.dim * im.dim);
    PT_CHECK_HIP(c, hipMemcpyAsync(im.mips[0].data(), c->dScratch4.p, sizeof(ptk::float4) * im.mips[0].size(), hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    for (uint l = 1; l < im.mipCount; l++) {
        uint pd = im.dim >> (l - 1), d = im.dim >> l;
        im.mips[l].resize((size_t)d * d);
        const std::vector<ptk::float4>& p = im.mips[l - 1];
        for (uint y = 0; y < d; y++) for (uint x = 0; x < d; x++) {
            ptk::float4 s = (p[(size_t)(2 * y) * pd + 2 * x] + p[(size_t)(2 * y) * pd + 2 * x + 1]) + (p[(size_t)(2 * y + 1) * pd + 2 * x] + p[(size_t)(2 * y + 1) * pd + 2 * x + 1]);
            // MipMapGenPass MODE_COLOR (Donut, not vendored: the 2x2 mean of the stored texels, stored as binary16)
            im.mips[l][(size_t)y * d + x] = ptk::env_round_rgba16f(s * 0.25f);
        }
    }
    std::vector<QTNode> base; std::vector<uint> packed;
    for (uint li = 0; li < QT_BASE_RES * QT_BASE_RES; li++) {
        QTNode n; n.dim = QT_BASE_RES; n.x = li / QT_BASE_RES; n.y = li % QT_BASE_RES;
        base.push_back(n); packed.push_back(qt_weight(im, n.dim, n.x, n.y, li, QT_BOOST_DPT));
    }
    qt_subdivide(im, base, packed, QT_SUBDIV, QT_BOOST_DPT);
    c->envLookupDim = im.dim; c->envLookup.assign((size_t)im.dim * im.dim, 0);
    const float distantVsLocal = 1.0f * 0.0002f;                                       // LightsBaker.cpp:1029-1030
    for (uint g = 0; g < QT_UNBOOSTED; g++) {
        std::vector<QTNode> nodes(1, base[g]); std::vector<uint> pk(1, qt_weight(im, base[g].dim, base[g].x, base[g].y, 0, 0));
        qt_subdivide(im, nodes, pk, QT_BOOST_SUBDIV, 0);
        for (uint li = 0; li < QT_BOOST_MULT; li++) {
            EnvironmentQuadLight e; e.NodeDim = nodes[li].dim; e.NodeX = nodes[li].x; e.NodeY = nodes[li].y;
            uint mipLevel = im.mipCount - firstbithigh(e.NodeDim) - 1;               // EnvironmentComputeRadianceAndWeight (LightsBaker.hlsl:167-175)
            float areaMul = (float)(1u << (mipLevel * 2));
            ptk::float4 value = im.mips[mipLevel][(size_t)e.NodeY * e.NodeDim + e.NodeX];
            e.Weight = areaMul * fmaxf_(0.f, value.w * Average(c->envColorMul) * distantVsLocal);
            e.Radiance = xyz(value) * c->envColorMul;
            PolymorphicLightInfoFull lf = e.Store(0);
            ptk::float2 sub = ptk::make_float2(((float)e.NodeX + 0.5f) / (float)e.NodeDim, ((float)e.NodeY + 0.5f) / (float)e.NodeDim);
            lf.Base.Center = mul_vec_mat3(oct_to_ndir_equal_area_unorm(sub), c->envToWorld) * DISTANT_LIGHT_DISTANCE;
            uint out = g * QT_BOOST_MULT + li;
            c->lights[out] = lf.Base; c->lightsEx[out] = lf.Extended;
            uint dimScale = im.dim / e.NodeDim;                                        // EnvLightsFillLookupMap (LightsBaker.hlsl:472-492)
            for (uint yy = 0; yy < dimScale; yy++) for (uint xx = 0; xx < dimScale; xx++)
                c->envLookup[(size_t)(e.NodeY * dimScale + yy) * im.dim + (e.NodeX * dimScale + xx)] = out;
        }
    }
    return PT_OK;
}
// geometryOnly: the instances / vertices moved but materials, environment, analytic lights and settings did not (pt_animate): the environment quad-tree lights
// are kept, only the emissive triangles are re-baked, and everything downstream (weights, proxy counts, proxy table) runs on the device. ComputeProxyCounts +
// the proxy fill (LightsBaker.hlsl:880-948, 1009-1060) from the weights in dWeights[0 .. N) (dWeights[N] receives their sum). usage != null: NEE-AT's feedback
// term. Leaves numProxies and the scene view current.
int build_light_proxies(pt_context* c, float* dWeights, const uint* dUsage, uint totalMaxFeedbackCount, float globalFeedbackUseWeight) {
    const uint N = (uint)c->lights.size(); if (!N) return PT_OK;
    const uint budget = RTXPT_LIGHTING_SAMPLING_PROXY_RATIO * std::max(N, RTXPT_LIGHTING_MAX_LIGHTS / 10);
    // sum of ceil((budget - N) w_i / W) <= budget - N + N (the feedback lerp keeps the weights' sum at W)
    const size_t proxyCapacity = (size_t)budget + N;
    PT_CHECK_HIP(c, c->dProxyCounters.resize(N)); PT_CHECK_HIP(c, c->dProxyOffsets.resize(N)); PT_CHECK_HIP(c, c->dProxyIndices.resize(proxyCapacity));
    launch_light_proxy_counts(dWeights, N, dWeights + N, budget, c->S.NEEType == 0, RTXPT_LIGHTING_MAX_SAMPLING_PROXIES_PER_LIGHT - 1, c->dProxyCounters.p, dUsage, totalMaxFeedbackCount, globalFeedbackUseWeight, c->stream);
    size_t need = 0;
    PT_CHECK_HIP(c, rocprim::exclusive_scan(nullptr, need, c->dProxyCounters.p, c->dProxyOffsets.p, 0u, (size_t)N, rocprim::plus<uint>(), c->stream));
    if (need > c->scanTempBytes) { if (c->dScanTemp) (void)hipFree(c->dScanTemp); c->dScanTemp = nullptr; PT_CHECK_HIP(c, hipMalloc(&c->dScanTemp, need)); c->scanTempBytes = need; }
    need = c->scanTempBytes;
    PT_CHECK_HIP(c, rocprim::exclusive_scan(c->dScanTemp, need, c->dProxyCounters.p, c->dProxyOffsets.p, 0u, (size_t)N, rocprim::plus<uint>(), c->stream));
    launch_light_proxy_fill(c->dProxyCounters.p, c->dProxyOffsets.p, N, c->dProxyIndices.p, (uint)proxyCapacity, c->stream);
    uint last[2] = {0u, 0u};                           // the proxy count is a field of the by-value scene view: one 8-byte read-back
    PT_CHECK_HIP(c, hipMemcpyAsync(&last[0], c->dProxyOffsets.p + (N - 1), 4, hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipMemcpyAsync(&last[1], c->dProxyCounters.p + (N - 1), 4, hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    c->numProxies = last[0] + last[1];
    if (c->numProxies > proxyCapacity) return fail(c, PT_ERROR_HIP, "light proxy table overflow");
    return PT_OK;
}
int bake_lights(pt_context* c, bool geometryOnly = false) {
    hipEvent_t e0, e1; PT_CHECK_HIP(c, hipEventCreate(&e0)); PT_CHECK_HIP(c, hipEventCreate(&e1));
    PT_CHECK_HIP(c, hipEventRecord(e0, c->stream));
    const bool keepEnv = geometryOnly && c->S.NEEEnabled && c->envEnabled && c->envLightsBaked == QT_TOTAL && c->lights.size() >= QT_TOTAL;
    if (!keepEnv) { c->lights.clear(); c->lightsEx.clear(); c->envLookup.clear(); c->envLookupDim = 0; c->envLightsBaked = 0; }
    else { c->lights.resize(QT_TOTAL); c->lightsEx.resize(QT_TOTAL); }
    c->numProxies = 0;
    for (auto& si : c->subInstances) { si.EmissiveLightMappingOffset = 0xFFFFFFFFu; si.AnalyticProxyLightIndex = 0xFFFFFFFFu; }
    if (c->S.NEEEnabled) {
        if (c->envEnabled && !keepEnv) { c->lights.resize(QT_TOTAL); c->lightsEx.resize(QT_TOTAL); int r = bake_env_quads(c); if (r != PT_OK) return r; c->envLightsBaked = QT_TOTAL; }
        const uint analyticBase = (uint)c->lights.size();
        for (auto& a : c->analyticLights) { c->lights.push_back(a.Base); c->lightsEx.push_back(a.Extended); }
        // analytic light proxies (LightsBaker.cpp:718-753): a mesh instance that stands in for an analytic light carries that light's index in its
        // sub-instances
        for (size_t s = 0; s < c->subInstances.size(); s++) {
            const uint proxy = c->instances[c->subInstToInstGeom[s].x].analyticProxyLight;
            if (proxy && proxy <= c->analyticLights.size() && (c->materials[c->subInstances[s].GlobalGeometryIndex_PTMaterialDataIndex & 0xFFFFu].Flags & PTMaterialFlags_EnableAsAnalyticLightProxy))
                c->subInstances[s].AnalyticProxyLightIndex = analyticBase + proxy - 1u;
        }
        // emissive triangles: host decides the layout (LightsBaker.cpp:663-827), the GPU bakes the records (LightsBaker.hlsl:544-716)
        std::vector<uint> list, offsets; uint total = 0; uint lightBase = (uint)c->lights.size();
        for (size_t s = 0; s < c->subInstances.size(); s++) {
            const GeometryDesc& g = c->geometries[c->subInstances[s].GlobalGeometryIndex_PTMaterialDataIndex >> 16];
            const ptk::PTMaterialData& mat = c->materials[g.materialIndex];
            bool isEmissive = any_gt0(mat.EmissiveColor);                            // PTMaterial::IsEmissive (MaterialsBaker.cpp:511-514)
            uint ntri = g.numIndices / 3;
            if (!isEmissive || (size_t)lightBase + total + ntri >= RTXPT_LIGHTING_MAX_LIGHTS) continue;
            c->subInstances[s].EmissiveLightMappingOffset = lightBase + total;
            list.push_back((uint)s); offsets.push_back(total); total += ntri;
        }
        // (the emissive part of the host mirror only holds the size)
        c->lights.resize((size_t)lightBase + total); c->lightsEx.resize((size_t)lightBase + total);
        const uint N = (uint)c->lights.size();
        PT_CHECK_HIP(c, c->dLights.resize(N)); PT_CHECK_HIP(c, c->dLightsEx.resize(N));
        if (lightBase && !keepEnv) {                       // environment quads + analytic lights: computed on the host, static under animation
            PT_CHECK_HIP(c, hipMemcpyAsync(c->dLights.p, c->lights.data(), sizeof(ptk::PolymorphicLightInfo) * lightBase, hipMemcpyHostToDevice, c->stream));
            PT_CHECK_HIP(c, hipMemcpyAsync(c->dLightsEx.p, c->lightsEx.data(), sizeof(ptk::PolymorphicLightInfoEx) * lightBase, hipMemcpyHostToDevice, c->stream));
        } else if (keepEnv && lightBase > QT_TOTAL) {
            PT_CHECK_HIP(c, hipMemcpyAsync(c->dLights.p + QT_TOTAL, c->lights.data() + QT_TOTAL, sizeof(ptk::PolymorphicLightInfo) * (lightBase - QT_TOTAL), hipMemcpyHostToDevice, c->stream));
            PT_CHECK_HIP(c, hipMemcpyAsync(c->dLightsEx.p + QT_TOTAL, c->lightsEx.data() + QT_TOTAL, sizeof(ptk::PolymorphicLightInfoEx) * (lightBase - QT_TOTAL), hipMemcpyHostToDevice, c->stream));
        }
        if (total) {
            // (a few KB; always: the list is not keyed on a capacity)
            PT_CHECK_HIP(c, c->dEmissiveList.upload(list, c->stream)); PT_CHECK_HIP(c, c->dEmissiveOffsets.upload(offsets, c->stream));
            launch_bake_emissive(c->dsc, c->dEmissiveList.p, c->dEmissiveOffsets.p, (uint)list.size(), total, lightBase, c->dLights.p, c->dLightsEx.p, c->stream);
        }
        // ComputeWeights + ComputeProxyCounts + proxy fill (LightsBaker.hlsl:738-751, 836-948) on the device; NEEType 0 = uniform (1 proxy per light)
        if (N) {
            const uint budget = RTXPT_LIGHTING_SAMPLING_PROXY_RATIO * std::max(N, RTXPT_LIGHTING_MAX_LIGHTS / 10);
            const size_t proxyCapacity = (size_t)budget + N;                       // sum of ceil((budget - N) w_i / W) <= budget - N + N
            (void)budget; (void)proxyCapacity;
            PT_CHECK_HIP(c, c->dLightW.resize(N + 1));
            launch_light_weights(c->dLights.p, c->dLightsEx.p, N, c->dLightW.p, c->lightBoost, c->stream);
            int pr = build_light_proxies(c, c->dLightW.p, nullptr, 0u, 0.f); if (pr != PT_OK) return pr;
        }
    } else { PT_CHECK_HIP(c, c->dLights.resize(1)); PT_CHECK_HIP(c, c->dLightsEx.resize(1)); }
    if (!keepEnv) PT_CHECK_HIP(c, c->dEnvLookup.upload(c->envLookup, c->stream));
    PT_CHECK_HIP(c, c->dSubInstances.upload(c->subInstances, c->stream));
    PT_CHECK_HIP(c, hipEventRecord(e1, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); c->lightBakeMs = ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    refresh_scene_view(c);
    c->lightsDirty = false;
    return PT_OK;
}
// EnvMapBaker::Update (EnvMapBaker.cpp:425-620): lat-long source + directional lights -> the RGBA16F cube the path tracer and the light baker sample
int bake_env_cube(pt_context* c) {
    ptk::EnvCube& e = c->envCube; memset(&e, 0, sizeof(e)); e.dim = c->envCubeDim; e.mipLevels = ptk::env_cube_mip_levels(e.dim);
    size_t total = 0; for (uint l = 0; l < e.mipLevels; l++) { e.mipOffset[l] = (uint)total; total += 6ull * (e.dim >> l) * (e.dim >> l); }
    PT_CHECK_HIP(c, c->dEnvCube.resize(total));
    // the lights drawn into the cube: what the host handed over in the environment's frame (pt_set_environment_bake), then the loaded scene's own directional
    // lights, taken there by Sample::UpdateLighting's step (pt_env_bake_lights) with THIS bake's cube size and the environment's current orientation;
    // EMB_MAXDIRLIGHTS in all
    std::vector<ptk::EnvDirectionalLight> dirLights = c->envDirLights;
    if (!c->sceneDirLights.empty()) {
        PtEnvMapSceneParams prm; memset(&prm, 0, sizeof(prm)); memcpy(prm.Transform, c->envToWorld.m, 48); prm.Enabled = 1.f;
        std::vector<ptk::EnvDirectionalLight> conv(c->sceneDirLights.size());
        if (pt_env_bake_lights(reinterpret_cast<const PtEnvDirectionalLight*>(c->sceneDirLights.data()), (uint32_t)conv.size(), &prm, e.dim, reinterpret_cast<PtEnvDirectionalLight*>(conv.data())) != PT_OK) return fail(c, PT_ERROR_INVALID_ARGUMENT, "scene directional lights");
        for (auto& l : conv) if (dirLights.size() < 16u) dirLights.push_back(l);
    }
    PT_CHECK_HIP(c, c->dEnvDirLights.upload(dirLights, c->stream));
    if (c->skyEnabled) {          // constants + texture views for the kernels, and room for the half-resolution cloud pre-pass
        ptk::ProceduralSkyContext h = c->sky; h.Transmittance.texels = c->dSkyTex[0].p; h.Scatter.texels = c->dSkyTex[1].p; h.Irradiance.texels = c->dSkyTex[2].p; h.Clouds.texels = c->dSkyTex[3].p;
        PT_CHECK_HIP(c, c->dSky.resize(1)); PT_CHECK_HIP(c, hipMemcpyAsync(c->dSky.p, &h, sizeof(h), hipMemcpyHostToDevice, c->stream)); PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));      // (h is a local)
        PT_CHECK_HIP(c, c->dSkyLowRes.resize(6ull * (e.dim / 2u) * (e.dim / 2u)));
    }
    refresh_scene_view(c);
    launch_env_cube_bake(c->dsc, c->dEnvDirLights.p, (uint)dirLights.size(), c->dEnvCube.p, c->dsc.envCube, c->stream);
    if (c->envCompression) {          // EnvMapBaker.cpp:593-633: the path tracer samples the BC6H cube, the importance baker keeps the uncompressed one
        PT_CHECK_HIP(c, c->dEnvCubeSource.resize(total));
        PT_CHECK_HIP(c, hipMemcpyAsync(c->dEnvCubeSource.p, c->dEnvCube.p, sizeof(ptk::uint2) * total, hipMemcpyDeviceToDevice, c->stream));
        launch_env_cube_compress(c->dEnvCube.p, c->dsc.envCube, c->envCompression, c->stream);
    } else c->dEnvCubeSource.free();
    refresh_scene_view(c);
    PT_CHECK_HIP(c, hipGetLastError());
    c->envCubeDirty = false; c->lightsDirty = true;
    return PT_OK;
}
int prepare(pt_context* c) {
    if (c->texDirty) { int r = upload_textures(c); if (r != PT_OK) return r; refresh_scene_view(c); c->lightsDirty = true; c->envCubeDirty = true; }
    if (c->envEnabled && c->envCubeDirty) { int r = bake_env_cube(c); if (r != PT_OK) return r; }
    if (c->geomDirty) { int r = finalize_geometry(c); if (r != PT_OK) return r; }
    if (c->lightsDirty) { int r = bake_lights(c); if (r != PT_OK) return r; }
    // the camera moved under a frustum boost: weights and proxies only (the lights themselves do not depend on it)
    else if (c->weightsDirty && !c->lights.empty()) {
        launch_light_weights(c->dLights.p, c->dLightsEx.p, (uint)c->lights.size(), c->dLightW.p, c->lightBoost, c->stream);
        int r = build_light_proxies(c, c->dLightW.p, nullptr, 0u, 0.f); if (r != PT_OK) return r;
        refresh_scene_view(c);
    }
    c->weightsDirty = false;
    return PT_OK;
}
int ensure_pool(pt_context* c, uint n, uint shadowPerPath) {      // shadowPerPath: shadow-queue entries a path vertex may emit (NEEFullSamples)
    if (n <= c->poolCapacity && (size_t)n * shadowPerPath <= c->shadowCapacity) return PT_OK;
    if (n < c->poolCapacity) n = c->poolCapacity;
    const size_t ns = (size_t)n * shadowPerPath;
    if (ns > 0xF0000000ull) return fail(c, PT_ERROR_INVALID_ARGUMENT, "too many shadow-queue entries in one pt_render call (paths x NEEFullSamples)");
    PT_CHECK_HIP(c, c->dS0.resize(n)); PT_CHECK_HIP(c, c->dS1.resize(n)); PT_CHECK_HIP(c, c->dS2.resize(n)); PT_CHECK_HIP(c, c->dS3.resize(n)); PT_CHECK_HIP(c, c->dS4.resize(n));
    PT_CHECK_HIP(c, c->dHit.resize(n)); PT_CHECK_HIP(c, c->dQueue[0].resize(n)); PT_CHECK_HIP(c, c->dQueue[1].resize(n));
    PT_CHECK_HIP(c, c->dSq0.resize(ns)); PT_CHECK_HIP(c, c->dSq1.resize(ns)); PT_CHECK_HIP(c, c->dSq2.resize(ns));
    PT_CHECK_HIP(c, c->dBestKey.resize(ns)); PT_CHECK_HIP(c, c->dResolveList.resize(ns));
    PT_CHECK_HIP(c, c->dTaskQ.resize((size_t)PT_PIPELINE_BATCHES * 2 * TASK_QUEUE_CAPACITY)); PT_CHECK_HIP(c, c->dTravCounts.resize(PT_PIPELINE_BATCHES * PASS_COUNTERS));
    c->poolCapacity = n; c->shadowCapacity = ns;
    return PT_OK;
}

// Tile-sharded frames (pt_create with shardCount > 1): a rank traces, and feeds back for, its own pixels only, but the baker's passes read whole
// neighbourhoods. Between two frames every rank therefore receives the other ranks' reservoirs and exported depth (12 bytes per pixel: 100 MB for a 4K frame;
// the depth is what the reprojection tests — a pixel another rank traced would otherwise read as depth 0, "valid" by NaN compare, whatever its owner sees) and
// then runs the same deterministic passes on the same planes as everybody else: identical tables and proxy counts on all ranks, identical to the unsharded run.
// With a communicator (pt_comm_init) the exchange is RCCL point-to-point inside one group, un-padded like pt_gather; without one the host moves the packed
// buffers (pt_neeat_pack_feedback / pt_neeat_unpack_feedback).
int neeat_exchange_feedback(pt_context* c) {
    pt_context::NeeAt& st = c->neeat;
    if (c->shardCount == 1 || !c->comm || !st.feedbackFilled || st.W != c->width || st.H != c->height) return PT_OK;
    hipStream_t s = c->stream;
    if (st.xW != c->width || st.xH != c->height) {
        std::vector<uint> others; for (uint r = 0; r < c->shardCount; r++) if (r != c->shardRank) others.insert(others.end(), c->shardPixels[r].begin(), c->shardPixels[r].end());
        PT_CHECK_HIP(c, st.xPixels.upload(others, s)); PT_CHECK_HIP(c, st.xRecv.resize(3 * others.size())); PT_CHECK_HIP(c, st.xSend.resize(3 * c->owned.size())); PT_CHECK_HIP(c, hipStreamSynchronize(s));
        st.xW = c->width; st.xH = c->height;
    }
    const size_t n = c->owned.size();
    launch_pack_feedback(st.fbW.p, st.fbC.p, st.depth.p, c->dOwned.p, (uint)n, c->width, st.xSend.p, s);
    PT_CHECK_NCCL(c, g_rccl.GroupStart());
    size_t off = 0; ncclResult_t bad = ncclSuccess;
    for (uint r = 0; r < c->shardCount && bad == ncclSuccess; r++) {
        if (r == c->shardRank) continue;
        const size_t m = c->shardPixels[r].size();
        if (n) bad = g_rccl.Send(st.xSend.p, 3 * n, ncclFloat, (int)r, c->comm, s);
        if (m && bad == ncclSuccess) bad = g_rccl.Recv(st.xRecv.p + 3 * off, 3 * m, ncclFloat, (int)r, c->comm, s);
        off += m;
    }
    ncclResult_t ge = g_rccl.GroupEnd();
    if (bad != ncclSuccess || ge != ncclSuccess) return fail(c, PT_ERROR_HIP, std::string("NEE-AT feedback exchange: ") + g_rccl.GetErrorString(bad != ncclSuccess ? bad : ge));
    launch_unpack_feedback(st.fbW.p, st.fbC.p, st.depth.p, st.xPixels.p, (uint)off, c->width, st.xRecv.p, s);
    return PT_OK;
}
// One frame of LightsBaker::UpdateBegin + UpdateEnd for the NEE-AT layer (LightsBaker.cpp:943-962, 985-1075, 1186-1213, 1335-1420) ahead of the frame's path
// tracing; the order and the constants are spelled out in pt_neeat.h. The light set is the baked one; what changes per frame is the global proxy table, the
// tile tables and the jitter. phases: NEEAT_BEGIN = LightsBaker::UpdateBegin (before the frame's G-buffer), NEEAT_END = UpdateEnd (after it, on the frame's
// depth and motion vectors: Sample.cpp:2491-2494; pt_realtime_frame), NEEAT_BOTH = reference mode: nothing happens in between, UpdateEnd reads the depth the
// last traced frame exported (depth == nullptr) and the motion vectors are zero
enum { NEEAT_BEGIN = 1, NEEAT_END = 2, NEEAT_BOTH = 3 };
int neeat_frame(pt_context* c, int phases = NEEAT_BOTH, const float* depth = nullptr, const ptk::uint2* motion = nullptr) {
    pt_context::NeeAt& st = c->neeat;
    const uint N = (uint)c->lights.size();
    // nothing to sample (no lights, or all of them dark): NEE does not run (LightSampler::IsEmpty), the frame is traced without a local layer
    if (!N || !c->numProxies) {
        c->localResX = c->localResY = c->localJitterX = c->localJitterY = c->localMaxLight = 0; c->localRatio = 0.f; c->feedbackRequired = false; st.feedbackFilled = st.lastFeedbackAvailable = false; st.frameOpen = false;
        refresh_scene_view(c); return PT_OK;
    }
    NeeAtFrame F; memset(&F, 0, sizeof(F));
    F.W = c->width; F.H = c->height; F.BW = (F.W + 1) / 2; F.BH = (F.H + 1) / 2; F.tilesX = (F.W + 7) / 8 + 1; F.tilesY = (F.H + 7) / 8 + 1;
    const size_t px = (size_t)F.W * F.H, bpx = (size_t)F.BW * F.BH, tiles = (size_t)F.tilesX * F.tilesY;
    if (phases & NEEAT_BEGIN) {
        // another light set: its indices mean nothing to the old reservoirs and tiles
        if (st.historicTotalLightCount && st.historicTotalLightCount != N) st.feedbackFilled = st.lastFeedbackAvailable = false;
        if (st.W != F.W || st.H != F.H) {               // (re)create the textures: LightsBaker::CreateRenderPasses (LightsBaker.cpp:300-345)
            st.W = F.W; st.H = F.H; st.feedbackFilled = false; st.lastFeedbackAvailable = false;
            PT_CHECK_HIP(c, st.fbW.resize(px)); PT_CHECK_HIP(c, st.fbC.resize(px)); PT_CHECK_HIP(c, st.scW.resize(px)); PT_CHECK_HIP(c, st.scC.resize(px)); PT_CHECK_HIP(c, st.snapW.resize(px)); PT_CHECK_HIP(c, st.snapC.resize(px));
            PT_CHECK_HIP(c, st.blW.resize(bpx)); PT_CHECK_HIP(c, st.blC.resize(bpx)); PT_CHECK_HIP(c, st.local.resize(tiles * RTXPT_LIGHTING_LOCAL_PROXY_COUNT));
            PT_CHECK_HIP(c, hipMemsetAsync(st.fbW.p, 0, 4 * px, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(st.fbC.p, 0xFF, 4 * px, c->stream));
            PT_CHECK_HIP(c, st.depth.resize(px)); PT_CHECK_HIP(c, st.histDepth.resize(px));
            PT_CHECK_HIP(c, hipMemsetAsync(st.depth.p, 0, 4 * px, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(st.histDepth.p, 0, 4 * px, c->stream));
        }
        st.prevJitter[0] = st.jitter[0]; st.prevJitter[1] = st.jitter[1];
        neeat_advance_jitter(st.updateCounter, st.jitterF, st.jitter);
        st.updateCounter++;
        st.frameLocalAvailable = st.lastFeedbackAvailable; st.frameFeedbackAvailable = st.feedbackFilled;
        st.framePrevLightCount = st.historicTotalLightCount; st.historicTotalLightCount = N;
        st.frameOpen = true;
    }
    if (!st.frameOpen || st.W != F.W || st.H != F.H) return fail(c, PT_ERROR_NOT_READY, "NEE-AT: UpdateEnd without UpdateBegin of this frame size");
    const bool lastFrameLocalSamplesAvailable = st.frameLocalAvailable, lastFrameFeedbackAvailable = st.frameFeedbackAvailable;
    F.jitterX = st.jitter[0]; F.jitterY = st.jitter[1]; F.jitterPrevX = st.prevJitter[0]; F.jitterPrevY = st.prevJitter[1];
    F.updateCounter = st.updateCounter; F.dropoff = st.dropoff; F.totalLightCount = N; F.historicTotalLightCount = st.framePrevLightCount;
    F.lastFrameFeedbackAvailable = lastFrameFeedbackAvailable ? 1u : 0u; F.lastFrameLocalSamplesAvailable = (lastFrameLocalSamplesAvailable && lastFrameFeedbackAvailable) ? 1u : 0u;
    F.fbW = st.fbW.p; F.fbC = st.fbC.p; F.scW = st.scW.p; F.scC = st.scC.p; F.blW = st.blW.p; F.blC = st.blC.p; F.local = st.local.p;
    F.depth = depth ? depth : st.depth.p; F.motion = motion; F.historyDepth = st.histDepth.p; F.depthDisocclusionThreshold = 1.5f;
    const uint totalMaxFeedbackCount = lastFrameFeedbackAvailable ? ((F.W + 7) / 8) * ((F.H + 7) / 8) * 64u : 0u;
    if (phases & NEEAT_BEGIN) {
        PT_CHECK_HIP(c, st.counters.resize(N + 1)); PT_CHECK_HIP(c, hipMemsetAsync(st.counters.p, 0, 4 * (size_t)(N + 1), c->stream)); F.perLightCounters = st.counters.p;      // ResetLightProxyCounters
        if (lastFrameFeedbackAvailable) launch_neeat_begin(F, st.snapW.p, st.snapC.p, st.preFilter, totalMaxFeedbackCount, c->stream);
        PT_CHECK_HIP(c, st.curW.resize(N + 1)); PT_CHECK_HIP(c, st.histW.resize(N));
        launch_neeat_boost_weights(c->dLightW.p, (lastFrameFeedbackAvailable && st.intensityDeltaMul > 0) ? st.histW.p : nullptr, st.nHist, N, st.intensityDeltaMul, st.curW.p, c->stream);
        int r = build_light_proxies(c, st.curW.p, lastFrameFeedbackAvailable ? st.counters.p : nullptr, totalMaxFeedbackCount, lastFrameFeedbackAvailable ? st.globalFeedbackWeight : 0.f); if (r != PT_OK) return r;
        PT_CHECK_HIP(c, hipMemcpyAsync(st.histW.p, st.curW.p, 4 * (size_t)N, hipMemcpyDeviceToDevice, c->stream)); st.nHist = N;
        st.lastFeedbackAvailable = lastFrameFeedbackAvailable;
    }
    if (!(phases & NEEAT_END)) { PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); return PT_OK; }
    F.perLightCounters = st.counters.p;
    F.samplingProxyCount = c->numProxies; F.proxies = c->dProxyIndices.p;
    // ---- UpdateEnd
    launch_neeat_end(F, c->stream);
    // reference mode: Bridge::ExportSurfaceInit of every pixel of the frame about to be traced
    if (!depth) PT_CHECK_HIP(c, hipMemsetAsync(st.depth.p, 0, 4 * px, c->stream));
    // (the fill passes of realtime mode export nothing: the build pass wrote this frame's depth)
    st.exportDepth = depth == nullptr;
    st.feedbackFilled = true; st.frameOpen = false;
    // what the path tracer binds this frame (the local layer is sampled only once feedback exists: LightsBaker.cpp:1048)
    c->localResX = F.tilesX; c->localResY = F.tilesY; c->localJitterX = st.jitter[0]; c->localJitterY = st.jitter[1]; c->localMaxLight = 0;
    c->localRatio = lastFrameFeedbackAvailable ? st.localRatio : 0.f; c->sscThreshold = st.sscThreshold; c->feedbackRequired = true;
    refresh_scene_view(c);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

} // namespace

extern "C" {

int32_t pt_create(const PtDeviceDesc* desc, pt_context** out) {
    if (!out) return PT_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PT_ERROR_NO_DEVICE;
    int dev = desc ? desc->deviceOrdinal : 0;
    if (dev < 0 || dev >= ndev) return PT_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(dev) != hipSuccess) return PT_ERROR_HIP;
    pt_context* c = new pt_context();
    c->device = dev; c->shardRank = desc ? desc->shardRank : 0; c->shardCount = (desc && desc->shardCount) ? desc->shardCount : 1;
    if (c->shardRank >= c->shardCount) { delete c; return PT_ERROR_INVALID_ARGUMENT; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PT_ERROR_HIP; }
    c->streams[0] = c->stream;
    for (uint b = 1; b < PT_PIPELINE_BATCHES; b++) if (hipStreamCreateWithFlags(&c->streams[b], hipStreamNonBlocking) != hipSuccess) { delete c; return PT_ERROR_HIP; }
    if (hipHostMalloc(&c->hostCounters, PT_PIPELINE_BATCHES * sizeof(WaveCounters), hipHostMallocDefault) != hipSuccess) { delete c; return PT_ERROR_HIP; }
    c->serialKernels = desc && (desc->flags & PT_DEVICE_SERIAL_KERNELS);
    // scene builds prefer fast trace, as the reference asks of its driver (Sample.cpp:1093): PLOC + parallel re-insertion + cost-driven wide nodes, all on the
    // device (round 3; PT_DEVICE_HOST_SAH_BUILDER: round 2's host-side binned SAH + re-insertion) unless the host asks for fast builds; pt_animate's rebuilds
    // are always plain PLOC
    c->bvhBuilder = (desc && (desc->flags & PT_DEVICE_PREFER_FAST_BUILD)) ? BVH_BUILDER_PLOC : ((desc && (desc->flags & PT_DEVICE_HOST_SAH_BUILDER)) ? BVH_BUILDER_SAH : BVH_BUILDER_PLOC_OPT);
    { const char* e = getenv("MI355PT_BVH_BUILDER");        // developer A/B switch
      if (e && !strcmp(e, "karras")) c->bvhBuilder = BVH_BUILDER_KARRAS; else if (e && !strcmp(e, "ploc")) c->bvhBuilder = BVH_BUILDER_PLOC; else if (e && !strcmp(e, "sah")) c->bvhBuilder = BVH_BUILDER_SAH; else if (e && (!strcmp(e, "ploc_opt") || !strcmp(e, "device"))) c->bvhBuilder = BVH_BUILDER_PLOC_OPT; }
    { const char* e = getenv("MI355PT_TAIL_PATHS"); if (e) c->tailBelow = (uint)strtoul(e, nullptr, 10); }      // developer A/B switch (pt_set_tail_paths)
    // developer A/B switch (pt_set_fused_traversal)
    { const char* e = getenv("MI355PT_COMPACT_POOL"); if (e) c->compactPool = atoi(e) != 0; }      // developer A/B / test switch, read at pt_create like the others
    { const char* e = getenv("MI355PT_FUSED_TRAVERSAL"); if (e) c->fusedTraversal = (uint)strtoul(e, nullptr, 10); }
    // test switch: iterations after which the tail kernel hands a ray back (0: T8_TAIL_DEFER); a small value sends most rays through the hand-back path
    { const char* e = getenv("MI355PT_TAIL_DEFER"); if (e) c->tailDefer = (uint)strtoul(e, nullptr, 10); }
    memset(&c->dsc, 0, sizeof(c->dsc)); memset(&c->cam, 0, sizeof(c->cam)); memset(&c->bvh, 0, sizeof(c->bvh));
    pt_default_settings(reinterpret_cast<::PtSettings*>(&c->S));
    const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; memcpy(c->envToWorld.m, I, 48); memcpy(c->envToLocal.m, I, 48); c->envColorMul = ptk::make_float3(1.0f / ptk::kEnvMapRadianceScale);      // no params = tint 1, intensity 1: the cube holds radiance x 1/4
    if (c->dCounters.resize(PT_PIPELINE_BATCHES) != hipSuccess) { delete c; return PT_ERROR_HIP; }
    *out = c;
    return PT_OK;
}
int32_t pt_destroy(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.lib) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    c->dSpHeader.free(); c->dSpThroughput.free(); c->dSpPlanes.free(); c->dSpRadiance.free(); c->dSpMotion.free(); c->dSpDepth.free(); c->dSpHitT.free(); c->dSpMark.free(); c->dSpNewL.free(); c->dSpScratch.free(); c->dSpGatherSend.free(); c->dSpGatherRecv.free(); c->dSpGatherPixels.free();
    c->neeat.free(); c->dLocalTable.free(); c->dFbWeight.free(); c->dFbCand.free(); c->dSq3.free();
    c->dGatherSend.free(); c->dGatherRecv.free(); c->dGatherPixels.free(); c->dLightW.free(); c->dProxyOffsets.free(); if (c->dScanTemp) (void)hipFree(c->dScanTemp);
    if (c->bvhAllocated) bvh_free(c->bvh);
    c->dIndices.free(); c->dNormals.free(); c->dTangents.free(); c->dProxyCounters.free(); c->dProxyIndices.free(); c->dEnvLookup.free(); c->dOwned.free(); c->dQueue[0].free(); c->dQueue[1].free();
    c->dPrevPositions.free(); c->dPrevInstances.free(); c->dEmissiveList.free(); c->dEmissiveOffsets.free(); c->dPositions.free(); c->dUvs.free(); c->dGeometries.free(); c->dInstances.free(); c->dSubInstances.free(); c->dSubInstToInstGeom.free();
    c->dPrimInfo.free(); c->dShadeTris.free(); c->dAlphaPlanes.free(); c->dAlphaPool.free(); c->dMaterials.free(); c->dTexInfos.free(); c->dTexels.free(); c->dEnvCube.free(); c->dEnvCubeSource.free(); c->dEnvImageCube.free(); c->dEnvDirLights.free(); c->dLights.free(); c->dLightsEx.free(); c->dS0.free(); c->dS1.free(); c->dS2.free(); c->dS3.free(); c->dS4.free();
    c->dHit.free(); c->dS0b.free(); c->dS1b.free(); c->dS3b.free(); c->dS4b.free(); c->dHitb.free(); c->dSq0.free(); c->dSq1.free(); c->dSq2.free(); c->dAccum.free(); c->dScratch4.free(); c->dCounters.free(); c->dTravSpill.free(); c->dTaskQ.free(); c->dTravCounts.free(); c->dResolveList.free(); c->dBestKey.free(); c->dResolveListSh.free(); c->dBestKeySh.free(); c->dTaskQSh.free();
    for (uint b = 1; b < PT_PIPELINE_BATCHES; b++) (void)hipStreamDestroy(c->streams[b]);
    (void)hipStreamDestroy(c->stream); (void)hipHostFree(c->hostCounters);
    delete c;
    return PT_OK;
}
const char* pt_get_last_error(pt_context* c) { return c ? c->lastError.c_str() : "null context"; }

int32_t pt_set_geometry(pt_context* c, const PtGeometryBuffers* b, const PtGeometryDesc* geoms, uint32_t nGeoms, const PtMeshDesc* meshes, uint32_t nMeshes) {
    if (!c || !b || !geoms || !meshes) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    if (!b->indices || !b->positions) return fail(c, PT_ERROR_INVALID_ARGUMENT, "indices and positions are required");
    for (uint32_t g = 0; g < nGeoms; g++) {
        if ((size_t)geoms[g].indexOffset + geoms[g].numIndices > b->numIndices || (size_t)geoms[g].vertexOffset + geoms[g].numVertices > b->numVertices || geoms[g].numIndices % 3)
            return fail(c, PT_ERROR_INVALID_ARGUMENT, "geometry range outside the supplied streams");
        if ((geoms[g].flags & PT_GEOM_HAS_UV) && !b->uvs) return fail(c, PT_ERROR_INVALID_ARGUMENT, "geometry has uv flag but no uv stream");
        if ((geoms[g].flags & PT_GEOM_HAS_NORMAL) && !b->normals) return fail(c, PT_ERROR_INVALID_ARGUMENT, "geometry has normal flag but no normal stream");
        if ((geoms[g].flags & PT_GEOM_HAS_TANGENT) && !b->tangents) return fail(c, PT_ERROR_INVALID_ARGUMENT, "geometry has tangent flag but no tangent stream");
        const uint32_t* idx = b->indices + geoms[g].indexOffset;      // an out-of-range index would be an out-of-bounds read on the device
        for (uint32_t k = 0; k < geoms[g].numIndices; k++) if (idx[k] >= geoms[g].numVertices) return fail(c, PT_ERROR_INVALID_ARGUMENT, "index outside the geometry's vertex range");
    }
    for (uint32_t m = 0; m < nMeshes; m++)
        if ((size_t)meshes[m].firstGeometry + meshes[m].numGeometries > nGeoms) return fail(c, PT_ERROR_INVALID_ARGUMENT, "mesh references geometries outside the supplied array");
    uint nv = b->numVertices;
    c->indices.assign(b->indices, b->indices + b->numIndices);
    c->positions.assign(b->positions, b->positions + 3 * (size_t)nv);
    c->uvs.assign(nv, ptk::make_float2(0, 0)); if (b->uvs) memcpy(c->uvs.data(), b->uvs, 8 * (size_t)nv);
    c->normals.assign(nv, 0); if (b->normals) memcpy(c->normals.data(), b->normals, 4 * (size_t)nv);
    c->tangents.assign(nv, 0); if (b->tangents) memcpy(c->tangents.data(), b->tangents, 4 * (size_t)nv);
    c->geometries.resize(nGeoms); memcpy(c->geometries.data(), geoms, sizeof(GeometryDesc) * nGeoms);
    c->meshes.resize(nMeshes); memcpy(c->meshes.data(), meshes, sizeof(MeshDesc) * nMeshes);
    c->geomDirty = true;
    return PT_OK;
}
int32_t pt_set_instances(pt_context* c, const PtInstanceDesc* inst, uint32_t n) {
    if (!c || (!inst && n)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    c->instances.resize(n); if (n) memcpy(c->instances.data(), inst, sizeof(InstanceDesc) * n);
    c->geomDirty = true;
    return PT_OK;
}
int32_t pt_set_materials(pt_context* c, const ::PTMaterialData* mats, uint32_t nMats, const PtTextureDesc* tex, uint32_t nTex) {
    if (!c || (!mats && nMats) || (!tex && nTex)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    c->materials.resize(nMats); if (nMats) memcpy(c->materials.data(), mats, 128 * (size_t)nMats);
    c->textures.clear();
    for (uint32_t i = 0; i < nTex; i++) {
        const PtTextureDesc& d = tex[i];
        if (!d.pixels || !d.width || !d.height || d.format > 2) return fail(c, PT_ERROR_INVALID_ARGUMENT, "bad texture descriptor");
        if (d.width > 32768u || d.height > 32768u) return fail(c, PT_ERROR_UNSUPPORTED, "textures above 32768 texels per side are not supported (16 mip levels)");
        HostTexture t; t.w = d.width; t.h = d.height; t.mips.resize(1); t.mips[0].resize((size_t)d.width * d.height);
        for (size_t k = 0; k < (size_t)d.width * d.height; k++) {
            ptk::float4 v;
            if (d.format == PT_TEX_RGBA32F) { const float* p = (const float*)d.pixels + 4 * k; v = ptk::make_float4(p[0], p[1], p[2], p[3]); }
            else {
                const uint8_t* p = (const uint8_t*)d.pixels + 4 * k;
                v = ptk::make_float4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
                if (d.format == PT_TEX_RGBA8_SRGB) { v.x = srgb_to_linear(v.x); v.y = srgb_to_linear(v.y); v.z = srgb_to_linear(v.z); }
            }
            t.mips[0][k] = v;
        }
        build_mips(t);
        c->textures.push_back(std::move(t));
    }
    for (uint32_t m = 0; m < nMats; m++) {
        const ptk::PTMaterialData& mm = c->materials[m];
        auto chk = [&](uint flag, uint word) { return !(mm.Flags & flag) || (word & 0xFFFFu) < nTex; };
        if (!chk(PTMaterialFlags_UseBaseOrDiffuseTexture, mm.BaseOrDiffuseTextureIndex) || !chk(PTMaterialFlags_UseEmissiveTexture, mm.EmissiveTextureIndex) ||
            !chk(PTMaterialFlags_UseNormalTexture, mm.NormalTextureIndex) || !chk(PTMaterialFlags_UseMetalRoughOrSpecularTexture, mm.MetalRoughOrSpecularTextureIndex) ||
            !chk(PTMaterialFlags_UseTransmissionTexture, mm.TransmissionTextureIndex))
            return fail(c, PT_ERROR_INVALID_ARGUMENT, "material references a missing texture");
    }
    c->texDirty = true; c->geomDirty = true;
    return PT_OK;
}
// orientation and colour multiplier of the environment (EnvMapSceneParams), shared by the two image setters
static void set_env_params(pt_context* c, const PtEnvMapSceneParams* params) {
    if (params) {
        memcpy(c->envToWorld.m, params->Transform, 48);
        memset(&c->envToLocal, 0, sizeof(float3x4));
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c->envToLocal.m[r * 4 + k] = c->envToWorld.m[k * 4 + r];   // rotation inverse = transpose
        c->envColorMul = ptk::make_float3(params->ColorMultiplier[0], params->ColorMultiplier[1], params->ColorMultiplier[2]);
    // no params: identity orientation, the supplied radiance as it is (ColorMultiplier = 1 / c_envMapRadianceScale undoes the cube's 1/4)
    } else {
        const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; memcpy(c->envToWorld.m, I, 48); memcpy(c->envToLocal.m, I, 48);
        c->envColorMul = ptk::make_float3(1.0f / ptk::kEnvMapRadianceScale);
    }
}
static bool env_has_image(const pt_context* c) { return c->envTex.w != 0u || c->envImageCubeDim != 0u; }
int32_t pt_set_environment(pt_context* c, const float* rgb, uint32_t w, uint32_t h, const PtEnvMapSceneParams* params) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    c->envImageCubeDim = 0; c->dEnvImageCube.free();          // one image source at a time: a lat-long image replaces a cube-map source
    c->envEnabled = (w != 0 && h != 0 && rgb && (!params || params->Enabled != 0.f));
    if (!c->envEnabled) { c->envTex.w = c->envTex.h = 0; c->envTex.mips.clear(); if (c->skyEnabled && (!params || params->Enabled != 0.f)) c->envEnabled = true; }      // (a procedural sky needs no image)
    else {
        HostTexture& t = c->envTex; t.w = w; t.h = h; t.mips.clear(); t.mips.resize(1); t.mips[0].resize((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; i++) t.mips[0][i] = ptk::make_float4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 1.f);
        t.mipLevels = 1;                                   // the bake reads mip 0 only (EnvMapBaker.hlsl:98-110: SampleLevel(.., 0))
    }
    set_env_params(c, params);
    c->texDirty = true; c->lightsDirty = true; c->envCubeDirty = true;
    return PT_OK;
}
int32_t pt_set_environment_cube(pt_context* c, const float* rgbaFaces, uint32_t dim, const PtEnvMapSceneParams* params) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (dim > 16384u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "cube-map source: at most 16384 texels per side");
    c->envTex.w = c->envTex.h = 0; c->envTex.mips.clear();    // one image source at a time
    c->envEnabled = (dim != 0 && rgbaFaces && (!params || params->Enabled != 0.f));
    if (!c->envEnabled) { c->envImageCubeDim = 0; c->dEnvImageCube.free(); if (c->skyEnabled && (!params || params->Enabled != 0.f)) c->envEnabled = true; }
    // stored as the RGBA16F texels a BC6H / RGBA16F cube file decodes to (a RGBA32F file is rounded to them)
    else {
        const size_t n = 6ull * dim * dim; std::vector<ptk::uint2> h(n);
        for (size_t i = 0; i < n; i++) h[i] = ptk::env_pack_rgba16f(ptk::make_float4(rgbaFaces[4 * i], rgbaFaces[4 * i + 1], rgbaFaces[4 * i + 2], rgbaFaces[4 * i + 3]));
        PT_CHECK_HIP(c, c->dEnvImageCube.resize(n)); PT_CHECK_HIP(c, hipMemcpy(c->dEnvImageCube.p, h.data(), n * sizeof(ptk::uint2), hipMemcpyHostToDevice));
        c->envImageCubeDim = dim;
    }
    set_env_params(c, params);
    c->texDirty = true; c->lightsDirty = true; c->envCubeDirty = true;
    return PT_OK;
}
int32_t pt_set_procedural_sky(pt_context* c, const PtProceduralSkyConstants* consts, const PtProceduralSkyTextures* tex) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    static_assert(sizeof(PtProceduralSkyConstants) == sizeof(ptk::ProceduralSkyConstants), "ProceduralSkyConstants layout");
    if (!consts) {
        if (c->skyEnabled) { c->skyEnabled = false; if (!env_has_image(c)) c->envEnabled = false; c->envCubeDirty = true; c->lightsDirty = true; c->texDirty = true; }
        return PT_OK;
    }
    if (tex) {
        const PtSkyTexture* t[4] = {&tex->transmittance, &tex->scattering, &tex->irradiance, &tex->clouds}; ptk::SkyTexture* dst[4] = {&c->sky.Transmittance, &c->sky.Scatter, &c->sky.Irradiance, &c->sky.Clouds};
        for (int i = 0; i < 4; i++) {
            if (!t[i]->rgba || !t[i]->width || !t[i]->height || !t[i]->depth || t[i]->width > 4096u || t[i]->height > 4096u || t[i]->depth > 4096u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "procedural sky: every look-up texture needs RGBA float texels and sizes in [1, 4096]");
            if ((i == 0 || i == 2) && t[i]->depth != 1u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "procedural sky: the transmittance and irradiance textures are 2-D (depth 1)");
        }
        for (int i = 0; i < 4; i++) {
            const size_t n = (size_t)t[i]->width * t[i]->height * t[i]->depth;
            PT_CHECK_HIP(c, c->dSkyTex[i].resize(n)); PT_CHECK_HIP(c, hipMemcpy(c->dSkyTex[i].p, t[i]->rgba, n * sizeof(ptk::float4), hipMemcpyHostToDevice));
            dst[i]->texels = nullptr; dst[i]->w = t[i]->width; dst[i]->h = t[i]->height; dst[i]->d = t[i]->depth; dst[i]->_pad = 0u;
        }
    } else if (!c->dSkyTex[0].p) return fail(c, PT_ERROR_INVALID_ARGUMENT, "procedural sky: no look-up textures have been set yet");
    memcpy(&c->sky.Consts, consts, sizeof(ptk::ProceduralSkyConstants));
    if (!c->envEnabled) {          // a sky without pt_set_environment: identity orientation, the baked radiance as it is
        if (!env_has_image(c)) { const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; memcpy(c->envToWorld.m, I, 48); memcpy(c->envToLocal.m, I, 48); c->envColorMul = ptk::make_float3(1.0f / ptk::kEnvMapRadianceScale); }
        c->envEnabled = true; c->texDirty = true;
    }
    c->skyEnabled = true; c->envCubeDirty = true; c->lightsDirty = true;
    return PT_OK;
}
int32_t pt_set_environment_bake(pt_context* c, uint32_t cubeDim, const PtEnvDirectionalLight* lights, uint32_t n) {
    if (!c || (!lights && n)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    if (cubeDim && (cubeDim < 16u || cubeDim > 8192u || (cubeDim & (cubeDim - 1u)))) return fail(c, PT_ERROR_INVALID_ARGUMENT, "cube resolution must be a power of two in [16, 8192]");
    if (n > 16u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "at most 16 directional lights are baked into the environment cube (EnvMapBaker.hlsl: EMB_MAXDIRLIGHTS)");
    if (cubeDim) c->envCubeDim = cubeDim;
    static_assert(sizeof(PtEnvDirectionalLight) == sizeof(ptk::EnvDirectionalLight), "EnvDirectionalLight layout");
    c->envDirLights.resize(n); if (n) memcpy(c->envDirLights.data(), lights, sizeof(ptk::EnvDirectionalLight) * n);
    c->envCubeDirty = true; c->lightsDirty = true;
    return PT_OK;
}
int32_t pt_set_scene_directional_lights(pt_context* c, const PtEnvDirectionalLight* worldLights, uint32_t n) {
    if (!c || (!worldLights && n)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    if (n > 16u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "at most 16 directional lights are baked into the environment cube (EnvMapBaker.hlsl: EMB_MAXDIRLIGHTS)");
    c->sceneDirLights.resize(n); if (n) memcpy(c->sceneDirLights.data(), worldLights, sizeof(ptk::EnvDirectionalLight) * n);
    c->envCubeDirty = true; c->lightsDirty = true;
    return PT_OK;
}
int32_t pt_set_environment_compression(pt_context* c, uint32_t quality) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (quality > 2u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "BC6U compression: 0 = off, 1 = fast (the reference's default on D3D12), 2 = quality (two-region modes)");
    if (c->envCompression != quality) { c->envCompression = quality; c->envCubeDirty = true; c->lightsDirty = true; }
    return PT_OK;
}
int32_t pt_set_local_light_sampling(pt_context* c, const uint32_t* table, uint32_t resX, uint32_t resY, uint32_t jitterX, uint32_t jitterY, float localToGlobalSampleRatio,
                                    float screenSpaceVsWorldSpaceThreshold, int32_t temporalFeedback) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (c->neeat.enabled) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_set_neeat is on: the baker in the loop writes the tile tables itself (pt_set_neeat(ctx, 0, ...) first)");
    (void)hipSetDevice(c->device);
    const uint N = ptk::RTXPT_LIGHTING_LOCAL_PROXY_COUNT, TILE_PX = ptk::RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE;
    if (!(localToGlobalSampleRatio >= 0.f && localToGlobalSampleRatio <= 0.95f)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "LocalToGlobalSampleRatio: 0 .. 0.95 (SampleUI.cpp:750)");
    if (table) {
        if (!resX || !resY || jitterX >= TILE_PX || jitterY >= TILE_PX) return fail(c, PT_ERROR_INVALID_ARGUMENT, "local sampling table: resolution in tiles of 8 x 8 pixels, jitter below the tile size");
        uint maxLight = 0;
        // SampleLocalPDF searches the tile by light index (LightingAlgorithms.hlsli:654): the entries must be sorted
        for (size_t t = 0; t < (size_t)resX * resY; t++) for (uint k = 0; k < N; k++) {
            const uint light = table[t * N + k] >> 9;
            if (k && light < (table[t * N + k - 1] >> 9)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "local sampling table: a tile's entries must be sorted by light index");
            if (light > maxLight) maxLight = light;
        }
        PT_CHECK_HIP(c, c->dLocalTable.upload(table, (size_t)resX * resY * N, c->stream));
        PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        c->localResX = resX; c->localResY = resY; c->localJitterX = jitterX; c->localJitterY = jitterY; c->localMaxLight = maxLight;
    } else { c->localResX = c->localResY = c->localJitterX = c->localJitterY = c->localMaxLight = 0; }
    c->localRatio = localToGlobalSampleRatio; c->sscThreshold = screenSpaceVsWorldSpaceThreshold; c->feedbackRequired = temporalFeedback != 0;
    refresh_scene_view(c);
    return PT_OK;
}
int32_t pt_get_light_feedback(pt_context* c, uint32_t sample, float* totalWeight, uint32_t* candidates) {
    if (!c || !totalWeight || !candidates) return PT_ERROR_INVALID_ARGUMENT;
    if (sample >= c->fbSamples) return fail(c, PT_ERROR_NOT_READY, "no feedback for that sample: pt_set_local_light_sampling(temporalFeedback = 1) or pt_set_neeat, then pt_render");
    if (c->neeat.enabled && (!c->neeat.fbW.p || c->neeat.W != c->width || c->neeat.H != c->height)) return fail(c, PT_ERROR_NOT_READY, "no NEE-AT frame of this size yet: pt_render first");
    (void)hipSetDevice(c->device);
    const size_t plane = (size_t)c->width * c->height;
    const float* w = c->neeat.enabled ? c->neeat.fbW.p : c->dFbWeight.p + plane * sample; const uint* cand = c->neeat.enabled ? c->neeat.fbC.p : c->dFbCand.p + plane * sample;
    PT_CHECK_HIP(c, hipMemcpy(totalWeight, w, 4 * plane, hipMemcpyDeviceToHost));
    PT_CHECK_HIP(c, hipMemcpy(candidates, cand, 4 * plane, hipMemcpyDeviceToHost));
    return PT_OK;
}
int32_t pt_set_neeat(pt_context* c, int32_t enable, float globalTemporalFeedbackWeight, float localToGlobalSampleRatio, float screenSpaceVsWorldSpaceThreshold, int32_t preFilter) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (enable && (!(globalTemporalFeedbackWeight >= 0.f && globalTemporalFeedbackWeight <= 0.95f) || !(localToGlobalSampleRatio >= 0.f && localToGlobalSampleRatio <= 0.95f)))
        return fail(c, PT_ERROR_INVALID_ARGUMENT, "global feedback weight and local-to-global ratio: 0 .. 0.95 (SampleUI.cpp:747-750)");
    (void)hipSetDevice(c->device);
    pt_context::NeeAt& st = c->neeat;
    if (st.enabled && !enable) {                 // back to the plain global sampler: no local layer, no feedback, the proxy table of the bake
        c->localResX = c->localResY = c->localJitterX = c->localJitterY = c->localMaxLight = 0; c->localRatio = 0.f; c->feedbackRequired = false; c->lightsDirty = true;
    }
    if (st.enabled != (enable != 0)) c->fbSamples = 0;      // the planes pt_get_light_feedback reads change hands: nothing is valid until the next pt_render
    st.enabled = enable != 0; st.globalFeedbackWeight = globalTemporalFeedbackWeight; st.localRatio = localToGlobalSampleRatio; st.sscThreshold = screenSpaceVsWorldSpaceThreshold; st.preFilter = preFilter != 0;
    refresh_scene_view(c);
    return PT_OK;
}
// the owned pixels' reservoirs and exported depth as (weight bits, candidate, depth bits) triples, 12 bytes per pixel in the order of the rank's pixel list —
// pt_pack_shard's order; device pointers
int32_t pt_neeat_pack_feedback(pt_context* c, void* dst, size_t bytes) {
    if (!c || !dst) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->neeat.enabled || !c->neeat.W || c->neeat.W != c->width || c->neeat.H != c->height) return fail(c, PT_ERROR_NOT_READY, "no NEE-AT frame yet: pt_set_neeat, then pt_render");
    if (bytes < c->owned.size() * 12) return fail(c, PT_ERROR_INVALID_ARGUMENT, "destination too small (12 bytes per owned pixel)");
    (void)hipSetDevice(c->device);
    launch_pack_feedback(c->neeat.fbW.p, c->neeat.fbC.p, c->neeat.depth.p, c->dOwned.p, (uint)c->owned.size(), c->width, (uint*)dst, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}
int32_t pt_neeat_unpack_feedback(pt_context* c, const void* src, size_t bytes, uint32_t rank) {
    if (!c || !src || rank >= c->shardCount) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->neeat.enabled || !c->neeat.W || c->neeat.W != c->width || c->neeat.H != c->height) return fail(c, PT_ERROR_NOT_READY, "no NEE-AT frame yet: pt_set_neeat, then pt_render");
    const std::vector<uint>& px = c->shardPixels[rank];
    if (bytes < px.size() * 12) return fail(c, PT_ERROR_INVALID_ARGUMENT, "source too small (12 bytes per pixel of that rank)");
    (void)hipSetDevice(c->device);
    DevBuf<uint> tmp; PT_CHECK_HIP(c, tmp.upload(px, c->stream));
    launch_unpack_feedback(c->neeat.fbW.p, c->neeat.fbC.p, c->neeat.depth.p, tmp.p, (uint)px.size(), c->width, (const uint*)src, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    tmp.free();
    return PT_OK;
}
int32_t pt_set_view_projection(pt_context* c, const float* worldToClipRowMajor16) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    pt_context::NeeAt& st = c->neeat;
    if (worldToClipRowMajor16) { for (int r = 0; r < 4; r++) { st.clipZ[r] = worldToClipRowMajor16[4 * r + 2]; st.clipW[r] = worldToClipRowMajor16[4 * r + 3]; } st.haveClip = true; }
    else { memset(st.clipZ, 0, 16); memset(st.clipW, 0, 16); st.haveClip = false; }
    refresh_scene_view(c);
    return PT_OK;
}
int32_t pt_neeat_reset(pt_context* c) { if (!c) return PT_ERROR_INVALID_ARGUMENT; c->neeat.reset(); c->fbSamples = 0; return PT_OK; }
int32_t pt_get_neeat_tables(pt_context* c, uint32_t tilesXY[2], uint32_t jitterXY[2], uint32_t* table, uint32_t tableCapacityWords) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    const pt_context::NeeAt& st = c->neeat;
    if (!st.enabled || !st.W) return fail(c, PT_ERROR_NOT_READY, "no NEE-AT frame yet: pt_set_neeat, then pt_render");
    (void)hipSetDevice(c->device);
    const uint tx = (st.W + 7) / 8 + 1, ty = (st.H + 7) / 8 + 1;
    if (tilesXY) { tilesXY[0] = tx; tilesXY[1] = ty; }
    if (jitterXY) { jitterXY[0] = st.jitter[0]; jitterXY[1] = st.jitter[1]; }
    if (table) {
        if ((size_t)tableCapacityWords < (size_t)tx * ty * RTXPT_LIGHTING_LOCAL_PROXY_COUNT) return fail(c, PT_ERROR_INVALID_ARGUMENT, "table buffer too small");
        PT_CHECK_HIP(c, hipMemcpy(table, st.local.p, 4 * (size_t)tx * ty * RTXPT_LIGHTING_LOCAL_PROXY_COUNT, hipMemcpyDeviceToHost));
    }
    return PT_OK;
}
int32_t pt_set_light_importance_boost(pt_context* c, const float* viewProjRowMajor16, float frustumMul, float frustumFadeDistance) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (viewProjRowMajor16 && !(frustumMul >= 0.f && frustumFadeDistance >= 0.f)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "frustum boost: multiplier and fade distance must not be negative");
    ptk::LightFrustumBoost b; memset(&b, 0, sizeof(b));
    if (viewProjRowMajor16 && frustumMul > 0.f) { ptk::light_frustum_planes_from_viewproj(viewProjRowMajor16, b.planes); b.mul = frustumMul; b.fadeDistance = frustumFadeDistance; }
    if (memcmp(&b, &c->lightBoost, sizeof(b)) != 0) { c->lightBoost = b; c->weightsDirty = true; }
    return PT_OK;
}
int32_t pt_set_lights(pt_context* c, const ::PolymorphicLightInfo* lights, const ::PolymorphicLightInfoEx* ex, uint32_t n) {
    if (!c || (!lights && n)) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    c->analyticLights.clear();
    for (uint32_t i = 0; i < n; i++) {
        PolymorphicLightInfoFull f; memcpy(&f.Base, &lights[i], 32);
        if (ex) memcpy(&f.Extended, &ex[i], 16); else memset(&f.Extended, 0, 16);
        uint type = DecodeLightType(f.Base);
        // sphere lights are the enabled analytic type; a point-type record (ConvertLight of a light without radius) has its type compiled out in the
        // reference (PolymorphicLightPTConfig.h:17-22) and is carried as the reference carries it: a slot in the buffer with no power and empty samples
        if (type != kSphere && type != kPoint) return fail(c, PT_ERROR_UNSUPPORTED, "only sphere (and inert point-type) analytic light records are accepted (PolymorphicLightPTConfig.h:17-22)");
        c->analyticLights.push_back(f);
    }
    c->lightsDirty = true;
    return PT_OK;
}

int32_t pt_bridge_camera(uint32_t w, uint32_t h, const float pos[3], const float dir[3], const float up[3], float fovY, float nearZ, float farZ, float focalDistance,
                         float apertureRadius, const float jitter[2], ::PathTracerCameraData* out) {
    if (!out || !pos || !dir || !up || !w || !h) return PT_ERROR_INVALID_ARGUMENT;
    // PathTracerShared.h:109-141 (donut::math normalize = v / length(v))
    auto nrm = [](ptk::float3 v) { float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); return ptk::make_float3(v.x / l, v.y / l, v.z / l); };
    ptk::PathTracerCameraData d; memset(&d, 0, sizeof(d));
    float aspect = (float)w / (float)h;
    d.FocalDistance = focalDistance; d.PosW = ptk::make_float3(pos[0], pos[1], pos[2]); d.NearZ = nearZ; d.FarZ = farZ; d.AspectRatio = aspect;
    d.ViewportSize = ptk::make_uint2(w, h);
    ptk::float3 camDir = ptk::make_float3(dir[0], dir[1], dir[2]), camUp = ptk::make_float3(up[0], up[1], up[2]);
    d.DirectionW = nrm(camDir);
    d.CameraW = nrm(camDir) * d.FocalDistance;
    d.CameraU = nrm(cross(d.CameraW, camUp));
    d.CameraV = nrm(cross(d.CameraU, d.CameraW));
    const float ulen = d.FocalDistance * std::tan(fovY * 0.5f) * d.AspectRatio;
    d.CameraU = d.CameraU * ulen;
    const float vlen = d.FocalDistance * std::tan(fovY * 0.5f);
    d.CameraV = d.CameraV * vlen;
    d.ApertureRadius = apertureRadius;
    d.PixelConeSpreadAngle = std::atan(2.0f * std::tan(fovY * 0.5f) / (float)h);
    d.Jitter = ptk::make_float2(jitter ? jitter[0] : 0.f, jitter ? -jitter[1] : 0.f);
    memcpy(out, &d, sizeof(d));
    return PT_OK;
}
int32_t pt_set_camera(pt_context* c, const ::PathTracerCameraData* cam) {
    if (!c || !cam) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    memcpy(&c->cam, cam, sizeof(c->cam));
    return PT_OK;
}
int32_t pt_default_settings(::PtSettings* s) {
    if (!s) return PT_ERROR_INVALID_ARGUMENT;
    memset(s, 0, sizeof(*s));
    s->bounceCount = 8; s->diffuseBounceCount = 8; s->perPixelJitterAAScale = 1.0f; s->texLODBias = -1.0f; s->fireflyFilterThreshold = 0.f; s->envMapDiffuseSampleMIPLevel = 0.f;
    s->NEEEnabled = 1; s->NEEType = 1; s->NEECandidateSamples = 5; s->NEEFullSamples = 1; s->enableRussianRoulette = 1; s->nestedDielectricsQuality = 1;
    s->enableLDSamplerForBSDF = 1; s->diffuseBrdf = 2; s->useFp16Types = 1;      // SampleUI.h:182: UseFp16Types = true
    return PT_OK;
}
int32_t pt_set_settings(pt_context* c, const ::PtSettings* s) {
    if (!c || !s) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    // NEEType 2 (NEE-AT): the global table is built as for type 1 (LightsBaker.hlsl:920-923 treats every type but 0 alike; the feedback-weighted boost of the
    // global proxies belongs to the baker's feedback passes, which the host does not have yet); the local layer and the feedback come from
    // pt_set_local_light_sampling
    if (s->NEEType > 2) return fail(c, PT_ERROR_INVALID_ARGUMENT, "NEEType: 0 (uniform), 1 (power) or 2 (NEE-AT)");
    if (s->NEECandidateSamples == 0 || s->NEECandidateSamples > 63) return fail(c, PT_ERROR_INVALID_ARGUMENT, "NEECandidateSamples must be in [1,63]");
    if (s->nestedDielectricsQuality > 2 || (s->diffuseBrdf != 0 && s->diffuseBrdf != 2) || s->bounceCount > 96 || s->useFp16Types > 1) return fail(c, PT_ERROR_INVALID_ARGUMENT, "setting out of range");
    if (c->S.NEEEnabled != s->NEEEnabled || c->S.NEEType != s->NEEType) c->lightsDirty = true;
    memcpy(&c->S, s, sizeof(c->S));
    return PT_OK;
}
int32_t pt_resize(pt_context* c, uint32_t w, uint32_t h) {
    if (!c || !w || !h || w > 65535 || h > 65535) return fail(c, PT_ERROR_INVALID_ARGUMENT, "bad size");
    (void)hipSetDevice(c->device);
    c->width = w; c->height = h; c->accumCount = 0; c->fbSamples = 0;
    build_shards(c);
    PT_CHECK_HIP(c, c->dAccum.resize((size_t)w * h));
    PT_CHECK_HIP(c, hipMemsetAsync(c->dAccum.p, 0, sizeof(ptk::float4) * (size_t)w * h, c->stream));
    PT_CHECK_HIP(c, c->dOwned.upload(c->owned, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}
int32_t pt_reset_accumulation(pt_context* c) {
    if (!c || !c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    (void)hipSetDevice(c->device);
    c->accumCount = 0;
    PT_CHECK_HIP(c, hipMemsetAsync(c->dAccum.p, 0, sizeof(ptk::float4) * (size_t)c->width * c->height, c->stream));
    return PT_OK;
}
int32_t pt_animate(pt_context* c, const PtInstanceDesc* inst, uint32_t nInst, const float* positions, uint32_t nVerts, int32_t rebuild) { return pt_animate_ranges(c, inst, nInst, positions, nVerts, nullptr, 0u, rebuild); }
// Motion history (Donut: SceneGraph::Refresh keeps every node's previous global transform, the skinning pass the previous positions of the meshes it rewrites;
// InstanceData.prevTransform, GeometryData.prevPositionOffset): with it on, every pt_animate / pt_animate_ranges call is one scene refresh — the pose it finds
// becomes the previous pose, the pose it brings the current one — and the stable-plane build pass's motion vectors carry the objects' motion
// (Bridge::loadSurface's prevPosW). A call without instances and positions only advances the history (a frame in which nothing moved: previous = current, no
// refit). Device-to-device copies of the instance table and of the vertex ranges that differ.
int32_t pt_set_motion_history(pt_context* c, int32_t enable) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (!enable) { c->motionHistory = false; c->dPrevPositions.free(); c->dPrevInstances.free(); c->prevStaleRanges.clear(); c->prevAllStale = false; refresh_scene_view(c); return PT_OK; }
    if (c->geomDirty || c->texDirty) { int r = prepare(c); if (r != PT_OK) return r; }
    // previous = current: nothing has moved yet
    if (!c->motionHistory) { c->motionHistory = true; c->prevAllStale = true; int r = motion_history_sync(c); if (r != PT_OK) return r; }
    refresh_scene_view(c);
    return PT_OK;
}
// the previous pose handed over directly (a host that keeps its own history, or a test): arrays shaped like the scene's; either may be NULL (= that part did
// not move). Turns the history on.
int32_t pt_set_previous_pose(pt_context* c, const PtInstanceDesc* inst, uint32_t nInst, const float* positions, uint32_t nVerts) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->geomDirty || c->texDirty) { int r = prepare(c); if (r != PT_OK) return r; }
    if (inst && nInst != c->instances.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_set_previous_pose: instance count differs from the scene's");
    if (positions && (size_t)nVerts * 3 != c->positions.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_set_previous_pose: vertex count differs from the scene's");
    c->motionHistory = true; c->prevAllStale = true;
    int r = motion_history_sync(c); if (r != PT_OK) return r;
    if (inst) { static_assert(sizeof(PtInstanceDesc) == sizeof(InstanceDesc), "instance layout"); PT_CHECK_HIP(c, hipMemcpyAsync(c->dPrevInstances.p, inst, sizeof(InstanceDesc) * nInst, hipMemcpyHostToDevice, c->stream)); }
    if (positions) { PT_CHECK_HIP(c, hipMemcpyAsync(c->dPrevPositions.p, positions, 12 * (size_t)nVerts, hipMemcpyHostToDevice, c->stream)); c->prevAllStale = true; }      // (differs anywhere: the next refresh copies everything)
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    refresh_scene_view(c);
    return PT_OK;
}

int32_t pt_animate_ranges(pt_context* c, const PtInstanceDesc* inst, uint32_t nInst, const float* positions, uint32_t nVerts, const uint32_t* vertexRanges, uint32_t nRanges, int32_t rebuild) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->geomDirty || c->texDirty) { int r = prepare(c); if (r != PT_OK) return r; }
    if (inst && nInst != c->instances.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_animate: instance count must not change");
    if (positions && (size_t)nVerts * 3 != c->positions.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_animate: vertex count must not change");
    if (positions && vertexRanges) for (uint32_t r = 0; r < nRanges; r++) if ((unsigned long long)vertexRanges[2 * r] + vertexRanges[2 * r + 1] > nVerts) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_animate_ranges: vertex range beyond the vertex count");
    if (inst) for (uint32_t i = 0; i < nInst; i++) if (inst[i].meshIndex != c->instances[i].meshIndex) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_animate: topology must not change");
    // (every argument is validated by now: a rejected call must not advance the motion history — the next build pass would report zero object motion for what
    // moved last frame)
    if (c->motionHistory) {      // one scene refresh: what is current becomes previous (before the uploads below overwrite it)
        int r = motion_history_sync(c); if (r != PT_OK) return r;
        if (positions) { if (vertexRanges) c->prevStaleRanges.assign(vertexRanges, vertexRanges + 2 * (size_t)nRanges); else c->prevAllStale = true; }
        if (!inst && !positions) { PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); refresh_scene_view(c); return PT_OK; }
    }
    static const bool animLog = getenv("MI355PT_ANIMATE_LOG") != nullptr;      // developer probe: host-side time of every step of the call (stderr)
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tA = now(); double tB = tA, tC = tA, tD = tA, tE = tA;
    if (inst) {
        memcpy(c->instances.data(), inst, sizeof(InstanceDesc) * nInst);
        PT_CHECK_HIP(c, c->dInstances.upload(c->instances, c->stream));
    }
    if (positions) {
        if (vertexRanges) {
            for (uint32_t r = 0; r < nRanges; r++) {
                const size_t first = 3 * (size_t)vertexRanges[2 * r], count = 3 * (size_t)vertexRanges[2 * r + 1];
                if (!count) continue;
                memcpy(c->positions.data() + first, positions + first, 4 * count);
                PT_CHECK_HIP(c, hipMemcpyAsync(c->dPositions.p + first, c->positions.data() + first, 4 * count, hipMemcpyHostToDevice, c->stream));
            }
        } else {
            memcpy(c->positions.data(), positions, 12 * (size_t)nVerts);
            PT_CHECK_HIP(c, c->dPositions.upload(c->positions, c->stream));
        }
    }
    tB = now();
    refresh_scene_view(c);
    if (positions) {      // the shading records hold object-space vertices: deformed meshes rewrite them, rigid motion does not
        if (!vertexRanges) launch_shade_tris(c->dsc, 0u, c->numTris, c->dShadeTris.p, c->stream);
        else for (size_t s = 0; s < c->subInstToInstGeom.size(); s++) {      // sub-instances (= primitive ranges) of the geometries that hold a moved vertex
            const GeometryDesc& gd = c->geometries[c->subInstToInstGeom[s].y];
            bool touched = false;
            for (uint32_t r = 0; r < nRanges && !touched; r++) touched = vertexRanges[2 * r + 1] && vertexRanges[2 * r] < gd.vertexOffset + gd.numVertices && gd.vertexOffset < vertexRanges[2 * r] + vertexRanges[2 * r + 1];
            if (touched) launch_shade_tris(c->dsc, c->subInstFirstPrim[s], gd.numIndices / 3u, c->dShadeTris.p, c->stream);
        }
    }
    hipEvent_t e0, e1; PT_CHECK_HIP(c, hipEventCreate(&e0)); PT_CHECK_HIP(c, hipEventCreate(&e1));
    PT_CHECK_HIP(c, hipEventRecord(e0, c->stream));
    if (rebuild) {                                     // a rebuild between animated frames prefers a fast build: PLOC on the device (15 ms at 2.8 M triangles)
        if (c->bvh.builder == BVH_BUILDER_SAH || c->bvh.builder == BVH_BUILDER_PLOC_OPT) c->bvh.builder = BVH_BUILDER_PLOC;
        PT_CHECK_HIP(c, bvh_build(c->bvh, c->dsc, c->numTris, c->stream));
    } else PT_CHECK_HIP(c, bvh_refit(c->bvh, c->dsc, c->numTris, c->stream));
    // (a fast refit leaves the BVH2 behind: nothing may read it until the next full build, pt_build.h bvh2Stale)
    c->dsc.nodes = c->bvh.bvh2Stale ? nullptr : c->bvh.nodes;
    PT_CHECK_HIP(c, hipEventRecord(e1, c->stream));
    tC = now();
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    tD = now();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); if (rebuild) c->buildMs = ms; else c->refitMs = ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const bool lightsWereDirty = c->lightsDirty;      // something besides the geometry changed since the last bake (environment, analytic lights, NEE settings)
    c->lightsDirty = true;                    // emissive triangle lights move with the geometry (Sample.cpp:1170-1198)
    c->accumCount = 0;                        // any scene change resets accumulation in reference mode
    PT_CHECK_HIP(c, hipMemsetAsync(c->dAccum.p, 0, sizeof(ptk::float4) * (size_t)c->width * c->height, c->stream));
    // geometry moved, nothing else: environment lights kept, emissive re-bake + weights + proxy table on the device
    const int32_t rb = bake_lights(c, !lightsWereDirty);
    tE = now();
    if (animLog) fprintf(stderr, "[animate] copy + upload %.3f ms, launches %.3f ms, wait %.3f ms (refit %.3f ms on the device), light re-bake %.3f ms: %.3f ms\n", tB - tA, tC - tB, tD - tC, ms, tE - tD, tE - tA);
    return rb;
}

int32_t pt_animate_normals(pt_context* c, const uint32_t* normals, const uint32_t* tangents, uint32_t nVerts) {
    if (!c || (!normals && !tangents)) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->geomDirty || c->texDirty) { int r = prepare(c); if (r != PT_OK) return r; }
    if ((normals && nVerts != c->normals.size()) || (tangents && nVerts != c->tangents.size())) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_animate_normals: vertex count must not change");
    if (normals) { memcpy(c->normals.data(), normals, 4 * (size_t)nVerts); PT_CHECK_HIP(c, c->dNormals.upload(c->normals, c->stream)); }
    if (tangents) { memcpy(c->tangents.data(), tangents, 4 * (size_t)nVerts); PT_CHECK_HIP(c, c->dTangents.upload(c->tangents, c->stream)); }
    refresh_scene_view(c);
    launch_shade_tris(c->dsc, 0u, c->numTris, c->dShadeTris.p, c->stream);      // the shading records hold the packed vertex normals and tangents
    c->accumCount = 0;
    PT_CHECK_HIP(c, hipMemsetAsync(c->dAccum.p, 0, sizeof(ptk::float4) * (size_t)c->width * c->height, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}
// every additive field of PtFrameStats (a call that traces its samples one frame at a time — NEE-AT — reports the sums; maxima stay maxima)
static void add_frame_stats(PtFrameStats& t, const PtFrameStats& o) {
    t.extendRays += o.extendRays; t.shadowRays += o.shadowRays; t.hits += o.hits;
    t.nodeVisitsExtend += o.nodeVisitsExtend; t.triTestsExtend += o.triTestsExtend; t.nodeVisitsShadow += o.nodeVisitsShadow; t.triTestsShadow += o.triTestsShadow;
    t.leafVisitsExtend += o.leafVisitsExtend; t.waveItersExtend += o.waveItersExtend; t.leafVisitsShadow += o.leafVisitsShadow; t.waveItersShadow += o.waveItersShadow;
    for (int q = 0; q < 4; q++) t.extendPhaseCycles[q] += o.extendPhaseCycles[q];
    t.leafBlocksExtend += o.leafBlocksExtend; if (o.waveItersMaxExtend > t.waveItersMaxExtend) t.waveItersMaxExtend = o.waveItersMaxExtend;
    for (int q = 0; q < 16; q++) t.extendRayIterHist[q] += o.extendRayIterHist[q];
    for (uint q = 0; q < o.longRayCount && q < 32u && t.longRayCount < 32u; q++) { memcpy(t.longRays[t.longRayCount], o.longRays[q], 32); t.longRayCount++; }
    for (int q = 0; q < 8; q++) t.extendEvents[q] += o.extendEvents[q];
    t.gpuMilliseconds += o.gpuMilliseconds; t.extendKernelMs += o.extendKernelMs; t.shadeKernelMs += o.shadeKernelMs; t.shadowKernelMs += o.shadowKernelMs;
    t.extendLaunches += o.extendLaunches; if (o.iterations > t.iterations) t.iterations = o.iterations;
    t.pathsTraced += o.pathsTraced; t.tailLaunches += o.tailLaunches;
}
int32_t pt_render(pt_context* c, uint32_t first, uint32_t count, PtFrameStats* stats) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    if (!count) return PT_OK;
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    // NEE-AT with the baker in the loop: every sample is a frame — baker passes, then the path tracer
    if (c->neeat.enabled && c->S.NEEEnabled && c->S.NEEFullSamples != 0u) {
        if (count > 1) {
            PtFrameStats total; memset(&total, 0, sizeof(total));
            for (uint32_t s = 0; s < count; s++) {
                PtFrameStats one; r = pt_render(c, first + s, 1, &one);
                // (the samples before the failing one were accumulated: their counts are reported)
                if (r != PT_OK) { if (stats) *stats = total; return r; }
                add_frame_stats(total, one);
            }
            if (stats) *stats = total;
            return PT_OK;
        }
        // (tile shards with a communicator; a host without one exchanges through pt_neeat_pack / unpack_feedback)
        r = neeat_exchange_feedback(c); if (r != PT_OK) return r;
        r = neeat_frame(c); if (r != PT_OK) return r;
    }
    uint numOwned = (uint)c->owned.size();
    if ((unsigned long long)numOwned * count > 0xF0000000ull) return fail(c, PT_ERROR_INVALID_ARGUMENT, "too many paths in one pt_render call");
    uint total = numOwned * count;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (total == 0) { c->accumCount += count; return PT_OK; }
    // min(RTXPT_LIGHTING_MAX_SAMPLE_COUNT, NEEFullSamples), PathTracerNEE.hlsli:312
    const uint neeSamples = c->S.NEEFullSamples < 63u ? c->S.NEEFullSamples : 63u;
    // 0: one shadow-queue entry per path vertex, written by k_shade itself
    const uint shadowGroup = (c->S.NEEEnabled && neeSamples > 1u) ? neeSamples : 0u;
    const uint shadowPerPath = shadowGroup ? shadowGroup : 1u;
    r = ensure_pool(c, total, shadowPerPath); if (r != PT_OK) return r;
    if (c->localResX) {                     // NEE-AT local layer: every pixel's (jittered) tile must exist, and a table can only name lights that were baked
        const uint TILE_PX = ptk::RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE;
        if ((c->width - 1u + c->localJitterX) / TILE_PX >= c->localResX || (c->height - 1u + c->localJitterY) / TILE_PX >= c->localResY) return fail(c, PT_ERROR_INVALID_ARGUMENT, "local sampling table smaller than the frame");
        if (c->localMaxLight >= c->lights.size()) return fail(c, PT_ERROR_INVALID_ARGUMENT, "local sampling table names a light index beyond the baked light table");
    }
    const bool feedback = c->feedbackRequired && c->S.NEEEnabled && neeSamples != 0u;
    c->fbSamples = 0;
    if (feedback) {
        if (shadowGroup) return fail(c, PT_ERROR_INVALID_ARGUMENT, "NEE-AT temporal feedback needs NEEFullSamples 1 (the reference's default): the feedback draw of one light sample shifts the random numbers of the next");
        const size_t plane = (size_t)c->width * c->height;
        PT_CHECK_HIP(c, c->dSq3.resize(c->shadowCapacity));
        if (!c->neeat.enabled) {      // (with the baker in the loop the run's own reservoirs are the target: they carry what the Clear pass kept)
            PT_CHECK_HIP(c, c->dFbWeight.resize(plane * count)); PT_CHECK_HIP(c, c->dFbCand.resize(plane * count));
            PT_CHECK_HIP(c, hipMemsetAsync(c->dFbWeight.p, 0, 4 * plane * count, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(c->dFbCand.p, 0xFF, 4 * plane * count, c->stream));      // LightFeedbackReservoir::Clear
        }
        PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    }
    PathKernelContext k; k.sc = c->dsc; k.S = c->S; k.cam = c->cam;
    // `applyNEE &= fullSamples > 0` (PathTracerNEE.hlsli:322): the vertices behave as without NEE; the light tables stay as baked
    if (neeSamples == 0u) k.S.NEEEnabled = 0;

    // The owned pixels are traced as up to PT_PIPELINE_BATCHES independent sub-frame batches, each on its own stream. Paths never interact, so this changes
    // nothing in the result; it lets the k_shade of one batch (3 waves per SIMD, mostly waiting on memory) overlap the traversal of the others and hides the
    // ~0.5 ms drain at the end of every launch (C3: 241 ms with one batch, 199 ms with four). The batches advance in lockstep (queue all, then service each as
    // its counts arrive): an event-driven variant that re-queued each batch independently was 7 % slower. Small frames use fewer batches,
    // PT_DEVICE_SERIAL_KERNELS one.
    struct Batch {
        uint pixFirst = 0, numPix = 0, total = 0, base = 0; hipStream_t st = nullptr; WaveCounters* wc = nullptr; WaveCounters* hwc = nullptr;
        PathPool pool; ShadowQueue sq; uint* queue[2] = {nullptr, nullptr}; DeviceScene sc; PathKernelContext k; TravAux aux;
        uint cur = 0, active = 0, iterations = 0, tailLaunches = 0; unsigned long long extendRays = 0, shadowRays = 0; bool waiting = false, afterTail = false, inTail = false; uint bound = 0, pendingShadow = 0; TravAux auxSh; PathPool poolSet[2]; uint set = 0; bool compact = false;      // poolSet / set / compact: the compacted pool (below)      // pendingShadow / auxSh: fused traversal launches (below)      // bound: wavefront passes so far (what maxIter limits; a tail launch is followed by one, so the loop ends)
        std::vector<hipEvent_t> ev; struct Span { size_t a, b; int kind; uint items; }; std::vector<Span> spans; size_t t0 = 0, t1 = 0;
        // per-launch HIP events: only when somebody reads them (serial-kernel steps, the pass log) — ten API calls per pass and batch otherwise
        bool timed = false;
        size_t mark() { if (!timed) return 0; hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); ev.push_back(e); return ev.size() - 1; }
    };
    uint numBatches = (c->serialKernels || total < (1u << 20)) ? 1u : ((total < PT_PIPELINE_FULL_AT) ? (uint)PT_PIPELINE_MID_BATCHES : PT_PIPELINE_BATCHES);
    { static const uint batchesOverride = []() { const char* e = getenv("MI355PT_BATCHES"); return e ? (uint)strtoul(e, nullptr, 10) : 0u; }(); if (batchesOverride && !c->serialKernels) numBatches = batchesOverride < (uint)PT_PIPELINE_BATCHES ? batchesOverride : (uint)PT_PIPELINE_BATCHES; }      // developer A/B switch
    Batch B[PT_PIPELINE_BATCHES];
    for (uint b = 0; b < numBatches; b++) {
        Batch& t = B[b];
        t.pixFirst = (uint)((unsigned long long)numOwned * b / numBatches); t.numPix = (uint)((unsigned long long)numOwned * (b + 1) / numBatches) - t.pixFirst;
        t.total = t.numPix * count; t.base = t.pixFirst * count; t.st = c->streams[b]; t.wc = c->dCounters.p + b; t.hwc = c->hostCounters + b;
        t.pool = PathPool{c->dS0.p + t.base, c->dS1.p + t.base, c->dS2.p + t.base, c->dS3.p + t.base, c->dS4.p + t.base, c->dHit.p + t.base};
        const size_t sbase = (size_t)t.base * shadowPerPath;
        t.sq = ShadowQueue{c->dSq0.p + sbase, c->dSq1.p + sbase, c->dSq2.p + sbase, shadowGroup, nullptr, nullptr, nullptr, 0u, 0u, 0u};
        if (feedback) { t.sq.q3 = c->dSq3.p + sbase; t.sq.fbTotalWeight = c->neeat.enabled ? c->neeat.fbW.p : c->dFbWeight.p; t.sq.fbCandidates = c->neeat.enabled ? c->neeat.fbC.p : c->dFbCand.p; t.sq.fbWidth = c->width; t.sq.fbPlane = c->width * c->height; t.sq.fbSampleFirst = first; }
        t.queue[0] = c->dQueue[0].p + t.base; t.queue[1] = c->dQueue[1].p + t.base;
        t.sc = c->dsc; t.sc.travSpill = c->dsc.travSpill + (size_t)b * T8_MAX_BLOCKS * T8_GROUPS_PER_BLOCK * T8_SPILL_DEPTH;
        t.k = k; t.k.sc = t.sc;
        t.aux.taskQ[0] = c->dTaskQ.p + (size_t)(2 * b) * TASK_QUEUE_CAPACITY; t.aux.taskQ[1] = t.aux.taskQ[0] + TASK_QUEUE_CAPACITY; t.aux.counts = c->dTravCounts.p + PASS_COUNTERS * b;
        // pipelined batches: one GPU-full of blocks per traversal launch (pt_scene.h PT_T8_MAX_BLOCKS) — and fewer for the launches of a small frame (one rank
        // of a sharded frame): every wave then works through more chunks before it runs dry and fewer of its rays are cut into sub-trees; the other batches
        // keep the GPU full. profiles/r05q_grid_cap_ab.txt: a rank of eight (4.1 M paths) 896 blocks -1 ... -3 %, a rank of four / two 1120 blocks -1 %, the
        // full frame (33 M paths) +1 % with either: hence by size.
        t.aux.maxBlocks = (numBatches >= 3u) ? (total < (6u << 20) ? 256u * 7u / 2u : (total < (24u << 20) ? 256u * 35u / 8u : 256u * 7u)) : 0u;
        { static const uint blocksOverride = []() { const char* e = getenv("MI355PT_MAX_BLOCKS"); return e ? (uint)strtoul(e, nullptr, 10) : 0u; }(); if (blocksOverride) t.aux.maxBlocks = blocksOverride < T8_MAX_BLOCKS ? blocksOverride : T8_MAX_BLOCKS; }      // (clamped: a batch's stack-tail slice is sized for T8_MAX_BLOCKS blocks)      // developer A/B switch
        t.aux.taskCap = TASK_QUEUE_CAPACITY; t.aux.bestKey = c->dBestKey.p + sbase; t.aux.resolveList = c->dResolveList.p + sbase; t.aux.primToSlot = c->bvh.primToSlot;
        t.timed = c->serialKernels || c->countersEnabled || getenv("MI355PT_PASS_LOG") != nullptr;
        memset(t.hwc, 0, sizeof(WaveCounters)); t.hwc->extendCount[0] = t.total;
        t.active = t.total;
    }
    // uploads issued on the main stream (prepare) must be visible to the second stream
    if (numBatches > 1) PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    hipEvent_t frame0, frame1; PT_CHECK_HIP(c, hipEventCreate(&frame0)); PT_CHECK_HIP(c, hipEventCreate(&frame1));
    PT_CHECK_HIP(c, hipEventRecord(frame0, c->stream));
    for (uint b = 0; b < numBatches; b++) {
        Batch& t = B[b];
        t.t0 = t.mark();
        PT_CHECK_HIP(c, hipMemcpyAsync(t.wc, t.hwc, sizeof(WaveCounters), hipMemcpyHostToDevice, t.st));
        launch_generate(t.k, t.pool, c->dOwned.p + t.pixFirst, t.numPix, first, count, 0u, t.total, t.queue[0], nullptr, t.st);
    }
    // upper bound on extend passes: bounceCount+1 vertices plus rejected (nested dielectric) re-traces
    uint maxIter = c->S.bounceCount + 2 + ((c->S.nestedDielectricsQuality == 2) ? 16u : (c->S.nestedDielectricsQuality == 1 ? 4u : 0u));
    // the tail kernel takes over a batch once it holds at most this many paths (0: never). Not in serial-kernel / counter frames (their per-kernel attribution
    // is the point), not with grouped NEE samples (NEEFullSamples > 1 folds a vertex's samples in k_resolve_nee) and not without a tree (the traversal's
    // empty-scene path is per launch, not per wave)
    const uint tailBelow = (!c->serialKernels && !c->countersEnabled && !shadowGroup && c->dsc.rootIsValid) ? c->tailBelow : 0u;
    // Fused traversal launches (round 6; pt_set_fused_traversal, k_trace_pair): the visibility rays a bounce's shading leaves in the shadow queue are not
    // traced in a launch of their own but wait (Batch::pendingShadow) for the next bounce's closest-hit launch and share it — and its task rounds and resolve
    // pass — block by block. Nothing of vertex k + 1 needs the visibility of vertex k before vertex k + 1 is shaded (the order of the fp16 additions into a
    // path's L), and that is exactly where the fused launch sits, so the image cannot change; the visibility rays need their own task queues, merge keys and
    // resolve list (auxSh). A batch whose paths have ended, or which goes to the tail kernel, traces what is pending in a plain visibility launch first. Not in
    // serial-kernel / counter frames (their per-kernel attribution is the point) and not with grouped NEE samples (k_resolve_nee).
    const bool fused = !c->serialKernels && !c->countersEnabled && !shadowGroup && c->dsc.rootIsValid && (c->fusedTraversal == 1u || (c->fusedTraversal == 2u && total < PT_FUSED_BELOW));
    if (fused) {
        PT_CHECK_HIP(c, c->dBestKeySh.resize(c->shadowCapacity)); PT_CHECK_HIP(c, c->dResolveListSh.resize(c->shadowCapacity)); PT_CHECK_HIP(c, c->dTaskQSh.resize((size_t)PT_PIPELINE_BATCHES * 2 * TASK_QUEUE_CAPACITY));
        for (uint b = 0; b < numBatches; b++) { Batch& t = B[b]; t.auxSh = t.aux; t.auxSh.counts = t.aux.counts + PASS_SHADOW_OFFSET; t.auxSh.taskQ[0] = c->dTaskQSh.p + (size_t)(2 * b) * TASK_QUEUE_CAPACITY; t.auxSh.taskQ[1] = t.auxSh.taskQ[0] + TASK_QUEUE_CAPACITY;
            t.auxSh.bestKey = c->dBestKeySh.p + (size_t)t.base * shadowPerPath; t.auxSh.resolveList = c->dResolveListSh.p + (size_t)t.base * shadowPerPath; }
    }
    // Batches run in lockstep: a batch's next half-pass is queued when ALL batches have delivered their counts, which keeps one batch's shading next to the
    // others' traversal (free-running streams drift into running the same kernel at the same time: 7 % slower on the full frame and no gain on a rank of a
    // sharded frame, DESIGN.md §4, profiles/r04i_event_loop_ab.txt).
    const bool passLog = getenv("MI355PT_PASS_LOG") != nullptr;
    // Compacted pool (round 6; ptk::PathPool::home). A path's state lives at its home slot (owned pixel x sample) for the whole frame in the layout above, and from the second bounce on
    // the survivors are scattered over the pool: a wave's 64 paths touch up to 64 lines per word group where the first bounce touches 8. Here k_shade writes a survivor's origin, direction,
    // interior list | counters | ray cone and {firefly K, MIS info, flags, sample index} at the POSITION it appends the path to, into the other of two array sets; the next bounce's
    // traversal reads rays, and writes hits, by position (the extend queue is the identity), k_classify and k_shade read dense arrays. Only throughput | radiance — what the
    // visibility resolve and k_accumulate address by path — stays at the home slot, which the extend queue keeps carrying (and the shadow queue names). Same values, another place:
    // the image cannot change. A batch that goes to the tail kernel is scattered back to its home slots first (k_uncompact) and continues in the home-slot layout. Not for NEE-AT
    // (its visibility resolve patches the path's flags), grouped NEE samples, serial-kernel and counter frames.
    const bool neeatShade = c->dsc.lights.LocalSamplingBuffer != nullptr || c->dsc.lights.TemporalFeedbackRequired != 0u;
    const bool compactPool = c->compactPool && !c->serialKernels && !c->countersEnabled && !shadowGroup && !neeatShade && !feedback && c->dsc.rootIsValid;
    if (compactPool) {
        PT_CHECK_HIP(c, c->dS0b.resize(c->poolCapacity)); PT_CHECK_HIP(c, c->dS1b.resize(c->poolCapacity)); PT_CHECK_HIP(c, c->dS3b.resize(c->poolCapacity)); PT_CHECK_HIP(c, c->dS4b.resize(c->poolCapacity)); PT_CHECK_HIP(c, c->dHitb.resize(c->poolCapacity));
        for (uint b = 0; b < numBatches; b++) { Batch& t = B[b]; t.compact = true; t.set = 0; t.poolSet[0] = t.pool;
            t.poolSet[1] = PathPool{c->dS0b.p + t.base, c->dS1b.p + t.base, t.pool.s2, c->dS3b.p + t.base, c->dS4b.p + t.base, c->dHitb.p + t.base, nullptr}; }
    }
    // One pass of a batch is queued by queue_pass (counter reset, traversal — fused with the pending visibility rays — classify + shade, read-back of the two
    // queue counts) and finished by finish_pass once those counts have arrived (the visibility rays become pending, or are traced if the batch ends here).
    uint wavefrontPasses = 0;
    auto queue_pass = [&](Batch& t) -> int32_t {
        uint nxt = t.cur ^ 1u;
        // the pass's traversal / class counters and the two queue counters it refills: one launch (fused: the shadow queue's counter still counts the pending
        // rays; k_resolve_pair zeroes it)
        launch_pass_reset(t.aux.counts, &t.wc->extendCount[nxt], fused ? nullptr : &t.wc->shadowCount, t.st);
        if (tailBelow && t.active <= tailBelow && !t.afterTail && t.compact) {      // the tail kernel works on home slots: scatter the live paths back, into the array set that is not being read
            PathPool in = t.poolSet[t.set]; in.home = t.queue[t.cur];
            launch_uncompact(in, t.poolSet[t.set ^ 1u], &t.wc->extendCount[t.cur], t.active, t.st);
            t.pool = t.poolSet[t.set ^ 1u]; t.compact = false;
        }
        if (tailBelow && t.active <= tailBelow && !t.afterTail) {
            if (t.pendingShadow) { launch_shadow(t.sc, t.pool, t.sq, &t.wc->shadowCount, t.pendingShadow, t.wc, false, t.auxSh, t.st); PT_CHECK_HIP(c, hipMemsetAsync(&t.wc->shadowCount, 0, 4, t.st)); t.pendingShadow = 0; }      // (the tail kernel adds to the paths' radiance itself: what is pending lands first)      // few paths left: one launch runs them to their end, wave by wave (pt_tail.hip); stragglers come back through queue[nxt] / the shadow queue
            size_t e0 = t.mark(); launch_tail(t.k, t.pool, t.queue[t.cur], &t.wc->extendCount[t.cur], t.active, t.queue[nxt], &t.wc->extendCount[nxt], t.sq, t.wc, maxIter - t.bound, c->tailDefer, t.aux.maxBlocks, t.st); size_t e1 = t.mark();
            if (t.timed) t.spans.push_back({e0, e1, 3, t.active});
            // what comes back — stragglers — is traced by a wavefront pass (task rounds included) before the tail kernel gets another turn
            t.tailLaunches++; t.afterTail = true; t.inTail = true;
            PT_CHECK_HIP(c, hipMemcpyAsync(t.hwc, t.wc, 16, hipMemcpyDeviceToHost, t.st));
            t.waiting = true;
            return PT_OK;
        }
        t.afterTail = false; t.bound++; wavefrontPasses++;
        size_t e0 = t.mark();
        PathPool pin = t.pool, pout = PathPool{};
        if (t.compact) { pin = t.poolSet[t.set]; pin.home = t.queue[t.cur]; pout = t.poolSet[t.set ^ 1u]; t.set ^= 1u; }
        if (t.pendingShadow) { launch_trace_pair(t.sc, pin, t.queue[t.cur], &t.wc->extendCount[t.cur], t.active, t.sq, &t.wc->shadowCount, t.pendingShadow, t.wc, t.aux, t.auxSh, t.st); t.pendingShadow = 0; }
        else launch_extend(t.sc, pin, t.queue[t.cur], &t.wc->extendCount[t.cur], t.active, t.wc, c->countersEnabled, t.aux, t.st);
        size_t e1 = t.mark(); if (t.timed) t.spans.push_back({e0, e1, 0, t.active});
        launch_shade(t.k, pin, t.queue[t.cur], &t.wc->extendCount[t.cur], t.active, t.queue[nxt], &t.wc->extendCount[nxt], t.sq, t.wc, t.active >= PT_CLASSIFY_FROM ? reinterpret_cast<uint*>(t.aux.bestKey) : nullptr /* the straggler keys are idle between k_resolve_extend and the shadow launch; a few thousand paths are shaded in queue order: one launch fewer */, t.aux.counts + PASS_CLASS_OFFSET, t.st, pout); size_t e2 = t.mark(); if (t.timed) t.spans.push_back({e1, e2, 1, t.active});
        t.extendRays += t.active;
        PT_CHECK_HIP(c, hipMemcpyAsync(t.hwc, t.wc, 16, hipMemcpyDeviceToHost, t.st));
        t.waiting = true;
        return PT_OK;
    };
    auto finish_pass = [&](Batch& t, uint b) -> int32_t {
        t.waiting = false; t.inTail = false;
        uint nxt = t.cur ^ 1u, nShadow = t.hwc->shadowCount;
        // the pass's straggler counters: sub-trees split off by k_extend and by task rounds 0..2, rays sent to the resolve pass (the previous pass's shadow
        // launch is reported with the next line)
        if (passLog) {
            uint pc[PASS_COUNTERS]; PT_CHECK_HIP(c, hipMemcpy(pc, t.aux.counts, sizeof(pc), hipMemcpyDeviceToHost));
            fprintf(stderr, "[pass log]   b%u pass %u: %u paths -> extend splits %u / %u / %u / %u sub-trees, %u rays resolved; %u visibility rays next\n", b, t.iterations, t.active, pc[0], pc[1], pc[2], pc[3], pc[TRAV_RESOLVE], nShadow);
        }
        TravAux auxShadow = t.aux; auxShadow.counts = t.aux.counts + PASS_SHADOW_OFFSET;
        t.active = t.hwc->extendCount[nxt];
        // they ride with the next closest-hit launch
        if (fused && nShadow && t.active && t.bound < maxIter) { t.pendingShadow = nShadow; t.shadowRays += nShadow; }
        else if (nShadow) { size_t s0 = t.mark(); launch_shadow(t.sc, t.pool, t.sq, &t.wc->shadowCount, nShadow, t.wc, c->countersEnabled, fused ? t.auxSh : auxShadow, t.st); size_t s1 = t.mark(); if (t.timed) t.spans.push_back({s0, s1, 2, nShadow}); if (!shadowGroup) t.shadowRays += nShadow;
                            if (fused) PT_CHECK_HIP(c, hipMemsetAsync(&t.wc->shadowCount, 0, 4, t.st)); }
        t.cur = nxt; t.iterations++;
        return PT_OK;
    };
    auto live = [&](const Batch& t) { return t.active && t.bound < maxIter; };
    // Small passes run free (round 6). The lockstep above pays while every pass fills the GPU; at the end of a frame — and for the whole of a small frame — a
    // pass is a chain of a dozen short launches, the batches no longer take equally long, and in lockstep three streams sit idle until the slowest has
    // delivered its counts (0.7 - 1 ms per late pass of the 4K frame, profiles/r06i_*). Once every live batch holds fewer than `freeRunBelow` paths the loop
    // turns event-driven: whichever batch's counts arrive first is finished and its next pass queued at once.
    static const uint freeRunBelow = []() { const char* e = getenv("MI355PT_FREE_RUN_BELOW"); return e ? (uint)strtoul(e, nullptr, 10) : (uint)PT_FREE_RUN_BELOW; }();
    bool any = true;
    while (any) {
        bool freeRun = freeRunBelow != 0u && numBatches > 1u;
        for (uint b = 0; b < numBatches; b++) if (live(B[b]) && B[b].active >= freeRunBelow) freeRun = false;
        // phase 1: every live batch queues extend + shade and the read-back of its queue counts
        wavefrontPasses = 0;
        for (uint b = 0; b < numBatches; b++) {
            Batch& t = B[b];
            if (t.waiting) continue;                        // a tail launch still in flight (below): the batch rejoins the lockstep when it is done
            if (!live(t)) continue;
            int32_t r1 = queue_pass(t); if (r1 != PT_OK) return r1;
        }
        any = false;
        if (freeRun) {      // event-driven until every batch has ended
            uint waiting = 0; for (uint b = 0; b < numBatches; b++) waiting += B[b].waiting ? 1u : 0u;
            while (waiting) {
                for (uint b = 0; b < numBatches; b++) {
                    Batch& t = B[b];
                    if (!t.waiting || hipStreamQuery(t.st) == hipErrorNotReady) continue;
                    PT_CHECK_HIP(c, hipStreamSynchronize(t.st));
                    int32_t r2 = finish_pass(t, b); if (r2 != PT_OK) return r2;
                    waiting--;
                    if (live(t)) { int32_t r1 = queue_pass(t); if (r1 != PT_OK) return r1; waiting++; }
                }
            }
            break;
        }
        // phase 2: as each batch's counts arrive, its visibility rays become pending (or are traced, if the batch ends); the other batches keep the GPU busy
        // meanwhile
        for (uint b = 0; b < numBatches; b++) {
            Batch& t = B[b];
            if (!t.waiting) continue;
            // a tail launch runs for about a millisecond — several of the other batches' passes: while those have wavefront passes to queue, it is only polled
            if (t.inTail && wavefrontPasses && hipStreamQuery(t.st) == hipErrorNotReady) { any = true; continue; }
            PT_CHECK_HIP(c, hipStreamSynchronize(t.st));
            int32_t r2 = finish_pass(t, b); if (r2 != PT_OK) return r2;
            if (live(t)) any = true;
        }
    }
    for (uint b = 0; b < numBatches; b++) {
        Batch& t = B[b];
        launch_accumulate(t.pool, c->dOwned.p + t.pixFirst, t.numPix, count, c->dAccum.p, c->accumCount, c->width, t.st);
        t.t1 = t.mark();
        PT_CHECK_HIP(c, hipMemcpyAsync(t.hwc, t.wc, sizeof(WaveCounters), hipMemcpyDeviceToHost, t.st));
    }
    for (uint b = 0; b < numBatches; b++) PT_CHECK_HIP(c, hipStreamSynchronize(B[b].st));
    PT_CHECK_HIP(c, hipEventRecord(frame1, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    PT_CHECK_HIP(c, hipGetLastError());
    bool overflow = false;
    for (uint b = 0; b < numBatches; b++) overflow = overflow || B[b].hwc->overflow;
    c->accumCount += count;
    if (feedback) c->fbSamples = count;
    if (stats) {
        // whole-call time: from the first batch's start to the later batch's end (both streams were idle before and are drained now)
        float ms = 0; (void)hipEventElapsedTime(&ms, frame0, frame1); stats->gpuMilliseconds = ms;
        for (uint b = 0; b < numBatches; b++) {
            Batch& t = B[b]; const WaveCounters& h = *t.hwc;
            for (auto& sp : t.spans) { float m = 0; (void)hipEventElapsedTime(&m, t.ev[sp.a], t.ev[sp.b]); if (sp.kind == 0) stats->extendKernelMs += m; else if (sp.kind == 1) stats->shadeKernelMs += m; else if (sp.kind == 2) stats->shadowKernelMs += m; }      // (zero in pipelined frames: no per-launch events there)
            stats->extendLaunches += t.iterations;
            stats->extendRays += t.extendRays + h.tailExtendRays; stats->shadowRays += shadowGroup ? h.shadowValid : t.shadowRays + h.tailShadowRays; stats->hits += h.hits; stats->tailLaunches += t.tailLaunches;
            if (getenv("MI355PT_PASS_LOG") && t.tailLaunches) fprintf(stderr, "[pass log] batch %u: %u tail launches traced %llu + %llu rays, handed back %llu extend stragglers, %llu visibility stragglers, %llu paths at the bounce bound\n", b, t.tailLaunches,
                (unsigned long long)h.tailExtendRays, (unsigned long long)h.tailShadowRays, (unsigned long long)h.tailHandedBack[0], (unsigned long long)h.tailHandedBack[1], (unsigned long long)h.tailHandedBack[2]);
            stats->nodeVisitsExtend += h.nodeVisitsExt; stats->triTestsExtend += h.triTestsExt;
            stats->nodeVisitsShadow += h.nodeVisitsSh; stats->triTestsShadow += h.triTestsSh;
            stats->leafVisitsExtend += h.leafVisitsExt; stats->waveItersExtend += h.itersExt; stats->leafVisitsShadow += h.leafVisitsSh; stats->waveItersShadow += h.itersSh;
            for (int q = 0; q < 4; q++) stats->extendPhaseCycles[q] += h.phaseCycExt[q];
            stats->leafBlocksExtend += h.leafBlocksExt; if (h.itersMaxExt > stats->waveItersMaxExtend) stats->waveItersMaxExtend = h.itersMaxExt;
            for (int q = 0; q < 8; q++) stats->extendEvents[q] += h.eventsExt[q];
            for (int q = 0; q < 16; q++) stats->extendRayIterHist[q] += h.rayIterHistExt[q];
            for (uint q = 0; q < h.longRayCount && q < 32u && stats->longRayCount < 32u; q++) { memcpy(stats->longRays[stats->longRayCount], h.longRays[q], 32); stats->longRayCount++; }
            if (t.iterations > stats->iterations) stats->iterations = t.iterations;
        }
        stats->pathsTraced = total;
    }
    if (getenv("MI355PT_PASS_LOG")) {        // developer probe: the launch sequence of every batch with item counts and HIP-event durations (stderr)
        for (uint b = 0; b < numBatches; b++) { Batch& t = B[b]; float whole = 0; if (!t.timed) continue; (void)hipEventElapsedTime(&whole, t.ev[t.t0], t.ev[t.t1]);
            fprintf(stderr, "[pass log] batch %u of %u: %u paths, %u passes, %.3f ms from first to last event\n", b, numBatches, t.total, t.iterations, whole);
            for (auto& sp : t.spans) { float m = 0, at = 0; (void)hipEventElapsedTime(&m, t.ev[sp.a], t.ev[sp.b]); (void)hipEventElapsedTime(&at, t.ev[t.t0], t.ev[sp.a]);
                fprintf(stderr, "[pass log]   b%u %-6s %9u items  start %8.3f ms  %7.3f ms\n", b, sp.kind == 0 ? "extend" : (sp.kind == 1 ? "shade" : (sp.kind == 2 ? "shadow" : "tail")), sp.items, at, m); } }
    }
    for (uint b = 0; b < numBatches; b++) for (auto e : B[b].ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(frame0); (void)hipEventDestroy(frame1);
    if (overflow) return fail(c, PT_ERROR_HIP, "BVH8 traversal: stack tail or straggler task queue overflow (raise T8_SPILL_DEPTH / TASK_QUEUE_CAPACITY)");
    // The pass bound (maxIter) is a safety net, never what ends a path: a path ends by its own bounce / rejected-hit counters (PathTracer.hlsli:40-45,
    // PathTracerNestedDielectrics.hlsli). Were a path still alive here, the set of dropped paths — the image — would depend on how the passes were composed
    // (tail threshold): reported, not swallowed.
    for (uint b = 0; b < numBatches; b++) if (B[b].active) return fail(c, PT_ERROR_HIP, "pt_render: paths still alive at the pass bound (bounceCount + 2 + the nested-dielectric allowance): the bound must be raised");
    return PT_OK;
}
static_assert(sizeof(::PtStablePlanesParams) == sizeof(ptk::StablePlanesParams) && sizeof(::PtStablePlane) == sizeof(ptk::StablePlane), "stable-plane ABI");
int32_t pt_stable_planes_plane_stride(uint32_t width, uint32_t height, uint32_t* stride) { if (!stride) return PT_ERROR_INVALID_ARGUMENT; *stride = ptk::GenericTSComputePlaneStride(width, height); return PT_OK; }
int32_t pt_build_stable_planes(pt_context* c, uint32_t sampleIndex, const PtStablePlanesParams* params, PtFrameStats* stats) {
    if (!c || !params) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    if (stats) memset(stats, 0, sizeof(*stats));
    const uint numOwned = (uint)c->owned.size();
    r = ensure_pool(c, numOwned ? numOwned : 1u, 1u); if (r != PT_OK) return r;
    const size_t N = (size_t)c->width * c->height;
    ptk::StablePlanesParams prm; memcpy(&prm, params, sizeof(prm));
    StablePlanesContext sp; sp.C = ptk::SP_make_consts(prm, c->width, c->height, c->S.bounceCount);
    PT_CHECK_HIP(c, c->dSpHeader.resize(4 * N)); PT_CHECK_HIP(c, c->dSpPlanes.resize((size_t)cStablePlaneCount * sp.C.genericTSPlaneStride)); PT_CHECK_HIP(c, c->dSpRadiance.resize(N)); PT_CHECK_HIP(c, c->dSpMotion.resize(N));
    PT_CHECK_HIP(c, c->dSpDepth.resize(N)); PT_CHECK_HIP(c, c->dSpHitT.resize(N)); PT_CHECK_HIP(c, c->dSpThroughput.resize(N));
    // a new size: nothing of the old frame is meaningful (pixels of other ranks' tiles and the records of planes that do not exist stay zero)
    if (c->spW != c->width || c->spH != c->height) {
        PT_CHECK_HIP(c, hipMemsetAsync(c->dSpHeader.p, 0xFF, 16 * N, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpPlanes.p, 0, sizeof(ptk::StablePlane) * cStablePlaneCount * sp.C.genericTSPlaneStride, c->stream));
        PT_CHECK_HIP(c, hipMemsetAsync(c->dSpRadiance.p, 0, 8 * N, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpMotion.p, 0, 8 * N, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpDepth.p, 0, 4 * N, c->stream));
        PT_CHECK_HIP(c, hipMemsetAsync(c->dSpHitT.p, 0, 4 * N, c->stream)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpThroughput.p, 0, 4 * N, c->stream));
        c->spW = c->width; c->spH = c->height;
    }
    sp.B.Header = c->dSpHeader.p; sp.B.Planes = c->dSpPlanes.p; sp.B.StableRadiance = c->dSpRadiance.p; sp.B.Depth = c->dSpDepth.p; sp.B.SpecularHitT = c->dSpHitT.p; sp.B.MotionVectors = c->dSpMotion.p; sp.B.Throughput = c->dSpThroughput.p;
    c->spGathered = false;
    if (!numOwned) return PT_OK;
    PathKernelContext k; k.sc = c->dsc; k.S = c->S; k.cam = c->cam;
    PathPool pool{c->dS0.p, c->dS1.p, c->dS2.p, c->dS3.p, c->dS4.p, c->dHit.p};
    uint* queue[2] = {c->dQueue[0].p, c->dQueue[1].p};
    WaveCounters* wc = c->dCounters.p; WaveCounters* hwc = c->hostCounters;
    TravAux aux; aux.taskQ[0] = c->dTaskQ.p; aux.taskQ[1] = aux.taskQ[0] + TASK_QUEUE_CAPACITY; aux.counts = c->dTravCounts.p; aux.maxBlocks = 0u;
    aux.taskCap = TASK_QUEUE_CAPACITY; aux.bestKey = c->dBestKey.p; aux.resolveList = c->dResolveList.p; aux.primToSlot = c->bvh.primToSlot;
    memset(hwc, 0, sizeof(WaveCounters)); hwc->extendCount[0] = numOwned;
    hipEvent_t e0, e1; PT_CHECK_HIP(c, hipEventCreate(&e0)); PT_CHECK_HIP(c, hipEventCreate(&e1));
    PT_CHECK_HIP(c, hipEventRecord(e0, c->stream));
    PT_CHECK_HIP(c, hipMemcpyAsync(wc, hwc, sizeof(WaveCounters), hipMemcpyHostToDevice, c->stream));
    launch_sp_generate(k, sp, pool, c->dOwned.p, numOwned, sampleIndex, queue[0], c->stream);
    // every pass is one vertex of every pixel that still explores: at most three planes of at most maxStablePlaneVertexDepth + 1 vertices, plus the false hits
    // nested dielectrics reject
    const uint maxIter = cStablePlaneCount * (sp.C.maxStablePlaneVertexDepth + 2u + ((c->S.nestedDielectricsQuality == 2) ? 16u : (c->S.nestedDielectricsQuality == 1 ? 4u * (sp.C.maxStablePlaneVertexDepth + 1u) : 0u)));
    uint cur = 0, active = numOwned, iterations = 0; unsigned long long rays = 0;
    while (active && iterations < maxIter) {
        const uint nxt = cur ^ 1u;
        launch_pass_reset(aux.counts, &wc->extendCount[nxt], &wc->shadowCount, c->stream);
        launch_extend(c->dsc, pool, queue[cur], &wc->extendCount[cur], active, wc, c->countersEnabled, aux, c->stream);
        launch_sp_build_shade(k, sp, pool, queue[cur], &wc->extendCount[cur], active, queue[nxt], &wc->extendCount[nxt], sampleIndex, wc, c->stream);
        rays += active;
        PT_CHECK_HIP(c, hipMemcpyAsync(hwc, wc, 16, hipMemcpyDeviceToHost, c->stream));
        PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        active = hwc->extendCount[nxt]; cur = nxt; iterations++;
    }
    PT_CHECK_HIP(c, hipEventRecord(e1, c->stream));
    PT_CHECK_HIP(c, hipMemcpyAsync(hwc, wc, sizeof(WaveCounters), hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    PT_CHECK_HIP(c, hipGetLastError());
    if (stats) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); stats->gpuMilliseconds = ms; stats->extendRays = rays; stats->hits = hwc->hits; stats->iterations = iterations; stats->extendLaunches = iterations; stats->pathsTraced = numOwned;
                 stats->nodeVisitsExtend = hwc->nodeVisitsExt; stats->triTestsExtend = hwc->triTestsExt; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (hwc->overflow) return fail(c, PT_ERROR_HIP, "BVH8 traversal: stack tail or straggler task queue overflow (raise T8_SPILL_DEPTH / TASK_QUEUE_CAPACITY)");
    if (active) return fail(c, PT_ERROR_HIP, "stable-plane build pass: paths still exploring after the iteration bound");
    return PT_OK;
}
int32_t pt_fill_stable_planes(pt_context* c, uint32_t sampleIndex, const PtStablePlanesParams* params, PtFrameStats* stats) {
    if (!c || !params) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes first");
    if (c->S.NEEEnabled && c->S.NEEFullSamples > 1u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "the fill pass traces one full NEE sample per vertex (NEEFullSamples 0 or 1, the reference's default)");
    // temporal feedback: with the baker in the loop (pt_set_neeat + pt_realtime_frame) the pass's visible light samples fill the run's reservoirs; a host that
    // runs its own baker (pt_set_local_light_sampling with temporalFeedback) gets its per-sample planes from pt_render only
    const bool feedback = c->neeat.enabled && c->feedbackRequired && c->S.NEEEnabled && c->S.NEEFullSamples != 0u;
    if (c->feedbackRequired && !c->neeat.enabled) return fail(c, PT_ERROR_INVALID_ARGUMENT, "the fill pass feeds NEE-AT's reservoirs only with the baker in the loop (pt_set_neeat, pt_realtime_frame): switch the temporal feedback of pt_set_local_light_sampling off");
    if (feedback && (!c->neeat.fbW.p || c->neeat.W != c->width || c->neeat.H != c->height)) return fail(c, PT_ERROR_NOT_READY, "NEE-AT: no baker frame of this size yet (pt_realtime_frame runs it)");
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    if (stats) memset(stats, 0, sizeof(*stats));
    const uint numOwned = (uint)c->owned.size();
    if (!numOwned) return PT_OK;
    r = ensure_pool(c, numOwned, 1u); if (r != PT_OK) return r;
    const bool freshMark = c->dSpMark.n < numOwned;
    PT_CHECK_HIP(c, c->dSpMark.resize(numOwned)); PT_CHECK_HIP(c, c->dSpNewL.resize(numOwned));
    if (feedback) PT_CHECK_HIP(c, c->dSq3.resize(c->shadowCapacity));
    // (k_sp_fill_resolve clears what a pass marked)
    if (freshMark) PT_CHECK_HIP(c, hipMemsetAsync(c->dSpMark.p, 0, sizeof(ptk::uint4) * c->dSpMark.n, c->stream));
    ptk::StablePlanesParams prm; memcpy(&prm, params, sizeof(prm));
    StablePlanesContext sp; sp.C = ptk::SP_make_consts(prm, c->width, c->height, c->S.bounceCount);
    sp.B.Header = c->dSpHeader.p; sp.B.Planes = c->dSpPlanes.p; sp.B.StableRadiance = c->dSpRadiance.p; sp.B.Depth = c->dSpDepth.p; sp.B.SpecularHitT = c->dSpHitT.p; sp.B.MotionVectors = c->dSpMotion.p; sp.B.Throughput = c->dSpThroughput.p;
    PathKernelContext k; k.sc = c->dsc; k.S = c->S; k.cam = c->cam;
    // As pt_render: the pixels are traced as up to PT_PIPELINE_BATCHES independent batches, each on its own stream with its own queues, counters, task queues
    // and slice of the pool, advancing in lockstep (queue all, then service each as its counts arrive) — the shading of one batch overlaps the traversal of the
    // others and the drain at the end of every launch is hidden.
    struct Batch { uint pixFirst = 0, numPix = 0; hipStream_t st = nullptr; WaveCounters* wc = nullptr; WaveCounters* hwc = nullptr; PathPool pool, markPool; ShadowQueue sq; ptk::float4* newL = nullptr; uint* queue[2] = {nullptr, nullptr};
                   DeviceScene sc; PathKernelContext k; TravAux aux; uint cur = 0, active = 0, iterations = 0; unsigned long long rays = 0, shadowRays = 0; bool waiting = false; };
    const uint numBatches = (c->serialKernels || numOwned < (1u << 20)) ? 1u : ((numOwned < PT_PIPELINE_FULL_AT) ? (uint)PT_PIPELINE_MID_BATCHES : PT_PIPELINE_BATCHES);
    Batch B[PT_PIPELINE_BATCHES];
    for (uint b = 0; b < numBatches; b++) {
        Batch& t = B[b];
        t.pixFirst = (uint)((unsigned long long)numOwned * b / numBatches); t.numPix = (uint)((unsigned long long)numOwned * (b + 1) / numBatches) - t.pixFirst;
        const uint base = t.pixFirst;
        t.st = c->streams[b]; t.wc = c->dCounters.p + b; t.hwc = c->hostCounters + b;
        t.pool = PathPool{c->dS0.p + base, c->dS1.p + base, c->dS2.p + base, c->dS3.p + base, c->dS4.p + base, c->dHit.p + base};
        t.markPool = t.pool; t.markPool.s2 = c->dSpMark.p + base; t.newL = c->dSpNewL.p + base;
        t.sq = ShadowQueue{c->dSq0.p + base, c->dSq1.p + base, c->dSq2.p + base, 0u, nullptr, nullptr, nullptr, 0u, 0u, 0u};
        // feedback: the fourth word group of an entry and the reservoir planes; the reference mode's shadow kernels then apply the reservoir update and the
        // roulette fix-up of a visible entry themselves (pt_wavefront.hip shadow_visible; one slot per pixel: plane stride 0)
        if (feedback) { t.sq.q3 = c->dSq3.p + base; t.sq.fbTotalWeight = c->neeat.fbW.p; t.sq.fbCandidates = c->neeat.fbC.p; t.sq.fbWidth = c->width; t.sq.fbPlane = 0u; t.sq.fbSampleFirst = 0u; }
        t.queue[0] = c->dQueue[0].p + base; t.queue[1] = c->dQueue[1].p + base;
        t.sc = c->dsc; t.sc.travSpill = c->dsc.travSpill + (size_t)b * T8_MAX_BLOCKS * T8_GROUPS_PER_BLOCK * T8_SPILL_DEPTH;
        t.k = k; t.k.sc = t.sc;
        t.aux.taskQ[0] = c->dTaskQ.p + (size_t)(2 * b) * TASK_QUEUE_CAPACITY; t.aux.taskQ[1] = t.aux.taskQ[0] + TASK_QUEUE_CAPACITY; t.aux.counts = c->dTravCounts.p + PASS_COUNTERS * b;
        t.aux.maxBlocks = (numBatches >= 3u) ? 256u * 7u : 0u;
        t.aux.taskCap = TASK_QUEUE_CAPACITY; t.aux.bestKey = c->dBestKey.p + base; t.aux.resolveList = c->dResolveList.p + base; t.aux.primToSlot = c->bvh.primToSlot;
        memset(t.hwc, 0, sizeof(WaveCounters));
    }
    // uploads / memsets issued on the main stream (prepare, the marks) must be visible to the batch streams
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    hipEvent_t e0, e1; PT_CHECK_HIP(c, hipEventCreate(&e0)); PT_CHECK_HIP(c, hipEventCreate(&e1));
    PT_CHECK_HIP(c, hipEventRecord(e0, c->stream));
    for (uint b = 0; b < numBatches; b++) {
        Batch& t = B[b];
        PT_CHECK_HIP(c, hipMemcpyAsync(t.wc, t.hwc, sizeof(WaveCounters), hipMemcpyHostToDevice, t.st));
        launch_sp_fill_generate(t.k, sp, t.pool, c->dOwned.p + t.pixFirst, t.numPix, sampleIndex, t.queue[0], &t.wc->extendCount[0], t.st);
        PT_CHECK_HIP(c, hipMemcpyAsync(t.hwc, t.wc, 16, hipMemcpyDeviceToHost, t.st));
    }
    for (uint b = 0; b < numBatches; b++) { PT_CHECK_HIP(c, hipStreamSynchronize(B[b].st)); B[b].active = B[b].hwc->extendCount[0]; }
    const uint maxIter = c->S.bounceCount + 2 + ((c->S.nestedDielectricsQuality == 2) ? 16u : (c->S.nestedDielectricsQuality == 1 ? 4u : 0u)) * (c->S.bounceCount + 1u);
    bool any = true;
    while (any) {
        for (uint b = 0; b < numBatches; b++) {      // phase 1: every live batch queues extend + shade and the read-back of its queue counts
            Batch& t = B[b]; t.waiting = false;
            if (!t.active || t.iterations >= maxIter) continue;
            const uint nxt = t.cur ^ 1u;
            launch_pass_reset(t.aux.counts, &t.wc->extendCount[nxt], &t.wc->shadowCount, t.st);
            launch_extend(t.sc, t.pool, t.queue[t.cur], &t.wc->extendCount[t.cur], t.active, t.wc, c->countersEnabled, t.aux, t.st, /*ranged*/ t.iterations == 0u && PT_SP_FILL_RANGED);
            launch_sp_fill_shade(t.k, sp, t.pool, t.queue[t.cur], &t.wc->extendCount[t.cur], t.active, t.queue[nxt], &t.wc->extendCount[nxt], t.sq, t.newL, sampleIndex, t.wc,
                                 (PT_SP_FILL_CLASSES && t.active >= PT_CLASSIFY_FROM) ? reinterpret_cast<uint*>(t.aux.bestKey) : nullptr, t.aux.counts + PASS_CLASS_OFFSET, t.st);      // (the straggler keys are idle between k_resolve_extend and the shadow launch, as in pt_render)
            t.rays += t.active;
            PT_CHECK_HIP(c, hipMemcpyAsync(t.hwc, t.wc, 16, hipMemcpyDeviceToHost, t.st));
            t.waiting = true;
        }
        any = false;
        // phase 2: as each batch's counts arrive, its visibility rays and their resolve; the other batches keep the GPU busy meanwhile
        for (uint b = 0; b < numBatches; b++) {
            Batch& t = B[b];
            if (!t.waiting) continue;
            PT_CHECK_HIP(c, hipStreamSynchronize(t.st));
            const uint nxt = t.cur ^ 1u, nShadow = t.hwc->shadowCount;
            if (nShadow) {
                TravAux auxShadow = t.aux; auxShadow.counts = t.aux.counts + PASS_SHADOW_OFFSET;
                launch_shadow(t.sc, t.markPool, t.sq, &t.wc->shadowCount, nShadow, t.wc, c->countersEnabled, auxShadow, t.st);
                launch_sp_fill_resolve(t.pool, t.markPool.s2, t.sq, t.newL, &t.wc->shadowCount, nShadow, t.st);
                t.shadowRays += nShadow;
            }
            t.active = t.hwc->extendCount[nxt]; t.cur = nxt; t.iterations++;
            if (t.active && t.iterations < maxIter) any = true;
        }
    }
    for (uint b = 0; b < numBatches; b++) {
        Batch& t = B[b];
        launch_sp_fill_commit(t.k, sp, t.pool, t.numPix, sampleIndex, t.st);
        PT_CHECK_HIP(c, hipMemcpyAsync(t.hwc, t.wc, sizeof(WaveCounters), hipMemcpyDeviceToHost, t.st));
    }
    for (uint b = 0; b < numBatches; b++) PT_CHECK_HIP(c, hipStreamSynchronize(B[b].st));
    PT_CHECK_HIP(c, hipEventRecord(e1, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    PT_CHECK_HIP(c, hipGetLastError());
    bool overflow = false; uint active = 0;
    if (stats) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); stats->gpuMilliseconds = ms; stats->pathsTraced = numOwned; }
    for (uint b = 0; b < numBatches; b++) {
        const Batch& t = B[b]; overflow = overflow || t.hwc->overflow; active += t.active;
        if (stats) { stats->extendRays += t.rays; stats->shadowRays += t.shadowRays; stats->hits += t.hwc->hits; stats->extendLaunches += t.iterations; if (t.iterations > stats->iterations) stats->iterations = t.iterations;
                     stats->nodeVisitsExtend += t.hwc->nodeVisitsExt; stats->triTestsExtend += t.hwc->triTestsExt; stats->nodeVisitsShadow += t.hwc->nodeVisitsSh; stats->triTestsShadow += t.hwc->triTestsSh; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (overflow) return fail(c, PT_ERROR_HIP, "BVH8 traversal: stack tail or straggler task queue overflow (raise T8_SPILL_DEPTH / TASK_QUEUE_CAPACITY)");
    if (active) return fail(c, PT_ERROR_HIP, "stable-plane fill pass: paths still alive after the iteration bound");
    return PT_OK;
}
// The realtime mode's frame with everything coupled (Sample.cpp:2438-2516; pt_set_neeat on): LightsBaker::UpdateBegin -> build pass -> LightsBaker::UpdateEnd
// on THAT frame's depth and screen-space motion vectors -> the fill passes, which sample the tables just made and fill the reservoirs the next frame's
// UpdateBegin reads.
static StablePlanesContext sp_buffers(pt_context* c) {
    ptk::StablePlanesParams prm; memset(&prm, 0, sizeof(prm)); prm.activeStablePlaneCount = cStablePlaneCount;
    StablePlanesContext sp; sp.C = ptk::SP_make_consts(prm, c->width, c->height, c->S.bounceCount);
    sp.B.Header = c->dSpHeader.p; sp.B.Planes = c->dSpPlanes.p; sp.B.StableRadiance = c->dSpRadiance.p; sp.B.Depth = c->dSpDepth.p; sp.B.SpecularHitT = c->dSpHitT.p; sp.B.MotionVectors = c->dSpMotion.p; sp.B.Throughput = c->dSpThroughput.p;
    return sp;
}
// ---- the realtime frame on tile shards (no reference analogue). The baker's passes read whole neighbourhoods of three things a rank only has for its own
// tiles: last frame's reservoirs (UpdateBegin resolves them into the history), and this frame's depth and motion vectors (UpdateEnd reprojects through them).
// So a sharded frame has two exchanges: the reservoirs before UpdateBegin (neeat_exchange_feedback, as between two reference-mode frames) and the build pass's
// guides — depth and motion vectors, 16 bytes per pixel with the hit-distance word that lies between them — before UpdateEnd; every rank then runs the same
// deterministic baker passes on the same planes. With a communicator pt_realtime_frame does both itself (RCCL point-to-point in one group, un-padded, like
// pt_gather); without one the host drives the parts and moves the packed buffers: pt_neeat_pack / unpack_feedback -> pt_neeat_update_begin ->
// pt_build_stable_planes -> pt_pack / unpack_stable_plane_guides -> pt_neeat_update_end -> pt_fill_stable_planes
static int sp_exchange_guides(pt_context* c) {
    if (c->shardCount == 1 || !c->comm) return PT_OK;
    hipStream_t s = c->stream;
    const size_t n = c->owned.size(), W = SP_GUIDE_WORDS;
    // the other ranks' pixel list and the staging buffers: once per frame size, not per realtime frame (the shard layout is a function of the size)
    if (c->spGatherW != c->width || c->spGatherH != c->height) {
        std::vector<uint> others; for (uint r = 0; r < c->shardCount; r++) if (r != c->shardRank) others.insert(others.end(), c->shardPixels[r].begin(), c->shardPixels[r].end());
        PT_CHECK_HIP(c, c->dSpGatherPixels.upload(others, s)); PT_CHECK_HIP(c, c->dSpGatherRecv.resize(others.size() * W)); PT_CHECK_HIP(c, c->dSpGatherSend.resize(n * W)); PT_CHECK_HIP(c, hipStreamSynchronize(s));
        c->spGatherW = c->width; c->spGatherH = c->height;
    }
    const StablePlanesContext sp = sp_buffers(c);
    launch_sp_pack(sp, c->dOwned.p, (uint)n, c->dSpGatherSend.p, false, s, SP_GUIDE_FIRST, SP_GUIDE_WORDS);
    PT_CHECK_NCCL(c, g_rccl.GroupStart());
    size_t off = 0; ncclResult_t bad = ncclSuccess;
    for (uint r = 0; r < c->shardCount && bad == ncclSuccess; r++) {
        if (r == c->shardRank) continue;
        const size_t m = c->shardPixels[r].size();
        if (n) bad = g_rccl.Send(c->dSpGatherSend.p, n * W, ncclFloat, (int)r, c->comm, s);
        if (m && bad == ncclSuccess) bad = g_rccl.Recv(c->dSpGatherRecv.p + off * W, m * W, ncclFloat, (int)r, c->comm, s);
        off += m;
    }
    ncclResult_t ge = g_rccl.GroupEnd();
    if (bad != ncclSuccess || ge != ncclSuccess) return fail(c, PT_ERROR_HIP, std::string("stable-plane guide exchange: ") + g_rccl.GetErrorString(bad != ncclSuccess ? bad : ge));
    launch_sp_pack(sp, c->dSpGatherPixels.p, (uint)off, c->dSpGatherRecv.p, true, s, SP_GUIDE_FIRST, SP_GUIDE_WORDS);
    return PT_OK;
}
int32_t pt_pack_stable_plane_guides(pt_context* c, void* dst, size_t bytes) {
    if (!c || !dst) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes");
    if (bytes < c->owned.size() * (size_t)SP_GUIDE_WORDS * 4u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "destination too small (16 bytes per owned pixel)");
    (void)hipSetDevice(c->device);
    launch_sp_pack(sp_buffers(c), c->dOwned.p, (uint)c->owned.size(), (uint*)dst, false, c->stream, SP_GUIDE_FIRST, SP_GUIDE_WORDS);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); PT_CHECK_HIP(c, hipGetLastError());
    return PT_OK;
}
int32_t pt_unpack_stable_plane_guides(pt_context* c, const void* src, size_t bytes, uint32_t rank) {
    if (!c || !src || rank >= c->shardCount) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes (it allocates the buffers)");
    const std::vector<uint>& px = c->shardPixels[rank];
    if (bytes < px.size() * (size_t)SP_GUIDE_WORDS * 4u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "source too small (16 bytes per pixel of that rank)");
    (void)hipSetDevice(c->device);
    DevBuf<uint> tmp; PT_CHECK_HIP(c, tmp.upload(px, c->stream));
    launch_sp_pack(sp_buffers(c), tmp.p, (uint)px.size(), (uint*)src, true, c->stream, SP_GUIDE_FIRST, SP_GUIDE_WORDS);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); PT_CHECK_HIP(c, hipGetLastError());
    tmp.free();
    return PT_OK;
}
// LightsBaker::UpdateBegin / UpdateEnd as calls of their own (Rtxpt/Sample.cpp:1380-1412, 2491-2494): what pt_realtime_frame runs around the build pass, for a
// host that puts something of its own in between (a tile-sharded frame without a communicator: the exchanges above)
int32_t pt_neeat_update_begin(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    if (!c->neeat.enabled) return fail(c, PT_ERROR_NOT_READY, "pt_set_neeat first");
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    return neeat_frame(c, NEEAT_BEGIN);
}
int32_t pt_neeat_update_end(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->neeat.enabled) return fail(c, PT_ERROR_NOT_READY, "pt_set_neeat first");
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "UpdateEnd reads the frame's depth and motion vectors: pt_build_stable_planes first");
    (void)hipSetDevice(c->device);
    int r = neeat_frame(c, NEEAT_END, c->dSpDepth.p, c->dSpMotion.p); if (r != PT_OK) return r;
    if (c->feedbackRequired) c->fbSamples = 1;
    return PT_OK;
}
int32_t pt_realtime_frame(pt_context* c, uint32_t sampleIndex, const PtStablePlanesParams* params, PtFrameStats* buildStats, PtFrameStats* fillStats) {
    if (!c || !params) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    const bool baker = c->neeat.enabled && c->S.NEEEnabled && c->S.NEEFullSamples != 0u;
    if (baker && c->shardCount > 1 && !c->comm) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_realtime_frame on tile shards: the baker reads the whole frame's reservoirs, depth and motion vectors — pt_comm_init first, or drive the parts (pt_neeat_update_begin, pt_build_stable_planes, pt_pack / pt_unpack_stable_plane_guides, pt_neeat_update_end, pt_fill_stable_planes)");
    if (baker) { r = neeat_exchange_feedback(c); if (r != PT_OK) return r; r = neeat_frame(c, NEEAT_BEGIN); if (r != PT_OK) return r; }
    r = pt_build_stable_planes(c, sampleIndex, params, buildStats); if (r != PT_OK) return r;
    if (baker) { r = sp_exchange_guides(c); if (r != PT_OK) return r; r = neeat_frame(c, NEEAT_END, c->dSpDepth.p, c->dSpMotion.p); if (r != PT_OK) return r; }
    const uint32_t subSamples = params->subSampleCount ? params->subSampleCount : 1u;
    PtFrameStats total; memset(&total, 0, sizeof(total));
    for (uint32_t s = 0; s < subSamples; s++) { PtFrameStats one; r = pt_fill_stable_planes(c, sampleIndex + s, params, &one); if (r != PT_OK) return r; add_frame_stats(total, one); }
    if (fillStats) *fillStats = total;
    if (baker && c->feedbackRequired) c->fbSamples = 1;      // pt_get_light_feedback(0): the reservoirs as the fill passes left them
    return PT_OK;
}
int32_t pt_stable_planes_merge(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes, pt_fill_stable_planes");
    (void)hipSetDevice(c->device);
    ptk::StablePlanesParams prm; memset(&prm, 0, sizeof(prm)); prm.activeStablePlaneCount = cStablePlaneCount;
    StablePlanesContext sp; sp.C = ptk::SP_make_consts(prm, c->width, c->height, c->S.bounceCount); memset(&sp.B, 0, sizeof(sp.B));
    sp.B.Header = c->dSpHeader.p; sp.B.Planes = c->dSpPlanes.p; sp.B.StableRadiance = c->dSpRadiance.p;
    const uint numOwned = (uint)c->owned.size();
    if (numOwned) launch_sp_merge(sp, c->dOwned.p, numOwned, c->dAccum.p, c->stream);
    // the buffer now holds one finished frame: pt_map_radiance / pt_tonemap / pt_gather read it, the next pt_render blends into it like into any first sample
    c->accumCount = 1;
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); PT_CHECK_HIP(c, hipGetLastError());
    return PT_OK;
}
// ---- the plane buffers of tile-sharded frames (no reference analogue): every rank builds and fills the planes of its own tiles; the rank that denoises or
// shows the frame needs them all. 284 bytes per pixel (SP_SHARD_WORDS): header, three plane records, stable radiance, depth, specular hit distance, motion
// vectors, throughput.
int32_t pt_stable_planes_shard_bytes(pt_context* c, uint32_t rank, size_t* bytes) {
    if (!c || !bytes || rank >= c->shardCount || !c->width) return PT_ERROR_INVALID_ARGUMENT;
    *bytes = c->shardPixels[rank].size() * (size_t)SP_SHARD_WORDS * 4u; return PT_OK;
}
int32_t pt_pack_stable_planes(pt_context* c, void* dst, size_t bytes) {
    if (!c || !dst) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes");
    if (bytes < c->owned.size() * (size_t)SP_SHARD_WORDS * 4u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "destination too small (pt_stable_planes_shard_bytes)");
    (void)hipSetDevice(c->device);
    launch_sp_pack(sp_buffers(c), c->dOwned.p, (uint)c->owned.size(), (uint*)dst, false, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); PT_CHECK_HIP(c, hipGetLastError());
    return PT_OK;
}
int32_t pt_unpack_stable_planes(pt_context* c, const void* src, size_t bytes, uint32_t rank) {
    if (!c || !src || rank >= c->shardCount) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes (it allocates the buffers)");
    const std::vector<uint>& px = c->shardPixels[rank];
    if (bytes < px.size() * (size_t)SP_SHARD_WORDS * 4u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "source too small (pt_stable_planes_shard_bytes)");
    (void)hipSetDevice(c->device);
    DevBuf<uint> tmp; PT_CHECK_HIP(c, tmp.upload(px, c->stream));
    launch_sp_pack(sp_buffers(c), tmp.p, (uint)px.size(), (uint*)src, true, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); PT_CHECK_HIP(c, hipGetLastError());
    tmp.free();
    c->spGathered = true;      // (the host says when all ranks are in: pt_denoise_spec_hit_t trusts it from here on)
    return PT_OK;
}
// pt_gather for the plane buffers: every rank sends its tiles' records to rank 0 (RCCL point-to-point inside one group, un-padded, on the library's stream); a
// world of one with a communicator runs the protocol as a loop-back with the buffers poisoned in between, like pt_gather
int32_t pt_gather_stable_planes(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes");
    if (c->shardCount == 1 && !c->comm) return PT_OK;
    if (!c->comm) return fail(c, PT_ERROR_NOT_READY, "pt_comm_init first");
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const StablePlanesContext sp = sp_buffers(c);
    const size_t W = SP_SHARD_WORDS, n = c->owned.size();
    if (c->shardCount == 1) {
        if (!n) return PT_OK;
        PT_CHECK_HIP(c, c->dSpGatherSend.resize(n * W)); PT_CHECK_HIP(c, c->dSpGatherRecv.resize(n * W));
        launch_sp_pack(sp, c->dOwned.p, (uint)n, c->dSpGatherSend.p, false, st);
        const size_t N = (size_t)c->width * c->height;
        PT_CHECK_HIP(c, hipMemsetAsync(c->dSpHeader.p, 0xEE, 16 * N, st)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpPlanes.p, 0xEE, sizeof(ptk::StablePlane) * cStablePlaneCount * sp.C.genericTSPlaneStride, st));
        PT_CHECK_HIP(c, hipMemsetAsync(c->dSpRadiance.p, 0xEE, 8 * N, st)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpDepth.p, 0xEE, 4 * N, st)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpHitT.p, 0xEE, 4 * N, st));
        PT_CHECK_HIP(c, hipMemsetAsync(c->dSpMotion.p, 0xEE, 8 * N, st)); PT_CHECK_HIP(c, hipMemsetAsync(c->dSpThroughput.p, 0xEE, 4 * N, st));
        PT_CHECK_NCCL(c, g_rccl.GroupStart());
        ncclResult_t rs = g_rccl.Send(c->dSpGatherSend.p, n * W, ncclFloat, 0, c->comm, st);
        ncclResult_t rr = (rs == ncclSuccess) ? g_rccl.Recv(c->dSpGatherRecv.p, n * W, ncclFloat, 0, c->comm, st) : rs;
        ncclResult_t re = g_rccl.GroupEnd();
        if (rr != ncclSuccess || re != ncclSuccess) return fail(c, PT_ERROR_HIP, std::string("pt_gather_stable_planes loop-back: ") + g_rccl.GetErrorString(rr != ncclSuccess ? rr : re));
        launch_sp_pack(sp, c->dOwned.p, (uint)n, c->dSpGatherRecv.p, true, st);
        PT_CHECK_HIP(c, hipStreamSynchronize(st));
        return PT_OK;
    }
    if (c->shardRank != 0) {
        if (n) { PT_CHECK_HIP(c, c->dSpGatherSend.resize(n * W)); launch_sp_pack(sp, c->dOwned.p, (uint)n, c->dSpGatherSend.p, false, st); PT_CHECK_NCCL(c, g_rccl.Send(c->dSpGatherSend.p, n * W, ncclFloat, 0, c->comm, st)); }
        PT_CHECK_HIP(c, hipStreamSynchronize(st));
        return PT_OK;
    }
    std::vector<uint> others; for (uint r = 1; r < c->shardCount; r++) others.insert(others.end(), c->shardPixels[r].begin(), c->shardPixels[r].end());
    PT_CHECK_HIP(c, c->dSpGatherPixels.upload(others, st)); PT_CHECK_HIP(c, c->dSpGatherRecv.resize(others.size() * W)); PT_CHECK_HIP(c, hipStreamSynchronize(st));
    size_t off = 0; ncclResult_t bad = ncclSuccess;
    PT_CHECK_NCCL(c, g_rccl.GroupStart());
    for (uint r = 1; r < c->shardCount && bad == ncclSuccess; r++) { const size_t m = c->shardPixels[r].size(); if (m) bad = g_rccl.Recv(c->dSpGatherRecv.p + off * W, m * W, ncclFloat, (int)r, c->comm, st); off += m; }
    ncclResult_t ge = g_rccl.GroupEnd();
    if (bad != ncclSuccess || ge != ncclSuccess) return fail(c, PT_ERROR_HIP, std::string("pt_gather_stable_planes: ") + g_rccl.GetErrorString(bad != ncclSuccess ? bad : ge));
    launch_sp_pack(sp, c->dSpGatherPixels.p, (uint)off, c->dSpGatherRecv.p, true, st);
    PT_CHECK_HIP(c, hipStreamSynchronize(st));
    c->spGathered = true;
    return PT_OK;
}
int32_t pt_denoise_spec_hit_t(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes, pt_fill_stable_planes");
    if (c->shardCount > 1 && !c->spGathered) return fail(c, PT_ERROR_INVALID_ARGUMENT, "the fill-in reads 5 x 5 neighbourhoods: run it on the gathered planes (pt_gather_stable_planes / pt_unpack_stable_planes on rank 0), not on one rank's tiles");
    (void)hipSetDevice(c->device);
    PT_CHECK_HIP(c, c->dSpScratch.resize((size_t)c->width * c->height));
    launch_sp_denoise_spec_hit_t(c->dSpHitT.p, c->dSpDepth.p, c->dSpScratch.p, c->width, c->height, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream)); PT_CHECK_HIP(c, hipGetLastError());
    return PT_OK;
}
int32_t pt_get_stable_planes(pt_context* c, uint32_t* header, PtStablePlane* planes, size_t planeCapacity, uint16_t* stableRadiance, float* depth, float* specularHitT, uint16_t* motionVectors, uint32_t* throughput) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->spW || c->spW != c->width || c->spH != c->height) return fail(c, PT_ERROR_NOT_READY, "no stable planes of this frame size yet: pt_build_stable_planes");
    (void)hipSetDevice(c->device);
    const size_t N = (size_t)c->width * c->height, nPlanes = (size_t)cStablePlaneCount * ptk::GenericTSComputePlaneStride(c->width, c->height);
    if (planes && planeCapacity < nPlanes) return fail(c, PT_ERROR_INVALID_ARGUMENT, "plane buffer smaller than 3 x the plane stride (pt_stable_planes_plane_stride)");
    if (header) PT_CHECK_HIP(c, hipMemcpy(header, c->dSpHeader.p, 16 * N, hipMemcpyDeviceToHost));
    if (planes) PT_CHECK_HIP(c, hipMemcpy(planes, c->dSpPlanes.p, sizeof(ptk::StablePlane) * nPlanes, hipMemcpyDeviceToHost));
    if (stableRadiance) PT_CHECK_HIP(c, hipMemcpy(stableRadiance, c->dSpRadiance.p, 8 * N, hipMemcpyDeviceToHost));
    if (depth) PT_CHECK_HIP(c, hipMemcpy(depth, c->dSpDepth.p, 4 * N, hipMemcpyDeviceToHost));
    if (specularHitT) PT_CHECK_HIP(c, hipMemcpy(specularHitT, c->dSpHitT.p, 4 * N, hipMemcpyDeviceToHost));
    if (motionVectors) PT_CHECK_HIP(c, hipMemcpy(motionVectors, c->dSpMotion.p, 8 * N, hipMemcpyDeviceToHost));
    if (throughput) PT_CHECK_HIP(c, hipMemcpy(throughput, c->dSpThroughput.p, 4 * N, hipMemcpyDeviceToHost));
    return PT_OK;
}
int32_t pt_map_radiance(pt_context* c, const float** rgba, size_t* pitch) {
    if (!c || !rgba) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    (void)hipSetDevice(c->device);
    c->hostRadiance.resize((size_t)c->width * c->height * 4);
    PT_CHECK_HIP(c, hipMemcpyAsync(c->hostRadiance.data(), c->dAccum.p, sizeof(float) * c->hostRadiance.size(), hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    *rgba = c->hostRadiance.data(); if (pitch) *pitch = (size_t)c->width * 16;
    return PT_OK;
}
int32_t pt_unmap_radiance(pt_context* c) { return c ? PT_OK : PT_ERROR_INVALID_ARGUMENT; }

int32_t pt_default_tonemap(PtToneMapParams* out, float exposureCompensation, float filmSpeed, float shutter, float fNumber) {
    if (!out || !(shutter > 0.f) || !(fNumber > 0.f)) return PT_ERROR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    out->whiteScale = 5.1f; out->whiteMaxLuminance = 1.0f; out->toneMapOperator = 5u; out->clamped = 1u; out->enabled = 1u;      // ToneMappingPasses.h:36-53
    // ToneMappingPasses.cpp:337-338
    out->autoExposure = 0u; out->avgLuminance = 1.0f; out->autoExposureLumValueMin = exp2f(-16.0f); out->autoExposureLumValueMax = exp2f(16.0f);
    // UpdateColorTransform (ToneMappingPasses.cpp:428-441), white balance off => identity * exposureScale * manualExposureScale
    float exposureScale = powf(2.f, exposureCompensation);
    float manualExposureScale = ((1.f / 100.f) * filmSpeed) / (shutter * fNumber * fNumber);
    float s = exposureScale * manualExposureScale;
    out->colorTransform[0] = s; out->colorTransform[4] = s; out->colorTransform[8] = s;
    return PT_OK;
}
int32_t pt_average_luminance(pt_context* c, float* avgLuminance) {
    if (!c || !avgLuminance) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    (void)hipSetDevice(c->device);
    size_t n = (size_t)ptk::tm_pow2_floor(c->width) * ptk::tm_pow2_floor(c->height);
    DevBuf<float> d; PT_CHECK_HIP(c, d.resize(2 * n));
    float* result = nullptr; float logLum = 0.f;
    launch_average_log_luminance(c->dAccum.p, c->width, c->height, d.p, &result, c->stream);
    PT_CHECK_HIP(c, hipMemcpyAsync(&logLum, result, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    PT_CHECK_HIP(c, hipGetLastError());
    d.free();
    *avgLuminance = exp2f(logLum);                      // ToneMappingPasses.cpp:284
    return PT_OK;
}
int32_t pt_tonemap(pt_context* c, const PtToneMapParams* params, uint8_t* rgba8, size_t bytes) {
    if (!c || !params || !rgba8) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    size_t n = (size_t)c->width * c->height;
    if (bytes < n * 4) return fail(c, PT_ERROR_INVALID_ARGUMENT, "rgba8 buffer too small");
    if (params->toneMapOperator > 5u) return fail(c, PT_ERROR_INVALID_ARGUMENT, "unknown tone map operator");
    (void)hipSetDevice(c->device);
    static_assert(sizeof(PtToneMapParams) == sizeof(ptk::ToneMapParams), "tone map parameter layout");
    ptk::ToneMapParams p; memcpy(&p, params, sizeof(p));
    DevBuf<uint> d; PT_CHECK_HIP(c, d.resize(n));
    launch_tonemap(c->dAccum.p, (uint)n, p, d.p, c->stream);
    PT_CHECK_HIP(c, hipMemcpyAsync(rgba8, d.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    PT_CHECK_HIP(c, hipGetLastError());
    d.free();
    return PT_OK;
}

int32_t pt_shard_info(pt_context* c, uint32_t* numOwned, size_t* bytes) {
    if (!c || !c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    if (numOwned) *numOwned = (uint32_t)c->owned.size(); if (bytes) *bytes = c->owned.size() * 16;
    return PT_OK;
}
int32_t pt_pack_shard(pt_context* c, void* dst, size_t bytes) {
    if (!c || !dst || !c->width) return fail(c, PT_ERROR_INVALID_ARGUMENT, "bad argument");
    if (bytes < c->owned.size() * 16) return fail(c, PT_ERROR_INVALID_ARGUMENT, "destination too small");
    (void)hipSetDevice(c->device);
    launch_pack(c->dAccum.p, c->dOwned.p, (uint)c->owned.size(), c->width, (ptk::float4*)dst, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}
int32_t pt_unpack_shard(pt_context* c, const void* src, size_t bytes, uint32_t rank) {
    if (!c || !src || !c->width || rank >= c->shardCount) return fail(c, PT_ERROR_INVALID_ARGUMENT, "bad argument");
    const std::vector<uint>& px = c->shardPixels[rank];
    if (bytes < px.size() * 16) return fail(c, PT_ERROR_INVALID_ARGUMENT, "source too small");
    (void)hipSetDevice(c->device);
    DevBuf<uint> tmp; PT_CHECK_HIP(c, tmp.upload(px, c->stream));
    launch_unpack(c->dAccum.p, tmp.p, (uint)px.size(), c->width, (const ptk::float4*)src, c->stream);
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    tmp.free();
    return PT_OK;
}
int32_t pt_device_radiance(pt_context* c, void** p) { if (!c || !p || !c->width) return PT_ERROR_INVALID_ARGUMENT; *p = c->dAccum.p; return PT_OK; }

static int trace_probe(pt_context* c, const float* rays, uint32_t n, float* outClosest, uint32_t* outVisible, double* kernelMs) {
    if (!c || !rays || !n) return fail(c, PT_ERROR_INVALID_ARGUMENT, "bad argument");
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    DevBuf<ptk::float4> dr, dc; DevBuf<uint> dv;
    PT_CHECK_HIP(c, dr.upload((const ptk::float4*)rays, 2 * (size_t)n, c->stream));
    if (outClosest) PT_CHECK_HIP(c, dc.resize(n)); else PT_CHECK_HIP(c, dv.resize(n));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    launch_trace_probe(c->dsc, dr.p, n, outClosest ? dc.p : nullptr, outClosest ? nullptr : dv.p, &c->dCounters.p->overflow, c->stream);
    (void)hipEventRecord(e1, c->stream);
    if (outClosest) PT_CHECK_HIP(c, hipMemcpyAsync(outClosest, dc.p, 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    else PT_CHECK_HIP(c, hipMemcpyAsync(outVisible, dv.p, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); if (kernelMs) *kernelMs = ms; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    dr.free(); dc.free(); dv.free();
    return PT_OK;
}
int32_t pt_trace_closest(pt_context* c, const float* rays, uint32_t n, float* out, double* ms) { if (!out) return PT_ERROR_INVALID_ARGUMENT; return trace_probe(c, rays, n, out, nullptr, ms); }
int32_t pt_trace_visibility(pt_context* c, const float* rays, uint32_t n, uint32_t* out, double* ms) { if (!out) return PT_ERROR_INVALID_ARGUMENT; return trace_probe(c, rays, n, nullptr, out, ms); }

int32_t pt_get_lights(pt_context* c, uint32_t* nLights, uint32_t* nProxies, void* lights, void* lightsEx, uint32_t* pc, uint32_t* pi, uint32_t* envLookup, uint32_t* envDim) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    if (nLights) *nLights = (uint32_t)c->lights.size(); if (nProxies) *nProxies = c->numProxies; if (envDim) *envDim = c->envLookupDim;
    // read back from the DEVICE copies: this is what the kernels sample from
    if (lights && c->lights.size()) PT_CHECK_HIP(c, hipMemcpy(lights, c->dLights.p, 32 * c->lights.size(), hipMemcpyDeviceToHost));
    if (lightsEx && c->lights.size()) PT_CHECK_HIP(c, hipMemcpy(lightsEx, c->dLightsEx.p, 16 * c->lights.size(), hipMemcpyDeviceToHost));
    if (pc && c->lights.size()) PT_CHECK_HIP(c, hipMemcpy(pc, c->dProxyCounters.p, 4 * c->lights.size(), hipMemcpyDeviceToHost));
    if (pi && c->numProxies) PT_CHECK_HIP(c, hipMemcpy(pi, c->dProxyIndices.p, 4 * (size_t)c->numProxies, hipMemcpyDeviceToHost));
    if (envLookup && c->envLookup.size()) PT_CHECK_HIP(c, hipMemcpy(envLookup, c->dEnvLookup.p, 4 * c->envLookup.size(), hipMemcpyDeviceToHost));
    return PT_OK;
}
int32_t pt_get_env_cube(pt_context* c, uint32_t* texelCount, uint32_t* dim, uint32_t* mipLevels, void* texels8B, uint32_t capacityTexels) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    size_t n = 0;
    if (c->envEnabled) { const uint last = c->envCube.mipLevels - 1u, d = c->envCube.dim >> last; n = (size_t)c->envCube.mipOffset[last] + 6ull * d * d; }
    if (texelCount) *texelCount = (uint32_t)n; if (dim) *dim = c->envEnabled ? c->envCube.dim : 0; if (mipLevels) *mipLevels = c->envEnabled ? c->envCube.mipLevels : 0;
    if (texels8B && n) {
        if (capacityTexels < n) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_get_env_cube: buffer too small");
        PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        PT_CHECK_HIP(c, hipMemcpy(texels8B, c->dEnvCube.p, 8 * n, hipMemcpyDeviceToHost));
    }
    return PT_OK;
}
int32_t pt_get_subinstances(pt_context* c, uint32_t* count, void* out) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    if (count) *count = (uint32_t)c->subInstances.size();
    if (out && c->subInstances.size()) PT_CHECK_HIP(c, hipMemcpy(out, c->dSubInstances.p, 32 * c->subInstances.size(), hipMemcpyDeviceToHost));
    return PT_OK;
}
int32_t pt_get_scene_info(pt_context* c, uint32_t* nTris, uint32_t* nNodes, uint32_t* nInst, uint32_t* nMat) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    if (nTris) *nTris = c->numTris; if (nNodes) *nNodes = c->numTris ? c->bvh.numNodes8 : 0;      /* BVH8 nodes */ if (nInst) *nInst = (uint32_t)c->instances.size(); if (nMat) *nMat = (uint32_t)c->materials.size();
    return PT_OK;
}
#ifdef MI355PT_TEST_HOOKS      // include/mi355pt_testhooks.h: exported by libmi355pt_testhooks.so only
int32_t pt_probe(pt_context* c, int32_t kind, const void* in, size_t inBytes, void* out, size_t outBytes, uint32_t n) {
    if (!c || !in || !out || !n) return fail(c, PT_ERROR_INVALID_ARGUMENT, "bad argument");
    (void)hipSetDevice(c->device);
    // the surface, environment and alpha-test probes read the scene
    if (kind == 8 || kind == 9 || kind == 10) { int r = prepare(c); if (r != PT_OK) return r; }
    DevBuf<unsigned char> di, dout;
    PT_CHECK_HIP(c, di.upload((const unsigned char*)in, inBytes, c->stream)); PT_CHECK_HIP(c, dout.resize(outBytes));
    PathKernelContext k; k.sc = c->dsc; k.S = c->S; k.cam = c->cam;
    launch_probe(k, kind, di.p, dout.p, n, c->stream);
    PT_CHECK_HIP(c, hipMemcpyAsync(out, dout.p, outBytes, hipMemcpyDeviceToHost, c->stream));
    PT_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    PT_CHECK_HIP(c, hipGetLastError());
    di.free(); dout.free();
    return PT_OK;
}
#endif
// ---- the frame gather (include/mi355pt.h "the frame gather itself")
int32_t pt_shard_layout(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t* pixels, uint32_t capacity, uint32_t* count) {
    if (!width || !height || width > 65535 || height > 65535 || !world || rank >= world) return PT_ERROR_INVALID_ARGUMENT;
    std::vector<std::vector<uint>> lists; shard_pixel_lists(width, height, world, lists);
    const std::vector<uint>& mine = lists[rank];
    if (count) *count = (uint32_t)mine.size();
    if (pixels) { if (capacity < mine.size()) return PT_ERROR_INVALID_ARGUMENT; if (!mine.empty()) memcpy(pixels, mine.data(), 4 * mine.size()); }
    return PT_OK;
}
int32_t pt_comm_unique_id(void* id128) {
    if (!id128) return PT_ERROR_INVALID_ARGUMENT;
    if (!g_rccl.load()) return PT_ERROR_HIP;
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id; if (g_rccl.GetUniqueId(&id) != ncclSuccess) return PT_ERROR_HIP;
    memcpy(id128, &id, sizeof(id));
    return PT_OK;
}
int32_t pt_comm_init(pt_context* c, const void* id128, uint32_t rank, uint32_t world) {
    if (!c || !id128) return fail(c, PT_ERROR_INVALID_ARGUMENT, "null argument");
    if (rank != c->shardRank || world != c->shardCount) return fail(c, PT_ERROR_INVALID_ARGUMENT, "pt_comm_init: rank / world must equal the context's shardRank / shardCount");
    if (!g_rccl.load()) return fail(c, PT_ERROR_HIP, g_rccl.error);
    (void)hipSetDevice(c->device);
    if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    PT_CHECK_NCCL(c, g_rccl.CommInitRank(&c->comm, (int)world, id, (int)rank));
    c->commRank = rank; c->commWorld = world; c->gatherW = c->gatherH = 0;
    return PT_OK;
}
int32_t pt_comm_destroy(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (c->comm && g_rccl.lib) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); (void)g_rccl.CommDestroy(c->comm); }
    c->comm = nullptr;
    return PT_OK;
}
int32_t pt_gather(pt_context* c) {
    if (!c) return PT_ERROR_INVALID_ARGUMENT;
    if (!c->width) return fail(c, PT_ERROR_NOT_READY, "pt_resize first");
    if (c->shardCount == 1 && !c->comm) return PT_OK;                           // nothing to gather
    if (!c->comm) return fail(c, PT_ERROR_NOT_READY, "pt_comm_init first");
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    if (c->shardCount == 1) {
        // a world of one WITH a communicator: the whole protocol as a loop-back — pack, ncclSend to self + ncclRecv from self inside one group, unpack — so
        // that the run-time-bound RCCL path can be exercised (and checked) on a one-GPU box. The frame is poisoned between pack and unpack: what
        // pt_map_radiance returns afterwards has been through RCCL.
        const size_t n = c->owned.size();
        if (!n) return PT_OK;
        PT_CHECK_HIP(c, c->dGatherSend.resize(n)); PT_CHECK_HIP(c, c->dGatherRecv.resize(n));
        launch_pack(c->dAccum.p, c->dOwned.p, (uint)n, c->width, c->dGatherSend.p, st);
        PT_CHECK_HIP(c, hipMemsetAsync(c->dAccum.p, 0xFF, sizeof(ptk::float4) * (size_t)c->width * c->height, st));
        PT_CHECK_NCCL(c, g_rccl.GroupStart());
        ncclResult_t rs = g_rccl.Send(c->dGatherSend.p, 4 * n, ncclFloat, 0, c->comm, st);
        ncclResult_t rr = (rs == ncclSuccess) ? g_rccl.Recv(c->dGatherRecv.p, 4 * n, ncclFloat, 0, c->comm, st) : rs;
        ncclResult_t re = g_rccl.GroupEnd();
        if (rr != ncclSuccess || re != ncclSuccess) return fail(c, PT_ERROR_HIP, std::string("pt_gather loop-back: ") + g_rccl.GetErrorString(rr != ncclSuccess ? rr : re));
        launch_unpack(c->dAccum.p, c->dOwned.p, (uint)n, c->width, c->dGatherRecv.p, st);
        return PT_OK;
    }
    // per-size state: counts of every rank; on rank 0 the other ranks' pixel lists on the device
    if (c->gatherW != c->width || c->gatherH != c->height) {
        c->gatherCounts.assign(c->shardCount, 0);
        std::vector<uint> others;
        for (uint r = 0; r < c->shardCount; r++) { c->gatherCounts[r] = c->shardPixels[r].size(); if (r != 0 && c->shardRank == 0) others.insert(others.end(), c->shardPixels[r].begin(), c->shardPixels[r].end()); }
        if (c->shardRank == 0) { PT_CHECK_HIP(c, c->dGatherPixels.upload(others, st)); PT_CHECK_HIP(c, c->dGatherRecv.resize(others.size())); PT_CHECK_HIP(c, hipStreamSynchronize(st)); }
        else PT_CHECK_HIP(c, c->dGatherSend.resize(c->owned.size()));
        c->gatherW = c->width; c->gatherH = c->height;
    }
    if (c->shardRank != 0) {
        const size_t n = c->owned.size();
        if (n) {
            launch_pack(c->dAccum.p, c->dOwned.p, (uint)n, c->width, c->dGatherSend.p, st);
            PT_CHECK_NCCL(c, g_rccl.Send(c->dGatherSend.p, 4 * n, ncclFloat, 0, c->comm, st));
        }
        return PT_OK;
    }
    size_t off = 0, total = 0;
    PT_CHECK_NCCL(c, g_rccl.GroupStart());
    for (uint r = 1; r < c->shardCount; r++) {
        const size_t n = c->gatherCounts[r];
        if (n) { ncclResult_t rr = g_rccl.Recv(c->dGatherRecv.p + off, 4 * n, ncclFloat, (int)r, c->comm, st); if (rr != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(c, PT_ERROR_HIP, std::string("ncclRecv: ") + g_rccl.GetErrorString(rr)); } }
        off += n;
    }
    PT_CHECK_NCCL(c, g_rccl.GroupEnd());
    total = off;
    if (total) launch_unpack(c->dAccum.p, c->dGatherPixels.p, (uint)total, c->width, c->dGatherRecv.p, st);
    return PT_OK;
}
// The NEE-AT feedback exchange of tile-sharded frames (neeat_exchange_feedback) over HOST memory and the caller's transport: every rank sends the reservoirs
// and the exported depth of its own pixels (12 bytes each) to every other rank and receives theirs. Pairs meet in rank order (the lower rank sends first), so
// blocking transports cannot deadlock.
int32_t pt_neeat_exchange_host(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, float* totalWeight, uint32_t* candidates, float* depth, const PtTransport* t) {
    if (!totalWeight || !candidates || !depth || !t || !t->send || !t->recv || !width || !height || width > 65535 || height > 65535 || !world || rank >= world) return PT_ERROR_INVALID_ARGUMENT;
    if (world == 1) return PT_OK;
    try {
        std::vector<std::vector<uint>> lists; shard_pixel_lists(width, height, world, lists);
        auto slot = [&](uint px) { return (size_t)(px & 0xFFFFu) * width + (px >> 16); };
        const std::vector<uint>& mine = lists[rank];
        std::vector<uint> sendbuf(3 * mine.size()), recvbuf;
        for (size_t i = 0; i < mine.size(); i++) { memcpy(&sendbuf[3 * i], &totalWeight[slot(mine[i])], 4); sendbuf[3 * i + 1] = candidates[slot(mine[i])]; memcpy(&sendbuf[3 * i + 2], &depth[slot(mine[i])], 4); }
        for (uint p = 0; p < world; p++) {
            if (p == rank) continue;
            recvbuf.resize(3 * lists[p].size());
            for (int step = 0; step < 2; step++) {
                const bool sendNow = (rank < p) == (step == 0);
                if (sendNow) { if (!mine.empty() && t->send(t->user, sendbuf.data(), 12 * mine.size(), p) != 0) return PT_ERROR_IO; }
                else if (!lists[p].empty() && t->recv(t->user, recvbuf.data(), 12 * lists[p].size(), p) != 0) return PT_ERROR_IO;
            }
            for (size_t i = 0; i < lists[p].size(); i++) { memcpy(&totalWeight[slot(lists[p][i])], &recvbuf[3 * i], 4); candidates[slot(lists[p][i])] = recvbuf[3 * i + 1]; memcpy(&depth[slot(lists[p][i])], &recvbuf[3 * i + 2], 4); }
        }
        return PT_OK;
    } catch (...) { return PT_ERROR_IO; }
}
// The general form over HOST memory: numPlanes row-major width x height planes of bytesPerPixel[k] bytes per pixel each (depth: 4, motion vectors: 8, a header
// plane: 4 ...). toRoot = 0: every rank sends the records of its own pixels to every other rank and receives theirs (the guide exchange of a tile-sharded
// realtime frame; pt_neeat_exchange_host is this with three 4-byte planes). toRoot = 1: every rank sends to rank 0 only (the plane-buffer gather). A record =
// the planes' bytes of one pixel back to back, in the rank's pixel order (pt_shard_layout); transfers are un-padded; pairs meet in rank order, so blocking
// transports cannot deadlock.
int32_t pt_exchange_planes_host(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, void* const* planes, const uint32_t* bytesPerPixel, uint32_t numPlanes, int32_t toRoot, const PtTransport* t) {
    if (!planes || !bytesPerPixel || !numPlanes || !t || !t->send || !t->recv || !width || !height || width > 65535 || height > 65535 || !world || rank >= world) return PT_ERROR_INVALID_ARGUMENT;
    size_t rec = 0; for (uint32_t k = 0; k < numPlanes; k++) { if (!planes[k] || !bytesPerPixel[k]) return PT_ERROR_INVALID_ARGUMENT; rec += bytesPerPixel[k]; }
    if (world == 1) return PT_OK;
    try {
        std::vector<std::vector<uint>> lists; shard_pixel_lists(width, height, world, lists);
        auto slot = [&](uint px) { return (size_t)(px & 0xFFFFu) * width + (px >> 16); };
        auto pack = [&](const std::vector<uint>& px, std::vector<unsigned char>& buf) { buf.resize(rec * px.size()); size_t o = 0; for (uint q : px) for (uint32_t k = 0; k < numPlanes; k++) { memcpy(&buf[o], (const unsigned char*)planes[k] + slot(q) * bytesPerPixel[k], bytesPerPixel[k]); o += bytesPerPixel[k]; } };
        auto unpack = [&](const std::vector<uint>& px, const std::vector<unsigned char>& buf) { size_t o = 0; for (uint q : px) for (uint32_t k = 0; k < numPlanes; k++) { memcpy((unsigned char*)planes[k] + slot(q) * bytesPerPixel[k], &buf[o], bytesPerPixel[k]); o += bytesPerPixel[k]; } };
        const std::vector<uint>& mine = lists[rank];
        std::vector<unsigned char> sendbuf, recvbuf; pack(mine, sendbuf);
        for (uint p = 0; p < world; p++) {
            if (p == rank) continue;
            const bool iSend = !toRoot || p == 0u, iRecv = !toRoot || rank == 0u;
            recvbuf.resize(rec * lists[p].size());
            for (int step = 0; step < 2; step++) {
                const bool sendNow = (rank < p) == (step == 0);
                if (sendNow) { if (iSend && !mine.empty() && t->send(t->user, sendbuf.data(), sendbuf.size(), p) != 0) return PT_ERROR_IO; }
                else if (iRecv && !lists[p].empty() && t->recv(t->user, recvbuf.data(), recvbuf.size(), p) != 0) return PT_ERROR_IO;
            }
            if (iRecv) unpack(lists[p], recvbuf);
        }
        return PT_OK;
    } catch (...) { return PT_ERROR_IO; }
}
int32_t pt_gather_host(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, float* rgba, const PtTransport* t) {
    if (!rgba || !t || !t->send || !t->recv || !width || !height || width > 65535 || height > 65535 || !world || rank >= world) return PT_ERROR_INVALID_ARGUMENT;
    if (world == 1) return PT_OK;
    try {
        std::vector<std::vector<uint>> lists; shard_pixel_lists(width, height, world, lists);
        auto at = [&](uint px) { return rgba + 4 * ((size_t)(px & 0xFFFFu) * width + (px >> 16)); };
        if (rank != 0) {
            const std::vector<uint>& mine = lists[rank];
            std::vector<float> sendbuf(4 * mine.size());
            for (size_t i = 0; i < mine.size(); i++) memcpy(&sendbuf[4 * i], at(mine[i]), 16);
            if (!mine.empty() && t->send(t->user, sendbuf.data(), 16 * mine.size(), 0) != 0) return PT_ERROR_IO;
            return PT_OK;
        }
        std::vector<std::vector<float>> recvbuf(world);
        if (t->group_begin && t->group_begin(t->user) != 0) return PT_ERROR_IO;
        for (uint r = 1; r < world; r++) { recvbuf[r].resize(4 * lists[r].size()); if (!lists[r].empty() && t->recv(t->user, recvbuf[r].data(), 16 * lists[r].size(), r) != 0) return PT_ERROR_IO; }
        if (t->group_end && t->group_end(t->user) != 0) return PT_ERROR_IO;
        for (uint r = 1; r < world; r++) for (size_t i = 0; i < lists[r].size(); i++) memcpy(at(lists[r][i]), &recvbuf[r][4 * i], 16);
        return PT_OK;
    } catch (...) { return PT_ERROR_IO; }
}

int32_t pt_get_build_stats(pt_context* c, double* b, double* r, double* l) { if (!c) return PT_ERROR_INVALID_ARGUMENT; if (b) *b = c->buildMs; if (r) *r = c->refitMs; if (l) *l = c->lightBakeMs; return PT_OK; }
int32_t pt_get_bvh_info(pt_context* c, PtBvhInfo* out) {
    if (!c || !out) return PT_ERROR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    int r = prepare(c); if (r != PT_OK) return r;
    memset(out, 0, sizeof(*out));
    out->builder = c->bvh.builder; out->builtOnDevice = (c->bvh.builder == BVH_BUILDER_SAH && c->bvh.hostBuildMs > 0.f) ? 0u : 1u; out->numTriangles = c->numTris;
    out->numWideNodes = c->numTris ? c->bvh.numNodes8 : 0u; out->collapseLevels = c->bvh.collapseLevels; out->optimiserPasses = c->bvh.optimiserPasses; out->hostMs = c->bvh.hostBuildMs; out->buildMs = (float)c->buildMs;
    return PT_OK;
}
int32_t pt_set_counters(pt_context* c, int32_t enable) { if (!c) return PT_ERROR_INVALID_ARGUMENT; c->countersEnabled = enable != 0; return PT_OK; }
int32_t pt_set_serial_kernels(pt_context* c, int32_t enable) { if (!c) return PT_ERROR_INVALID_ARGUMENT; c->serialKernels = enable != 0; return PT_OK; }
int32_t pt_set_tail_paths(pt_context* c, uint32_t maxPaths) { if (!c) return PT_ERROR_INVALID_ARGUMENT; c->tailBelow = maxPaths; return PT_OK; }
int32_t pt_set_fused_traversal(pt_context* c, uint32_t mode) { if (!c || mode > 2u) return PT_ERROR_INVALID_ARGUMENT; c->fusedTraversal = mode; return PT_OK; }

} // extern "C"
