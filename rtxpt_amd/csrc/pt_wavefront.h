// mi355pt — wavefront kernels (declarations + HBM stream layout of the path pool and queues).
//
// Path pool (N = owned pixels x samples in flight; ALL samples of a pt_render call are resident at once — HBM is 288 GB):
//   s0[N] uint4  origin.xyz | pixel id (x<<16|y)          s1[N] uint4  dir.xyz | sceneLength
//   s2[N] uint4  thp fp16x4 (2 words) | L fp16x4 (2 words) s3[N] uint4  interiorList[2] | packedCounters | rayCone(fp16x2)
//   s4[N] uint4  {fireflyK,bsdfPdf} | {MISinfo,RRcorr} | flags+vertexIndex | sampleIndex
//   = 80 B per path, the reference's PathPayload size (Rtxpt/Shaders/PathTracer/PathTracerShared.h:21).
//   hit[N] uint4 t | global primitive | bary u | bary v   (written by k_extend, read by k_shade)
// Queues: extend queue = compacted u32 path indices (ping-pong); shadow queue = 3 x float4 per entry
//   q0 origin.xyz|tmax, q1 dir.xyz|path index, q2 radiance.rgb (already multiplied by throughput, MIS, BSDF).
// Queue appends use one atomic per wave64: ballot -> popcount prefix -> lane-0 atomicAdd -> broadcast.
#pragma once
#include "pt_path.h"
#include "pt_neeat.h"
#include "pt_tonemap.h"
#include <hip/hip_runtime.h>

namespace ptk {

struct PathPool { uint4* s0; uint4* s1; uint4* s2; uint4* s3; uint4* s4; uint4* hit;
    // Compacted pool (round 6, pt_render): home == nullptr — every array is indexed by the path's home slot (owned pixel x sample), the extend queue holds home slots. home != nullptr —
    // s0, s1, s3, s4 and hit are indexed by the path's POSITION in the extend queue (k_shade writes a surviving path's state at the position it appends it to, into the other of two
    // array sets), home[position] is its home slot, and s2 — throughput and radiance, what the visibility resolve and k_accumulate address — stays at the home slot.
    const uint* home; };
struct ShadowQueue { float4* q0; float4* q1; float4* q2; uint group;           // group: 0 = one entry per path vertex (NEEFullSamples 1), else entries come in groups of `group` (pt_path.h ShadowSink)
    // NEE-AT feedback (null: off): q3 = {weight, random, light | SSC flag, roulette fix-up} per entry (pt_path.h ShadowRequest); the sub-frame's feedback reservoirs, one slot per pixel
    float4* q3; float* fbTotalWeight; uint* fbCandidates; uint fbWidth, fbPlane, fbSampleFirst; };      // slot = (sample - fbSampleFirst) * fbPlane + y * fbWidth + x
struct WaveCounters {           // device-resident counters / stats (one 256 B block)
    uint extendCount[2]; uint shadowCount; uint overflow;
    unsigned long long hits, nodeVisitsExt, triTestsExt, nodeVisitsSh, triTestsSh, leafVisitsExt, itersExt, leafVisitsSh, itersSh, phaseCycExt[4], leafBlocksExt, eventsExt[8], itersMaxExt, rayIterHistExt[16]; uint longRayCount, _padLong; float longRays[32][8];
    unsigned long long shadowValid;     // grouped shadow queue: entries that carry a light sample (= shadow rays in the reference's sense)
    unsigned long long tailExtendRays, tailShadowRays;      // rays the tail kernel traced itself (pt_tail.hip); what it hands back is counted by the launches that trace it
    unsigned long long tailHandedBack[3];                   // paths the tail kernel handed back: extend stragglers, visibility stragglers, still alive at the bounce bound
};

// straggler splitting (pt_traverse8.h): per pipelined batch, two task queues (ping-pong), the per-ray merge keys and the list of rays to resolve.
// bestKey: closest hit = float bits of t << 32 | primitive (atomicMin = min t, ties to the
// lower primitive id); occlusion = 0 visible so far / 1 occluded.
// counts: TRAV_COUNTERS words per traversal launch, one counter per producer so that nothing has to be reset between the launches of a pass —
//   [0] sub-trees split off by k_extend / k_shadow (task queue 0), [1..3] by task rounds 0..2 (queues 1, 0, 1), [TRAV_RESOLVE] rays to resolve.
// pt_render keeps one PASS_COUNTERS block per batch — {extend launch, shadow launch, k_classify's three class counts} — and zeroes it once per pass.
static const uint TRAV_COUNTERS = 5, TRAV_RESOLVE = 4, PASS_COUNTERS = 16, PASS_SHADOW_OFFSET = 5, PASS_CLASS_OFFSET = 10;
struct TravAux { TravTask* taskQ[2]; uint* counts; uint taskCap; unsigned long long* bestKey; uint* resolveList; const uint* primToSlot; uint maxBlocks; };      // maxBlocks: grid bound of the traversal launches (0: T8_MAX_BLOCKS)

// paths [first, first + n) of the pool region -> queue[0 .. n); countPtr (may be null): the counter of the queue `queue` is the end of, incremented by n (k_generate)
void launch_generate(const PathKernelContext& k, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleFirst, uint spp, uint first, uint n, uint* queue, uint* countPtr, hipStream_t st);
void launch_extend(const DeviceScene& sc, PathPool pool, const uint* queue, const uint* countPtr, uint count, WaveCounters* wc, bool counters, TravAux aux, hipStream_t st, bool ranged = false);      // ranged: the rays' intervals wait in pool.hit[p].xy (k_extend)
// classScratch (2 x countIn words: memory that is free between the extend and the shadow launches of a bounce) + classCount (3 words, zero on entry): k_classify's output; null = shade in queue order
// launch_extend / launch_shade / launch_shadow expect the pass's counter block zeroed by the caller (launch_pass_reset)
void launch_shade(const PathKernelContext& k, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr,
                  ShadowQueue sq, WaveCounters* wc, uint* classScratch, uint* classCount, hipStream_t st, PathPool outPool = PathPool{});      // outPool: a compacted pool's other array set (pool.home != nullptr)
// a compacted pool's live paths back to their home slots: out.s0 / s1 / s3 / s4 [home[i]] = in...[i] for the *countPtr positions (out: another array set than in's); the queue `in.home` is then an ordinary extend queue
void launch_uncompact(PathPool in, PathPool out, const uint* countPtr, uint count, hipStream_t st);
void launch_classify(PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* classScratch, uint* classCount, hipStream_t st);      // k_classify: {continuing hit, terminating hit, miss} made contiguous
void launch_shadow(const DeviceScene& sc, PathPool pool, ShadowQueue sq, const uint* countPtr, uint count, WaveCounters* wc, bool counters, TravAux aux, hipStream_t st);
// the closest-hit rays of the extend queue and the visibility rays of the previous vertex (shadow queue) in ONE traversal launch, their task rounds and resolve passes in shared launches
// too (k_trace_pair; sq.group == 0 only). auxE / auxS must not share task queues, counters, keys or resolve lists; *shCountPtr is zero afterwards
void launch_trace_pair(const DeviceScene& sc, PathPool pool, const uint* queue, const uint* extCountPtr, uint extCount, ShadowQueue sq, uint* shCountPtr, uint shCount, WaveCounters* wc, TravAux auxE, TravAux auxS, hipStream_t st);
// the late bounces of a batch in one launch: every wave runs 32 paths of queueIn to their end (at most maxBounces bounces each); stragglers and paths beyond the bound come back through
// queueOut / countOutPtr (and, for visibility rays, the shadow queue / wc->shadowCount). NEEFullSamples 1 only (sq.group == 0); the pass's counters zeroed by the caller as for launch_shade
void launch_tail(const PathKernelContext& k, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr, ShadowQueue sq, WaveCounters* wc, uint maxBounces, uint deferIters /* 0: T8_TAIL_DEFER */, uint maxBlocks, hipStream_t st);
void launch_pass_reset(uint* passCounters, uint* nextCount, uint* shadowCount, hipStream_t st);      // one launch: the batch's PASS_COUNTERS words and the two queue counters the pass refills
void launch_accumulate(PathPool pool, const uint* ownedPixels, uint numOwned, uint spp, float4* accum, uint accumCountBase, uint width, hipStream_t st);
void launch_trace_probe(const DeviceScene& sc, const float4* rays, uint n, float4* outClosest, uint* outVisible, uint* overflow, hipStream_t st);
void launch_pack(const float4* accum, const uint* ownedPixels, uint numOwned, uint width, float4* dst, hipStream_t st);
void launch_unpack(float4* accum, const uint* pixels, uint num, uint width, const float4* src, hipStream_t st);
void launch_pack_feedback(const float* fbW, const uint* fbC, const float* depth, const uint* pixels, uint num, uint width, uint* dst, hipStream_t st);      // 3 words per pixel: weight bits, candidate, depth bits
void launch_unpack_feedback(float* fbW, uint* fbC, float* depth, const uint* pixels, uint num, uint width, const uint* src, hipStream_t st);
// EnvMapBaker: lat-long source (sc.envTex) + directional lights -> RGBA16F cube with mips (cube.mipOffset / dim / mipLevels filled by the caller)
void launch_env_cube_bake(const DeviceScene& sc, const EnvDirectionalLight* lights, uint nLights, uint2* texels, const EnvCube& cube, hipStream_t st);
void launch_env_cube_compress(uint2* texels, const EnvCube& cube, uint quality, hipStream_t st);      // every level through BC6UCompress.hlsl's encoder (quality 1: EncodeP1; 2: + the two-region modes) and the BC6H decode, in place
void launch_env_importance(const DeviceScene& sc, uint dim, uint sx, uint sy, float4* out, hipStream_t st);
void launch_bake_emissive(const DeviceScene& sc, const uint* subInstList, const uint* subInstTriOffset, uint numEmissiveSubInst, uint totalTris, uint lightBase,
                          PolymorphicLightInfo* lights, PolymorphicLightInfoEx* lightsEx, hipStream_t st);
// light weights (power^0.8), their in-order sum, proxy counts; then (after an exclusive scan of the counts by the caller) the proxy index fill
void launch_light_weights(const PolymorphicLightInfo* lights, const PolymorphicLightInfoEx* lightsEx, uint n, float* w, const LightFrustumBoost& boost, hipStream_t st);      // ComputeWeight (+ the frustum boost) per light
// weight sum (in light order) + ComputeProxyCounts; usage != null: NEE-AT's feedback term (n + 1 usage counts, the last one = pixels without valid feedback)
void launch_light_proxy_counts(const float* w, uint n, float* sum, uint budget, bool uniform, uint maxPerLight, uint* counts, const uint* usage, uint totalMaxFeedbackCount, float globalFeedbackUseWeight, hipStream_t st);
void launch_neeat_boost_weights(const float* base, const float* hist, uint nHist, uint n, float mul, float* cur, hipStream_t st);
void launch_neeat_begin(const NeeAtFrame& F, float* snapW, uint* snapC, bool preFilter, uint totalThreads, hipStream_t st);
void launch_neeat_end(const NeeAtFrame& F, hipStream_t st);
void launch_light_proxy_fill(const uint* counts, const uint* offsets, uint n, uint* proxies, uint capacity, hipStream_t st);
void launch_tonemap(const float4* accum, uint num, const ToneMapParams& p, uint* outRgba8, hipStream_t st);
// scratch: 2 * pow2floor(W) * pow2floor(H) floats; *result points at the 1x1 mip inside scratch once the stream has drained
void launch_average_log_luminance(const float4* accum, uint W, uint H, float* scratch, float** result, hipStream_t st);
void launch_probe(const PathKernelContext& k, int kind, const void* dIn, void* dOut, uint n, hipStream_t st);

} // namespace ptk
