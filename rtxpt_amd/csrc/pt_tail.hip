// mi355pt — the tail kernel: the late bounces of a batch without leaving the GPU.
//
// A wavefront pass is a chain of ~12 dependent launches (counters, traversal, task rounds, resolve, classify, shade, count read-back, the same again for the
// shadow rays). That is the right shape while a pass holds millions of paths; the last passes of a frame hold thousands (one rank of an 8-way sharded 4K frame:
// 34 k, 7 k, 1 k, 170 paths per batch in passes 6-9) and then the chain itself — launch boundaries, the host round trip for the counts, the fixed task rounds —
// is what the frame waits for: ~0.45 ms per pass whatever it carries (profiles/r03r_rank8.txt). Here a WAVE takes 32 paths and runs them to their end on its own:
// closest-hit traversal (two lanes per ray, pt_traverse8p.h) -> shading on the path's own lane (PathKernelContext::HandleHit / HandleMiss, the code k_shade runs)
// -> any-hit traversal of the NEE visibility rays -> the deferred NEE contribution (ResolveShadow) -> next bounce. Rays and hits travel between the phases through
// the wave's own slice of LDS, the path state stays in registers; nothing is synchronised across waves and no queue is touched. This is the reference's own
// shape (one thread per pixel looping over bounces, PathTracerSample.hlsl:200-250) applied where it fits: few paths, latency-bound, divergence irrelevant.
//
// The image cannot change: paths do not interact, the closest hit is traversal-order free (min t, ties to the lower primitive id), and the order of a path's own
// radiance terms — emission of vertex k, NEE of vertex k, emission of vertex k + 1 (fp16 sums do not commute) — is the order of the loop below.
//
// Stragglers (pt_traverse8.h CAN_SPLIT): a ray that is still in flight T8_TAIL_DEFER iterations after its wave ran dry (a wave is dry as soon as one of its 32 rays has finished) is not split into sub-trees here; its path
// leaves the kernel — state stored as of the start of the bounce, index appended to the pass's output queue (a visibility ray: the request goes to the shadow
// queue, the path to the output queue) — and the host loop traces it with the wavefront kernels, task rounds included. So does every path that is still alive
// after `maxBounces` bounces (the host's iteration bound stays the only bound).
#include "pt_wavefront.h"
#include "pt_traverse8.h"
#include "pt_traverse8p.h"
#include "pt_wavefront_device.h"

namespace ptk {

#ifndef T8_TAIL_DEFER
#define T8_TAIL_DEFER 512u        // iterations after which a ray of the tail kernel is handed back to the wavefront path (average extend ray: ~25, 99.99 % below 256)
#endif
static const uint TAIL_DEFERRED = 0xFFFFFFFEu;      // wHit[].y of a deferred ray (no primitive: pt_build checks the triangle count against 89 M)

template <class PKC, bool NEEAT>
__global__ void __launch_bounds__(T8_BLOCK, 2) k_tail(PKC k, PathPool pool, const uint* __restrict__ queueIn, const uint* __restrict__ countInPtr, uint* __restrict__ queueOut, uint* countOutPtr,
                                                      ShadowQueue sq, WaveCounters* wc, uint maxBounces, uint deferIters) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_RAYBUF_WORDS];
    __shared__ float2 mineUV[T8_BLOCK];
    __shared__ float4 wRayAll[T8_BLOCK / 64u][T8_CHUNK][2];      // per wave: the bounce's rays, compacted (origin | tmax, direction)
    __shared__ uint4 wHitAll[T8_BLOCK / 64u][T8_CHUNK];          // per wave: closest hit of ray r (t, primitive, u, v) / visibility of shadow ray r (.x = 1 visible, .y = TAIL_DEFERRED)
    const uint count = *countInPtr;
    const uint lane = threadIdx.x & 63u, w4 = threadIdx.x >> 6;
    const uint waveId = blockIdx.x * (T8_BLOCK / 64u) + w4, numWaves = gridDim.x * (T8_BLOCK / 64u);
    const uint vbase = waveId * T8_CHUNK;                          // the traversal deals chunk `waveId` of a virtual index space to this wave: [vbase, vbase + n) are its rays
    float4 (*wRay)[2] = wRayAll[w4]; uint4* wHit = wHitAll[w4];
    const unsigned long long below = (1ull << lane) - 1ull;
    const DeviceScene& sc = k.sc;
    Traverse8Counters ctr; t8_counters_init(ctr);
    unsigned long long nExtend = 0ull, nShadow = 0ull, nHits = 0ull, nBack0 = 0ull, nBack1 = 0ull, nBack2 = 0ull;

    auto fetchRay = [&](uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> uint {
        const uint r = i - vbase; const float4 a = wRay[r][0], b = wRay[r][1];
        o = make_float3(a.x, a.y, a.z); d = make_float3(b.x, b.y, b.z); tmin = 0.0f; tmax = a.w; startRef = 0u; bestT0 = a.w; bestPrim0 = 0xFFFFFFFFu;
        return r;
    };
    auto defer = [&](uint r, float, uint) { wHit[r] = make_uint4(0u, TAIL_DEFERRED, 0u, 0u); };

    for (uint chunk = waveId; chunk * T8_CHUNK < count; chunk += numWaves) {
        const uint first = chunk * T8_CHUNK;
        bool alive = lane < T8_CHUNK && first + lane < count;
        uint p = 0u; PathState path;
        if (alive) { p = queueIn[first + lane]; path = load_path(pool, p); }
        else __builtin_memset(&path, 0, sizeof(path));
        for (uint bounce = 0u; ; bounce++) {
            const unsigned long long mAlive = __builtin_amdgcn_ballot_w64(alive);
            if (mAlive == 0ull) break;
            if (bounce >= maxBounces) {                            // the host's iteration bound: what is still alive goes back to its loop
                const uint slot = wave_append(alive, countOutPtr);
                if (alive) { store_path(pool, p, path); queueOut[slot] = p; nBack2++; }
                break;
            }
            // ---- closest hit of every live path (Bridge::traceScatterRay)
            const uint n = (uint)__popcll(mAlive), r = (uint)__popcll(mAlive & below);
            if (alive) { wRay[r][0] = make_float4(path.origin.x, path.origin.y, path.origin.z, kMaxRayTravel); wRay[r][1] = make_float4(path.dir.x, path.dir.y, path.dir.z, 0.f); }
            {
                auto commit = [&](uint rr, const HitInfo& h) { wHit[rr] = make_uint4(asuint(h.t), h.prim, asuint(h.u), asuint(h.v)); };
                traverse8_pairs<false, false, true, false, true, true>(sc, vbase + n, T8_CHUNK, stack, rayBuf, mineUV, fetchRay, commit, defer, TravTaskOut{nullptr, nullptr, deferIters}, ctr, &wc->overflow, blockIdx.x, gridDim.x);
            }
            ShadowRequest req; req.valid = false;
            bool deferred = false;
            if (alive) {
                const uint4 hr = wHit[r];
                if (hr.y == TAIL_DEFERRED) deferred = true;
                else {
                    HitInfo h; h.t = asfloat(hr.x); h.prim = hr.y; h.u = asfloat(hr.z); h.v = asfloat(hr.w);
                    nExtend++;
                    if (h.prim == 0xFFFFFFFFu) k.template HandleMiss<NEEAT>(path, path.dir, kMaxRayTravel);
                    else { nHits++; k.template HandleHit<false, NEEAT>(path, h, req, nullptr); }
                }
            }
            {   // stragglers of the extend phase leave with the state they came with
                const uint slot = wave_append(deferred, countOutPtr);
                if (deferred) { store_path(pool, p, path); queueOut[slot] = p; alive = false; nBack0++; }
            }
            // ---- visibility of the NEE samples (Bridge::traceVisibilityRay), then the deferred half of HandleNEE
            const bool wantShadow = alive && req.valid;
            const unsigned long long mShadow = __builtin_amdgcn_ballot_w64(wantShadow);
            if (mShadow != 0ull) {
                const uint ns = (uint)__popcll(mShadow), rs = (uint)__popcll(mShadow & below);
                if (wantShadow) { wRay[rs][0] = make_float4(req.origin.x, req.origin.y, req.origin.z, req.tmax); wRay[rs][1] = make_float4(req.dir.x, req.dir.y, req.dir.z, 0.f); }
                {
                    auto commit = [&](uint rr, const HitInfo& h) { wHit[rr] = make_uint4(h.prim == 0xFFFFFFFFu ? 1u : 0u, 0u, 0u, 0u); };
                    traverse8_pairs<true, false, false, false, true, true>(sc, vbase + ns, T8_CHUNK, stack, rayBuf, nullptr, fetchRay, commit, defer, TravTaskOut{nullptr, nullptr, deferIters}, ctr, &wc->overflow, blockIdx.x, gridDim.x);
                }
                bool shadowDeferred = false;
                if (wantShadow) {
                    const uint4 v = wHit[rs];
                    if (v.y == TAIL_DEFERRED) shadowDeferred = true;
                    else {
                        nShadow++;
                        if (v.x) {                                // visible: the contribution lands (pt_wavefront.hip shadow_visible)
                            PKC::ResolveShadow(path.pack45, req.radiance);
                            if (NEEAT && sq.q3 && req.fbLight != RTXPT_INVALID_LIGHT_INDEX) {
                                const uint fslot = (path.sampleIndex - sq.fbSampleFirst) * sq.fbPlane + (path.id & 0xFFFFu) * sq.fbWidth + (path.id >> 16);
                                float total = sq.fbTotalWeight[fslot]; uint cand = sq.fbCandidates[fslot];
                                LightFeedbackReservoir_Add(total, cand, req.fbRandom, req.fbLight & ~LFR_SCREEN_SPACE_COHERENT_FLAG, req.fbWeight, (req.fbLight & LFR_SCREEN_SPACE_COHERENT_FLAG) != 0u);
                                sq.fbTotalWeight[fslot] = total; sq.fbCandidates[fslot] = cand;
                                if (req.rrFix & 1u) {
                                    const uint bit = (uint)PF_terminateAtNextBounce << kVertexIndexBitCount;
                                    if (req.rrFix & 2u) path.flagsAndVertexIndex |= bit; else { path.flagsAndVertexIndex &= ~bit; path.pack1 = (path.pack1 & 0xFFFF0000u) | (req.rrFix >> 16); }
                                }
                            }
                        }
                    }
                }
                // a straggler among the visibility rays: request to the shadow queue, path (if it goes on) to the output queue; the host's shadow launch resolves it before the next pass
                const unsigned long long mSd = __builtin_amdgcn_ballot_w64(shadowDeferred);
                if (mSd != 0ull) {
                    const uint sslot = wave_append(shadowDeferred, &wc->shadowCount);
                    const bool goesOn = shadowDeferred && path.isActive();
                    const uint qslot = wave_append(goesOn, countOutPtr);
                    if (shadowDeferred) {
                        store_path(pool, p, path);
                        sq.q0[sslot] = make_float4(req.origin.x, req.origin.y, req.origin.z, req.tmax);
                        sq.q1[sslot] = make_float4(req.dir.x, req.dir.y, req.dir.z, asfloat(p));
                        sq.q2[sslot] = make_float4(req.radiance.x, req.radiance.y, req.radiance.z, 0.f);
                        if (NEEAT && sq.q3) sq.q3[sslot] = make_float4(req.fbWeight, req.fbRandom, asfloat(req.fbLight), asfloat(req.rrFix));
                        if (goesOn) queueOut[qslot] = p;
                        alive = false; nBack1++;
                    }
                }
            }
            if (alive && !path.isActive()) { store_path(pool, p, path); alive = false; }      // the path has ended: its radiance waits in the pool for k_accumulate
        }
    }
    wave_add64(nExtend, &wc->tailExtendRays); wave_add64(nShadow, &wc->tailShadowRays); wave_add64(nHits, &wc->hits);
    wave_add64(nBack0, &wc->tailHandedBack[0]); wave_add64(nBack1, &wc->tailHandedBack[1]); wave_add64(nBack2, &wc->tailHandedBack[2]);
}

// grid: one wave per 32 paths up to what the GPU holds at this kernel's occupancy (2 blocks per CU); beyond that the waves stride over the queue
void launch_tail(const PathKernelContext& k, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr, ShadowQueue sq, WaveCounters* wc, uint maxBounces, uint deferIters, uint maxBlocks, hipStream_t st) {
    const uint wavesPerBlock = T8_BLOCK / 64u;
    uint g = (countIn + T8_CHUNK * wavesPerBlock - 1u) / (T8_CHUNK * wavesPerBlock); if (g < 1u) g = 1u;
    const uint cap = (maxBlocks && maxBlocks < T8_MAX_BLOCKS) ? maxBlocks : T8_MAX_BLOCKS; if (g > cap) g = cap;
    if (!deferIters) deferIters = T8_TAIL_DEFER;
    const bool neeat = k.sc.lights.LocalSamplingBuffer != nullptr || k.sc.lights.TemporalFeedbackRequired != 0u;
#define PT_LAUNCH_TAIL(PKC, CTX) do { if (neeat) hipLaunchKernelGGL((k_tail<PKC, true>), dim3(g), dim3(T8_BLOCK), 0, st, CTX, pool, queueIn, countInPtr, queueOut, countOutPtr, sq, wc, maxBounces, deferIters); \
                                      else hipLaunchKernelGGL((k_tail<PKC, false>), dim3(g), dim3(T8_BLOCK), 0, st, CTX, pool, queueIn, countInPtr, queueOut, countOutPtr, sq, wc, maxBounces, deferIters); } while (0)
    if (k.S.useFp16Types) { PathKernelContextT<true> k16; __builtin_memcpy(&k16, &k, sizeof(k16)); PT_LAUNCH_TAIL(PathKernelContextT<true>, k16); }
    else PT_LAUNCH_TAIL(PathKernelContext, k);
#undef PT_LAUNCH_TAIL
}

} // namespace ptk
