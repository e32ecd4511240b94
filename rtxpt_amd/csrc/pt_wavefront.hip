// mi355pt — wavefront kernels: generate / extend / shade / shadow / accumulate (+ bake and probe kernels).
// The reference runs all of this as ONE DXR dispatch, one thread per pixel looping over bounces
// (Rtxpt/Shaders/PathTracerSample.hlsl:200-250) and relies on SER to re-sort threads (:136-148). Here each stage is its own
// kernel over a compacted queue, so every wave starts full regardless of how many paths died in the previous bounce.
#include "pt_wavefront.h"
#include "pt_traverse8.h"
#include "pt_traverse8p.h"
#include "pt_wavefront_device.h"
#include <cstdlib>
#define T8_TRAVERSE traverse8_pairs

namespace ptk {

#ifndef PT_SAMPLE_MINOR
#define PT_SAMPLE_MINOR 1       // 1: the samples of a pixel are neighbours in the path pool (pixel-major), 0: all pixels of sample 0, then of sample 1, ...
#endif
#ifndef T8_CHUNKS_PER_WAVE_MIN
#define T8_CHUNKS_PER_WAVE_MIN 1     // small launches: fewer waves, each working through this many 64-ray chunks (idle quads refill from the next chunk)
#endif
#ifndef T8_SHORT_TAIL_BELOW
#define T8_SHORT_TAIL_BELOW 32768u    // traversal launches of at most this many rays run two task rounds instead of four (launch_extend)
#endif
#ifndef T8_TASK_BLOCKS_N
#define T8_TASK_BLOCKS_N 512        // blocks of a task-round launch: task rounds hold thousands of sub-trees, not millions
#endif

// Paths [first, first + n) of the batch's pool region (slot i = sample i % spp of owned pixel i / spp) are generated and their indices written to queue[0 ..
// n).
__global__ void __launch_bounds__(256) k_generate(PathKernelContext k, PathPool pool, const uint* __restrict__ ownedPixels, uint numOwned, uint sampleFirst, uint spp, uint first, uint n, uint* __restrict__ queue, uint* countPtr) {
    const uint t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint i = first + t;
#if PT_SAMPLE_MINOR
    // the samples of a pixel are neighbours in the pool: a 64-path chunk is 16 pixels x 4 samples
    uint kpx = i / spp, s = i - kpx * spp, px = ownedPixels[kpx];
#else
    uint s = i / numOwned, px = ownedPixels[i - s * numOwned];
#endif
    PathState p = k.generate(px >> 16, px & 0xFFFFu, sampleFirst + s);
    store_path(pool, i, p);
    queue[t] = i;
    if (countPtr && t == 0u) atomicAdd(countPtr, n);
}

// The closest-hit launch of a bounce (Bridge::traceScatterRay for every path of the extend queue) as a device function, so that two kernels can run it:
// k_extend, and k_trace_pair next to the visibility rays of the previous vertex. vBlock / vGrid: this block's place among the blocks that work on the extend
// queue (traverse8_pairs).
template <bool COUNT, bool RANGED>
__device__ __forceinline__ void t8_extend_body(const DeviceScene& sc, const PathPool& pool, const uint* __restrict__ queue, const uint count, WaveCounters* wc, const TravAux& aux, const uint rpc,
                                               uint2* stack, uint* rayBuf, float2* mineUV, const uint vBlock, const uint vGrid) {
    // RANGED: every ray brings its own interval in the first two words of its (not yet written) hit record — the stable-plane fill pass's first launch,
    // FirstHitFromVBuffer (pt_stableplanes.h firstHitInterval). A ray of such a launch that is cut into sub-trees continues over [0, best hit so far]: the
    // lower bound is a hint, not part of the query.
    Traverse8Counters ctr; t8_counters_init(ctr);
    auto fetch = [&](uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> uint {
        uint p = pool.home ? i : queue[i];      // (a compacted pool's ray i is the path at position i)
        uint4 a = pool.s0[p], b = pool.s1[p];
        o = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)); d = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z));
        tmin = 0.0f; tmax = kMaxRayTravel; startRef = 0u; bestT0 = kMaxRayTravel; bestPrim0 = 0xFFFFFFFFu;
        if (RANGED) { const uint4 r = pool.hit[p]; tmin = asfloat(r.x); tmax = asfloat(r.y); bestT0 = tmax; }
        return p;
    };
    auto commit = [&](uint p, const HitInfo& h) { pool.hit[p] = make_uint4(asuint(h.t), h.prim, asuint(h.u), asuint(h.v)); };
    // a split ray: its best hit so far seeds the merge key, the resolve pass will write pool.hit (k_resolve_extend)
    auto publish = [&](uint p, float bestT, uint bestPrim) { aux.bestKey[p] = t8_hit_key(bestT, bestPrim); aux.resolveList[atomicAdd(&aux.counts[TRAV_RESOLVE], 1u)] = p; };
    if (COUNT) { ctr.rayIterHist = wc->rayIterHistExt; ctr.longRayCount = &wc->longRayCount; ctr.longRays = &wc->longRays[0][0]; }
    T8_TRAVERSE<false, COUNT, !RANGED, false, true>(sc, count, rpc, stack, rayBuf, mineUV, fetch, commit, publish, TravTaskOut{aux.taskQ[0], &aux.counts[0], aux.taskCap}, ctr, &wc->overflow, vBlock, vGrid);
    if (COUNT) { wave_add64(ctr.nodeVisits, &wc->nodeVisitsExt); wave_add64(ctr.triTests, &wc->triTestsExt); wave_add64(ctr.leafVisits, &wc->leafVisitsExt); wave_add64(ctr.iters, &wc->itersExt); wave_add64(ctr.leafBlocks, &wc->leafBlocksExt); if ((threadIdx.x & 63u) == 0u) atomicMax(&wc->itersMaxExt, (unsigned long long)ctr.iters);
                 if ((threadIdx.x & 63u) == 0u) for (int q = 0; q < 4; q++) atomicAdd(&wc->phaseCycExt[q], ctr.cyc[q]);
                 for (int q = 0; q < 8; q++) wave_add64(ctr.ev[q], &wc->eventsExt[q]); }
}
template <bool COUNT, bool RANGED = false>
__global__ void __launch_bounds__(T8_BLOCK, T8_EXTEND_MIN_BLOCKS) k_extend(DeviceScene sc, PathPool pool, const uint* __restrict__ queue, const uint* __restrict__ countPtr, WaveCounters* wc, TravAux aux, uint rpc) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_RAYBUF_WORDS];
    __shared__ float2 mineUV[T8_BLOCK];
    t8_extend_body<COUNT, RANGED>(sc, pool, queue, *countPtr, wc, aux, rpc, stack, rayBuf, mineUV, blockIdx.x, gridDim.x);
}

// Task rounds hold hundreds to thousands of sub-trees, far fewer than the launch has quads. A 64-item chunk would put them on count/64 waves (one rank of
// an 8-way sharded frame: ~90 us per round on a handful of waves, 40 % on top of k_extend itself); with few tasks a chunk carries only 16 real ones —
// one per quad — and 48 empty slots that cost an iteration each, so the sub-trees spread over 4x as many waves.
#ifndef T8_TASK_SPREAD
#define T8_TASK_SPREAD 1
#endif
__device__ __forceinline__ uint t8_tasks_per_chunk(uint count) { return (!T8_TASK_SPREAD || count > T8_GROUPS_PER_WAVE * 4u * T8_TASK_BLOCKS_N || T8_GROUPS_PER_WAVE > T8_CHUNK) ? T8_CHUNK : T8_GROUPS_PER_WAVE; }
// sub-trees of split extend rays, round STAGE (0..3) of a traversal launch: reads queue STAGE & 1 (count: counts[STAGE]); unless it is the final round,
// stragglers among the sub-trees are split again into the other queue (count: counts[STAGE + 1]). One counter per round: the whole block is zeroed once per
// pass (pt_wavefront.h TravAux). Task i of the launch is queue entry (i % 64) * ceil(count / 64) + i / 64: the sub-trees of one ray sit next to each other in
// the queue and would otherwise land in one 64-item chunk, i.e. on one wave.
template <int STAGE, bool FINAL>
__device__ __forceinline__ void t8_extend_tasks_body(const DeviceScene& sc, const PathPool& pool, WaveCounters* wc, const TravAux& aux, uint2* stack, uint* rayBuf, const uint vBlock, const uint vGrid) {
    constexpr int IN = STAGE & 1;
    uint count = aux.counts[STAGE]; if (count > aux.taskCap) count = aux.taskCap;
    if (count == 0u) return;
    const TravTask* tasks = aux.taskQ[IN];
    const uint real = t8_tasks_per_chunk(count), per = (count + real - 1u) / real;
    Traverse8Counters ctr; t8_counters_init(ctr);
    auto fetch = [&](uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> uint {
        const uint lane = i % T8_CHUNK, j = lane * per + i / T8_CHUNK;
        const bool pad = lane >= real || j >= count;                                  // padding of the transposed index space: an empty task
        TravTask t = tasks[pad ? 0u : j];
        if (pad) t.tbits = 0x7F800000u;
        uint4 a = pool.s0[t.tag], b = pool.s1[t.tag];
        o = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)); d = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z));
        unsigned long long key = aux.bestKey[t.tag];
        tmin = 0.0f; tmax = kMaxRayTravel; bestT0 = __uint_as_float((uint)(key >> 32)); bestPrim0 = (uint)key;
        startRef = (__uint_as_float(t.tbits) <= bestT0) ? t.ref : BVH_EMPTY;            // a sub-tree behind the current best hit is dropped here
        return t.tag;
    };
    auto commit = [&](uint p, const HitInfo& h) { atomicMin(&aux.bestKey[p], t8_hit_key(h.t, h.prim)); };
    auto publish = [&](uint p, float bestT, uint bestPrim) { if (bestPrim != 0xFFFFFFFFu) atomicMin(&aux.bestKey[p], t8_hit_key(bestT, bestPrim)); };
    T8_TRAVERSE<false, false, true, true, !FINAL>(sc, per * T8_CHUNK, T8_CHUNK, stack, rayBuf, nullptr, fetch, commit, publish, TravTaskOut{aux.taskQ[IN ^ 1], &aux.counts[FINAL ? STAGE : STAGE + 1], FINAL ? 0u : aux.taskCap}, ctr, &wc->overflow, vBlock, vGrid);
}
template <int STAGE, bool FINAL = (STAGE == 3)>
__global__ void __launch_bounds__(T8_BLOCK, T8_EXTEND_MIN_BLOCKS) k_extend_tasks(DeviceScene sc, PathPool pool, WaveCounters* wc, TravAux aux) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_TASKBUF_WORDS];
    t8_extend_tasks_body<STAGE, FINAL>(sc, pool, wc, aux, stack, rayBuf, blockIdx.x, gridDim.x);
}

// split extend rays: the merged key -> hit record; the barycentrics come from re-intersecting the winning triangle (same arithmetic, same operands)
__device__ __forceinline__ void t8_resolve_extend_body(const DeviceScene& sc, const PathPool& pool, const TravAux& aux, const uint vBlock, const uint vGrid) {
    const uint n = aux.counts[TRAV_RESOLVE];
    for (uint i = vBlock * 256u + threadIdx.x; i < n; i += vGrid * 256u) {
        uint p = aux.resolveList[i];
        unsigned long long key = aux.bestKey[p];
        uint prim = (uint)key;
        uint4 out = make_uint4((uint)(key >> 32), prim, 0u, 0u);
        if (prim != 0xFFFFFFFFu) {
            uint4 a = pool.s0[p], b = pool.s1[p];
            float3 o = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)), d = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z));
            float t, u = 0.f, v = 0.f;
            // the winner was accepted by the traversal; only (u, v) are needed
            const TriRecord tr = sc.tris[aux.primToSlot[prim]]; (void)intersect_tri_wt(tri_v0(tr), tri_v1(tr), tri_v2(tr), o, d, 0.0f, kMaxRayTravel, t, u, v);
            out.z = asuint(u); out.w = asuint(v);
        }
        pool.hit[p] = out;
    }
}
__global__ void __launch_bounds__(256) k_resolve_extend(DeviceScene sc, PathPool pool, TravAux aux) { t8_resolve_extend_body(sc, pool, aux, blockIdx.x, gridDim.x); }

#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 256       // threads per k_shade block (the queue appends meet per block: one atomic per block and counter)
#endif
#ifndef PT_SHADE_MIN_BLOCKS
#define PT_SHADE_MIN_BLOCKS 1
#endif
#ifndef PT_SHADE_BLOCK_APPEND
#define PT_SHADE_BLOCK_APPEND 1      // 1: k_shade appends to the extend / shadow queues with one atomic per block and counter, 0: one per wave
#endif
#ifndef PT_SHADE_PROBE
#define PT_SHADE_PROBE 0
#endif
#ifndef PT_SHADE_CLASSES
#define PT_SHADE_CLASSES 1      // 1: k_classify sorts the bounce's paths into {hit that goes on, hit that terminates after its emission, miss} before k_shade
#endif

// Path classes for k_shade. A shading wave lives ~90 us, nearly all of it waiting on dependent loads, and as long as ONE lane runs the whole of HandleHit
// (surface, scatter, light sampling) the wave stays for all of it — while 13 % of a bounce's paths are misses and ~20 % are hits that terminate right after
// their emission term (PF_terminateAtNextBounce). k_classify makes the classes contiguous (continuing hits from the front of one array, terminating hits from
// its back, misses in a second one; one atomic per class per 1024 paths), so all but two waves of a launch are of one class and the short classes leave early.
// Radiance, queues and counters do not depend on the order in which paths are shaded.
// paths per thread: 4096 per block, one atomic per class and block (with 1024 per block the 32 000 blocks of a 4K bounce spent 0.3 ms queueing on three L2
// lines)
#define PT_CLASSIFY_ITEMS 4u
__global__ void __launch_bounds__(1024) k_classify(PathPool pool, const uint* __restrict__ queueIn, const uint* __restrict__ countInPtr, uint* __restrict__ classQ, uint* __restrict__ classCount) {
    __shared__ uint waveCnt[PT_CLASSIFY_ITEMS][16][3]; __shared__ uint blockBase[3];
    const uint count = *countInPtr, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // cls: 0 continuing hit, 1 terminating hit, 2 miss, 3 out of range
    uint p[PT_CLASSIFY_ITEMS], cls[PT_CLASSIFY_ITEMS]; unsigned long long mine[PT_CLASSIFY_ITEMS];
#pragma unroll
    for (uint j = 0; j < PT_CLASSIFY_ITEMS; j++) {
        const uint i = (blockIdx.x * PT_CLASSIFY_ITEMS + j) * 1024u + threadIdx.x;
        p[j] = 0u; cls[j] = 3u;
        if (i < count) {
            p[j] = pool.home ? i : queueIn[i];
            const uint prim = reinterpret_cast<const uint*>(pool.hit)[4u * (size_t)p[j] + 1u], flags = reinterpret_cast<const uint*>(pool.s4)[4u * (size_t)p[j] + 2u];
            cls[j] = (prim == 0xFFFFFFFFu) ? 2u : (((flags >> kVertexIndexBitCount) & PF_terminateAtNextBounce) ? 1u : 0u);
        }
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cls[j] == 0u), m1 = __builtin_amdgcn_ballot_w64(cls[j] == 1u), m2 = __builtin_amdgcn_ballot_w64(cls[j] == 2u);
        if (lane == 0u) { waveCnt[j][wave][0] = (uint)__popcll(m0); waveCnt[j][wave][1] = (uint)__popcll(m1); waveCnt[j][wave][2] = (uint)__popcll(m2); }
        mine[j] = cls[j] == 0u ? m0 : (cls[j] == 1u ? m1 : m2);
    }
    __syncthreads();
    if (threadIdx.x < 3u) {      // exclusive prefix over (item, wave) in queue order, then the block's base
        uint tot = 0;
        for (uint j = 0; j < PT_CLASSIFY_ITEMS; j++) for (uint w = 0; w < 16u; w++) { const uint c = waveCnt[j][w][threadIdx.x]; waveCnt[j][w][threadIdx.x] = tot; tot += c; }
        blockBase[threadIdx.x] = tot ? atomicAdd(&classCount[threadIdx.x], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint j = 0; j < PT_CLASSIFY_ITEMS; j++) {
        if (cls[j] > 2u) continue;
        const uint rank = blockBase[cls[j]] + waveCnt[j][wave][cls[j]] + (uint)__popcll(mine[j] & ((1ull << lane) - 1ull));
        // continuing hits from the front of [0, count), terminating hits from its back (they cannot meet: together they are at most count), misses in [count, 2
        // count)
        classQ[cls[j] == 0u ? rank : (cls[j] == 1u ? count - 1u - rank : count + rank)] = p[j];
    }
}

// PKC: PathKernelContextT<false> (lp types in fp32) or PathKernelContextT<true> (the reference's default build, lp types in binary16)
// COMPACT: the pool is compacted (PathPool::home): the path at position p of this bounce's array set is written, if it survives, to the position it is appended at in `outPool`'s set;
// its throughput | radiance word group stays at its home slot, which is also what the extend queue keeps and the shadow queue names
template <bool MULTI, class PKC, bool NEEAT, bool COMPACT = false>
__global__ void __launch_bounds__(PT_SHADE_BLOCK, PT_SHADE_MIN_BLOCKS) k_shade(PKC k, PathPool pool, const uint* __restrict__ queueIn, const uint* __restrict__ countInPtr,
                                               uint* __restrict__ queueOut, uint* countOutPtr, ShadowQueue sq, WaveCounters* wc, const uint* __restrict__ classCount, PathPool outPool) {
    const uint count = *countInPtr;
    uint i = blockIdx.x * (uint)PT_SHADE_BLOCK + threadIdx.x;
    bool inRange = i < count;
    bool alive = false; bool isHit = false; uint p = 0, hp = 0;
    ShadowRequest req; req.valid = false;
    PathState cpath;      // COMPACT: the shaded path, kept until its position is known
    if (COMPACT) __builtin_memset(&cpath, 0, sizeof(cpath));
    if (inRange) {
        // queueIn = k_classify's arrays: thread i takes the i-th path of the order {continuing, terminating, miss}
        if (classCount) {
            const uint nGo = classCount[0], nEnd = classCount[1];
            p = queueIn[i < nGo ? i : (i < nGo + nEnd ? count - 1u - (i - nGo) : count + (i - nGo - nEnd))];
        } else p = COMPACT ? i : queueIn[i];
        hp = COMPACT ? pool.home[p] : p;
        uint4 hr = pool.hit[p];
        HitInfo h; h.t = asfloat(hr.x); h.prim = hr.y; h.u = asfloat(hr.z); h.v = asfloat(hr.w);
#if PT_SHADE_PROBE
        PathState path = load_path(pool, p);
#if PT_SHADE_PROBE == 1          // timing probes (developer builds only, the image is wrong by design): 1 = stream the path state through, nothing else
        if (h.prim != 0xFFFFFFFFu) { isHit = true; path.sceneLength += h.t; } path.terminate();
#elif PT_SHADE_PROBE == 2        // 2 = the surface gather (record, instance, material, textures) and nothing after it
        if (h.prim != 0xFFFFFFFFu) { isHit = true; SurfaceData sfd = k.loadSurface(h.prim, h.u, h.v, path.dir, path.rayCone); path.origin = sfd.shadingData.posW + sfd.shadingData.N * sfd.bsdf.data.roughness + sfd.bsdf.data.diffuse + sfd.shadingData.T; } path.terminate();
#endif
        store_path(pool, p, path);
        alive = path.isActive();
#else
        if (COMPACT) {
            const PathCompactIO io{pool, p, hp};
            if (h.prim == 0xFFFFFFFFu) { cpath = io.load_all(); k.template HandleMiss<NEEAT>(cpath, cpath.dir, kMaxRayTravel); }
            else { isHit = true; cpath = io.load_first(); k.template HandleHit<false, NEEAT>(cpath, h, req, nullptr, io); }
            alive = cpath.isActive();
            pool.s2[hp] = make_uint4(cpath.pack23[0], cpath.pack23[1], cpath.pack45[0], cpath.pack45[1]);      // throughput | radiance: at home, alive or not
        }
        else if (h.prim == 0xFFFFFFFFu) { PathState path = load_path(pool, p); k.template HandleMiss<NEEAT>(path, path.dir, kMaxRayTravel); store_path(pool, p, path); alive = path.isActive(); }
        else {
            isHit = true;
            const PathPoolIO io{pool, p};      // the path streams through HandleHit: late loads, early stores (pt_wavefront_device.h)
            PathState path = io.load_first();
            if (MULTI) { ShadowSink sink{sq.q0, sq.q1, sq.q2, &wc->shadowCount, &wc->shadowValid, p}; k.template HandleHit<true, NEEAT>(path, h, req, &sink, io); }
            else k.template HandleHit<false, NEEAT>(path, h, req, nullptr, io);
            alive = path.isActive();
#ifdef PT_SHADE_PHASE_PROBE
            for (int q = 1; q < 6; q++) wave_add64(io.tk[q] ? (unsigned long long)(uint)(io.tk[q] - io.tk[q - 1 - (io.tk[q - 1] ? 0 : 1)]) : 0ull, &wc->eventsExt[q - 1]);      // (a vertex that returns early leaves later stamps at 0)
#endif
        }
#endif
    }
#if PT_SHADE_BLOCK_APPEND
    // Queue appends, one atomic per BLOCK and counter: the four waves' counts meet in LDS, thread 0 reserves both ranges. A launch of 33 M paths has 518 k
    // waves; one returning atomic per wave on each of three words — all waves of the GPU on the same three addresses — is what the kernel waited for
    // (same-address atomics serialise in the L2: ~10^8 per second and address). The hit count needs no atomic at all when the paths were classified: it is the
    // size of two classes.
    __shared__ uint sCnt[PT_SHADE_BLOCK / 64][2]; __shared__ uint sBase[2];
    const uint wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned long long mAlive = __builtin_amdgcn_ballot_w64(alive), mReq = __builtin_amdgcn_ballot_w64(!MULTI && req.valid);
    if (lane == 0u) { sCnt[wave][0] = (uint)__popcll(mAlive); sCnt[wave][1] = (uint)__popcll(mReq); }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint tot = 0; for (uint w = 0; w < (uint)(PT_SHADE_BLOCK / 64); w++) { const uint c = sCnt[w][threadIdx.x]; sCnt[w][threadIdx.x] = tot; tot += c; }
        sBase[threadIdx.x] = tot ? atomicAdd(threadIdx.x == 0u ? countOutPtr : &wc->shadowCount, tot) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (alive) {
        const uint slot = sBase[0] + sCnt[wave][0] + (uint)__popcll(mAlive & below);
        queueOut[slot] = hp;
        if (COMPACT) {      // the survivor's state at its new position, in the other array set
            outPool.s0[slot] = make_uint4(asuint(cpath.origin.x), asuint(cpath.origin.y), asuint(cpath.origin.z), cpath.id);
            outPool.s1[slot] = make_uint4(asuint(cpath.dir.x), asuint(cpath.dir.y), asuint(cpath.dir.z), asuint(cpath.sceneLength));
            outPool.s3[slot] = make_uint4(cpath.interiorList.slots[0], cpath.interiorList.slots[1], cpath.packedCounters, cpath.rayCone.widthSpreadAngleFP16);
            outPool.s4[slot] = make_uint4(cpath.pack0, cpath.pack1, cpath.flagsAndVertexIndex, cpath.sampleIndex);
        }
    }
    if (!MULTI && req.valid) {
        const uint sslot = sBase[1] + sCnt[wave][1] + (uint)__popcll(mReq & below);
        sq.q0[sslot] = make_float4(req.origin.x, req.origin.y, req.origin.z, req.tmax);
        sq.q1[sslot] = make_float4(req.dir.x, req.dir.y, req.dir.z, asfloat(hp));
        sq.q2[sslot] = make_float4(req.radiance.x, req.radiance.y, req.radiance.z, 0.f);
        if (NEEAT && sq.q3) sq.q3[sslot] = make_float4(req.fbWeight, req.fbRandom, asfloat(req.fbLight), asfloat(req.rrFix));
    }
    if (classCount) { if (blockIdx.x == 0u && threadIdx.x == 0u) atomicAdd(&wc->hits, (unsigned long long)classCount[0] + classCount[1]); }
    else wave_add64(isHit ? 1ull : 0ull, &wc->hits);
#else
    uint slot = wave_append(alive, countOutPtr);
    if (alive) queueOut[slot] = p;
    uint sslot = MULTI ? 0u : wave_append(req.valid, &wc->shadowCount);
    if (!MULTI && req.valid) {
        sq.q0[sslot] = make_float4(req.origin.x, req.origin.y, req.origin.z, req.tmax);
        sq.q1[sslot] = make_float4(req.dir.x, req.dir.y, req.dir.z, asfloat(p));
        sq.q2[sslot] = make_float4(req.radiance.x, req.radiance.y, req.radiance.z, 0.f);
        if (NEEAT && sq.q3) sq.q3[sslot] = make_float4(req.fbWeight, req.fbRandom, asfloat(req.fbLight), asfloat(req.rrFix));
    }
    wave_add64(isHit ? 1ull : 0ull, &wc->hits);
#endif
}

template <bool GROUPED>
// visible == the deferred NEE contribution lands (BridgeDonut:1026)
__device__ __forceinline__ void shadow_visible(PathPool pool, ShadowQueue sq, uint i) {
    // grouped queue: only mark, k_resolve_nee folds the group in sample order
    if (GROUPED) { reinterpret_cast<float*>(sq.q2 + i)[3] = 1.0f; return; }
    float4 r = sq.q2[i];
    uint p = asuint(sq.q1[i].w);
    uint4 c = pool.s2[p];
    uint pack45[2] = {c.z, c.w};
    PathKernelContext::ResolveShadow(pack45, make_float3(r.x, r.y, r.z));
    c.z = pack45[0]; c.w = pack45[1];
    pool.s2[p] = c;
    // NEE-AT: the visible light feeds the pixel's reservoir (LightSampler.hlsli:184-200), and the path continues as k_shade worked out for this case
    if (sq.q3) {
        const float4 f = sq.q3[i];
        const uint light = asuint(f.z), fix = asuint(f.w);
        if (light != RTXPT_INVALID_LIGHT_INDEX) {
            uint4 e = pool.s4[p];
            const uint id = pool.s0[p].w, slot = (e.w - sq.fbSampleFirst) * sq.fbPlane + (id & 0xFFFFu) * sq.fbWidth + (id >> 16);
            float total = sq.fbTotalWeight[slot]; uint cand = sq.fbCandidates[slot];
            LightFeedbackReservoir_Add(total, cand, f.y, light & ~LFR_SCREEN_SPACE_COHERENT_FLAG, f.x, (light & LFR_SCREEN_SPACE_COHERENT_FLAG) != 0u);
            sq.fbTotalWeight[slot] = total; sq.fbCandidates[slot] = cand;
            if (fix & 1u) {
                const uint bit = (uint)PF_terminateAtNextBounce << kVertexIndexBitCount;
                if (fix & 2u) e.z |= bit; else { e.z &= ~bit; e.y = (e.y & 0xFFFF0000u) | (fix >> 16); }
                pool.s4[p] = e;
            }
        }
    }
}

// The visibility launch of a path vertex (Bridge::traceVisibilityRay for every entry of the shadow queue) as a device function: k_shadow, and k_trace_pair next
// to the closest-hit rays of the next vertex.
template <bool COUNT, bool GROUPED>
__device__ __forceinline__ void t8_shadow_body(const DeviceScene& sc, const PathPool& pool, const ShadowQueue& sq, const uint count, WaveCounters* wc, const TravAux& aux, const uint rpc,
                                               uint2* stack, uint* rayBuf, const uint vBlock, const uint vGrid) {
    Traverse8Counters ctr; t8_counters_init(ctr);
    auto fetch = [&](uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> uint {
        float4 a = sq.q0[i], b = sq.q1[i];
        o = make_float3(a.x, a.y, a.z); d = make_float3(b.x, b.y, b.z); tmin = 0.0f; tmax = a.w; startRef = 0u; bestT0 = a.w; bestPrim0 = 0xFFFFFFFFu;
        return i;
    };
    auto commit = [&](uint i, const HitInfo& h) { if (h.prim == 0xFFFFFFFFu) shadow_visible<GROUPED>(pool, sq, i); };      // occluded: nothing is committed
    // a split shadow ray: "visible so far"; its sub-trees may set the flag, k_resolve_shadow applies the contribution if none did
    auto publish = [&](uint i, float, uint) { aux.bestKey[i] = 0ull; aux.resolveList[atomicAdd(&aux.counts[TRAV_RESOLVE], 1u)] = i; };
    T8_TRAVERSE<true, COUNT, false, false, true>(sc, count, rpc, stack, rayBuf, nullptr, fetch, commit, publish, TravTaskOut{aux.taskQ[0], &aux.counts[0], aux.taskCap}, ctr, &wc->overflow, vBlock, vGrid);
    if (COUNT) { wave_add64(ctr.nodeVisits, &wc->nodeVisitsSh); wave_add64(ctr.triTests, &wc->triTestsSh); wave_add64(ctr.leafVisits, &wc->leafVisitsSh); wave_add64(ctr.iters, &wc->itersSh); }
}
template <bool COUNT, bool GROUPED>
__global__ void __launch_bounds__(T8_BLOCK, COUNT ? 1 : T8_SHADOW_MIN_WAVES) k_shadow(DeviceScene sc, PathPool pool, ShadowQueue sq, const uint* __restrict__ countPtr, WaveCounters* wc, TravAux aux, uint rpc) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_RAYBUF_WORDS];
    t8_shadow_body<COUNT, GROUPED>(sc, pool, sq, *countPtr, wc, aux, rpc, stack, rayBuf, blockIdx.x, gridDim.x);
}

template <int STAGE, bool FINAL>
__device__ __forceinline__ void t8_shadow_tasks_body(const DeviceScene& sc, const ShadowQueue& sq, WaveCounters* wc, const TravAux& aux, uint2* stack, uint* rayBuf, const uint vBlock, const uint vGrid) {
    constexpr int IN = STAGE & 1;
    uint count = aux.counts[STAGE]; if (count > aux.taskCap) count = aux.taskCap;
    if (count == 0u) return;
    const TravTask* tasks = aux.taskQ[IN];
    const uint real = t8_tasks_per_chunk(count), per = (count + real - 1u) / real;
    Traverse8Counters ctr; t8_counters_init(ctr);
    auto fetch = [&](uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> uint {
        const uint lane = i % T8_CHUNK, j = lane * per + i / T8_CHUNK;
        const bool pad = lane >= real || j >= count;
        TravTask t = tasks[pad ? 0u : j];
        float4 a = sq.q0[t.tag], b = sq.q1[t.tag];
        o = make_float3(a.x, a.y, a.z); d = make_float3(b.x, b.y, b.z); tmin = 0.0f; tmax = a.w; bestT0 = a.w; bestPrim0 = 0xFFFFFFFFu;
        startRef = (!pad && aux.bestKey[t.tag] == 0ull) ? t.ref : BVH_EMPTY;              // padding, or another sub-tree already found an occluder
        return t.tag;
    };
    auto commit = [&](uint i, const HitInfo& h) { if (h.prim != 0xFFFFFFFFu) aux.bestKey[i] = 1ull; };
    auto publish = [&](uint, float, uint) {};
    T8_TRAVERSE<true, false, false, true, !FINAL>(sc, per * T8_CHUNK, T8_CHUNK, stack, rayBuf, nullptr, fetch, commit, publish, TravTaskOut{aux.taskQ[IN ^ 1], &aux.counts[FINAL ? STAGE : STAGE + 1], FINAL ? 0u : aux.taskCap}, ctr, &wc->overflow, vBlock, vGrid);
}
template <int STAGE, bool FINAL = (STAGE == 3)>
__global__ void __launch_bounds__(T8_BLOCK) k_shadow_tasks(DeviceScene sc, ShadowQueue sq, WaveCounters* wc, TravAux aux) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_TASKBUF_WORDS];
    t8_shadow_tasks_body<STAGE, FINAL>(sc, sq, wc, aux, stack, rayBuf, blockIdx.x, gridDim.x);
}

template <bool GROUPED>
__device__ __forceinline__ void t8_resolve_shadow_body(const PathPool& pool, const ShadowQueue& sq, const TravAux& aux, const uint vBlock, const uint vGrid) {
    const uint n = aux.counts[TRAV_RESOLVE];
    for (uint k = vBlock * 256u + threadIdx.x; k < n; k += vGrid * 256u) {
        uint i = aux.resolveList[k];
        if (aux.bestKey[i] == 0ull) shadow_visible<GROUPED>(pool, sq, i);
    }
}
template <bool GROUPED>
__global__ void __launch_bounds__(256) k_resolve_shadow(PathPool pool, ShadowQueue sq, TravAux aux) { t8_resolve_shadow_body<GROUPED>(pool, sq, aux, blockIdx.x, gridDim.x); }

// ---- Fused traversal launches (round 6). The visibility rays of path vertex k and the closest-hit rays of vertex k + 1 are independent of each other — the
// visibility results only have to be in the paths' radiance before vertex k + 1 is SHADED (the order of the fp16 additions into PathState::L: the light sample
// of vertex k, then the emission found at vertex k + 1; PathTracer.hlsli:505-762, PathTracerNEE.hlsli:185-275) — so one launch traces both: blocks [0, blocksE)
// work through the extend queue, the others through the shadow queue; every block is of one kind, the traversal loops themselves are the ones k_extend /
// k_shadow run (no per-ray kind, not one instruction more in the loop). What it buys a small frame (one rank of a tile-sharded frame): half the traversal
// launches of a pass and half the straggler rounds behind them — the two kinds' task rounds and resolve passes share their launches too (k_tasks_pair,
// k_resolve_pair) — and each kind's dry tail runs beside the other kind's work instead of in a launch of its own.
__global__ void __launch_bounds__(T8_BLOCK, T8_EXTEND_MIN_BLOCKS) k_trace_pair(DeviceScene sc, PathPool pool, const uint* __restrict__ queue, const uint* __restrict__ extCountPtr, ShadowQueue sq, const uint* __restrict__ shCountPtr,
                                                                              WaveCounters* wc, TravAux auxE, TravAux auxS, uint rpcE, uint rpcS, uint blocksE) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_RAYBUF_WORDS];
    __shared__ float2 mineUV[T8_BLOCK];
    if (blockIdx.x < blocksE) t8_extend_body<false, false>(sc, pool, queue, *extCountPtr, wc, auxE, rpcE, stack, rayBuf, mineUV, blockIdx.x, blocksE);
    else t8_shadow_body<false, false>(sc, pool, sq, *shCountPtr, wc, auxS, rpcS, stack, rayBuf, blockIdx.x - blocksE, gridDim.x - blocksE);
}
template <int STAGE, bool FINAL = (STAGE == 3)>
__global__ void __launch_bounds__(T8_BLOCK, T8_EXTEND_MIN_BLOCKS) k_tasks_pair(DeviceScene sc, PathPool pool, ShadowQueue sq, WaveCounters* wc, TravAux auxE, TravAux auxS) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_TASKBUF_WORDS];
    const uint half = gridDim.x >> 1;
    if (blockIdx.x < half) t8_extend_tasks_body<STAGE, FINAL>(sc, pool, wc, auxE, stack, rayBuf, blockIdx.x, half);
    else t8_shadow_tasks_body<STAGE, FINAL>(sc, sq, wc, auxS, stack, rayBuf, blockIdx.x - half, gridDim.x - half);
}
// ... and the two resolve passes; the shadow queue's counter is zeroed here for the k_shade that follows (every reader of it — k_trace_pair — is an earlier
// launch of the stream)
__global__ void __launch_bounds__(256) k_resolve_pair(DeviceScene sc, PathPool pool, ShadowQueue sq, TravAux auxE, TravAux auxS, uint* shadowCount) {
    const uint half = gridDim.x >> 1;
    if (blockIdx.x < half) t8_resolve_extend_body(sc, pool, auxE, blockIdx.x, half);
    else { t8_resolve_shadow_body<false>(pool, sq, auxS, blockIdx.x - half, gridDim.x - half); if (blockIdx.x == half && threadIdx.x == 0u) *shadowCount = 0u; }
}

// NEEFullSamples != 1: NEEResult accumulates the visible samples of a path vertex in sample order (fp16, PathTracerTypes.hlsli:170-207), then the
// vertex adds the sum to the path once (PathTracer.hlsli:722-746). One thread per group of the shadow queue.
__global__ void __launch_bounds__(256) k_resolve_nee(PathPool pool, ShadowQueue sq, const uint* __restrict__ countPtr) {
    const uint groups = *countPtr / sq.group;
    for (uint g = blockIdx.x * 256u + threadIdx.x; g < groups; g += gridDim.x * 256u) {
        const uint first = g * sq.group;
        uint nee[2] = {0u, 0u};
        for (uint s = 0; s < sq.group; s++) {
            float4 r = sq.q2[first + s];
            if (r.w != 0.f) PathKernelContext::NeeAccumulate(nee, make_float3(r.x, r.y, r.z));
        }
        uint p = asuint(sq.q1[first].w);
        uint4 c = pool.s2[p];
        uint pack45[2] = {c.z, c.w};
        PathKernelContext::NeeCommit(pack45, nee);
        c.z = pack45[0]; c.w = pack45[1];
        pool.s2[p] = c;
    }
}

__global__ void __launch_bounds__(256) k_accumulate(PathPool pool, const uint* __restrict__ ownedPixels, uint numOwned, uint spp, float4* __restrict__ accum, uint accumCountBase, uint width) {
    uint kpx = blockIdx.x * 256u + threadIdx.x;
    if (kpx >= numOwned) return;
    uint px = ownedPixels[kpx];
    uint addr = (px & 0xFFFFu) * width + (px >> 16);
    float4 acc = accum[addr];
    for (uint s = 0; s < spp; s++) {
        uint4 c = pool.s2[PT_SAMPLE_MINOR ? kpx * spp + s : s * numOwned + kpx];
        float2 l0 = Fp16ToFp32(c.z), l1 = Fp16ToFp32(c.w);
        float4 col = make_float4(l0.x, l0.y, l1.x, 1.0f);
        float blend = 1.0f / (float)(accumCountBase + s + 1u);
        acc = (blend < 1.f) ? lerp4(acc, col, blend) : col;
    }
    accum[addr] = acc;
}

__global__ void __launch_bounds__(T8_BLOCK) k_trace_probe(DeviceScene sc, const float4* __restrict__ rays, uint n, float4* __restrict__ outClosest, uint* __restrict__ outVisible, uint* overflow) {
    __shared__ uint2 stack[T8_GROUPS_PER_BLOCK * BVH8_STACK_STRIDE];
    __shared__ uint rayBuf[T8_RAYBUF_WORDS];
    __shared__ float2 mineUV[T8_BLOCK];
    Traverse8Counters ctr; t8_counters_init(ctr);
    auto fetch = [&](uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> uint {
        float4 a = rays[2 * i], b = rays[2 * i + 1];
        o = make_float3(a.x, a.y, a.z); d = make_float3(b.x, b.y, b.z); tmin = a.w; tmax = b.w; startRef = 0u; bestT0 = b.w; bestPrim0 = 0xFFFFFFFFu;
        return i;
    };
    auto publish = [&](uint, float, uint) {};
    if (outClosest) {
        auto commit = [&](uint i, const HitInfo& h) { outClosest[i] = make_float4(h.t, asfloat(h.prim), h.u, h.v); };
        T8_TRAVERSE<false, false, false, false, false>(sc, n, T8_CHUNK, stack, rayBuf, mineUV, fetch, commit, publish, TravTaskOut{nullptr, nullptr, 0u}, ctr, overflow, blockIdx.x, gridDim.x);
    } else {
        auto commit = [&](uint i, const HitInfo& h) { outVisible[i] = (h.prim == 0xFFFFFFFFu) ? 1u : 0u; };
        T8_TRAVERSE<true, false, false, false, false>(sc, n, T8_CHUNK, stack, rayBuf, mineUV, fetch, commit, publish, TravTaskOut{nullptr, nullptr, 0u}, ctr, overflow, blockIdx.x, gridDim.x);
    }
}

__global__ void __launch_bounds__(256) k_pack(const float4* __restrict__ accum, const uint* __restrict__ pixels, uint num, uint width, float4* __restrict__ dst) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= num) return;
    uint px = pixels[i]; dst[i] = accum[(px & 0xFFFFu) * width + (px >> 16)];
}
__global__ void __launch_bounds__(256) k_unpack(float4* __restrict__ accum, const uint* __restrict__ pixels, uint num, uint width, const float4* __restrict__ src) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= num) return;
    uint px = pixels[i]; accum[(px & 0xFFFFu) * width + (px >> 16)] = src[i];
}

// ---- EnvMapBaker on the device (EnvMapBaker.hlsl:194-246, 268-371; pt_envcube.h): BaseLayerCS makes four texels of mip 0 and their mip-1 texel per thread,
// MIPReduceCS the further levels from the stored (fp16) texels; solid-angle weighted, summation order of the shader
__device__ __forceinline__ float4 env_generate_texel(const DeviceScene& sc, const EnvDirectionalLight* __restrict__ lights, uint nLights, uint px, uint py, uint face, uint dim) {
    float3 envCol = env_sample_source(sc, CubemapGetDirectionFor(face, make_float2(((float)px + 0.0f + 0.5f) / (float)dim, ((float)py + 0.0f + 0.5f) / (float)dim)));
    for (uint i = 0; i < nLights; i++) envCol = envCol + EnvComputeLightContribution(px, py, face, lights[i], dim);
    if (sc.sky) {                                       // g_Const.ProcSkyEnabled (EnvMapBaker.hlsl:224-236): toLocal swaps y and z
        const float3 cubeDir = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f) / (float)dim, ((float)py + 0.5f) / (float)dim));
        const float3 cubeDirRight = CubemapGetDirectionFor(face, make_float2((((float)px + 1.0f) + 0.5f) / (float)dim, ((float)py + 0.5f) / (float)dim)) - cubeDir;
        const float3 cubeDirBottom = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f) / (float)dim, (((float)py + 1.0f) + 0.5f) / (float)dim)) - cubeDir;
        envCol = envCol + ProceduralSky(make_float3(cubeDir.x, cubeDir.z, cubeDir.y), *sc.sky, sc.skyLowRes, cubeDir, cubeDirRight, cubeDirBottom);
    }
    envCol = envCol * kEnvMapRadianceScale;
    envCol = clamp3(envCol, 0.0f, HLF_MAX);
    return make_float4(envCol.x, envCol.y, envCol.z, 1.0f);
}
__device__ __forceinline__ float4 env_reduce(float4 e00, float4 e01, float4 e10, float4 e11, float4 wsa) {
    float wsum = wsa.x + wsa.y + wsa.z + wsa.w;
    float4 s = (e00 * wsa.x + e01 * wsa.y) + e10 * wsa.z + e11 * wsa.w;
    return make_float4(s.x / wsum, s.y / wsum, s.z / wsum, s.w / wsum);
}
// LowResPrePassLayerCS (EnvMapBaker.hlsl:247-265): the clouds of the procedural sky at half the cube's resolution, one thread per texel, RGBA16F
__global__ void __launch_bounds__(256) k_env_sky_lowres(DeviceScene sc, uint2* __restrict__ texels, uint res) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= 6u * res * res) return;
    const uint face = i / (res * res), r = i - face * res * res, y = r / res, x = r - y * res;
    const float3 direction = CubemapGetDirectionFor(face, make_float2(((float)x + 0.5f) / (float)res, ((float)y + 0.5f) / (float)res));
    texels[i] = env_pack_rgba16f(ProceduralSkyLowRes(x, y, face, make_float3(direction.x, direction.z, direction.y), *sc.sky));
}
__global__ void __launch_bounds__(256) k_env_cube_base(DeviceScene sc, const EnvDirectionalLight* __restrict__ lights, uint nLights, uint2* __restrict__ texels, EnvCube cube) {
    const uint dim = cube.dim, h = dim / 2u;
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= 6u * h * h) return;
    const uint face = i / (h * h), r = i - face * h * h, y = r / h, x = r - y * h;
    float4 e00 = env_generate_texel(sc, lights, nLights, 2 * x, 2 * y, face, dim), e01 = env_generate_texel(sc, lights, nLights, 2 * x, 2 * y + 1, face, dim),
           e10 = env_generate_texel(sc, lights, nLights, 2 * x + 1, 2 * y, face, dim), e11 = env_generate_texel(sc, lights, nLights, 2 * x + 1, 2 * y + 1, face, dim);
    uint2* m0 = texels + cube.mipOffset[0] + (size_t)face * dim * dim;
    m0[(size_t)(2 * y) * dim + 2 * x] = env_pack_rgba16f(e00); m0[(size_t)(2 * y + 1) * dim + 2 * x] = env_pack_rgba16f(e01);
    m0[(size_t)(2 * y) * dim + 2 * x + 1] = env_pack_rgba16f(e10); m0[(size_t)(2 * y + 1) * dim + 2 * x + 1] = env_pack_rgba16f(e11);
    if (cube.mipLevels > 1u) texels[cube.mipOffset[1] + ((size_t)face * h + y) * h + x] = env_pack_rgba16f(env_reduce(e00, e01, e10, e11, CubemapTexelSolidAngle4((float)dim, 2 * x, 2 * y)));
}
__global__ void __launch_bounds__(256) k_env_cube_mip(uint2* __restrict__ texels, EnvCube cube, uint level) {
    const uint d = cube.dim >> level, s = d * 2u;
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= 6u * d * d) return;
    const uint face = i / (d * d), r = i - face * d * d, y = r / d, x = r - y * d;
    const uint2* src = texels + cube.mipOffset[level - 1u] + (size_t)face * s * s;
    texels[cube.mipOffset[level] + ((size_t)face * d + y) * d + x] = env_pack_rgba16f(env_reduce(
        env_unpack_rgba16f(src[(size_t)(2 * y) * s + 2 * x]), env_unpack_rgba16f(src[(size_t)(2 * y + 1) * s + 2 * x]),
        env_unpack_rgba16f(src[(size_t)(2 * y) * s + 2 * x + 1]), env_unpack_rgba16f(src[(size_t)(2 * y + 1) * s + 2 * x + 1]), CubemapTexelSolidAngle4((float)s, 2 * x, 2 * y)));
}
// BC6UCompress.hlsl CSMain (QUALITY 0) + the texture unit's BC6H_UF16 decode, in place: one thread per 4x4 block of one cube level (pt_envcube.h)
__global__ void __launch_bounds__(64) k_env_cube_bc6(uint2* __restrict__ level, uint dim, uint quality) {
    const uint nb = dim / 4u; uint i = blockIdx.x * 64u + threadIdx.x; if (i >= 6u * nb * nb) return;
    const uint face = i / (nb * nb), r = i - face * nb * nb, by = r / nb, bx = r - by * nb;
    env_cube_bc6_round_trip_block(level, dim, face, bx, by, quality);
}
void launch_env_cube_compress(uint2* texels, const EnvCube& cube, uint quality, hipStream_t st) {
    for (uint l = 0; l < cube.mipLevels; l++) { const uint d = cube.dim >> l, nb = d / 4u; hipLaunchKernelGGL(k_env_cube_bc6, dim3((6u * nb * nb + 63u) / 64u), dim3(64), 0, st, texels + cube.mipOffset[l], d, quality); }
}
void launch_env_cube_bake(const DeviceScene& sc, const EnvDirectionalLight* lights, uint nLights, uint2* texels, const EnvCube& cube, hipStream_t st) {
    const uint h = cube.dim / 2u;
    if (sc.sky) hipLaunchKernelGGL(k_env_sky_lowres, dim3((6u * h * h + 255u) / 256u), dim3(256), 0, st, sc, const_cast<uint2*>(sc.skyLowRes.texels), h);
    hipLaunchKernelGGL(k_env_cube_base, dim3((6u * h * h + 255u) / 256u), dim3(256), 0, st, sc, lights, nLights, texels, cube);
    for (uint l = 2; l < cube.mipLevels; l++) { const uint d = cube.dim >> l; hipLaunchKernelGGL(k_env_cube_mip, dim3((6u * d * d + 255u) / 256u), dim3(256), 0, st, texels, cube, l); }
}

// BuildMIPDescentImportanceMapCS (Rtxpt/Lighting/Distant/EnvMapImportanceSamplingBaker.hlsl:57-90) on the baked cube
__global__ void __launch_bounds__(256) k_env_importance(DeviceScene sc, uint dim, uint sx, uint sy, float4* __restrict__ out) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= dim * dim) return;
    uint x = i % dim, y = i / dim;
    float L = 0.f; float3 R = make_float3(0.f);
    const float invSamples = 1.f / (float)(sx * sy);
    for (uint j = 0; j < sy; j++) for (uint ii = 0; ii < sx; ii++) {
        float2 p = make_float2(((float)(x * sx + ii) + 0.5f) / (float)(dim * sx), ((float)(y * sy + j) + 0.5f) / (float)(dim * sy));
        float3 dir = oct_to_ndir_equal_area_unorm(p);
        // t_EnvMapCube.SampleLevel(s_LinearWrap, dir, 0) (:77) — the uncompressed cube (EnvMapBaker.cpp:635)
        float3 radiance = xyz(env_cube_sample_level(sc.envCubeSource, dir, 0.f));
        L += (Luminance(radiance) + Average(radiance)) * 0.5f;
        R += radiance;
    }
    // u_RadianceMap is RGBA16_FLOAT (EnvMapImportanceSamplingBaker.cpp:170)
    out[i] = env_round_rgba16f(make_float4(R.x * invSamples, R.y * invSamples, R.z * invSamples, L * invSamples));
}

// BakeEmissiveTriangles (Rtxpt/Lighting/LightsBaker.hlsl:544-716): one thread per emissive triangle, output in sub-instance order
__global__ void __launch_bounds__(256) k_bake_emissive(DeviceScene sc, const uint* __restrict__ subInstList, const uint* __restrict__ subInstTriOffset, uint numEmissiveSubInst,
                                                       uint totalTris, uint lightBase, PolymorphicLightInfo* __restrict__ lights, PolymorphicLightInfoEx* __restrict__ lightsEx) {
    uint t = blockIdx.x * 256u + threadIdx.x; if (t >= totalTris) return;
    uint lo = 0, hi = numEmissiveSubInst;                 // last k with offset[k] <= t
    while (hi - lo > 1) { uint mid = (lo + hi) >> 1; if (subInstTriOffset[mid] <= t) lo = mid; else hi = mid; }
    uint s = subInstList[lo], tri = t - subInstTriOffset[lo];
    uint2 ig = sc.subInstToInstGeom[s];
    const InstanceDesc& inst = sc.instances[ig.x];
    const GeometryDesc& g = sc.geometries[ig.y];
    const PTMaterialData& mat = sc.materials[g.materialIndex];
    const uint* idx = sc.indices + g.indexOffset + 3u * tri;
    uint i0 = g.vertexOffset + idx[0], i1 = g.vertexOffset + idx[1], i2 = g.vertexOffset + idx[2];
    const float* P = sc.positions;
    float3 p0 = xform_point(inst.transform, make_float3(P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2]));
    float3 p1 = xform_point(inst.transform, make_float3(P[3 * i1], P[3 * i1 + 1], P[3 * i1 + 2]));
    float3 p2 = xform_point(inst.transform, make_float3(P[3 * i2], P[3 * i2 + 1], P[3 * i2 + 2]));
    float3 radiance = mat.EmissiveColor;
    if ((mat.Flags & PTMaterialFlags_UseEmissiveTexture) && (g.flags & GEOM_HAS_UV)) {
        float2 uv0 = sc.uvs[i0], uv1 = sc.uvs[i1], uv2 = sc.uvs[i2];
        float2 e0 = uv1 - uv0, e1 = uv2 - uv1, e2 = uv0 - uv2;
        float l0 = length(e0), l1 = length(e1), l2 = length(e2);
        float2 shortE, longE1, longE2;
        if (l0 < l1 && l0 < l2) { shortE = e0; longE1 = e1; longE2 = e2; } else if (l1 < l2) { shortE = e1; longE1 = e2; longE2 = e0; } else { shortE = e2; longE1 = e0; longE2 = e1; }
        float2 sg = shortE * (2.0f / 3.0f); float2 lg = (longE1 + longE2) * (1.0f / 3.0f);
        const TexInfo& tex = sc.textures[mat.EmissiveTextureIndex & 0xFFFFu];
        float2 c = (uv0 + uv1 + uv2) * (1.0f / 3.0f);
        // emissiveTexture.SampleGrad(s_materialSampler, centerUV, shortGradient, longGradient) (:647)
        radiance = radiance * xyz(sample_grad_anisotropic(sc, tex, c, sg, lg));
    }
    radiance = max3v(radiance, make_float3(0.f));
    bool isFlipped = det3(inst.transform) < 0.f;
    TriangleLight tl; tl.base = p0;
    if (!isFlipped) { tl.edge1 = p1 - p0; tl.edge2 = p2 - p0; } else { tl.edge1 = p2 - p0; tl.edge2 = p1 - p0; }
    if (fmaxf_(radiance.x, fmaxf_(radiance.y, radiance.z)) < 1e-7f) radiance = make_float3(0.f);
    tl.radiance = radiance; tl.normal = make_float3(0.f); tl.surfaceArea = 0;
    PolymorphicLightInfoFull lf = tl.Store(0);
    lights[lightBase + t] = lf.Base; lightsEx[lightBase + t] = lf.Extended;
}

#ifdef MI355PT_TEST_HOOKS      // libmi355pt_testhooks.so only (the tests' build of the library): the shipped library carries no evaluation hooks
// known-answer probes for the tests: the device evaluates leaf functions so they can be compared bit-for-bit with the oracle
__global__ void __launch_bounds__(64) k_probe(PathKernelContext k, int kind, const float* __restrict__ in, float* __restrict__ out, uint n) {
    uint i = blockIdx.x * 64u + threadIdx.x; if (i >= n) return;
    switch (kind) {
    case 0: { const float* p = in + 3 * i; int fn = (int)p[0]; float x = p[1], y = p[2]; float r;      // dmath: (fn, x, y)
        switch (fn) { case 0: r = dm_sin(x); break; case 1: r = dm_cos(x); break; case 2: r = dm_exp2(x); break; case 3: r = dm_log2(x); break;
                      case 4: r = dm_atan2(y, x); break; case 5: r = dm_pow(x, y); break; case 6: r = FastACos(x); break; default: r = FastSqrt(x); break; }
        out[i] = r; } break;
    case 1: { float x = in[i]; uint hbits = f32tof16(x); out[2 * i] = asfloat(hbits); out[2 * i + 1] = f16tof32(hbits); } break;          // fp16 round trip
    // sample stream: pixel, vertex, sample, seed, kind, count(<=8)
    case 2: { const uint* p = reinterpret_cast<const uint*>(in) + 6 * i;
        SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make(p[0], p[1], p[2]);
        uint cnt = p[5] > 8 ? 8 : p[5];
        if (p[4] == 0) { float4 v = SampleSequenceGenerator::Generate(cnt, vb, p[3]); out[8 * i] = v.x; out[8 * i + 1] = v.y; out[8 * i + 2] = v.z; out[8 * i + 3] = v.w; }
        else if (p[4] == 1) { float4 v = UniformSampleSequenceGenerator::Generate(cnt, vb, p[3]); out[8 * i] = v.x; out[8 * i + 1] = v.y; out[8 * i + 2] = v.z; out[8 * i + 3] = v.w; }
        else if (p[4] == 2) { UniformSampleSequenceGenerator g = UniformSampleSequenceGenerator::make(vb, p[3]); for (uint j = 0; j < cnt; j++) out[8 * i + j] = sampleNext1D(g); }
        else { SampleSequenceGenerator g = SampleSequenceGenerator::make(vb, p[3], p[4] == 4); for (uint j = 0; j < cnt; j++) out[8 * i + j] = sampleNext1D(g); } } break;
    // bsdf probe: params[14], thin, model, wi[3], w[3], mode
    case 3: { const float* p = in + 24 * i;
        ShadingData sd; __builtin_memset(&sd, 0, sizeof(sd));
        sd.N = make_float3(0, 0, 1); sd.T = make_float3(1, 0, 0); sd.B = make_float3(0, 1, 0); sd.V = make_float3(p[16], p[17], p[18]);
        sd.faceNCorrected = sd.N; sd.vertexN = sd.N; sd.frontFacing = true; sd.mtl = MaterialHeader::make(); sd.mtl.setActiveLobes(Lobe_All); sd.mtl.setThinSurface(p[14] != 0.f);
        StandardBSDF b; b.diffuseModel = (int)p[15];
        b.data.diffuse = make_float3(p[0], p[1], p[2]); b.data.specular = make_float3(p[3], p[4], p[5]); b.data.roughness = p[6]; b.data.metallic = p[7];
        b.data.transmission = make_float3(p[8], p[9], p[10]); b.data.diffuseTransmission = p[11]; b.data.specularTransmission = p[12]; b.data.eta = p[13];
        float* o = out + 10 * i;
        if (p[22] == 0.f) { float3 wo = make_float3(p[19], p[20], p[21]); float4 e = b.eval(sd, wo); o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = e.w; o[4] = b.evalPdf(sd, wo); o[5] = (float)b.getLobes(); o[6] = o[7] = o[8] = o[9] = 0.f; }
        else { BSDFSample s; __builtin_memset(&s, 0, sizeof(s)); bool v = b.sample(sd, make_float4(p[19], p[20], p[21], 0), s);
               o[0] = s.wo.x; o[1] = s.wo.y; o[2] = s.wo.z; o[3] = s.pdf; o[4] = s.weight.x; o[5] = s.weight.y; o[6] = s.weight.z; o[7] = (float)s.lobe; o[8] = s.lobeP; o[9] = v ? 1.f : 0.f; } } break;
    case 4: { const uint* p = reinterpret_cast<const uint*>(in) + 3 * i;                                   // camera ray: px, py, sampleIndex
        PathState ps = k.generate(p[0], p[1], p[2]);
        out[6 * i] = ps.origin.x; out[6 * i + 1] = ps.origin.y; out[6 * i + 2] = ps.origin.z; out[6 * i + 3] = ps.dir.x; out[6 * i + 4] = ps.dir.y; out[6 * i + 5] = ps.dir.z; } break;
    // leaf functions pinned to the reference text (tests/golden/refpin_hlsl_golden.npz): (fn, 8 args) -> 4 results
    case 5: { const float* a = in + 9 * i + 1; int fn = (int)in[9 * i]; float* o = out + 4 * i; o[0] = o[1] = o[2] = o[3] = 0.f;
        switch (fn) {
        case 0: o[0] = evalFresnelSchlick(a[0], a[1], a[2]); break;
        case 1: { float3 r = evalFresnelSchlick(make_float3(a[0], a[1], a[2]), a[3], a[4]); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 2: { float ct = 0.f; o[0] = evalFresnelDielectric(a[0], a[1], ct); o[1] = ct; } break;
        case 3: o[0] = evalNdfGGX(a[0], a[1]); break;
        case 4: o[0] = evalPdfGGX_BVNDF(a[0], make_float3(a[1], a[2], a[3]), make_float3(a[4], a[5], a[6])); break;
        case 5: { float3 r = sampleGGX_BVNDF(a[0], make_float3(a[1], a[2], a[3]), make_float2(a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 6: o[0] = evalLambdaGGX(a[0], a[1]); break;
        case 7: o[0] = evalMaskingSmithGGXCorrelated(a[0], a[1], a[2]); break;
        case 8: { float2 r = ndir_to_oct_equal_area_unorm(make_float3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; } break;
        case 9: { float3 r = oct_to_ndir_equal_area_unorm(make_float2(a[0], a[1])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 10: { float2 r = sample_disk(make_float2(a[0], a[1])); o[0] = r.x; o[1] = r.y; } break;
        case 11: { float2 r = sample_disk_concentric(make_float2(a[0], a[1])); o[0] = r.x; o[1] = r.y; } break;
        case 12: { float pdf = 0.f; float3 r = sample_cosine_hemisphere_concentric(make_float2(a[0], a[1]), pdf); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf; } break;
        case 13: { float3 r = perp_stark(make_float3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 14: { float3 r = ComputeRayOrigin(make_float3(a[0], a[1], a[2]), make_float3(a[3], a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 15: o[0] = FastSqrt(a[0]); break;
        case 16: o[0] = FastACos(a[0]); break;
        case 17: o[0] = ComputeRayConeSpreadAngleExpansionByScatterPDF(a[0], a[1]); break;
        case 18: o[0] = ComputeNewScatterFireflyFilterK<LPOps<false>>(a[0], a[1], a[2]); break;
        case 19: { float3 r = FireflyFilter<LPOps<false>>(make_float3(a[0], a[1], a[2]), a[3], a[4]); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 20: o[0] = FireflyFilterShort(a[0], a[1], a[2]); break;
        case 21: o[0] = ComputeLowGrazingAngleFalloff(make_float3(a[0], a[1], a[2]), make_float3(a[3], a[4], a[5]), a[6], a[7]); break;
        default: break;
        } } break;
    case 6: { const uint* a = reinterpret_cast<const uint*>(in) + 19 * i + 1; const uint lk = reinterpret_cast<const uint*>(in)[19 * i]; uint* o = reinterpret_cast<uint*>(out) + 12 * i;      // polymorphic lights: (kind, 18 words) -> 12 words
        for (int q = 0; q < 12; q++) o[q] = 0u;
        const float3x4 I = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
        auto info = [&](const uint* w) { PolymorphicLightInfoFull li; __builtin_memset(&li, 0, sizeof(li));
            li.Base.Center = make_float3(asfloat(w[0]), asfloat(w[1]), asfloat(w[2])); li.Base.ColorTypeAndFlags = w[3]; li.Base.Direction1 = w[4]; li.Base.Direction2 = w[5]; li.Base.Scalars = w[6]; li.Base.LogRadiance = w[7];
            li.Extended.IesProfileIndex = w[8]; li.Extended.PrimaryAxis = w[9]; li.Extended.CosConeAngleAndSoftness = w[10]; li.Extended.UniqueID = w[11]; return li; };
        if (lk == 0u) { PolymorphicLightInfo b; __builtin_memset(&b, 0, sizeof(b)); PackLightColor(make_float3(asfloat(a[0]), asfloat(a[1]), asfloat(a[2])), b); float3 c = UnpackLightColor(b);
            o[0] = b.ColorTypeAndFlags; o[1] = b.LogRadiance; o[2] = asuint(c.x); o[3] = asuint(c.y); o[4] = asuint(c.z); }
        else if (lk == 1u) { TriangleLight t; t.base = make_float3(asfloat(a[0]), asfloat(a[1]), asfloat(a[2])); t.edge1 = make_float3(asfloat(a[3]), asfloat(a[4]), asfloat(a[5])); t.edge2 = make_float3(asfloat(a[6]), asfloat(a[7]), asfloat(a[8]));
            t.radiance = make_float3(asfloat(a[9]), asfloat(a[10]), asfloat(a[11])); t.normal = make_float3(0.f); t.surfaceArea = 0;
            PolymorphicLightInfoFull li = t.Store(7u);
            o[0] = asuint(li.Base.Center.x); o[1] = asuint(li.Base.Center.y); o[2] = asuint(li.Base.Center.z); o[3] = li.Base.ColorTypeAndFlags; o[4] = li.Base.Direction1; o[5] = li.Base.Direction2; o[6] = li.Base.Scalars; o[7] = li.Base.LogRadiance;
            o[8] = li.Extended.IesProfileIndex; o[9] = li.Extended.PrimaryAxis; o[10] = li.Extended.CosConeAngleAndSoftness; o[11] = li.Extended.UniqueID; }
        else if (lk == 2u) { PolymorphicLightInfoFull li = info(a);
            PolymorphicLightSample sm = PolymorphicLight_CalcSample(li, make_float2(asfloat(a[12]), asfloat(a[13])), make_float3(asfloat(a[14]), asfloat(a[15]), asfloat(a[16])), I);
            o[0] = asuint(sm.Position.x); o[1] = asuint(sm.Position.y); o[2] = asuint(sm.Position.z); o[3] = asuint(sm.Normal.x); o[4] = asuint(sm.Normal.y); o[5] = asuint(sm.Normal.z);
            o[6] = asuint(sm.Radiance.x); o[7] = asuint(sm.Radiance.y); o[8] = asuint(sm.Radiance.z); o[9] = asuint(sm.SolidAnglePdf); o[10] = sm.LightSampleableByBSDF ? 1u : 0u; o[11] = asuint(PolymorphicLight_GetPower(li)); }
        else if (lk == 3u) { TriangleLight t = TriangleLight::Create(info(a));
            o[0] = asuint(t.CalcSolidAnglePdfForMIS(make_float3(asfloat(a[12]), asfloat(a[13]), asfloat(a[14])), make_float3(asfloat(a[15]), asfloat(a[16]), asfloat(a[17])))); }
        else { uint pk = NDirToOctUnorm32(make_float3(asfloat(a[0]), asfloat(a[1]), asfloat(a[2]))); float3 d = OctToNDirUnorm32(pk); o[0] = pk; o[1] = asuint(d.x); o[2] = asuint(d.y); o[3] = asuint(d.z); }
        } break;
    // loadSurface: (prim bits, u, v, dir.xyz, coneWidth, coneSpread) -> 45 words (layout of the oracle's surface probe)
    case 8: { const float* a = in + 8 * i; uint* o = reinterpret_cast<uint*>(out) + 45 * i;
        RayCone rc = RayCone::make(a[6], a[7]);
        auto emit = [&](const SurfaceData& q) {
            const ShadingData& s = q.shadingData; const StandardBSDFData& b = q.bsdf.data;
            auto put3 = [&](float3 v) { *o++ = asuint(v.x); *o++ = asuint(v.y); *o++ = asuint(v.z); };
            put3(s.posW); put3(s.faceNCorrected); put3(s.V); put3(s.N); put3(s.T); put3(s.B); put3(s.vertexN);
            *o++ = s.frontFacing ? 1u : 0u; *o++ = s.mtl.packedData; *o++ = s.materialID; *o++ = asuint(s.IoR); *o++ = asuint(s.shadowNoLFadeout); put3(s.emission);
            put3(b.diffuse); *o++ = asuint(b.roughness); put3(b.specular); *o++ = asuint(b.metallic); put3(b.transmission);
            *o++ = asuint(b.diffuseTransmission); *o++ = asuint(b.specularTransmission); *o++ = asuint(b.eta); *o++ = asuint(q.interiorIoR); *o++ = q.neeTriangleLightIndex; };
        if (k.S.useFp16Types) { PathKernelContextT<true> k16; __builtin_memcpy(&k16, &k, sizeof(k16)); emit(k16.loadSurface(asuint(a[0]), a[1], a[2], make_float3(a[3], a[4], a[5]), rc)); }
        else emit(k.loadSurface(asuint(a[0]), a[1], a[2], make_float3(a[3], a[4], a[5]), rc));
        } break;
    // the half-typed operators of the lp16 build: (op, a, b, c) -> result
    case 7: { const float* a = in + 4 * i; const int op = (int)a[0]; typedef LPOps<true> H; float r = 0.f;
        switch (op) { case 0: r = H::r(a[1]); break; case 1: r = H::add(H::r(a[1]), H::r(a[2])); break; case 2: r = H::sub(H::r(a[1]), H::r(a[2])); break; case 3: r = H::mul(H::r(a[1]), H::r(a[2])); break;
                      case 4: r = H::div(H::r(a[1]), H::r(a[2])); break; case 5: r = H::lerp(H::r(a[1]), H::r(a[2]), H::r(a[3])); break;
                      case 7: r = H::r(a[1] * a[2]); break; case 8: r = H::r(H::r(a[1]) * a[2]); break; case 9: r = H::r(a[1] + a[2]); break; case 10: r = H::r(a[1] / a[2]); break;      // lpfloat(float expression)
                      default: r = H::average3(make_float3(H::r(a[1]), H::r(a[2]), H::r(a[3]))); break; }
        out[i] = r; } break;
    // the traversal's alpha test: (primitive, u, v) -> (scatter ray accepts, visibility ray accepts)
    case 10: { const uint* a = reinterpret_cast<const uint*>(in) + 3 * i; uint* o = reinterpret_cast<uint*>(out) + 2 * i;
        // AlphaTestImpl / Bridge::AlphaTest / AlphaTestVisibilityRay (BridgeDonut:929-989) as the leaf block applies them
        const uint slot = k.sc.primToSlot[a[0]]; const TriRecord tr = k.sc.tris[slot];
        const bool solid = !(tr.flags & 1u) || alpha_test_slot(k.sc, slot, asfloat(a[1]), asfloat(a[2]));
        o[0] = solid ? 1u : 0u; o[1] = (!(tr.flags & 1u) || (!(tr.flags & 2u) && solid)) ? 1u : 0u; } break;
    // EnvMap::EvalLocal on the baked cube: (localDir.xyz, lod)
    case 9: { const float* a = in + 4 * i; float* o = out + 3 * i;
        float3 r = k.sc.envEnabled ? env_eval_local(k.sc, make_float3(a[0], a[1], a[2]), a[3]) : make_float3(0.f);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
    default: break;
    }
}
#endif

#ifndef T8_ADAPTIVE_CHUNKS
#define T8_ADAPTIVE_CHUNKS 1        // 1: launches below (resident waves x 64) rays use shorter chunks (see traverse8_pairs), 0: always 64 rays per chunk
#endif
// rays per chunk of a traversal launch: the waves the GPU can hold (256 CUs x 4 SIMDs x 8 waves) should cover the launch in one go, 16 .. 64 rays each, in
// steps of 16
static inline uint rays_per_chunk(uint count) {
    if (!T8_ADAPTIVE_CHUNKS) return T8_CHUNK;
    const uint resident = 256u * 4u * 8u;
    const uint step = T8_GROUPS_PER_WAVE;                   // one ray per lane group
    uint r = ((count + resident - 1u) / resident + step - 1u) / step * step;
    return r < step ? (step > T8_CHUNK ? T8_CHUNK : step) : (r > T8_CHUNK ? T8_CHUNK : r);
}
static inline uint grid_for(uint count, uint block, uint maxBlocks) { uint g = (count + block - 1) / block; if (g < 1) g = 1; if (g > maxBlocks) g = maxBlocks; return g; }

// ToneMappingPass::Render + SRGBA8 store (ToneMapping.ps.hlsli:136-174): one thread per pixel, streaming 16 B in / 4 B out
__global__ void __launch_bounds__(256) k_tonemap(const float4* __restrict__ accum, uint num, ToneMapParams p, uint* __restrict__ out) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i < num) out[i] = tm_pixel(p, accum[i]);
}
void launch_tonemap(const float4* accum, uint num, const ToneMapParams& p, uint* outRgba8, hipStream_t st) {
    hipLaunchKernelGGL(k_tonemap, dim3((num + 255) / 256), dim3(256), 0, st, accum, num, p, outRgba8);
}
// auto-exposure luminance capture: log-luminance target, then its mip chain down to 1x1 (ping-pong between the two halves of `scratch`)
__global__ void __launch_bounds__(256) k_log_luminance(const float4* __restrict__ accum, uint W, uint H, uint LW, uint LH, float* __restrict__ out) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i < LW * LH) out[i] = tm_log_luminance_texel(accum, W, H, LW, LH, i % LW, i / LW);
}
__global__ void __launch_bounds__(256) k_luminance_mip(const float* __restrict__ src, uint w, uint h, uint ow, uint oh, float* __restrict__ dst) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i < ow * oh) dst[i] = tm_mip_texel(src, w, h, i % ow, i / ow);
}
void launch_average_log_luminance(const float4* accum, uint W, uint H, float* scratch, float** result, hipStream_t st) {
    uint w = tm_pow2_floor(W), h = tm_pow2_floor(H);
    float* a = scratch; float* b = scratch + (size_t)w * h;
    hipLaunchKernelGGL(k_log_luminance, dim3((w * h + 255) / 256), dim3(256), 0, st, accum, W, H, w, h, a);
    while (w > 1u || h > 1u) {
        uint ow = w > 1u ? w / 2u : 1u, oh = h > 1u ? h / 2u : 1u;
        hipLaunchKernelGGL(k_luminance_mip, dim3((ow * oh + 255) / 256), dim3(256), 0, st, a, w, h, ow, oh, b);
        float* t = a; a = b; b = t; w = ow; h = oh;
    }
    *result = a;
}
void launch_generate(const PathKernelContext& k, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleFirst, uint spp, uint first, uint n, uint* queue, uint* countPtr, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_generate, dim3((n + 255) / 256), dim3(256), 0, st, k, pool, ownedPixels, numOwned, sampleFirst, spp, first, n, queue, countPtr);
}
// task rounds + resolve pass of one traversal launch; all counts live on the device, so the grids are fixed (empty rounds return at once)
static const uint T8_TASK_BLOCKS = T8_TASK_BLOCKS_N, T8_RESOLVE_BLOCKS = 256;
void launch_extend(const DeviceScene& sc, PathPool pool, const uint* queue, const uint* countPtr, uint count, WaveCounters* wc, bool counters, TravAux aux, hipStream_t st, bool ranged) {
    const uint rpc = rays_per_chunk(count);
    uint g = grid_for(count, (T8_BLOCK / 64u) * rpc * T8_CHUNKS_PER_WAVE_MIN, (aux.maxBlocks && aux.maxBlocks < T8_MAX_BLOCKS) ? aux.maxBlocks : T8_MAX_BLOCKS);
    if (ranged) { if (counters) hipLaunchKernelGGL((k_extend<true, true>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, queue, countPtr, wc, aux, rpc);
                  else hipLaunchKernelGGL((k_extend<false, true>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, queue, countPtr, wc, aux, rpc); }
    else if (counters) hipLaunchKernelGGL((k_extend<true>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, queue, countPtr, wc, aux, rpc);
    else hipLaunchKernelGGL((k_extend<false>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, queue, countPtr, wc, aux, rpc);
    // a small launch holds few stragglers and short ones: two task rounds (split once more, then finish) instead of four — late bounces are
    if (count <= T8_SHORT_TAIL_BELOW) {
        // bound by the host's launch rate, not by the GPU
        hipLaunchKernelGGL((k_extend_tasks<0>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, pool, wc, aux);
        hipLaunchKernelGGL((k_extend_tasks<1, true>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, pool, wc, aux);
        hipLaunchKernelGGL(k_resolve_extend, dim3(T8_RESOLVE_BLOCKS), dim3(256), 0, st, sc, pool, aux);
        return;
    }
    hipLaunchKernelGGL((k_extend_tasks<0>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, pool, wc, aux);      // queue 0 -> 1
    hipLaunchKernelGGL((k_extend_tasks<1>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, pool, wc, aux);      // queue 1 -> 0
    hipLaunchKernelGGL((k_extend_tasks<2>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, pool, wc, aux);      // queue 0 -> 1
    hipLaunchKernelGGL((k_extend_tasks<3>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, pool, wc, aux);      // queue 1, to the end
    hipLaunchKernelGGL(k_resolve_extend, dim3(T8_RESOLVE_BLOCKS), dim3(256), 0, st, sc, pool, aux);
}
// One launch for the closest-hit rays of the extend queue AND the visibility rays the previous vertex left in the shadow queue (k_trace_pair), then the task
// rounds and resolve passes of both. auxE / auxS: separate task queues, counters, merge keys and resolve lists. The grid is what the two launches would use if
// it fits the bound, else the bound split by ray count (a visibility ray costs about what a closest-hit ray costs: 0.50 against 0.44 ns on C3). Zeroes
// *shCountPtr at its end.
void launch_trace_pair(const DeviceScene& sc, PathPool pool, const uint* queue, const uint* extCountPtr, uint extCount, ShadowQueue sq, uint* shCountPtr, uint shCount, WaveCounters* wc, TravAux auxE, TravAux auxS, hipStream_t st) {
    const uint rpc = rays_per_chunk(extCount + shCount);
    const uint bound = (auxE.maxBlocks && auxE.maxBlocks < T8_MAX_BLOCKS) ? auxE.maxBlocks : T8_MAX_BLOCKS;
    uint gE = grid_for(extCount, (T8_BLOCK / 64u) * rpc * T8_CHUNKS_PER_WAVE_MIN, bound), gS = grid_for(shCount, (T8_BLOCK / 64u) * rpc * T8_CHUNKS_PER_WAVE_MIN, bound);
    if (gE + gS > bound) {
        uint s = (uint)(((unsigned long long)bound * shCount + (extCount + shCount) / 2u) / (extCount + shCount)); if (s < 1u) s = 1u; if (s > bound - 1u) s = bound - 1u;
        if (s > gS) s = gS;
        uint e = bound - s; if (e > gE) { e = gE; s = (bound - e < gS) ? bound - e : gS; }
        gE = e; gS = s;
    }
    hipLaunchKernelGGL(k_trace_pair, dim3(gE + gS), dim3(T8_BLOCK), 0, st, sc, pool, queue, extCountPtr, sq, shCountPtr, wc, auxE, auxS, rpc, rpc, gE);
    const dim3 tg(2u * T8_TASK_BLOCKS), tb(T8_BLOCK);
    if (extCount <= T8_SHORT_TAIL_BELOW && shCount <= T8_SHORT_TAIL_BELOW) {      // (as launch_extend: two task rounds behind a small launch)
        hipLaunchKernelGGL((k_tasks_pair<0>), tg, tb, 0, st, sc, pool, sq, wc, auxE, auxS);
        hipLaunchKernelGGL((k_tasks_pair<1, true>), tg, tb, 0, st, sc, pool, sq, wc, auxE, auxS);
    } else {
        hipLaunchKernelGGL((k_tasks_pair<0>), tg, tb, 0, st, sc, pool, sq, wc, auxE, auxS);
        hipLaunchKernelGGL((k_tasks_pair<1>), tg, tb, 0, st, sc, pool, sq, wc, auxE, auxS);
        hipLaunchKernelGGL((k_tasks_pair<2>), tg, tb, 0, st, sc, pool, sq, wc, auxE, auxS);
        hipLaunchKernelGGL((k_tasks_pair<3>), tg, tb, 0, st, sc, pool, sq, wc, auxE, auxS);
    }
    hipLaunchKernelGGL(k_resolve_pair, dim3(2u * T8_RESOLVE_BLOCKS), dim3(256), 0, st, sc, pool, sq, auxE, auxS, shCountPtr);
}
// k_classify for a caller in another translation unit (the stable-plane fill pass): classScratch 2 x countIn words, classCount 3 words (zero on entry)
void launch_classify(PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* classScratch, uint* classCount, hipStream_t st) {
    hipLaunchKernelGGL(k_classify, dim3((countIn + 1024u * PT_CLASSIFY_ITEMS - 1u) / (1024u * PT_CLASSIFY_ITEMS)), dim3(1024), 0, st, pool, queueIn, countInPtr, classScratch, classCount);
}
void launch_shade(const PathKernelContext& k, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr, ShadowQueue sq, WaveCounters* wc,
                  uint* classScratch, uint* classCount, hipStream_t st, PathPool outPool) {
    const dim3 g((countIn + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK), b(PT_SHADE_BLOCK);
    // (scratch: 2 x countIn words, free between the extend and the shadow launches; classCount: 3 words, zeroed with the pass's traversal counters)
    if (PT_SHADE_CLASSES && classScratch) {
        hipLaunchKernelGGL(k_classify, dim3((countIn + 1024u * PT_CLASSIFY_ITEMS - 1u) / (1024u * PT_CLASSIFY_ITEMS)), dim3(1024), 0, st, pool, queueIn, countInPtr, classScratch, classCount);
        queueIn = classScratch;
    } else classCount = nullptr;
    // NEE-AT (a local sampling table and / or temporal feedback, pt_set_local_light_sampling) runs its own instantiations: the frames without it keep their
    // kernels unchanged
    const bool neeat = k.sc.lights.LocalSamplingBuffer != nullptr || k.sc.lights.TemporalFeedbackRequired != 0u;
#define PT_LAUNCH_SHADE(MULTI, PKC, CTX) do { if (neeat) hipLaunchKernelGGL((k_shade<MULTI, PKC, true>), g, b, 0, st, CTX, pool, queueIn, countInPtr, queueOut, countOutPtr, sq, wc, classCount, outPool); \
                                              else if (pool.home && !MULTI) hipLaunchKernelGGL((k_shade<false, PKC, false, true>), g, b, 0, st, CTX, pool, queueIn, countInPtr, queueOut, countOutPtr, sq, wc, classCount, outPool); \
                                              else hipLaunchKernelGGL((k_shade<MULTI, PKC, false>), g, b, 0, st, CTX, pool, queueIn, countInPtr, queueOut, countOutPtr, sq, wc, classCount, outPool); } while (0)
    if (k.S.useFp16Types) {          // the reference's default build of its lp types (binary16): same context data, the other instantiation of the shading code
        static_assert(sizeof(PathKernelContextT<true>) == sizeof(PathKernelContext), "the two lp builds share one context layout");
        PathKernelContextT<true> k16; __builtin_memcpy(&k16, &k, sizeof(k16));
        if (sq.group) PT_LAUNCH_SHADE(true, PathKernelContextT<true>, k16); else PT_LAUNCH_SHADE(false, PathKernelContextT<true>, k16);
    } else {
        if (sq.group) PT_LAUNCH_SHADE(true, PathKernelContext, k); else PT_LAUNCH_SHADE(false, PathKernelContext, k);
    }
#undef PT_LAUNCH_SHADE
}
void launch_shadow(const DeviceScene& sc, PathPool pool, ShadowQueue sq, const uint* countPtr, uint count, WaveCounters* wc, bool counters, TravAux aux, hipStream_t st) {
    const uint rpc = rays_per_chunk(count);
    uint g = grid_for(count, (T8_BLOCK / 64u) * rpc * T8_CHUNKS_PER_WAVE_MIN, (aux.maxBlocks && aux.maxBlocks < T8_MAX_BLOCKS) ? aux.maxBlocks : T8_MAX_BLOCKS);
    // (no traversal counters in the grouped mode)
    if (sq.group) hipLaunchKernelGGL((k_shadow<false, true>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, sq, countPtr, wc, aux, rpc);
    else if (counters) hipLaunchKernelGGL((k_shadow<true, false>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, sq, countPtr, wc, aux, rpc);
    else hipLaunchKernelGGL((k_shadow<false, false>), dim3(g), dim3(T8_BLOCK), 0, st, sc, pool, sq, countPtr, wc, aux, rpc);
    hipLaunchKernelGGL((k_shadow_tasks<0>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, sq, wc, aux);
    // (as launch_extend: two rounds for a small launch)
    if (count <= T8_SHORT_TAIL_BELOW) hipLaunchKernelGGL((k_shadow_tasks<1, true>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, sq, wc, aux);
    else {
    hipLaunchKernelGGL((k_shadow_tasks<1>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, sq, wc, aux);
    hipLaunchKernelGGL((k_shadow_tasks<2>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, sq, wc, aux);
    hipLaunchKernelGGL((k_shadow_tasks<3>), dim3(T8_TASK_BLOCKS), dim3(T8_BLOCK), 0, st, sc, sq, wc, aux);
    }
    if (sq.group) hipLaunchKernelGGL((k_resolve_shadow<true>), dim3(T8_RESOLVE_BLOCKS), dim3(256), 0, st, pool, sq, aux);
    else hipLaunchKernelGGL((k_resolve_shadow<false>), dim3(T8_RESOLVE_BLOCKS), dim3(256), 0, st, pool, sq, aux);
    if (sq.group) hipLaunchKernelGGL(k_resolve_nee, dim3(grid_for(count / sq.group, 256, 4096)), dim3(256), 0, st, pool, sq, countPtr);
}
// start of a pass: the batch's PASS_COUNTERS words and the two queue counters the pass refills, zeroed by one launch (three memsets were three launches)
__global__ void __launch_bounds__(64) k_pass_begin(uint* __restrict__ passCounters, uint* __restrict__ nextCount, uint* __restrict__ shadowCount) {
    if (threadIdx.x < PASS_COUNTERS) passCounters[threadIdx.x] = 0u;
    if (threadIdx.x == 32u) *nextCount = 0u;
    if (threadIdx.x == 33u && shadowCount) *shadowCount = 0u;      // (null: a frame of fused traversal launches, whose k_resolve_pair zeroes it)
}
void launch_pass_reset(uint* passCounters, uint* nextCount, uint* shadowCount, hipStream_t st) { static_assert(PASS_COUNTERS <= 32u, "k_pass_begin"); hipLaunchKernelGGL(k_pass_begin, dim3(1), dim3(64), 0, st, passCounters, nextCount, shadowCount); }
__global__ void __launch_bounds__(256) k_uncompact(PathPool in, PathPool out, const uint* __restrict__ countPtr) {
    const uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= *countPtr) return;
    const uint hp = in.home[i];
    out.s0[hp] = in.s0[i]; out.s1[hp] = in.s1[i]; out.s3[hp] = in.s3[i]; out.s4[hp] = in.s4[i];
}
void launch_uncompact(PathPool in, PathPool out, const uint* countPtr, uint count, hipStream_t st) {
    if (count) hipLaunchKernelGGL(k_uncompact, dim3((count + 255u) / 256u), dim3(256), 0, st, in, out, countPtr);
}
void launch_accumulate(PathPool pool, const uint* ownedPixels, uint numOwned, uint spp, float4* accum, uint accumCountBase, uint width, hipStream_t st) {
    hipLaunchKernelGGL(k_accumulate, dim3((numOwned + 255) / 256), dim3(256), 0, st, pool, ownedPixels, numOwned, spp, accum, accumCountBase, width);
}
void launch_trace_probe(const DeviceScene& sc, const float4* rays, uint n, float4* outClosest, uint* outVisible, uint* overflow, hipStream_t st) {
    hipLaunchKernelGGL(k_trace_probe, dim3(grid_for(n, T8_BLOCK, T8_MAX_BLOCKS)), dim3(T8_BLOCK), 0, st, sc, rays, n, outClosest, outVisible, overflow);
}
// the NEE-AT feedback of a list of pixels as (weight bits, candidate, exported depth bits) triples, 12 bytes per pixel: what the ranks of a tile-sharded frame
// exchange between frames — the reservoirs AND the depth the baker's reprojection tests (neeat_reproject): a neighbourhood crosses shard borders, so every rank
// needs every pixel's depth
__global__ void __launch_bounds__(256) k_pack_feedback(const float* __restrict__ fbW, const uint* __restrict__ fbC, const float* __restrict__ depth, const uint* __restrict__ pixels, uint num, uint width, uint* __restrict__ dst) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= num) return;
    const uint px = pixels[i], slot = (px & 0xFFFFu) * width + (px >> 16);
    dst[3u * i] = __float_as_uint(fbW[slot]); dst[3u * i + 1u] = fbC[slot]; dst[3u * i + 2u] = depth ? __float_as_uint(depth[slot]) : 0u;
}
__global__ void __launch_bounds__(256) k_unpack_feedback(float* __restrict__ fbW, uint* __restrict__ fbC, float* __restrict__ depth, const uint* __restrict__ pixels, uint num, uint width, const uint* __restrict__ src) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= num) return;
    const uint px = pixels[i], slot = (px & 0xFFFFu) * width + (px >> 16);
    fbW[slot] = __uint_as_float(src[3u * i]); fbC[slot] = src[3u * i + 1u]; if (depth) depth[slot] = __uint_as_float(src[3u * i + 2u]);
}
void launch_pack_feedback(const float* fbW, const uint* fbC, const float* depth, const uint* pixels, uint num, uint width, uint* dst, hipStream_t st) {
    if (num) hipLaunchKernelGGL(k_pack_feedback, dim3((num + 255) / 256), dim3(256), 0, st, fbW, fbC, depth, pixels, num, width, dst);
}
void launch_unpack_feedback(float* fbW, uint* fbC, float* depth, const uint* pixels, uint num, uint width, const uint* src, hipStream_t st) {
    if (num) hipLaunchKernelGGL(k_unpack_feedback, dim3((num + 255) / 256), dim3(256), 0, st, fbW, fbC, depth, pixels, num, width, src);
}
void launch_pack(const float4* accum, const uint* pixels, uint num, uint width, float4* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_pack, dim3((num + 255) / 256), dim3(256), 0, st, accum, pixels, num, width, dst);
}
void launch_unpack(float4* accum, const uint* pixels, uint num, uint width, const float4* src, hipStream_t st) {
    hipLaunchKernelGGL(k_unpack, dim3((num + 255) / 256), dim3(256), 0, st, accum, pixels, num, width, src);
}
void launch_env_importance(const DeviceScene& sc, uint dim, uint sx, uint sy, float4* out, hipStream_t st) {
    hipLaunchKernelGGL(k_env_importance, dim3((dim * dim + 255) / 256), dim3(256), 0, st, sc, dim, sx, sy, out);
}
// ---- light weights and the sampling-proxy table on the device (ComputeWeights / ComputeProxyCounts / the proxy fill of LightsBaker.hlsl:738-751, 836-948), so
// that a per-frame re-bake of animated emissives (C5) needs no D2H / host loop / H2D. Arithmetic = the host loop it replaces (and the oracle's): weight =
// power^0.8 with the deterministic pow, thresholded; the weight SUM is taken in light order by ONE lane, because a float sum is only reproducible in a fixed
// order (7 k lights: ~10 us; the table limit of 512 k lights: ~1 ms); counts = ceil((budget - N) * w / sum), capped; offsets by an exclusive scan (integers);
// the fill is per proxy.
__global__ void __launch_bounds__(256) k_light_weights(const PolymorphicLightInfo* __restrict__ lights, const PolymorphicLightInfoEx* __restrict__ lightsEx, uint n, float* __restrict__ w, LightFrustumBoost boost) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= n) return;
    PolymorphicLightInfoFull lf; lf.Base = lights[i]; lf.Extended = lightsEx[i];
    float wt = dm_pow(PolymorphicLight_GetPower(lf), 0.8f);
    if (!(wt >= 1e-8f)) wt = 0.f;                         // RTXPT_LIGHTING_MIN_WEIGHT_THRESHOLD
    w[i] = light_importance_frustum_boost(boost, lf, wt);      // ImportanceBooster, frustum term (off unless the host supplied its view-projection matrix)
}
// the sum in light order: the block stages 256 weights at a time in LDS (coalesced loads), lane 0 adds them one after the other
__global__ void __launch_bounds__(256) k_light_weight_sum(const float* __restrict__ w, uint n, float* __restrict__ sum) {
    __shared__ float stage[256];
    float s = 0.f;
    for (uint base = 0; base < n; base += 256u) {
        const uint i = base + threadIdx.x;
        stage[threadIdx.x] = (i < n) ? w[i] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) { const uint m = (n - base < 256u) ? n - base : 256u; for (uint k = 0; k < m; k++) s += stage[k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *sum = s;
}
// usage != null (NEE-AT, last frame left feedback): the weight is pulled towards the number of pixels that asked for the light (pt_neeat.h
// neeat_feedback_light_weight)
__global__ void __launch_bounds__(256) k_light_proxy_counts(const float* __restrict__ w, uint n, const float* __restrict__ sum, uint budget, uint uniform, uint maxPerLight, uint* __restrict__ counts,
                                                            const uint* __restrict__ usage, uint totalMaxFeedbackCount, float globalFeedbackUseWeight) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= n) return;
    float lightWeight = w[i];
    if (usage) lightWeight = neeat_feedback_light_weight(lightWeight, usage[i], *sum, totalMaxFeedbackCount, usage[n], globalFeedbackUseWeight);
    uint cnt = 0;
    if (lightWeight > 0) cnt = uniform ? 1u : (uint)ceilf(((float)(budget - n) * lightWeight) / *sum);
    counts[i] = cnt < maxPerLight ? cnt : maxPerLight;
}
// ComputeWeights of a NEE-AT frame: the baked weight, boosted where the light got brighter than 1.1 x what it weighed last frame (ImportanceBooster,
// LightsBaker.hlsl:137-147)
__global__ void __launch_bounds__(256) k_neeat_boost_weights(const float* __restrict__ base, const float* __restrict__ hist, uint nHist, uint n, float mul, float* __restrict__ cur) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= n) return;
    cur[i] = hist ? neeat_intensity_delta_boost(base[i], i < nHist ? hist[i] : 0.f, mul) : base[i];
}
// ---- NEE-AT feedback passes (pt_neeat.h): one thread per pixel / low-resolution pixel / tile; every pass reads what the previous one wrote and writes only
// its own slot
__global__ void __launch_bounds__(256) k_neeat_prefilter(NeeAtFrame F, const float* __restrict__ snapW, const uint* __restrict__ snapC) {
    uint i = blockIdx.x * 256u + threadIdx.x; if (i >= F.W * F.H) return;
    neeat_prefilter_pixel(F, snapW, snapC, (int)(i % F.W), (int)(i / F.W));
}
// P0's counts: neighbouring pixels mostly ask for the same few lights, and same-address atomics serialise in the L2. Two merges before anything reaches memory
// (the reference merges equal lanes with WaveMatch, LightsBaker.hlsl:1287-1305): the lanes of a wave (an 8 x 8 pixel block) that count the same light become
// one entry, and the sixteen waves of a block (32 x 32 pixels) add their entries into a small LDS hash table that is flushed once — one global atomic per
// distinct light and 32 x 32 pixels.
static const uint NEEAT_P0_TABLE = 512u;      // LDS hash slots per block (a full table falls back to a global atomic per entry)
__global__ void __launch_bounds__(1024) k_neeat_p0(NeeAtFrame F, uint totalThreads) {
    __shared__ uint hKey[NEEAT_P0_TABLE]; __shared__ uint hCnt[NEEAT_P0_TABLE];
    const uint NONE = 0xFFFFFFFFu, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint k = threadIdx.x; k < NEEAT_P0_TABLE; k += 1024u) { hKey[k] = NONE; hCnt[k] = 0u; }
    // the dispatch's threads beyond the frame count as "no valid feedback"
    if (blockIdx.x == 0u && threadIdx.x == 0u && totalThreads > F.W * F.H) atomicAdd(&F.perLightCounters[F.totalLightCount], totalThreads - F.W * F.H);
    __syncthreads();
    const uint regionsX = (F.W + 31u) / 32u;
    const uint px = (blockIdx.x % regionsX) * 32u + (wave & 3u) * 8u + (lane & 7u), py = (blockIdx.x / regionsX) * 32u + (wave >> 2) * 8u + (lane >> 3);
    const uint slot = (px < F.W && py < F.H) ? neeat_p0_pixel(F, px, py) : NONE;
    unsigned long long active = __builtin_amdgcn_ballot_w64(slot != NONE);
    while (active) {
        const uint leader = (uint)__builtin_ctzll(active);
        const uint v = (uint)__builtin_amdgcn_readlane((int)slot, (int)leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(slot == v);
        if (lane == leader) {
            const uint n = (uint)__popcll(same);
            uint h = (v * 2654435761u) >> 23; bool placed = false;                 // 9-bit multiplicative hash, linear probing
            for (uint probe = 0; probe < 16u && !placed; probe++, h = (h + 1u) & (NEEAT_P0_TABLE - 1u)) {
                const uint prev = atomicCAS(&hKey[h], NONE, v);
                if (prev == NONE || prev == v) { atomicAdd(&hCnt[h], n); placed = true; }
            }
            if (!placed) atomicAdd(&F.perLightCounters[v], n);
        }
        active &= ~same;
    }
    __syncthreads();
    for (uint k = threadIdx.x; k < NEEAT_P0_TABLE; k += 1024u) if (hKey[k] != NONE) atomicAdd(&F.perLightCounters[hKey[k]], hCnt[k]);
}
__global__ void __launch_bounds__(256) k_neeat_p1a(NeeAtFrame F) { uint i = blockIdx.x * 256u + threadIdx.x; if (i < F.BW * F.BH) neeat_p1a_pixel(F, i % F.BW, i / F.BW); }
__global__ void __launch_bounds__(256) k_neeat_p1b(NeeAtFrame F) { uint i = blockIdx.x * 256u + threadIdx.x; if (i < F.W * F.H) neeat_p1b_pixel(F, i % F.W, i / F.W); }
__global__ void __launch_bounds__(64) k_neeat_p2(NeeAtFrame F) { uint i = blockIdx.x * 64u + threadIdx.x; if (i < F.tilesX * F.tilesY) neeat_fill_tile(F, i % F.tilesX, i / F.tilesX); }
// P3: one 64-lane block per tile. The 128 light indices are sorted in LDS with a bitonic network (compare-exchange pairs as in MiniEngine's Bitonic32PreSortCS,
// which the reference cites: element `hi` = lane with a one inserted at bit j, partner = hi ^ (first step of a merge ? k - 1 : j)); then every entry finds the
// ends of its run.
__global__ void __launch_bounds__(64) k_neeat_p3(NeeAtFrame F) {
    __shared__ uint key[RTXPT_LIGHTING_LOCAL_PROXY_COUNT];
    const uint N = RTXPT_LIGHTING_LOCAL_PROXY_COUNT, t = threadIdx.x;
    uint* tile = F.local + (size_t)blockIdx.x * N;
    key[t] = UnpackMiniListLight(tile[t]); key[t + N / 2] = UnpackMiniListLight(tile[t + N / 2]);
    __syncthreads();
    for (uint k = 2; k <= N; k <<= 1) for (uint j = k / 2; j > 0; j >>= 1) {
        const uint mask = j - 1u, hi = ((t & ~mask) << 1) | (t & mask) | j, lo = hi ^ (k == 2u * j ? k - 1u : j);
        const uint a = key[lo], b = key[hi];
        if (a > b) { key[lo] = b; key[hi] = a; }
        __syncthreads();
    }
    for (uint e = t; e < N; e += N / 2) {
        const uint v = key[e]; uint l = e, r = e;
        while (l > 0u && key[l - 1u] == v) l--;
        while (r + 1u < N && key[r + 1u] == v) r++;
        tile[e] = PackMiniListLightAndCount(v, r - l + 1u);
    }
}
__global__ void __launch_bounds__(256) k_neeat_clear(NeeAtFrame F) { uint i = blockIdx.x * 256u + threadIdx.x; if (i < F.W * F.H) neeat_clear_pixel(F, i % F.W, i / F.W); }
// one thread per proxy slot: the light whose [offset, offset + count) range holds the slot (binary search over the scanned offsets)
__global__ void __launch_bounds__(256) k_light_proxy_fill(const uint* __restrict__ counts, const uint* __restrict__ offsets, uint n, uint* __restrict__ proxies, uint capacity) {
    const uint p = blockIdx.x * 256u + threadIdx.x;
    const uint total = offsets[n - 1u] + counts[n - 1u];
    if (p >= total || p >= capacity) return;
    // last light with offset <= p (lights without proxies share their successor's offset: skipped by taking the last)
    uint lo = 0u, hi = n;
    while (hi - lo > 1u) { uint mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid; else hi = mid; }
    proxies[p] = lo;
}
void launch_light_weights(const PolymorphicLightInfo* lights, const PolymorphicLightInfoEx* lightsEx, uint n, float* w, const LightFrustumBoost& boost, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_light_weights, dim3((n + 255) / 256), dim3(256), 0, st, lights, lightsEx, n, w, boost);
}
void launch_light_proxy_counts(const float* w, uint n, float* sum, uint budget, bool uniform, uint maxPerLight, uint* counts, const uint* usage, uint totalMaxFeedbackCount, float globalFeedbackUseWeight, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_light_weight_sum, dim3(1), dim3(256), 0, st, w, n, sum);
    hipLaunchKernelGGL(k_light_proxy_counts, dim3((n + 255) / 256), dim3(256), 0, st, w, n, sum, budget, uniform ? 1u : 0u, maxPerLight, counts, usage, totalMaxFeedbackCount, globalFeedbackUseWeight);
}
void launch_neeat_boost_weights(const float* base, const float* hist, uint nHist, uint n, float mul, float* cur, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_neeat_boost_weights, dim3((n + 255) / 256), dim3(256), 0, st, base, hist, nHist, n, mul, cur);
}
// UpdateBegin's feedback half: PreFilter from a snapshot of the reservoirs (snapW / snapC: W x H scratch), then P0's counts. The counters are expected zeroed
// (totalLightCount + 1 words).
void launch_neeat_begin(const NeeAtFrame& F, float* snapW, uint* snapC, bool preFilter, uint totalThreads, hipStream_t st) {
    const uint px = F.W * F.H;
    if (preFilter) {
        (void)hipMemcpyAsync(snapW, F.fbW, 4ull * px, hipMemcpyDeviceToDevice, st); (void)hipMemcpyAsync(snapC, F.fbC, 4ull * px, hipMemcpyDeviceToDevice, st);
        hipLaunchKernelGGL(k_neeat_prefilter, dim3((px + 255) / 256), dim3(256), 0, st, F, snapW, snapC);
        if (getenv("MI355PT_NEEAT_DEBUG")) fprintf(stderr, "[neeat] prefilter: %s\n", hipGetErrorString(hipStreamSynchronize(st)));
    }
    hipLaunchKernelGGL(k_neeat_p0, dim3(((F.W + 31) / 32) * ((F.H + 31) / 32)), dim3(1024), 0, st, F, totalThreads);
    if (getenv("MI355PT_NEEAT_DEBUG")) fprintf(stderr, "[neeat] p0: %s\n", hipGetErrorString(hipStreamSynchronize(st)));
}
// UpdateEnd: blended reservoirs, one candidate per pixel, the tiles (fill, sort + count), and the reservoirs cleared down to what the next frame keeps
void launch_neeat_end(const NeeAtFrame& F, hipStream_t st) {
    const uint px = F.W * F.H, bpx = F.BW * F.BH, tiles = F.tilesX * F.tilesY;
    const bool dbg = getenv("MI355PT_NEEAT_DEBUG") != nullptr;
#define NEEAT_DBG(what) do { if (dbg) { hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[neeat] %s: %s\n", what, hipGetErrorString(e)); } } while (0)
    hipLaunchKernelGGL(k_neeat_p1a, dim3((bpx + 255) / 256), dim3(256), 0, st, F); NEEAT_DBG("p1a");
    hipLaunchKernelGGL(k_neeat_p1b, dim3((px + 255) / 256), dim3(256), 0, st, F); NEEAT_DBG("p1b");
    hipLaunchKernelGGL(k_neeat_p2, dim3((tiles + 63) / 64), dim3(64), 0, st, F); NEEAT_DBG("p2");
    hipLaunchKernelGGL(k_neeat_p3, dim3(tiles), dim3(64), 0, st, F); NEEAT_DBG("p3");
    hipLaunchKernelGGL(k_neeat_clear, dim3((px + 255) / 256), dim3(256), 0, st, F); NEEAT_DBG("clear");
#undef NEEAT_DBG
}
void launch_light_proxy_fill(const uint* counts, const uint* offsets, uint n, uint* proxies, uint capacity, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_light_proxy_fill, dim3((capacity + 255) / 256), dim3(256), 0, st, counts, offsets, n, proxies, capacity);
}
void launch_bake_emissive(const DeviceScene& sc, const uint* subInstList, const uint* subInstTriOffset, uint numEmissiveSubInst, uint totalTris, uint lightBase,
                          PolymorphicLightInfo* lights, PolymorphicLightInfoEx* lightsEx, hipStream_t st) {
    if (!totalTris) return;
    hipLaunchKernelGGL(k_bake_emissive, dim3((totalTris + 255) / 256), dim3(256), 0, st, sc, subInstList, subInstTriOffset, numEmissiveSubInst, totalTris, lightBase, lights, lightsEx);
}
#ifdef MI355PT_TEST_HOOKS
void launch_probe(const PathKernelContext& k, int kind, const void* dIn, void* dOut, uint n, hipStream_t st) {
    hipLaunchKernelGGL(k_probe, dim3((n + 63) / 64), dim3(64), 0, st, k, kind, (const float*)dIn, (float*)dOut, n);
}
#endif

} // namespace ptk
