// mi355pt — cooperative BVH8 traversal for wave64: a wave carries 16 rays, each owned by a QUAD of lanes; lane q of a quad tests
// children 2q and 2q+1 of the current 128-byte node (one cache-line lookup per node per ray), the hit children are ranked by entry
// distance with quad-permute DPP compares, the nearest is followed directly and the rest go to the quad's stack in LDS (deep entries
// spill to a global-memory tail). Leaves hand their triangles (up to 4 by default: one round; up to 8: two) to the 4 lanes. Quads refill independently from
// the wave's 64-ray chunk (persistent threads), so a long ray never holds 63 idle lanes hostage. Replaces RayQuery::TraceRayInline /
// the DXR any-hit visibility query (PathTracerBridgeDonut.hlsli:993-1055). Results are traversal-order free (min t, ties to the lower
// primitive id).
//
// How the shape was arrived at on MI355X (profiles/r01a_*, r01b_*):
//  * one ray per lane over BVH2 was bound by divergent 16-byte gathers through the per-CU texture-address path (33 % L2 misses but only
//    ~0.4 TB/s of HBM traffic): 4 line lookups per lane per node. Cooperative 128 B nodes cut line lookups per ray by ~6-9x.
//  * 8 lanes per ray / 1 child per lane then turned out VALU-issue bound: at 6 waves per SIMD SQ_ACTIVE_INST_VALU covers 93 % of the
//    kernel time, and halving the occupancy (profiles/r01b_occupancy_experiment.txt) showed the latency side saturating exactly there.
//    Every per-ray scalar (addresses, stack, control) is replicated over the lanes of its group, so the cure is fewer lanes per ray:
//    4 lanes x 2 children per lane keeps the one-line-per-node access pattern and halves the replicated work.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_scene.h"

namespace ptk {

struct Traverse8Counters { uint nodeVisits, triTests, leafVisits, iters, leafBlocks; uint ev[8]; unsigned long long cyc[4]; unsigned long long* rayIterHist; uint* longRayCount; float* longRays; };

#define T8_EVENT(k, cond) do { if (COUNT) { unsigned long long m_ = t8_ballot(cond); if (m_ && lane == (uint)__ffsll((long long)m_) - 1u) ctr.ev[k]++; } } while (0)
#ifndef T8_PARK_RCP
#define T8_PARK_RCP 1           // (pairs build) 1: the ray's three reciprocals are formed once by the lane that fetches it and parked with it in LDS, 0: at every refill
#endif
static const uint T8_RAY_STRIDE = (PT_T8_LANES == 2 && T8_PARK_RCP) ? 13 : 9, T8_TASK_STRIDE = (PT_T8_LANES == 2 && T8_PARK_RCP) ? 15 : 11;                                              // per-wave LDS parking lot for a chunk's rays / tasks (odd strides)
static const uint T8_RAYBUF_WORDS = (T8_BLOCK / 64u) * T8_CHUNK * T8_RAY_STRIDE, T8_TASKBUF_WORDS = (T8_BLOCK / 64u) * T8_CHUNK * T8_TASK_STRIDE;
static const uint T8_LEAF_ROUNDS = (BVH_MAX_LEAF + T8_LANES - 1u) / T8_LANES;

#ifndef T8_EXTEND_MIN_BLOCKS
#define T8_EXTEND_MIN_BLOCKS (PT_T8_LANES == 2 ? 7 : 8)   // waves per SIMD the register allocator must leave room for in k_extend (64 VGPRs, 2 spilled outside the loop): 6 -> 7 -> 8 waves were
                                  // 1161 -> 1187 -> 1239 Mrays/s in round 2 (profiles/r02n_occupancy_ab.txt). At 8 waves the loop is VALU-issue bound (round 3: extra v_nop
                                  // slots lengthen it one for one, profiles/r03i_valu_bound_probe.txt): from here on instructions per ray count, not waves in flight
#endif
#ifndef T8_SHADOW_MIN_WAVES
#define T8_SHADOW_MIN_WAVES (PT_T8_LANES == 2 ? 7 : 8)    // the same for k_shadow
#endif
#ifndef T8_FAST_INNER
#define T8_FAST_INNER 1          // near/far planes by byte permute, float scales from the node tail, quad hit count by DPP adds
#endif
#ifndef T8_TAIL_ITERS
#define T8_TAIL_ITERS 32         // loop iterations a wave keeps going after its last chunk before it splits what is still in flight into tasks
#endif
#ifndef T8_TAIL_ITERS_TASKS
#define T8_TAIL_ITERS_TASKS 8    // the same for task rounds: sub-trees are short, and every round of every launch pays this tail once
#endif
#ifndef T8_ANYHIT_UNORDERED
#define T8_ANYHIT_UNORDERED 1     // 1: occlusion queries number a node's hit children by child index instead of ranking them by entry distance
#endif
#ifndef T8_LEAF_QUEUE
#define T8_LEAF_QUEUE (PT_T8_LANES == 2 ? 3 : 2)        // postponed leaves a ray may hold (1..3) before it has to wait for the wave's next leaf block (A/B: within noise on extend, -4 % on shadow)
#endif
#ifndef T8_LEAF_BATCH
#define T8_LEAF_BATCH (PT_T8_LANES == 2 ? 20u : 8u)        // quads (of 16) that must hold a postponed leaf before the wave runs the leaf block (17 = only when a quad is blocked; A/B in profiles/)
#endif

// the ray's reciprocal direction is the correctly rounded one of the hit definition (pt_scene.h tri_box_accepts): inner nodes and the triangle's own
// box are then tested with the same arithmetic, which is what makes the closest hit independent of the tree (three divisions per ray, not per node)
__device__ __forceinline__ float t8_rcp_dir(float d) { return ray_safe_rcp(d); }
// tri_box_accepts (pt_scene.h) with the hardware's min/max (v_min3/v_max3 instead of compare + select pairs: 41 instead of 71 VALU instructions). The
// operands are finite here — Moeller-Trumbore has already accepted the triangle, so neither the ray nor the vertices hold a NaN — and on finite
// operands minNum / maxNum differ from `(a < b) ? a : b` only in the sign of a zero, which no comparison below can see: same boolean, bit for bit.
__device__ __forceinline__ bool t8_tri_box_accepts(const TriRecord& tr, float3 o, float ix, float iy, float iz, float t) {
    const float3 q1 = tr.v0 + tr.e1, q2 = tr.v0 + tr.e2;
    const float mnx = fminf(tr.v0.x, fminf(q1.x, q2.x)) - tr.pad, mny = fminf(tr.v0.y, fminf(q1.y, q2.y)) - tr.pad, mnz = fminf(tr.v0.z, fminf(q1.z, q2.z)) - tr.pad;
    const float mxx = fmaxf(tr.v0.x, fmaxf(q1.x, q2.x)) + tr.pad, mxy = fmaxf(tr.v0.y, fmaxf(q1.y, q2.y)) + tr.pad, mxz = fmaxf(tr.v0.z, fmaxf(q1.z, q2.z)) + tr.pad;
    const float ax = (mnx - o.x) * ix, bx = (mxx - o.x) * ix, ay = (mny - o.y) * iy, by = (mxy - o.y) * iy, az = (mnz - o.z) * iz, bz = (mxz - o.z) * iz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return (tn <= t) && (t <= tf);
}
__device__ __forceinline__ unsigned long long t8_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }   // v_cmp straight into an SGPR pair
__device__ __forceinline__ uint quad_bits(unsigned long long m, uint gl) { return (uint)(m >> gl) & 0xFu; }
// In-quad data exchange with DPP quad permutes only — no LDS crossbar (ds_bpermute) latency on the critical path.
#define DPP_QP_XOR1 0xB1
#define DPP_QP_XOR2 0x4E
#define DPP_QP_XOR3 0x1B
template <int CTRL> __device__ __forceinline__ uint dpp_u(uint v) { return (uint)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }

struct __attribute__((packed, aligned(8))) Bvh8ChildPair { uint refA, q0A, q1A, refB, q0B, q1B; };   // this lane's two 12 B child slots (24 B, 8-byte aligned)

// FIXED_RANGE: every ray of the launch has tmin = 0, tmax = kMaxRayTravel (extend rays), whatever fetch() reports.
// TASKS: the work items are sub-trees of rays (TravTask) instead of whole rays: fetch() also reports the start node and the ray's best hit so far.
// CAN_SPLIT: straggler handling. One ray is a serial chain on one quad, and C3 has ~300 rays (of 10^8) that need 10^3..10^4 iterations (rays
//   running along the street inside the tree crowns): alone they kept a launch alive for up to 20 ms and cost 15-25 % of the frame and half of
//   the 8-GPU scaling. So a wave that has been out of fresh rays for T8_TAIL_ITERS iterations stops: every ray still in flight is cut into its
//   pending sub-trees (node slot, postponed leaves, stack entries), which go to a task queue together with the best hit so far (publish()),
//   and a follow-up launch spreads them over the whole GPU. The result is the minimum over all sub-trees, so nothing changes in the image.
// Src: uint fetch(uint i, float3& o, float3& d, float& tmin, float& tmax, uint& startRef, float& bestT0, uint& bestPrim0) -> user tag (e.g. path
//        index); called once per item by one lane; tmin >= 0. Rays: startRef = 0, bestT0 = tmax, bestPrim0 = ~0.
// Dst: void commit(uint tag, const HitInfo& h) ; called by ONE lane of the quad (closest: best hit or prim == ~0; any-hit: prim != ~0 when occluded)
// Pub: void publish(uint tag, float bestT, uint bestPrim) ; called by one lane for every ray that is split (CAN_SPLIT only)
//
#if PT_T8_LANES == 4
// (DEFER belongs to the two-lane build, pt_traverse8p.h; accepted and ignored here so that the kernels read the same for both)
template <bool ANYHIT, bool COUNT, bool FIXED_RANGE, bool TASKS, bool CAN_SPLIT, bool DEFER = false, class Src, class Dst, class Pub>
__device__ __forceinline__ void traverse8_persistent(const DeviceScene& sc, uint count, uint raysPerChunk, uint2* stackBase, uint* rayBufBase, float2* mineUV, Src fetch, Dst commit, Pub publish, TravTaskOut taskOut,
                                                     Traverse8Counters& ctr, uint* overflowFlag) {
    const uint RAY_STRIDE = TASKS ? T8_TASK_STRIDE : T8_RAY_STRIDE;
    const uint lane = threadIdx.x & 63u, q = lane & 3u, gl = lane & ~3u;
    const uint grp = threadIdx.x >> 2;
    uint2* stack = stackBase + grp * BVH8_STACK_STRIDE;
    const uint wavesPerBlock = T8_BLOCK / 64u;
    const uint waveId = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6), numWaves = gridDim.x * wavesPerBlock;
    const char* nodesBase = reinterpret_cast<const char*>(sc.nodes8);
    const char* trisBase = reinterpret_cast<const char*>(sc.tris);
    // 32-bit byte offsets from the (uniform, SGPR) buffer bases: global_load takes saddr + 32-bit voffset, which saves 64-bit address
    // arithmetic and a VGPR pair per pointer (limits: 32 M BVH8 nodes, 89 M triangles per scene — checked at build time)
    const uint laneChildOff = 16u + 24u * q, laneTriOff = 48u * q;
    const uint INF_BITS = 0x7F800000u;

    // chunk cursor: wave-uniform, kept in SGPRs (readfirstlane); the first refill advances to chunk `waveId`. Chunks are dealt round robin over all waves
    // of the launch: giving every XCD (blocks b with b % 8 == x) one contiguous eighth of the queue, so that its private L2 sees one band of the frame,
    // was 3 % SLOWER on C3 (profiles/r02k_isa_experiments.txt): the bands differ in cost and the rays of later bounces are incoherent anyway.
    const uint numWavesU = (uint)__builtin_amdgcn_readfirstlane((int)numWaves);
    uint chunk = (uint)__builtin_amdgcn_readfirstlane((int)(waveId - numWaves)), chunkPos = 0u, chunkEnd = 0u;
    // rays a wave takes per chunk (16 .. T8_CHUNK, wave-uniform): a launch too small to fill the GPU with 64-ray chunks hands every wave fewer rays, down to one per
    // quad, so that all of its rays are in flight at once instead of four in a row per quad (a 64-ray chunk is ~0.2 ms of dependent fetches however small the launch)
    const uint rpc = (uint)__builtin_amdgcn_readfirstlane((int)raysPerChunk);
    bool exhausted = (waveId * rpc >= count) || !sc.rootIsValid;
    uint* rayBuf = rayBufBase + (threadIdx.x >> 6) * (T8_CHUNK * RAY_STRIDE);
    uint tailIters = 0u; bool waveDry = false;                // wave-uniform: a refill found the chunk list empty / iterations since then (CAN_SPLIT)
    if (!sc.rootIsValid && waveId == 0 && count) {            // empty scene: every ray misses
        for (uint i = lane; i < count; i += 64u) { float3 o, d; float a, b, bt; uint sr, bp; uint tag = fetch(i, o, d, a, b, sr, bt, bp); HitInfo h; h.t = b; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); }
    }
    bool active = false;
    float3 o = make_float3(0.f), d = make_float3(0.f);
    float ix = 0.f, iy = 0.f, iz = 0.f;
    uint selN = 0u, selF = 0u;                                // (T8_FAST_INNER) byte selectors of the near / far planes for this ray's direction signs
    float tmin = 0.f, tmax = FIXED_RANGE ? kMaxRayTravel : 0.f;
    float bestT = 0.f; uint bestPrim = 0xFFFFFFFFu;           // quad-uniform closest hit so far
    uint minePrim = 0xFFFFFFFFu;                              // the best hit THIS lane found; its barycentrics wait in LDS (mineUV) and never travel between lanes
    // Two work slots per ray so that one loop iteration advances BOTH an inner node and a leaf: `cur` is the node being descended,
    // `pend` a postponed leaf; the postponed leaf is tested while the next inner node is already being intersected. The result does
    // not depend on the visiting order (min t, ties to the lower primitive id).
    uint cur = BVH_EMPTY, pend = BVH_EMPTY, sp = 0, tag = 0;
    uint rayIters = 0;                                        // (counters build) iterations spent on the current ray
    float taskT0 = 0.f; uint taskPrim0 = 0xFFFFFFFFu;         // (TASKS) the ray's best hit when the task was fetched
    uint pend1 = BVH_EMPTY, pend2 = BVH_EMPTY;          // younger postponed leaves (T8_LEAF_QUEUE > 1): pend is tested first

    // LDS for the top BVH8_STACK entries, global memory behind them. The tail store is non-temporal on purpose: it keeps the compiler from
    // merging the two paths into one flat_store through a generic pointer (seen in the ISA), which would put every push on the slow flat path.
    auto stackStore = [&](uint idx, uint ref, uint tbits) {
        if (idx < BVH8_STACK) stack[idx] = make_uint2(ref, tbits);
        else {
            unsigned long long* tail = reinterpret_cast<unsigned long long*>(sc.travSpill + ((size_t)(blockIdx.x * T8_GROUPS_PER_BLOCK + grp) * T8_SPILL_DEPTH + (idx - BVH8_STACK)));
            __builtin_nontemporal_store(((unsigned long long)tbits << 32) | ref, tail);
        }
    };

    // One back edge, one exit (`stop` is wave-uniform). With a `continue` and two `break`s the compiler kept two copies of the loop-carried ray state and
    // moved one into the other at the top and at the bottom of every iteration (20 of ~340 VALU instructions, tools/isa_stats.sh); the single-exit form
    // also needs 2 VGPRs fewer (78), and is 2 % (k_extend) / 7 % (k_shadow) faster. See DESIGN.md 4 "What the ISA experiments of round 2 say".
    bool splitNow = false, stop = false;
    while (!stop) {
        unsigned long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
        if (COUNT) tc0 = __builtin_readcyclecounter();
        // ---- refill idle quads from the wave's current chunk
        bool need = !active && !exhausted;
        unsigned long long needMask = t8_ballot(need && q == 0u);
        if (needMask) {
            T8_EVENT(0, true);
            if (chunkPos >= chunkEnd) {
                T8_EVENT(1, true);
                // next 64-ray chunk: every lane fetches one ray and parks it in LDS, so the two dependent global loads of a fetch are paid once
                // per chunk by the whole wave instead of at every refill event
                chunk += numWavesU;
                chunkPos = chunk * rpc; chunkEnd = (chunkPos + rpc < count) ? chunkPos + rpc : count;
                if (chunkPos >= count) { chunkPos = chunkEnd = count; }
                if (chunkPos + lane < chunkEnd) {
                    float3 ro, rd; float rtmin, rtmax, rbestT; uint rstart, rbestPrim;
                    uint rtag = fetch(chunkPos + lane, ro, rd, rtmin, rtmax, rstart, rbestT, rbestPrim);
                    uint* slot = rayBuf + lane * RAY_STRIDE;
                    slot[0] = __float_as_uint(ro.x); slot[1] = __float_as_uint(ro.y); slot[2] = __float_as_uint(ro.z);
                    slot[3] = __float_as_uint(rd.x); slot[4] = __float_as_uint(rd.y); slot[5] = __float_as_uint(rd.z);
                    slot[6] = rtag; slot[7] = __float_as_uint(TASKS ? rtmax : rtmin); slot[8] = TASKS ? __float_as_uint(rbestT) : __float_as_uint(rtmax);
                    if (TASKS) { slot[9] = rbestPrim; slot[10] = rstart; }      // (tasks: tmin is 0)
                }
            }
            uint avail = chunkEnd - chunkPos;
            if (avail == 0u) { if (need) exhausted = true; waveDry = true; }
            else {
                uint rank = (uint)__popcll(needMask & ((1ull << gl) - 1ull));        // rank of my quad among the needing quads
                uint n = (uint)__popcll(needMask);
                if (need && rank < avail) {
                    const uint* slot = rayBuf + (((chunkPos - chunk * rpc) + rank) * RAY_STRIDE);
                    o = make_float3(__uint_as_float(slot[0]), __uint_as_float(slot[1]), __uint_as_float(slot[2]));
                    d = make_float3(__uint_as_float(slot[3]), __uint_as_float(slot[4]), __uint_as_float(slot[5]));
                    tag = slot[6];
                    {   // three correctly rounded divisions per ray: lane q of the quad does component q, quad-permute broadcasts hand the results round
                        const float mine = t8_rcp_dir(q == 0u ? d.x : (q == 1u ? d.y : d.z));
                        ix = __uint_as_float((uint)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(mine), 0x00, 0xF, 0xF, true));      // quad_perm [0,0,0,0]
                        iy = __uint_as_float((uint)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(mine), 0x55, 0xF, 0xF, true));      // quad_perm [1,1,1,1]
                        iz = __uint_as_float((uint)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(mine), 0xAA, 0xF, 0xF, true));      // quad_perm [2,2,2,2]
                    }
#if T8_FAST_INNER
                    {   // child bytes: q0 = lo.x lo.y lo.z hi.x (selector values 0..3), q1 = hi.y hi.z (4, 5)
                        const uint nxb = ix < 0.f ? 3u : 0u, fxb = ix < 0.f ? 0u : 3u, nyb = iy < 0.f ? 4u : 1u, fyb = iy < 0.f ? 1u : 4u, nzb = iz < 0.f ? 5u : 2u, fzb = iz < 0.f ? 2u : 5u;
                        selN = nxb | (nyb << 8) | (nzb << 16) | (fxb << 24); selF = fyb | (fzb << 8);
                    }
#endif
                    if (TASKS) {
                        if (!FIXED_RANGE) tmax = __uint_as_float(slot[7]);
                        bestT = taskT0 = __uint_as_float(slot[8]); bestPrim = taskPrim0 = slot[9]; cur = slot[10];
                    } else {
                        if (!FIXED_RANGE) { tmin = __uint_as_float(slot[7]); tmax = __uint_as_float(slot[8]); }      // FIXED_RANGE: [0, kMaxRayTravel] stays a compile-time constant (2 VGPRs)
                        bestT = tmax; bestPrim = 0xFFFFFFFFu; cur = 0u;
                    }
                    minePrim = 0xFFFFFFFFu; rayIters = 0u;
                    pend = BVH_EMPTY; pend1 = BVH_EMPTY; pend2 = BVH_EMPTY; sp = 0u; active = true;
                }
                chunkPos = (uint)__builtin_amdgcn_readfirstlane((int)(chunkPos + ((n < avail) ? n : avail)));
            }
        }
        bool run = t8_ballot(active) != 0ull;                       // wave-uniform
        if (!run) { if (t8_ballot(!exhausted) == 0ull) stop = true; }

        // ---- straggler splitting (after the loop, see below): out of fresh work for T8_TAIL_ITERS iterations -> stop and hand over what is in flight
        else if (CAN_SPLIT) {
            if (waveDry) tailIters++;
            if (tailIters > (uint)(TASKS ? T8_TAIL_ITERS_TASKS : T8_TAIL_ITERS)) { splitNow = true; stop = true; run = false; }
        }
        if (run) {

#ifdef T8_PROBE_VNOPS          // developer probe: T8_PROBE_VNOPS extra VALU issue slots per wave iteration (is the loop VALU-issue bound?)
#pragma unroll
        for (int k_ = 0; k_ < T8_PROBE_VNOPS; k_++) asm volatile("v_nop");
#endif
        if (COUNT && lane == 0u) ctr.iters++;
        if (COUNT && active) rayIters++;
        if (COUNT) tc1 = __builtin_readcyclecounter();
        const bool inner = active && !(cur & BVH_LEAF_BIT);
        // The leaf block is run when T8_LEAF_BATCH quads have a leaf waiting, or when some quad cannot advance without it (its node
        // slot holds a second leaf, or is empty with an empty stack). Meanwhile the descent continues against a slightly stale closest distance.
        const bool leafReady = active && (pend != BVH_EMPTY);
        // blocked: the node slot holds a leaf and the queue is full, or the descent is finished (empty node slot, empty stack) and only leaves remain
        const bool queueFull = (T8_LEAF_QUEUE == 1) ? true : ((T8_LEAF_QUEUE == 2) ? (pend1 != BVH_EMPTY) : (pend2 != BVH_EMPTY));
        const bool leafBlocked = leafReady && (cur & BVH_LEAF_BIT) && (cur == BVH_EMPTY || queueFull);
        const bool runLeaves = ((uint)__popcll(t8_ballot(leafReady && q == 0u)) >= (uint)T8_LEAF_BATCH) || (t8_ballot(leafBlocked) != 0ull);
        const bool leaf = leafReady && runLeaves;
        if (COUNT && leaf && q == 0u) ctr.leafVisits++;
        T8_EVENT(2, inner); T8_EVENT(3, leaf);

        // ---- inner node: lane q tests children 2q and 2q+1
        if (inner) {
            const uint nodeOff = cur * 128u;
            const u32x4 hdr = *reinterpret_cast<const u32x4*>(nodesBase + nodeOff);
            const Bvh8ChildPair ch = *reinterpret_cast<const Bvh8ChildPair*>(nodesBase + (nodeOff + laneChildOff));
            if (COUNT && q == 0u) ctr.nodeVisits++;
            const float nx = __uint_as_float(hdr.x), ny = __uint_as_float(hdr.y), nz = __uint_as_float(hdr.z);
#if T8_FAST_INNER
            // scales as floats from the node's last 16 bytes (one more 16 B load per quad, 6 VALU decode instructions fewer)
            const f32x4 scl = *reinterpret_cast<const f32x4*>(nodesBase + (nodeOff + 112u));
            const float sx = scl.x, sy = scl.y, sz = scl.z;
            // the ray's direction signs pick the near / far plane of every axis up front (two byte permutes per child instead of six min/max):
            // N = near.x | near.y | near.z | far.x, F = far.y | far.z
            auto slab = [&](uint q0, uint q1, float& tn, float& tf) {
                const uint N = __builtin_amdgcn_perm(q1, q0, selN), F = __builtin_amdgcn_perm(q1, q0, selF);
                f32x2 px = __builtin_elementwise_fma((f32x2){(float)(N & 0xFFu), (float)(N >> 24)}, (f32x2){sx, sx}, (f32x2){nx, nx});
                f32x2 py = __builtin_elementwise_fma((f32x2){(float)((N >> 8) & 0xFFu), (float)(F & 0xFFu)}, (f32x2){sy, sy}, (f32x2){ny, ny});
                f32x2 pz = __builtin_elementwise_fma((f32x2){(float)((N >> 16) & 0xFFu), (float)((F >> 8) & 0xFFu)}, (f32x2){sz, sz}, (f32x2){nz, nz});
                f32x2 tx = (px - (f32x2){o.x, o.x}) * (f32x2){ix, ix}, ty = (py - (f32x2){o.y, o.y}) * (f32x2){iy, iy}, tz = (pz - (f32x2){o.z, o.z}) * (f32x2){iz, iz};
                tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
                tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, bestT));
            };
#else
            const float sx = __uint_as_float((hdr.w << 23) & INF_BITS), sy = __uint_as_float((hdr.w << 15) & INF_BITS), sz = __uint_as_float((hdr.w << 7) & INF_BITS);
            // {lo, hi} pairs per axis: plane = fma(code, scale, origin) (the builder verified conservativeness with this exact expression)
            auto slab = [&](uint q0, uint q1, float& tn, float& tf) {
                f32x2 px = __builtin_elementwise_fma((f32x2){(float)(q0 & 0xFFu), (float)(q0 >> 24)}, (f32x2){sx, sx}, (f32x2){nx, nx});
                f32x2 py = __builtin_elementwise_fma((f32x2){(float)((q0 >> 8) & 0xFFu), (float)(q1 & 0xFFu)}, (f32x2){sy, sy}, (f32x2){ny, ny});
                f32x2 pz = __builtin_elementwise_fma((f32x2){(float)((q0 >> 16) & 0xFFu), (float)((q1 >> 8) & 0xFFu)}, (f32x2){sz, sz}, (f32x2){nz, nz});
                f32x2 tx = (px - (f32x2){o.x, o.x}) * (f32x2){ix, ix}, ty = (py - (f32x2){o.y, o.y}) * (f32x2){iy, iy}, tz = (pz - (f32x2){o.z, o.z}) * (f32x2){iz, iz};
                tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), tmin));
                tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), bestT));
            };
#endif
            float tnA, tfA, tnB, tfB;
            slab(ch.q0A, ch.q1A, tnA, tfA); slab(ch.q0B, ch.q1B, tnB, tfB);
            const bool hitA = (ch.refA != BVH_EMPTY) && (tnA <= tfA * 1.0000012f), hitB = (ch.refB != BVH_EMPTY) && (tnB <= tfB * 1.0000012f);
            // integer sort keys: tn >= 0 so its bits order like the value; the low 3 mantissa bits carry the child index (unique keys, ties to the lower child)
            const uint tbA = __float_as_uint(tnA) & ~7u, tbB = __float_as_uint(tnB) & ~7u;
            const uint keyA = (hitA ? tbA : INF_BITS) | (2u * q), keyB = (hitB ? tbB : INF_BITS) | (2u * q + 1u);
#if T8_FAST_INNER
            uint nhit = (hitA ? 1u : 0u) + (hitB ? 1u : 0u);                  // quad sum by two DPP adds (the ballot route costs two 64-bit shifts)
            nhit += dpp_u<DPP_QP_XOR1>(nhit); nhit += dpp_u<DPP_QP_XOR2>(nhit);
#else
            const uint nhit = (uint)__popc(quad_bits(t8_ballot(hitA), gl) | (quad_bits(t8_ballot(hitB), gl) << 4));
#endif
            uint rankA, rankB;
            if (ANYHIT && T8_ANYHIT_UNORDERED) {
                // an occlusion query has no use for a front-to-back order — a visible ray visits every node its segment touches whatever the order, an occluded one stops at the
                // first occluder it happens to meet: the hit children are numbered by child index (the quad's eight hit bits and a popcount instead of twelve key comparisons)
                const uint bitsA = quad_bits(t8_ballot(hitA), gl), bitsB = quad_bits(t8_ballot(hitB), gl);
                const uint all = (bitsA & 1u) | ((bitsB & 1u) << 1) | ((bitsA & 2u) << 1) | ((bitsB & 2u) << 2) | ((bitsA & 4u) << 2) | ((bitsB & 4u) << 3) | ((bitsA & 8u) << 3) | ((bitsB & 8u) << 4);      // bit 2q = child 2q (lane q's A), bit 2q + 1 = its B
                rankA = (uint)__popc(all & ((1u << (2u * q)) - 1u)); rankB = (uint)__popc(all & ((2u << (2u * q)) - 1u));
            } else {
            const uint a1 = dpp_u<DPP_QP_XOR1>(keyA), a2 = dpp_u<DPP_QP_XOR2>(keyA), a3 = dpp_u<DPP_QP_XOR3>(keyA);
            const uint b1 = dpp_u<DPP_QP_XOR1>(keyB), b2 = dpp_u<DPP_QP_XOR2>(keyB), b3 = dpp_u<DPP_QP_XOR3>(keyB);
            rankA = (keyB < keyA) ? 1u : 0u; rankB = (keyA < keyB) ? 1u : 0u;
            rankA += (a1 < keyA) ? 1u : 0u; rankA += (a2 < keyA) ? 1u : 0u; rankA += (a3 < keyA) ? 1u : 0u;
            rankA += (b1 < keyA) ? 1u : 0u; rankA += (b2 < keyA) ? 1u : 0u; rankA += (b3 < keyA) ? 1u : 0u;
            rankB += (a1 < keyB) ? 1u : 0u; rankB += (a2 < keyB) ? 1u : 0u; rankB += (a3 < keyB) ? 1u : 0u;
            rankB += (b1 < keyB) ? 1u : 0u; rankB += (b2 < keyB) ? 1u : 0u; rankB += (b3 < keyB) ? 1u : 0u;
            }
            // the nearest child's reference reaches every lane through an AND butterfly (only the hit child of rank 0 contributes; no hit -> BVH_EMPTY)
            uint next = (hitA && rankA == 0u) ? ch.refA : ((hitB && rankB == 0u) ? ch.refB : BVH_EMPTY);
            next &= dpp_u<DPP_QP_XOR1>(next); next &= dpp_u<DPP_QP_XOR2>(next);
            if (nhit > 1u) {
                if (sp + nhit - 1u > BVH8_STACK + T8_SPILL_DEPTH) { if (q == 0u) atomicOr(overflowFlag, 1u); }
                else {      // far to near: nearest on top
                    if (hitA && rankA > 0u) stackStore(sp + (nhit - 1u - rankA), ch.refA, tbA);
                    if (hitB && rankB > 0u) stackStore(sp + (nhit - 1u - rankB), ch.refB, tbB);
                    sp += nhit - 1u;
                }
            }
            cur = next;
        }

        if (COUNT) { tc2 = __builtin_readcyclecounter(); if (t8_ballot(leaf) != 0ull && lane == 0u) ctr.leafBlocks++; }
        // ---- postponed leaf: lane q tests triangles q, q + 4 (two rounds when the leaf holds more than 4)
        if (leaf) {
            const uint cnt = (pend & 7u) + 1u;
            const uint triOff0 = ((pend & 0x7FFFFFFFu) >> 3) * 48u + laneTriOff;
            float lt = __uint_as_float(INF_BITS), lu = 0.f, lv = 0.f; uint lp = 0xFFFFFFFFu;      // this lane's best candidate of this leaf
            bool alphaRan = false;
#pragma unroll 1
            for (uint r = 0; r < T8_LEAF_ROUNDS; r++) {
                const bool doit = (q + T8_LANES * r) < cnt;
                if (r > 0u && t8_ballot(doit) == 0ull) break;
                if (doit) {
                    const char* tp = trisBase + (triOff0 + (T8_LANES * 48u) * r);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(tp), tb = *reinterpret_cast<const f32x4*>(tp + 16), tc = *reinterpret_cast<const f32x4*>(tp + 32);
                    TriRecord tr; tr.v0 = make_float3(ta.x, ta.y, ta.z); tr.prim = __float_as_uint(ta.w);
                    tr.e1 = make_float3(tb.x, tb.y, tb.z); tr.flags = __float_as_uint(tb.w); tr.e2 = make_float3(tc.x, tc.y, tc.z); tr.pad = tc.w;
                    if (COUNT) ctr.triTests++;
                    float t, u, v;
                    if (intersect_tri_mt(tr, o, d, tmin, tmax, t, u, v)) {
                        bool c;
                        if (ANYHIT) {
                            c = t8_tri_box_accepts(tr, o, ix, iy, iz, t);
                            if (c && (tr.flags & 1u)) { if (COUNT && !(tr.flags & 2u)) alphaRan = true; c = !(tr.flags & 2u) && alpha_test_slot(sc, ((pend & 0x7FFFFFFFu) >> 3) + q + T8_LANES * r, u, v); }      // AlphaTestVisibilityRay (BridgeDonut:981-989)
                        } else {
                            c = ((t < bestT) || (t == bestT && tr.prim < bestPrim)) && ((t < lt) || (t == lt && tr.prim < lp));
                            if (c) c = t8_tri_box_accepts(tr, o, ix, iy, iz, t);      // only a candidate that would become the best needs the second half of the hit definition
                            if (c && (tr.flags & 1u)) { if (COUNT) alphaRan = true; c = alpha_test_slot(sc, ((pend & 0x7FFFFFFFu) >> 3) + q + T8_LANES * r, u, v); }
                        }
                        if (c) { lt = t; lp = tr.prim; lu = u; lv = v; }
                    }
                }
            }
            const bool cand = (lp != 0xFFFFFFFFu);
            pend = pend1; pend1 = pend2; pend2 = BVH_EMPTY;
            uint candBits = quad_bits(t8_ballot(cand), gl);
            T8_EVENT(4, alphaRan); T8_EVENT(5, candBits != 0u);
            if (candBits) {
                if (ANYHIT) {
                    if (q == (uint)__ffs((int)candBits) - 1u) { HitInfo h; h.t = lt; h.prim = lp; h.u = h.v = 0.f; commit(tag, h); }      // (occlusion queries carry no barycentrics)
                    active = false;
                } else {
                    if (cand && !TASKS) { minePrim = lp; mineUV[threadIdx.x] = make_float2(lu, lv); }      // beats the quad's best, hence this lane's earlier find too
                    // lexicographic min of (t, prim) over the quad: 2 butterfly steps, branch-free
                    float tk = lt; uint pk = lp;
                    {   float ot = dpp_f<DPP_QP_XOR1>(tk); uint op = dpp_u<DPP_QP_XOR1>(pk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; }
                    {   float ot = dpp_f<DPP_QP_XOR2>(tk); uint op = dpp_u<DPP_QP_XOR2>(pk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; }
                    bestT = tk; bestPrim = pk;
                }
            }
        }

        if (COUNT) tc3 = __builtin_readcyclecounter();
        // ---- slot bookkeeping: a leaf reached by the descent moves to the free leaf slot; an empty node slot pops the stack
        if (active) {
            if ((cur & BVH_LEAF_BIT) && cur != BVH_EMPTY) {
                if (pend == BVH_EMPTY) { pend = cur; cur = BVH_EMPTY; }
                else if (T8_LEAF_QUEUE > 1 && pend1 == BVH_EMPTY) { pend1 = cur; cur = BVH_EMPTY; }
                else if (T8_LEAF_QUEUE > 2 && pend2 == BVH_EMPTY) { pend2 = cur; cur = BVH_EMPTY; }
            }
            if (cur == BVH_EMPTY) {
                T8_EVENT(6, true);
                while (sp > 0u) {
                    T8_EVENT(7, true);
                    sp--;
                    uint2 e;
                    if (sp < BVH8_STACK) e = stack[sp];
                    else {      // (non-temporal for the same reason as in stackStore: keeps this a global load, not a flat one)
                        unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(sc.travSpill + ((size_t)(blockIdx.x * T8_GROUPS_PER_BLOCK + grp) * T8_SPILL_DEPTH + (sp - BVH8_STACK))));
                        e = make_uint2((uint)w, (uint)(w >> 32));
                    }
                    if (ANYHIT || __uint_as_float(e.y) <= bestT) { cur = e.x; break; }
                }
                if (cur == BVH_EMPTY && pend == BVH_EMPTY) {          // nothing left: report
                    if (COUNT && q == 0u && ctr.rayIterHist) {
                        if (rayIters > 2048u) { uint k = atomicAdd(ctr.longRayCount, 1u); if (k < 32u) { float* r = ctr.longRays + 8u * k; r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = (float)rayIters; r[7] = __uint_as_float(tag); } }
                        if (rayIters >= 128u) { uint bin = 31u - (uint)__clz((int)rayIters); atomicAdd(&ctr.rayIterHist[bin < 15u ? bin : 15u], 1ull); }      // only the tail is histogrammed (bins 7..15): 10^8 atomics on 16 words would dominate the counters step
                    }
                    if (TASKS) {                                     // a sub-tree reports only an improvement over what the ray already had
                        if (ANYHIT) { /* visible sub-tree: nothing to report */ }
                        else if (q == 0u && (bestPrim != taskPrim0 || bestT != taskT0)) { HitInfo h; h.t = bestT; h.prim = bestPrim; h.u = h.v = 0.f; commit(tag, h); }
                    }
                    else if (ANYHIT) { if (q == 0u) { HitInfo h; h.t = tmax; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); } }
                    else if (bestPrim == 0xFFFFFFFFu) { if (q == 0u) { HitInfo h; h.t = bestT; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); } }
                    else if (minePrim == bestPrim) { float2 uv = mineUV[threadIdx.x]; HitInfo h; h.t = bestT; h.prim = bestPrim; h.u = uv.x; h.v = uv.y; commit(tag, h); }
                    active = false;
                }
            }
        }
        if (COUNT) { unsigned long long tc4 = __builtin_readcyclecounter(); ctr.cyc[0] += tc1 - tc0; ctr.cyc[1] += tc2 - tc1; ctr.cyc[2] += tc3 - tc2; ctr.cyc[3] += tc4 - tc3; }
        }       // run
    }
    if (CAN_SPLIT && splitNow)
    // ---- every ray still in flight becomes a list of sub-tree tasks: node slot, postponed leaves, stack entries (outside the loop: the loop
    //      body's temporaries are dead here, so this rare path does not cost the hot kernel registers)
    {
        const uint nSlots = (cur != BVH_EMPTY ? 1u : 0u) + (pend != BVH_EMPTY ? 1u : 0u) + (pend1 != BVH_EMPTY ? 1u : 0u) + (pend2 != BVH_EMPTY ? 1u : 0u);
        const uint n = active ? nSlots + sp : 0u;
        uint base = 0u;
        if (q == 0u && n) base = atomicAdd(taskOut.count, n);
        base = dpp_u<0x00>(base);                                   // quad_perm [0,0,0,0]: lane 0 of the quad
        const bool fits = n && (base + n <= taskOut.capacity);
        if (active && !fits && q == 0u) atomicOr(overflowFlag, 2u);      // a full task queue is reported as an error by pt_render (raise TASK_QUEUE_CAPACITY)
        if (active && fits) {
            uint* tq = reinterpret_cast<uint*>(taskOut.tasks);      // scalar dword stores: no 4-register TravTask temporary
            if (q == 0u) {
                uint k = base;
                if (cur != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = cur; tq[4u * k + 2u] = 0u; k++; }
                if (pend != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = pend; tq[4u * k + 2u] = 0u; k++; }
                if (pend1 != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = pend1; tq[4u * k + 2u] = 0u; k++; }
                if (pend2 != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = pend2; tq[4u * k + 2u] = 0u; k++; }
                publish(tag, bestT, bestPrim);
            }
            for (uint i = q; i < sp; i += T8_LANES) {
                uint2 e;
                if (i < BVH8_STACK) e = stack[i];
                else { unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(sc.travSpill + ((size_t)(blockIdx.x * T8_GROUPS_PER_BLOCK + grp) * T8_SPILL_DEPTH + (i - BVH8_STACK)))); e = make_uint2((uint)w, (uint)(w >> 32)); }
                const uint k = base + nSlots + i;
                tq[4u * k] = tag; tq[4u * k + 1u] = e.x; tq[4u * k + 2u] = e.y;
            }
        }
    }
}
#endif      // PT_T8_LANES == 4

} // namespace ptk
