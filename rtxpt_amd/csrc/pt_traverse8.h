// mi355pt — cooperative BVH8 traversal for wave64: a wave carries 8 rays, each owned by an 8-lane group; lane j of a group tests
// child j of the current 128-byte node (one cache-line lookup per node per ray instead of one 64-byte gather per lane), the hit
// children are ranked by entry distance with 7 in-group shuffles, the nearest is followed directly and the rest go to the group's
// stack in LDS. Leaves hand up to 8 triangles to the 8 lanes at once. Groups refill independently from the wave's 64-ray chunk
// (persistent threads), so a long ray never holds 63 idle lanes hostage. Replaces RayQuery::TraceRayInline / the DXR any-hit
// visibility query (PathTracerBridgeDonut.hlsli:993-1055). Results are traversal-order free (min t, ties to the lower primitive id).
//
// Why this shape on MI355X: profiling the one-ray-per-lane BVH2 kernel (profiles/r01a_*) showed 33 % L2 misses but only ~0.4 TB/s
// of HBM traffic — the kernel was bound by divergent 16-byte gathers through the per-CU texture-address path (4 line lookups per
// lane per node), not by HBM bandwidth or ALU. Cooperative 128 B nodes cut line lookups per ray by ~6-9x.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_scene.h"

namespace ptk {

struct Traverse8Counters { uint nodeVisits, triTests, leafVisits, iters, leafBlocks; uint ev[8]; unsigned long long cyc[4]; };
#define T8_EVENT(k, cond) do { if (COUNT) { unsigned long long m_ = t8_ballot(cond); if (m_ && lane == (uint)__ffsll((long long)m_) - 1u) ctr.ev[k]++; } } while (0)
static const uint T8_GROUPS_PER_BLOCK = 32, T8_BLOCK = 256, T8_CHUNK = 64;
static const uint T8_RAY_STRIDE = 9, T8_RAYBUF_WORDS = (T8_BLOCK / 64u) * T8_CHUNK * T8_RAY_STRIDE;   // per-wave LDS parking lot for a chunk's rays (odd stride)

__device__ __forceinline__ float t8_rcp_dir(float d) {     // v_rcp_f32 (1 ulp): the slab test is only required to be conservative, see the tf padding
    float a = fabsf(d);
    float s = (a < 7.888609e-31f) ? 7.888609e-31f : a;
    return __builtin_amdgcn_rcpf((d < 0.0f) ? -s : s);
}
__device__ __forceinline__ uint group_bits(unsigned long long m, uint gl) { return (uint)(m >> gl) & 0xFFu; }
__device__ __forceinline__ unsigned long long t8_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }   // v_cmp straight into an SGPR pair
// In-group (8 lanes) data exchange with DPP modifiers only — no LDS crossbar (ds_bpermute) latency on the critical path.
// xor-1/2/3 are quad permutes; row_half_mirror maps lane i -> 7-i (= i^7), so i^m for m = 4..7 is half_mirror followed by the quad permute of 7^m.
#define DPP_QP_XOR1 0xB1
#define DPP_QP_XOR2 0x4E
#define DPP_QP_XOR3 0x1B
#define DPP_HALF_MIRROR 0x141
template <int CTRL> __device__ __forceinline__ uint dpp_u(uint v) { return (uint)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }

// compile-time knobs (A/B tested on the GPU, see profiles/)

#ifndef T8_LEAF_BATCH
#define T8_LEAF_BATCH 9         // groups (of 8) that must hold a postponed leaf before the wave runs the leaf block; 9 = only when a group is blocked (A/B: 1..6 within noise, 9 best)
#endif
// Src: uint fetch(uint i, float3& o, float3& d, float& tmin, float& tmax) -> user tag (e.g. path index); called by all 8 lanes of a group; tmin >= 0
// Dst: void commit(uint tag, const HitInfo& h) ; called by ONE lane of the group (closest: best hit or prim == ~0; any-hit: prim != ~0 when occluded)
//
// The kernel is VALU-issue bound (a wave64 instruction occupies a SIMD16 for 4 cycles and every per-ray scalar is replicated over the 8
// lanes of its group), so the step is written for instruction count: one 16 B header + one 12 B child slot load per lane, packed-fp32
// decode and slab arithmetic (v_pk_fma/add/mul_f32 — the box test has no parity constraint, only conservativeness), integer sort keys
// (entry distance bits with the lane id in the low 3 bits: unique, so the rank is 7 DPP compares), and hit attributes that stay in the
// lane that found them instead of being broadcast.
template <bool ANYHIT, bool COUNT, class Src, class Dst>
__device__ __forceinline__ void traverse8_persistent(const DeviceScene& sc, uint count, uint2* stackBase, uint* rayBufBase, Src fetch, Dst commit, Traverse8Counters& ctr, uint* overflowFlag) {
    const uint lane = threadIdx.x & 63u, j = lane & 7u, gl = lane & ~7u;
    const uint grp = threadIdx.x >> 3;
    uint2* stack = stackBase + grp * BVH8_STACK_STRIDE;
    const uint wavesPerBlock = T8_BLOCK / 64u;
    const uint waveId = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6), numWaves = gridDim.x * wavesPerBlock;
    const char* nodesLane = reinterpret_cast<const char*>(sc.nodes8) + 16u + 12u * j;      // this lane's child slot in node 0
    const char* trisLane = reinterpret_cast<const char*>(sc.tris) + 48u * j;               // this lane's triangle in leaf range 0
    const uint INF_BITS = 0x7F800000u;

    // chunk cursor: wave-uniform, kept in SGPRs (readfirstlane); the first refill advances to chunk `waveId`
    const uint numWavesU = (uint)__builtin_amdgcn_readfirstlane((int)numWaves);
    uint chunk = (uint)__builtin_amdgcn_readfirstlane((int)(waveId - numWaves)), chunkPos = 0u, chunkEnd = 0u;
    bool exhausted = (waveId * T8_CHUNK >= count) || !sc.rootIsValid;
    uint* rayBuf = rayBufBase + (threadIdx.x >> 6) * (T8_CHUNK * T8_RAY_STRIDE);
    if (!sc.rootIsValid && waveId == 0 && count) {            // empty scene: every ray misses
        for (uint i = lane; i < count; i += 64u) { float3 o, d; float a, b; uint tag = fetch(i, o, d, a, b); HitInfo h; h.t = b; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); }
    }
    bool active = false;
    float3 o = make_float3(0.f), d = make_float3(0.f);
    float ix = 0.f, iy = 0.f, iz = 0.f;
    float tmin = 0.f, tmax = 0.f;
    float bestT = 0.f; uint bestPrim = 0xFFFFFFFFu;           // group-uniform closest hit so far
    HitInfo mine; mine.t = 0.f; mine.prim = 0xFFFFFFFFu; mine.u = mine.v = 0.f;   // the best hit THIS lane found (attributes never leave the lane)
    // Two work slots per ray so that one loop iteration advances BOTH an inner node and a leaf: `cur` is the node being descended,
    // `pend` a postponed leaf. With a single slot every iteration runs the inner block and the leaf block at ~50 % lane use each
    // (measured: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = 30 of 64 lanes); the postponed leaf is tested while the next inner node is
    // already being intersected. The result does not depend on the visiting order (min t, ties to the lower primitive id).
    uint cur = BVH_EMPTY, pend = BVH_EMPTY, sp = 0, tag = 0;

    while (true) {
        unsigned long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
        if (COUNT) tc0 = __builtin_readcyclecounter();
        // ---- refill idle groups from the wave's current chunk
        bool need = !active && !exhausted;
        unsigned long long needMask = t8_ballot(need && j == 0u);
        if (needMask) {
            T8_EVENT(0, true);
            if (chunkPos >= chunkEnd) {
                T8_EVENT(1, true);
                // next 64-ray chunk: every lane fetches one ray and parks it in LDS, so the two dependent global loads of a fetch are paid once
                // per chunk by the whole wave instead of at every refill event (>= 8 per chunk, each stalling all 8 groups)
                chunk += numWavesU;
                chunkPos = chunk * T8_CHUNK; chunkEnd = (chunkPos + T8_CHUNK < count) ? chunkPos + T8_CHUNK : count;
                if (chunkPos >= count) { chunkPos = chunkEnd = count; }
                if (chunkPos + lane < chunkEnd) {
                    float3 ro, rd; float rtmin, rtmax;
                    uint rtag = fetch(chunkPos + lane, ro, rd, rtmin, rtmax);
                    uint* slot = rayBuf + lane * T8_RAY_STRIDE;
                    slot[0] = __float_as_uint(ro.x); slot[1] = __float_as_uint(ro.y); slot[2] = __float_as_uint(ro.z);
                    slot[3] = __float_as_uint(rd.x); slot[4] = __float_as_uint(rd.y); slot[5] = __float_as_uint(rd.z);
                    slot[6] = rtag; slot[7] = __float_as_uint(rtmin); slot[8] = __float_as_uint(rtmax);
                }
            }
            uint avail = chunkEnd - chunkPos;
            if (avail == 0u) { if (need) exhausted = true; }
            else {
                uint rank = (uint)__popcll(needMask & ((1ull << gl) - 1ull));        // rank of my group among the needing groups
                uint n = (uint)__popcll(needMask);
                if (need && rank < avail) {
                    const uint* slot = rayBuf + (((chunkPos & (T8_CHUNK - 1u)) + rank) * T8_RAY_STRIDE);
                    o = make_float3(__uint_as_float(slot[0]), __uint_as_float(slot[1]), __uint_as_float(slot[2]));
                    d = make_float3(__uint_as_float(slot[3]), __uint_as_float(slot[4]), __uint_as_float(slot[5]));
                    tag = slot[6]; tmin = __uint_as_float(slot[7]); tmax = __uint_as_float(slot[8]);
                    ix = t8_rcp_dir(d.x); iy = t8_rcp_dir(d.y); iz = t8_rcp_dir(d.z);
                    bestT = tmax; bestPrim = 0xFFFFFFFFu; mine.prim = 0xFFFFFFFFu;
                    cur = 0u; pend = BVH_EMPTY; sp = 0u; active = true;
                }
                chunkPos = (uint)__builtin_amdgcn_readfirstlane((int)(chunkPos + ((n < avail) ? n : avail)));
            }
        }
        if (t8_ballot(active) == 0ull) { if (t8_ballot(!exhausted) == 0ull) break; else continue; }

        if (COUNT && lane == 0u) ctr.iters++;
        if (COUNT) tc1 = __builtin_readcyclecounter();
        const bool inner = active && !(cur & BVH_LEAF_BIT);
        // The leaf block is ~half of an iteration's instructions but a group holds a leaf only every third or fourth step, so it is run
        // when T8_LEAF_BATCH groups have one waiting, or when some group cannot advance without it (its node slot holds a second leaf,
        // or is empty with an empty stack). Meanwhile the descent continues against a slightly stale closest distance.
        const bool leafReady = active && (pend != BVH_EMPTY);
        const bool leafBlocked = leafReady && (cur & BVH_LEAF_BIT);
        const bool runLeaves = ((uint)__popcll(t8_ballot(leafReady && j == 0u)) >= (uint)T8_LEAF_BATCH) || (t8_ballot(leafBlocked) != 0ull);
        const bool leaf = leafReady && runLeaves;
        if (COUNT && leaf && j == 0u) ctr.leafVisits++;
        const bool leafLane = leaf && (j <= (pend & 7u));

        // ---- issue this iteration's loads up front: 16 B header + 12 B child slot of `cur`, 48 B triangle of `pend`
        // (leaving these uninitialised lets the loads fly across the inner block, but the longer live ranges cost 34 VGPRs = 2 waves/SIMD
        //  and measured 15 % slower; with defaults the compiler waits where the loads are issued)
        u32x4 hdr = {0u, 0u, 0u, 0u}; u32x3p ch = {BVH_EMPTY, 0u, 0u};
        f32x4 ta = {0.f, 0.f, 0.f, 0.f}, tb = ta, tc = ta;
        if (inner) {
            const char* np = nodesLane + (size_t)cur * 128u;
            hdr = *reinterpret_cast<const u32x4*>(np - (16u + 12u * j));
            ch = *reinterpret_cast<const u32x3p*>(np);
        }
        if (leafLane) {
            const char* tp = trisLane + (size_t)((pend & 0x7FFFFFFFu) >> 3) * 48u;
            ta = *reinterpret_cast<const f32x4*>(tp); tb = *reinterpret_cast<const f32x4*>(tp + 16); tc = *reinterpret_cast<const f32x4*>(tp + 32);
        }

        T8_EVENT(2, inner); T8_EVENT(3, leaf);
        // ---- inner node: lane j tests child j
        if (inner) {
            if (COUNT && j == 0u) ctr.nodeVisits++;
            const float sx = __uint_as_float((hdr.w << 23) & INF_BITS), sy = __uint_as_float((hdr.w << 15) & INF_BITS), sz = __uint_as_float((hdr.w << 7) & INF_BITS);
            const float nx = __uint_as_float(hdr.x), ny = __uint_as_float(hdr.y), nz = __uint_as_float(hdr.z);
            // {lo, hi} pairs per axis: plane = fma(q, scale, origin) (the builder verified conservativeness with this exact expression)
            f32x2 px = __builtin_elementwise_fma((f32x2){(float)(ch.y & 0xFFu), (float)(ch.y >> 24)}, (f32x2){sx, sx}, (f32x2){nx, nx});
            f32x2 py = __builtin_elementwise_fma((f32x2){(float)((ch.y >> 8) & 0xFFu), (float)(ch.z & 0xFFu)}, (f32x2){sy, sy}, (f32x2){ny, ny});
            f32x2 pz = __builtin_elementwise_fma((f32x2){(float)((ch.y >> 16) & 0xFFu), (float)((ch.z >> 8) & 0xFFu)}, (f32x2){sz, sz}, (f32x2){nz, nz});
            f32x2 tx = (px - (f32x2){o.x, o.x}) * (f32x2){ix, ix}, ty = (py - (f32x2){o.y, o.y}) * (f32x2){iy, iy}, tz = (pz - (f32x2){o.z, o.z}) * (f32x2){iz, iz};
            float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), tmin));
            float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), bestT));
            bool hit = (ch.x != BVH_EMPTY) && (tn <= tf * 1.0000012f);
            // integer sort key: tn >= 0 so its bits order like the value; the low 3 mantissa bits carry the lane id (unique keys, ties to the lower lane)
            uint tnBits = __float_as_uint(tn) & ~7u;
            uint key = (hit ? tnBits : INF_BITS) | j;
            uint hitBits = group_bits(t8_ballot(hit), gl);
            uint nhit = (uint)__popc(hitBits);
            const uint hm = dpp_u<DPP_HALF_MIRROR>(key);
            uint rank = 0;
            rank += (dpp_u<DPP_QP_XOR1>(key) < key) ? 1u : 0u;
            rank += (dpp_u<DPP_QP_XOR2>(key) < key) ? 1u : 0u;
            rank += (dpp_u<DPP_QP_XOR3>(key) < key) ? 1u : 0u;
            rank += (dpp_u<DPP_QP_XOR3>(hm) < key) ? 1u : 0u;
            rank += (dpp_u<DPP_QP_XOR2>(hm) < key) ? 1u : 0u;
            rank += (dpp_u<DPP_QP_XOR1>(hm) < key) ? 1u : 0u;
            rank += (hm < key) ? 1u : 0u;
            // the nearest child's reference reaches every lane through an AND butterfly (only a hit lane of rank 0 contributes; no hit -> BVH_EMPTY)
            uint next = (hit && rank == 0u) ? ch.x : BVH_EMPTY;
            next &= dpp_u<DPP_QP_XOR1>(next); next &= dpp_u<DPP_QP_XOR2>(next); next &= dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(next));
            if (nhit > 1u) {
                if (sp + nhit - 1u > BVH8_STACK) { if (j == 0u) atomicOr(overflowFlag, 1u); }
                else { if (hit && rank > 0u) stack[sp + (nhit - 1u - rank)] = make_uint2(ch.x, tnBits); sp += nhit - 1u; }     // far to near: nearest on top
            }
            cur = next;
        }

        if (COUNT) { tc2 = __builtin_readcyclecounter(); if (lane == 0u && t8_ballot(leaf) != 0ull) ctr.leafBlocks++; }
        // ---- postponed leaf: lane j tests triangle j
        if (leaf) {
            bool cand = false; float t = 0.f, u = 0.f, v = 0.f; uint prim = 0xFFFFFFFFu; bool alphaRan = false;
            if (leafLane) {
                TriRecord tr; tr.v0 = make_float3(ta.x, ta.y, ta.z); tr.prim = __float_as_uint(ta.w);
                tr.e1 = make_float3(tb.x, tb.y, tb.z); tr.flags = __float_as_uint(tb.w); tr.e2 = make_float3(tc.x, tc.y, tc.z);
                if (COUNT) ctr.triTests++;
                if (intersect_tri(tr, o, d, tmin, tmax, t, u, v)) {
                    prim = tr.prim;
                    if (ANYHIT) {
                        cand = true;
                        if (tr.flags & 1u) { if (COUNT && !(tr.flags & 2u)) alphaRan = true; cand = !(tr.flags & 2u) && alpha_test(sc, prim, u, v); }      // AlphaTestVisibilityRay (BridgeDonut:981-989)
                    } else {
                        cand = (t < bestT) || (t == bestT && prim < bestPrim);
                        if (cand && (tr.flags & 1u)) { if (COUNT) alphaRan = true; cand = alpha_test(sc, prim, u, v); }
                    }
                }
            }
            pend = BVH_EMPTY;
            uint candBits = group_bits(t8_ballot(cand), gl);
            T8_EVENT(4, alphaRan); T8_EVENT(5, candBits != 0u);
            if (candBits) {
                if (ANYHIT) {
                    if (j == (uint)__ffs((int)candBits) - 1u) { HitInfo h; h.t = t; h.prim = prim; h.u = u; h.v = v; commit(tag, h); }
                    active = false;
                } else {
                    if (cand) { mine.t = t; mine.prim = prim; mine.u = u; mine.v = v; }      // beats the group's best, hence this lane's earlier find too
                    // lexicographic min of (t, prim) over the group: 3 butterfly steps, branch-free
                    float tk = cand ? t : __uint_as_float(INF_BITS); uint pk = cand ? prim : 0xFFFFFFFFu;
                    {   float ot = dpp_f<DPP_QP_XOR1>(tk); uint op = dpp_u<DPP_QP_XOR1>(pk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; }
                    {   float ot = dpp_f<DPP_QP_XOR2>(tk); uint op = dpp_u<DPP_QP_XOR2>(pk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; }
                    {   float ot = dpp_f<DPP_QP_XOR3>(dpp_f<DPP_HALF_MIRROR>(tk)); uint op = dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(pk));
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; }
                    bestT = tk; bestPrim = pk;
                }
            }
        }

        if (COUNT) tc3 = __builtin_readcyclecounter();
        // ---- slot bookkeeping: a leaf reached by the descent moves to the free leaf slot; an empty node slot pops the stack
        if (active) {
            if ((cur & BVH_LEAF_BIT) && cur != BVH_EMPTY && pend == BVH_EMPTY) { pend = cur; cur = BVH_EMPTY; }
            if (cur == BVH_EMPTY) {
                T8_EVENT(6, true);
                while (sp > 0u) {
                    T8_EVENT(7, true);
                    sp--;
                    uint2 e = stack[sp];
                    if (ANYHIT || __uint_as_float(e.y) <= bestT) { cur = e.x; break; }
                }
                if (cur == BVH_EMPTY && pend == BVH_EMPTY) {          // nothing left: report
                    if (ANYHIT) { if (j == 0u) { HitInfo h; h.t = tmax; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); } }
                    else if (bestPrim == 0xFFFFFFFFu) { if (j == 0u) { HitInfo h; h.t = bestT; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); } }
                    else if (mine.prim == bestPrim) commit(tag, mine);
                    active = false;
                }
            }
        }
        if (COUNT) { unsigned long long tc4 = __builtin_readcyclecounter(); ctr.cyc[0] += tc1 - tc0; ctr.cyc[1] += tc2 - tc1; ctr.cyc[2] += tc3 - tc2; ctr.cyc[3] += tc4 - tc3; }
    }
}

} // namespace ptk
