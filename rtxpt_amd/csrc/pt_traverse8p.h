// mi355pt — cooperative BVH8 traversal for wave64, two lanes per ray: a wave carries 32 rays, each owned by a PAIR of lanes; lane h of a pair tests
// children 4h .. 4h + 3 of the current 128-byte node (48 contiguous bytes: three 16-byte loads) and triangle h of a leaf.
//
// Why pairs (round 3): with four lanes per ray (the kernel of rounds 1-3, in the history) the loop is VALU-issue bound at 8 waves per SIMD — extra v_nop slots
// lengthen k_extend one for one (profiles/r03i_valu_bound_probe.txt) — and a wave64 VALU instruction costs its four cycles whatever the lanes do. Of the ~325
// VALU instructions of an average wave iteration only the slab tests (~50) and the triangle tests (~110 when the leaf block runs) are work that belongs to a
// child or a triangle; the rest — refill, child ranking, stack, slot bookkeeping, the alpha test's addressing, the hit reduction — is per-RAY work that every
// lane of the ray's group repeats. Two lanes per ray halve the replicated share per ray: per lane the slab and triangle work doubles (four children, two
// triangle rounds), per wave iteration the instruction count rises by about a third, and the iteration advances 32 rays instead of 16. The closest hit is
// traversal-order free (min t, ties to the lower primitive id); an occlusion query reports whether any accepted hit exists. Straggler splitting, Src / Dst /
// Pub and the template flags: pt_traverse8.h.
#pragma once
#include "pt_traverse8.h"

namespace ptk {

#define DPP_PAIR_LO 0xA0      // quad_perm [0,0,2,2]: lane 0 of the pair
#define DPP_PAIR_HI 0xF5      // quad_perm [1,1,3,3]: lane 1 of the pair
__device__ __forceinline__ uint pair_bits(unsigned long long m, uint pl) { return (uint)(m >> pl) & 0x3u; }

#ifndef T8_EMPTY_SLOT_CHECK
// 0: an empty child slot needs no test of its own — the builder writes it as an inverted box (lo = 255, hi = 0 on every axis: pt_build.hip k_collapse8),
#define T8_EMPTY_SLOT_CHECK 0
// which no ray's slab interval enters; were it ever "hit" (a node of zero extent), its reference is BVH_EMPTY, which the slot bookkeeping skips
#endif
#if T8_EMPTY_SLOT_CHECK
#define T8_HIT(ref, tn, tf) (((ref) != BVH_EMPTY) && ((tn) <= (tf) * 1.0000012f))
#else
#define T8_HIT(ref, tn, tf) ((tn) <= (tf) * 1.0000012f)
#endif
// Measured and removed (history: commit af4c2b2 has the code; numbers in DESIGN.md §4): a dense leaf block (profiles/r04y_dense_leaf_ab.txt, +13 %), batched
// refills (r04y_refill_batch_ab.txt), one pop trip per iteration (r04v_pop_once_ab.txt), two entries per pop trip, octant-ordered child slots
// (r04u_octant_order_ab.txt, +25 %); round 5: a ray with nothing left but postponed leaves waiting one or three iterations (or for a second / fourth such ray)
// before it forces the leaf block — the block then runs in 0.50 instead of 0.65 of the iterations and the rays take 0.63 instead of 0.58 iterations: k_extend
// unchanged (profiles/r05s_leaf_patience_ab.txt). DEFER (with CAN_SPLIT): a dry wave keeps going for taskOut.capacity iterations (instead of T8_TAIL_ITERS),
// and the rays then still in flight are not cut into sub-trees, only reported through publish() — the caller has them traced again elsewhere (the tail kernel,
// pt_tail.hip, hands their paths back to the host loop). No task queue is touched.
template <bool ANYHIT, bool COUNT, bool FIXED_RANGE, bool TASKS, bool CAN_SPLIT, bool DEFER = false, class Src, class Dst, class Pub>
__device__ __forceinline__ void traverse8_pairs(const DeviceScene& sc, uint count, uint raysPerChunk, uint2* stackBase, uint* rayBufBase, float2* mineUV, Src fetch, Dst commit, Pub publish, TravTaskOut taskOut,
                                                Traverse8Counters& ctr, uint* overflowFlag, const uint vBlock, const uint vGrid) {
    // vBlock / vGrid: the block's index among, and the number of, the blocks that work on THIS launch's items — blockIdx.x / gridDim.x for a kernel of one
    // kind; a fused launch (k_trace_pair: closest-hit blocks next to visibility blocks, pt_wavefront.hip) hands each kind its own range. The stack tails are
    // addressed by the physical block index.
    static_assert(T8_LANES == 2u, "traverse8_pairs is the two-lanes-per-ray build");
    const uint RAY_STRIDE = TASKS ? T8_TASK_STRIDE : T8_RAY_STRIDE;
    const uint lane = threadIdx.x & 63u, h = lane & 1u, pl = lane & ~1u;
    const uint grp = threadIdx.x >> 1;
    uint2* stack = stackBase + grp * BVH8_STACK_STRIDE;
    const uint wavesPerBlock = T8_BLOCK / 64u;
    const uint waveId = vBlock * wavesPerBlock + (threadIdx.x >> 6), numWaves = vGrid * wavesPerBlock;
    const char* nodesBase = reinterpret_cast<const char*>(sc.nodes8);
    const char* trisBase = reinterpret_cast<const char*>(sc.tris);
    const uint laneChildOff = 16u + 48u * h, laneTriOff = 48u * h;
    const uint INF_BITS = 0x7F800000u;

    const uint numWavesU = (uint)__builtin_amdgcn_readfirstlane((int)numWaves);
    uint chunk = (uint)__builtin_amdgcn_readfirstlane((int)(waveId - numWaves)), chunkPos = 0u, chunkEnd = 0u;
    const uint rpc = (uint)__builtin_amdgcn_readfirstlane((int)raysPerChunk);
    bool exhausted = (waveId * rpc >= count) || !sc.rootIsValid;
    uint* rayBuf = rayBufBase + (threadIdx.x >> 6) * (T8_CHUNK * RAY_STRIDE);
    uint tailIters = 0u, waveDry = 0u;      // wave-uniform, kept as scalars: the number of iterations since the wave found its queue empty
    if (!sc.rootIsValid && waveId == 0 && count) {            // empty scene: every ray misses
        for (uint i = lane; i < count; i += 64u) { float3 o, d; float a, b, bt; uint sr, bp; uint tag = fetch(i, o, d, a, b, sr, bt, bp); HitInfo hh; hh.t = b; hh.prim = 0xFFFFFFFFu; hh.u = hh.v = 0.f; commit(tag, hh); }
    }
    bool active = false;
    float3 o = make_float3(0.f);
    // the ray's shear (d[kx] / d[kz], d[ky] / d[kz]) and the byte offsets of the record groups of its axes kx | ky << 8 | kz << 16 (leaf block)
    float Sx = 0.f, Sy = 0.f; uint axes = 0u;
    float ix = 0.f, iy = 0.f, iz = 0.f;
    uint selN = 0u, selF = 0u;
    float tmin = 0.f, tmax = FIXED_RANGE ? kMaxRayTravel : 0.f;
    float bestT = 0.f; uint bestPrim = 0xFFFFFFFFu;
    uint minePrim = 0xFFFFFFFFu;
    uint cur = BVH_EMPTY, pend = BVH_EMPTY, sp = 0, tag = 0;
    uint rayIters = 0;
    float taskT0 = 0.f; uint taskPrim0 = 0xFFFFFFFFu;
    uint pend1 = BVH_EMPTY, pend2 = BVH_EMPTY;
#ifdef T8_PROBE_INSTANCE_SWITCHES
    uint lastInst_ = 0xFFFFFFFFu;
#endif

    // sb: the pair's stack base. The caller forms it anew for every node that pushes (stack_base(): two instructions) instead of keeping it across the loop:
    // the loop is one register short since the watertight leaf block, and the allocator's choice was to spill exactly this value — a scratch load and a full
    // vmcnt(0) wait at every push.
    auto stack_base = [&]() -> uint2* { uint g_ = threadIdx.x >> 1; asm volatile("" : "+v"(g_)); return stackBase + g_ * BVH8_STACK_STRIDE; };
    // the pair's stack tail in global memory (entries BVH8_STACK and up: rare). Formed where it is used, for the same reason: hoisted out of the loop, the
    // address is the value the allocator spills.
    auto spill_slot = [&](uint idx) -> uint2* { uint g_ = threadIdx.x >> 1; asm volatile("" : "+v"(g_)); return sc.travSpill + ((size_t)(blockIdx.x * T8_GROUPS_PER_BLOCK + g_) * T8_SPILL_DEPTH + (idx - BVH8_STACK)); };
    auto stackStore = [&](uint2* sb, uint idx, uint ref, uint tbits) {
        if (idx < BVH8_STACK) sb[idx] = make_uint2(ref, tbits);
        else {
            unsigned long long* tail = reinterpret_cast<unsigned long long*>(spill_slot(idx));
            __builtin_nontemporal_store(((unsigned long long)tbits << 32) | ref, tail);
        }
    };

    bool splitNow = false, stop = false;
    while (!stop) {
        unsigned long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
        if (COUNT) tc0 = __builtin_readcyclecounter();
        // ---- refill idle pairs from the wave's current chunk
        bool need = !active && !exhausted;
        unsigned long long needMask = t8_ballot(need && h == 0u);
        if (needMask) {
            T8_EVENT(0, true);
            if (chunkPos >= chunkEnd) {
                T8_EVENT(1, true);
                chunk += numWavesU;
                chunkPos = chunk * rpc; chunkEnd = (chunkPos + rpc < count) ? chunkPos + rpc : count;
                if (chunkPos >= count) { chunkPos = chunkEnd = count; }
                if (chunkPos + lane < chunkEnd) {
                    float3 ro, rd; float rtmin, rtmax, rbestT; uint rstart, rbestPrim;
                    uint rtag = fetch(chunkPos + lane, ro, rd, rtmin, rtmax, rstart, rbestT, rbestPrim);
                    uint* slot = rayBuf + lane * RAY_STRIDE;
                    slot[0] = __float_as_uint(ro.x); slot[1] = __float_as_uint(ro.y); slot[2] = __float_as_uint(ro.z);
                    // what the leaf block needs of the direction (intersect_tri_wt, pt_scene.h), formed once by the fetching lane: kz = the axis of the largest
                    // |d| (ties to the lower axis), kx / ky the next two in cyclic order; Sz = the reciprocal of d[kz], Sx = d[kx] * Sz, Sy = d[ky] * Sz. The
                    // direction itself is not kept.
                    const float rix = t8_rcp_dir(rd.x), riy = t8_rcp_dir(rd.y), riz = t8_rcp_dir(rd.z);
                    const float adx = fabsf(rd.x), ady = fabsf(rd.y), adz = fabsf(rd.z);
                    const bool rk2 = adz > adx && adz > ady, rk1 = !rk2 && ady > adx;
                    const float rSz = rk2 ? riz : (rk1 ? riy : rix);
                    slot[3] = __float_as_uint((rk2 ? rd.x : (rk1 ? rd.z : rd.y)) * rSz); slot[4] = __float_as_uint((rk2 ? rd.y : (rk1 ? rd.x : rd.z)) * rSz);
                    slot[5] = rk2 ? (0u | (16u << 8) | (32u << 16)) : (rk1 ? (32u | (0u << 8) | (16u << 16)) : (16u | (32u << 8) | (0u << 16)));
                    slot[6] = rtag;
                    // the interval's two words are free (every ray of the launch has [0, kMaxRayTravel]): they carry the byte selectors of the slab test
                    if (FIXED_RANGE && !TASKS) {
                        const uint nxb = rix < 0.f ? 3u : 0u, fxb = rix < 0.f ? 0u : 3u, nyb = riy < 0.f ? 4u : 1u, fyb = riy < 0.f ? 1u : 4u, nzb = riz < 0.f ? 5u : 2u, fzb = riz < 0.f ? 2u : 5u;
                        slot[7] = nxb | (nyb << 8) | (nzb << 16) | (fxb << 24); slot[8] = fyb | (fzb << 8);
                    } else { slot[7] = __float_as_uint(TASKS ? rtmax : rtmin); slot[8] = TASKS ? __float_as_uint(rbestT) : __float_as_uint(rtmax); }
                    if (TASKS) { slot[9] = rbestPrim; slot[10] = rstart; }
                    slot[RAY_STRIDE - 3u] = __float_as_uint(rix); slot[RAY_STRIDE - 2u] = __float_as_uint(riy); slot[RAY_STRIDE - 1u] = __float_as_uint(riz);
                }
            }
            uint avail = chunkEnd - chunkPos;
            // Two statements, not if / else: with "queue empty -> flags, else -> take a ray" every loop-carried value of the ray is a three-way phi at the
            // join, which the compiler lowers as a copy of the loop's registers out to temporaries and back on every pass through this block (38 v_mov). An
            // empty queue simply makes the take condition below false (rank < 0) and the cursor advance zero (profiles/r05t_refill_flat_ab.txt: k_extend 46.2
            // -> 45.1 ms).
            if (avail == 0u) { if (need) exhausted = true; waveDry = 1u; }
            {
                uint rank = (uint)__popcll(needMask & ((1ull << pl) - 1ull));
                uint n = (uint)__popcll(needMask);
                if (need && rank < avail) {
                    const uint* slot = rayBuf + (((chunkPos - chunk * rpc) + rank) * RAY_STRIDE);
                    o = make_float3(__uint_as_float(slot[0]), __uint_as_float(slot[1]), __uint_as_float(slot[2]));
                    Sx = __uint_as_float(slot[3]); Sy = __uint_as_float(slot[4]); axes = slot[5];
                    tag = slot[6];
                    ix = __uint_as_float(slot[RAY_STRIDE - 3u]); iy = __uint_as_float(slot[RAY_STRIDE - 2u]); iz = __uint_as_float(slot[RAY_STRIDE - 1u]);
                    if (FIXED_RANGE && !TASKS) { selN = slot[7]; selF = slot[8]; }
                    else {   // child bytes: q0 = lo.x lo.y lo.z hi.x (selector values 0..3), q1 = hi.y hi.z (4, 5)
                        const uint nxb = ix < 0.f ? 3u : 0u, fxb = ix < 0.f ? 0u : 3u, nyb = iy < 0.f ? 4u : 1u, fyb = iy < 0.f ? 1u : 4u, nzb = iz < 0.f ? 5u : 2u, fzb = iz < 0.f ? 2u : 5u;
                        selN = nxb | (nyb << 8) | (nzb << 16) | (fxb << 24); selF = fyb | (fzb << 8);
                    }
                    if (TASKS) {
                        if (!FIXED_RANGE) tmax = __uint_as_float(slot[7]);
                        bestT = taskT0 = __uint_as_float(slot[8]); bestPrim = taskPrim0 = slot[9]; cur = slot[10];
                    } else {
                        if (!FIXED_RANGE) { tmin = __uint_as_float(slot[7]); tmax = __uint_as_float(slot[8]); }
                        bestT = tmax; bestPrim = 0xFFFFFFFFu; cur = 0u;
                    }
                    minePrim = 0xFFFFFFFFu; rayIters = 0u;
#ifdef T8_PROBE_INSTANCE_SWITCHES
                    lastInst_ = 0xFFFFFFFFu;
#endif
                    pend = BVH_EMPTY; pend1 = BVH_EMPTY; pend2 = BVH_EMPTY; sp = 0u; active = true;
                }
                chunkPos = (uint)__builtin_amdgcn_readfirstlane((int)(chunkPos + ((n < avail) ? n : avail)));
            }
        }
        bool run = t8_ballot(active) != 0ull;
        if (!run) { if (t8_ballot(!exhausted) == 0ull) stop = true; }
        else if (CAN_SPLIT) {
            tailIters += waveDry;
            if (tailIters > (DEFER ? taskOut.capacity : (uint)(TASKS ? T8_TAIL_ITERS_TASKS : T8_TAIL_ITERS))) { splitNow = true; stop = true; run = false; }
        }
        if (run) {
#ifdef T8_PROBE_VNOPS
#pragma unroll
        for (int k_ = 0; k_ < T8_PROBE_VNOPS; k_++) asm volatile("v_nop");
#endif
// issue-slot probes (developer builds; tools/valu_ceiling): T8_PROBE_FILL_N independent filler instructions per wave iteration — 1: of the class that issues
// every
#ifdef T8_PROBE_FILL
        // 2 cycles per SIMD (v_add_u32), 2: of the class that issues every 4 (v_max_f32), 3: v_cndmask_b32_e64. Outputs are dead; no register lives across the
        // loop.
        {
            uint f0_, f1_;
#pragma unroll
            for (int k_ = 0; k_ < T8_PROBE_FILL_N / 2; k_++) {
                if (T8_PROBE_FILL == 1) asm volatile("v_add_u32 %0, %2, %2\n v_add_u32 %1, %2, %2" : "=v"(f0_), "=v"(f1_) : "v"(lane));
                else if (T8_PROBE_FILL == 2) asm volatile("v_max_f32 %0, %2, %2\n v_max_f32 %1, %2, %2" : "=v"(f0_), "=v"(f1_) : "v"(lane));
                else asm volatile("v_cndmask_b32_e64 %0, %2, %2, s[0:1]\n v_cndmask_b32_e64 %1, %2, %2, s[0:1]" : "=v"(f0_), "=v"(f1_) : "v"(lane));
            }
        }
#endif
        if (COUNT && lane == 0u) ctr.iters++;
        if (COUNT && active) rayIters++;
        if (COUNT) tc1 = __builtin_readcyclecounter();
        const bool inner = active && !(cur & BVH_LEAF_BIT);
        const bool leafReady = active && (pend != BVH_EMPTY);
        const bool queueFull = (T8_LEAF_QUEUE == 1) ? true : ((T8_LEAF_QUEUE == 2) ? (pend1 != BVH_EMPTY) : (pend2 != BVH_EMPTY));
        const bool leafBlocked = leafReady && (cur & BVH_LEAF_BIT) && (cur == BVH_EMPTY || queueFull);
        // (a second trigger — "n pairs hold a postponed leaf" — never paid: profiles/r05o_leaf_batch_ab.txt, r05t_refill_flat_ab.txt)
        const bool runLeaves = t8_ballot(leafBlocked) != 0ull;
        const bool leaf = leafReady && runLeaves;
        if (COUNT && leaf && h == 0u) ctr.leafVisits++;
// developer probe (counter builds): how often would a two-level traversal have to enter an instance? Counted per ray as the visited leaves whose
#ifdef T8_PROBE_INSTANCE_SWITCHES
        // instance differs from the previous visited leaf's (event slot 4 — the alpha-test count — carries it in such a build)
        if (COUNT && leaf && h == 0u) {
            const uint prim_ = *reinterpret_cast<const uint*>(trisBase + (((pend & 0x7FFFFFFFu) >> 3) * 48u + 12u));
            const uint inst_ = *reinterpret_cast<const uint*>(reinterpret_cast<const char*>(sc.shadeTris) + (size_t)prim_ * 128u);
            if (inst_ != lastInst_) ctr.ev[4]++;
            lastInst_ = inst_;
        }
#endif
        T8_EVENT(2, inner); T8_EVENT(3, leaf);

        // ---- inner node: lane h tests children 4h .. 4h + 3
        if (inner) {
            const uint nodeOff = cur * 128u;
            const u32x4 hdr = *reinterpret_cast<const u32x4*>(nodesBase + nodeOff);
            const u32x4 c0 = *reinterpret_cast<const u32x4*>(nodesBase + (nodeOff + laneChildOff));
            const u32x4 c1 = *reinterpret_cast<const u32x4*>(nodesBase + (nodeOff + laneChildOff + 16u));
            const u32x4 c2 = *reinterpret_cast<const u32x4*>(nodesBase + (nodeOff + laneChildOff + 32u));
            const f32x4 scl = *reinterpret_cast<const f32x4*>(nodesBase + (nodeOff + 112u));
            if (COUNT && h == 0u) ctr.nodeVisits++;
            const float nx = __uint_as_float(hdr.x), ny = __uint_as_float(hdr.y), nz = __uint_as_float(hdr.z);
            const float sx = scl.x, sy = scl.y, sz = scl.z;
            auto slab = [&](uint q0, uint q1, float& tn, float& tf) {
                const uint N = __builtin_amdgcn_perm(q1, q0, selN), F = __builtin_amdgcn_perm(q1, q0, selF);
                f32x2 px = __builtin_elementwise_fma((f32x2){(float)(N & 0xFFu), (float)(N >> 24)}, (f32x2){sx, sx}, (f32x2){nx, nx});
                f32x2 py = __builtin_elementwise_fma((f32x2){(float)((N >> 8) & 0xFFu), (float)(F & 0xFFu)}, (f32x2){sy, sy}, (f32x2){ny, ny});
                f32x2 pz = __builtin_elementwise_fma((f32x2){(float)((N >> 16) & 0xFFu), (float)((F >> 8) & 0xFFu)}, (f32x2){sz, sz}, (f32x2){nz, nz});
                f32x2 tx = (px - (f32x2){o.x, o.x}) * (f32x2){ix, ix}, ty = (py - (f32x2){o.y, o.y}) * (f32x2){iy, iy}, tz = (pz - (f32x2){o.z, o.z}) * (f32x2){iz, iz};
                tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
                tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, bestT));
            };
            uint ref[4] = {c0.x, c0.w, c1.z, c2.y};
            // integer sort keys: tn >= 0 so its bits order like the value; the low 3 mantissa bits carry the child index (unique keys, ties to the lower
            // child); a child that is not hit gets +inf. The stack entry's distance is the key without its index bits.
            uint key[4]; bool hit[4];
            {   float tn, tf;
                slab(c0.y, c0.z, tn, tf); hit[0] = T8_HIT(ref[0], tn, tf); key[0] = (hit[0] ? (__float_as_uint(tn) & ~7u) : INF_BITS) | (4u * h);
                slab(c1.x, c1.y, tn, tf); hit[1] = T8_HIT(ref[1], tn, tf); key[1] = (hit[1] ? (__float_as_uint(tn) & ~7u) : INF_BITS) | (4u * h + 1u);
                slab(c1.w, c2.x, tn, tf); hit[2] = T8_HIT(ref[2], tn, tf); key[2] = (hit[2] ? (__float_as_uint(tn) & ~7u) : INF_BITS) | (4u * h + 2u);
                slab(c2.z, c2.w, tn, tf); hit[3] = T8_HIT(ref[3], tn, tf); key[3] = (hit[3] ? (__float_as_uint(tn) & ~7u) : INF_BITS) | (4u * h + 3u);
            }
            uint nhit, rank[4];
            if (ANYHIT && T8_ANYHIT_UNORDERED) {
                // an occlusion query has no use for a front-to-back order: the hit children are numbered by child index
                const uint own = (hit[0] ? 1u : 0u) | (hit[1] ? 2u : 0u) | (hit[2] ? 4u : 0u) | (hit[3] ? 8u : 0u);
                const uint oth = dpp_u<DPP_QP_XOR1>(own);
                const uint all = h ? (oth | (own << 4)) : (own | (oth << 4));
                nhit = (uint)__popc(all);
                const uint below = (1u << (4u * h)) - 1u;            // children of the other lane that come first
#pragma unroll
                for (int k = 0; k < 4; k++) rank[k] = (uint)__popc(all & (below | (((1u << k) - 1u) << (4u * h))));
            } else {
                nhit = (hit[0] ? 1u : 0u) + (hit[1] ? 1u : 0u) + (hit[2] ? 1u : 0u) + (hit[3] ? 1u : 0u);
                nhit += dpp_u<DPP_QP_XOR1>(nhit);
                // rank = number of keys below mine: six compares among my own four (each decides two ranks), sixteen against the other lane's
                rank[0] = 3u; rank[1] = 2u; rank[2] = 1u; rank[3] = 0u;
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = i + 1; j < 4; j++) { const uint c = (key[i] < key[j]) ? 1u : 0u; rank[j] += c; rank[i] -= c; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint ok = dpp_u<DPP_QP_XOR1>(key[j]);
#pragma unroll
                    for (int k = 0; k < 4; k++) rank[k] += (ok < key[k]) ? 1u : 0u;
                }
            }
            uint next = BVH_EMPTY;
#pragma unroll
            for (int k = 0; k < 4; k++) next = (hit[k] && rank[k] == 0u) ? ref[k] : next;
            next &= dpp_u<DPP_QP_XOR1>(next);
            if (nhit > 1u) {
                if (sp + nhit - 1u > BVH8_STACK + T8_SPILL_DEPTH) { if (h == 0u) atomicOr(overflowFlag, 1u); }
                else {      // far to near: nearest on top
                    uint2* sb = stack_base();
#pragma unroll
                    for (int k = 0; k < 4; k++) if (hit[k] && rank[k] > 0u) stackStore(sb, sp + (nhit - 1u - rank[k]), ref[k], key[k] & ~7u);
                    sp += nhit - 1u;
                }
            }
            cur = next;
        }

        if (COUNT) { tc2 = __builtin_readcyclecounter(); if (t8_ballot(leaf) != 0ull && lane == 0u) ctr.leafBlocks++; }
        // ---- postponed leaf: lane h tests triangle h (h + 2, ... when leaves hold more than two). The watertight test of pt_scene.h (intersect_tri_wt) on the
        // same operands: the axis permutation (kx, ky, kz) is the ORDER in which the record's three 16-byte groups are loaded; the ray's own quantities are
        // selected once per leaf block.
        if (leaf) {
            const uint cnt = (pend & 7u) + 1u;
            const uint slot0 = (pend & 0x7FFFFFFFu) >> 3;
            const uint triOff0 = slot0 * 48u + laneTriOff;
            float lt = __uint_as_float(INF_BITS), lu = 0.f, lv = 0.f; uint lp = 0xFFFFFFFFu;
            bool alphaRan = false;
            const uint gx = axes & 0xFFu, gy = (axes >> 8) & 0xFFu, gz = axes >> 16;          // byte offsets of the groups of axes kx, ky, kz
            const bool k2 = gz == 32u, k1 = gz == 16u;                                      // kz = 2 / 1 / 0
            const float okx = k2 ? o.x : (k1 ? o.z : o.y), oky = k2 ? o.y : (k1 ? o.x : o.z), okz = k2 ? o.z : (k1 ? o.y : o.x);
            const float Sz = k2 ? iz : (k1 ? iy : ix);                                       // ix / iy / iz = ray_safe_rcp of the direction's components
#pragma unroll 1
            for (uint r = 0; r < T8_LEAF_ROUNDS; r++) {
                const bool doit = (h + T8_LANES * r) < cnt;
                if (r > 0u && t8_ballot(doit) == 0ull) break;
                if (doit) {
                    const uint toff = triOff0 + (T8_LANES * 48u) * r;      // (32-bit byte offsets from the SGPR base: global_load ... saddr)
                    // twelve bytes per group (global_load_dwordx3): the fourth word is not needed here, and a 16-byte destination whose last register the
                    // allocator then reuses for the next load's address makes that load wait for this one — two memory latencies in a row instead of one (seen
                    // in the ISA, measured: +10 % on k_extend)
                    struct __attribute__((packed, aligned(4))) f32x3p { float x, y, z; };
                    const f32x3p g0 = *reinterpret_cast<const f32x3p*>(trisBase + (toff + gx)), g1 = *reinterpret_cast<const f32x3p*>(trisBase + (toff + gy)), g2 = *reinterpret_cast<const f32x3p*>(trisBase + (toff + gz));
                    if (COUNT) ctr.triTests++;
                    // intersect_tri_wt's arithmetic, two values per instruction where the operands already sit in neighbouring registers (vertices 0 and 1 of a
                    // loaded group): packed fp32 operations round each half like the scalar ones
                    const f32x2 ABkz = (f32x2){g2.x, g2.y} - (f32x2){okz, okz}; const float Ckz = g2.z - okz;
                    const f32x2 ABx = __builtin_elementwise_fma((f32x2){-Sx, -Sx}, ABkz, (f32x2){g0.x, g0.y} - (f32x2){okx, okx}), ABy = __builtin_elementwise_fma((f32x2){-Sy, -Sy}, ABkz, (f32x2){g1.x, g1.y} - (f32x2){oky, oky});
                    const float Cx = fmaf(-Sx, Ckz, g0.z - okx), Cy = fmaf(-Sy, Ckz, g1.z - oky);
                    const float Ax = ABx.x, Bx = ABx.y, Ay = ABy.x, By = ABy.y, Akz = ABkz.x, Bkz = ABkz.y;
                    const f32x2 m1 = ABx * (f32x2){Cy, Cy}, m2 = ABy * (f32x2){Cx, Cx};      // (Ax Cy, Bx Cy), (Ay Cx, By Cx)
                    float V = m1.x - m2.x, U = m2.y - m1.y, W = Bx * Ay - By * Ax;            // U = Cx By - Cy Bx, V = Ax Cy - Ay Cx, W = Bx Ay - By Ax
                    // on an edge or a vertex (or a degenerate triangle): the exact sign from the products' rounding errors (wt_edge)
                    if (U == 0.0f || V == 0.0f || W == 0.0f) {
                        if (U == 0.0f) U = fmaf(Cx, By, -(Cx * By)) - fmaf(Cy, Bx, -(Cy * Bx));
                        if (V == 0.0f) V = fmaf(Ax, Cy, -(Ax * Cy)) - fmaf(Ay, Cx, -(Ay * Cx));
                        if (W == 0.0f) W = fmaf(Bx, Ay, -(Bx * Ay)) - fmaf(By, Ax, -(By * Ax));
                    }
                    const float det = (U + V) + W;
                    // no edge function negative, or none positive (pt_scene.h writes it as !((U < 0 || V < 0 || W < 0) && (U > 0 || V > 0 || W > 0)): the same
                    // boolean for numbers; a NaN ends as "no hit" either way, through t); without short-circuits: v_min3, v_max3 and three compares, no branch
                    const bool inside = (bool)(((int)(fminf(U, fminf(V, W)) >= 0.0f) | (int)(fmaxf(U, fmaxf(V, W)) <= 0.0f)) & (int)(det != 0.0f));
                    if (inside) {
                        const float T = fmaf(W, Sz * Ckz, fmaf(V, Sz * Bkz, U * (Sz * Akz)));
                        const float inv = 1.0f / det;
                        const float t = T * inv, u = V * inv, v = W * inv;
                        if (t > tmin && t < tmax) {
                            // a candidate (about one test in six): the record again, in its own order this time — prim rides with the x group, flags with y,
                            // pad with z. Reloaded (the line is in the L1) instead of kept: twelve registers that the loop does not have — with them live
                            // across the test the kernels spill inside the loop.
                            uint off2 = toff; asm volatile("" : "+v"(off2));      // (an address the compiler cannot match with the loads above)
                            const f32x4 rx = *reinterpret_cast<const f32x4*>(trisBase + off2), ry = *reinterpret_cast<const f32x4*>(trisBase + (off2 + 16u)), rz = *reinterpret_cast<const f32x4*>(trisBase + (off2 + 32u));
                            const uint prim = __float_as_uint(rx.w), flags = __float_as_uint(ry.w);
                            bool c;
                            if (ANYHIT) {
                                c = t8_tri_box_accepts(rx, ry, rz, rz.w, o.x, o.y, o.z, ix, iy, iz, t);
                                if (c && (flags & 1u)) { if (COUNT && !(flags & 2u)) alphaRan = true; c = !(flags & 2u) && alpha_test_slot(sc, slot0 + h + T8_LANES * r, u, v); }      // AlphaTestVisibilityRay (BridgeDonut:981-989)
                            } else {
                                c = ((t < bestT) || (t == bestT && prim < bestPrim)) && ((t < lt) || (t == lt && prim < lp));
                                if (c) c = t8_tri_box_accepts(rx, ry, rz, rz.w, o.x, o.y, o.z, ix, iy, iz, t);
                                if (c && (flags & 1u)) { if (COUNT) alphaRan = true; c = alpha_test_slot(sc, slot0 + h + T8_LANES * r, u, v); }
                            }
                            if (c) { lt = t; lp = prim; lu = u; lv = v; }
                        }
                    }
                }
            }
            const bool cand = (lp != 0xFFFFFFFFu);
            pend = pend1; pend1 = pend2; pend2 = BVH_EMPTY;
            uint candBits = pair_bits(t8_ballot(cand), pl);
#ifndef T8_PROBE_INSTANCE_SWITCHES
            T8_EVENT(4, alphaRan);
#endif
            T8_EVENT(5, candBits != 0u);
            if (candBits) {
                if (ANYHIT) {
                    if (h == (uint)__ffs((int)candBits) - 1u) { HitInfo hh; hh.t = lt; hh.prim = lp; hh.u = hh.v = 0.f; commit(tag, hh); }
                    active = false;
                } else {
                    if (cand && !TASKS) { minePrim = lp; mineUV[threadIdx.x] = make_float2(lu, lv); }
                    float tk = lt; uint pk = lp;      // lexicographic min of (t, prim) over the pair
                    {   float ot = dpp_f<DPP_QP_XOR1>(tk); uint op = dpp_u<DPP_QP_XOR1>(pk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; }
                    bestT = tk; bestPrim = pk;
                }
            }
        }

        if (COUNT) tc3 = __builtin_readcyclecounter();
        // ---- slot bookkeeping: a leaf reached by the descent moves to the free leaf slot; an empty node slot pops the stack
        if (active) {
            {   // the first free leaf slot takes the leaf: compares and selects, no branches (profiles/r05u_queue_select_ab.txt)
                const bool isLeaf = (cur & BVH_LEAF_BIT) && cur != BVH_EMPTY;
                const bool e0 = pend == BVH_EMPTY, e1 = (T8_LEAF_QUEUE > 1) && pend1 == BVH_EMPTY, e2 = (T8_LEAF_QUEUE > 2) && pend2 == BVH_EMPTY;
                const bool to0 = isLeaf && e0, to1 = isLeaf && !e0 && e1, to2 = isLeaf && !e0 && !e1 && e2;
                pend = to0 ? cur : pend; pend1 = to1 ? cur : pend1; pend2 = to2 ? cur : pend2;
                cur = (to0 || to1 || to2) ? BVH_EMPTY : cur;
            }
            if (cur == BVH_EMPTY) {
                T8_EVENT(6, true);
                while (sp > 0u) {
                    T8_EVENT(7, true);
                    sp--;
                    uint2 e;
                    if (sp < BVH8_STACK) e = stack[sp];
                    else {
                        unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(spill_slot(sp)));
                        e = make_uint2((uint)w, (uint)(w >> 32));
                    }
                    if (ANYHIT || __uint_as_float(e.y) <= bestT) { cur = e.x; break; }
                }
                if (cur == BVH_EMPTY && pend == BVH_EMPTY) {          // nothing left: report
                    if (COUNT && h == 0u && ctr.rayIterHist) {
                        if (rayIters > 2048u) { uint k = atomicAdd(ctr.longRayCount, 1u); if (k < 32u) { float* r = ctr.longRays + 8u * k; r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = Sx; r[4] = Sy; r[5] = __uint_as_float(axes); r[6] = (float)rayIters; r[7] = __uint_as_float(tag); } }
                        if (rayIters >= 128u) { uint bin = 31u - (uint)__clz((int)rayIters); atomicAdd(&ctr.rayIterHist[bin < 15u ? bin : 15u], 1ull); }
                    }
                    if (TASKS) {
                        if (ANYHIT) { /* visible sub-tree: nothing to report */ }
                        else if (h == 0u && (bestPrim != taskPrim0 || bestT != taskT0)) { HitInfo hh; hh.t = bestT; hh.prim = bestPrim; hh.u = hh.v = 0.f; commit(tag, hh); }
                    }
                    else if (ANYHIT) { if (h == 0u) { HitInfo hh; hh.t = tmax; hh.prim = 0xFFFFFFFFu; hh.u = hh.v = 0.f; commit(tag, hh); } }
                    else if (bestPrim == 0xFFFFFFFFu) { if (h == 0u) { HitInfo hh; hh.t = bestT; hh.prim = 0xFFFFFFFFu; hh.u = hh.v = 0.f; commit(tag, hh); } }
                    else if (minePrim == bestPrim) { float2 uv = mineUV[threadIdx.x]; HitInfo hh; hh.t = bestT; hh.prim = bestPrim; hh.u = uv.x; hh.v = uv.y; commit(tag, hh); }
                    active = false;
                }
            }
        }
        if (COUNT) { unsigned long long tc4 = __builtin_readcyclecounter(); ctr.cyc[0] += tc1 - tc0; ctr.cyc[1] += tc2 - tc1; ctr.cyc[2] += tc3 - tc2; ctr.cyc[3] += tc4 - tc3; }
        }       // run
    }
    if (CAN_SPLIT && splitNow)
    // ---- every ray still in flight becomes a list of sub-tree tasks: node slot, postponed leaves, stack entries
    {
        const uint nSlots = (cur != BVH_EMPTY ? 1u : 0u) + (pend != BVH_EMPTY ? 1u : 0u) + (pend1 != BVH_EMPTY ? 1u : 0u) + (pend2 != BVH_EMPTY ? 1u : 0u);
        const uint n = active ? nSlots + sp : 0u;
        if (DEFER) { if (active && h == 0u) publish(tag, bestT, bestPrim); return; }
        uint base = 0u;
        if (h == 0u && n) base = atomicAdd(taskOut.count, n);
        base = dpp_u<DPP_PAIR_LO>(base);
        const bool fits = n && (base + n <= taskOut.capacity);
        if (active && !fits && h == 0u) atomicOr(overflowFlag, 2u);
        if (active && fits) {
            uint* tq = reinterpret_cast<uint*>(taskOut.tasks);
            if (h == 0u) {
                uint k = base;
                if (cur != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = cur; tq[4u * k + 2u] = 0u; k++; }
                if (pend != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = pend; tq[4u * k + 2u] = 0u; k++; }
                if (pend1 != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = pend1; tq[4u * k + 2u] = 0u; k++; }
                if (pend2 != BVH_EMPTY) { tq[4u * k] = tag; tq[4u * k + 1u] = pend2; tq[4u * k + 2u] = 0u; k++; }
                publish(tag, bestT, bestPrim);
            }
            for (uint i = h; i < sp; i += T8_LANES) {
                uint2 e;
                if (i < BVH8_STACK) e = stack[i];
                else { unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(sc.travSpill + ((size_t)(blockIdx.x * T8_GROUPS_PER_BLOCK + grp) * T8_SPILL_DEPTH + (i - BVH8_STACK)))); e = make_uint2((uint)w, (uint)(w >> 32)); }
                const uint k = base + nSlots + i;
                tq[4u * k] = tag; tq[4u * k + 1u] = e.x; tq[4u * k + 2u] = e.y;
            }
        }
    }
}

} // namespace ptk
