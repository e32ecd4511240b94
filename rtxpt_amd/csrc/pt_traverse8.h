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

struct Traverse8Counters { uint nodeVisits, triTests; };
static const uint T8_GROUPS_PER_BLOCK = 32, T8_BLOCK = 256, T8_CHUNK = 64;

__device__ __forceinline__ float t8_rcp_dir(float d) {
    float a = fabsf(d);
    float s = (a < 7.888609e-31f) ? 7.888609e-31f : a;
    return 1.0f / ((d < 0.0f) ? -s : s);
}
__device__ __forceinline__ uint group_bits(unsigned long long m, uint gl) { return (uint)(m >> gl) & 0xFFu; }
// In-group (8 lanes) data exchange with DPP modifiers only — no LDS crossbar (ds_bpermute) latency on the critical path.
// xor-1/2/3 are quad permutes; row_half_mirror maps lane i -> 7-i (= i^7), so i^m for m = 4..7 is half_mirror followed by the quad permute of 7^m.
#define DPP_QP_XOR1 0xB1
#define DPP_QP_XOR2 0x4E
#define DPP_QP_XOR3 0x1B
#define DPP_HALF_MIRROR 0x141
template <int CTRL> __device__ __forceinline__ uint dpp_u(uint v) { return (uint)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }

// compile-time knobs (A/B tested on the GPU, see profiles/)
#ifndef T8_ORDERED_PUSH
#define T8_ORDERED_PUSH 1        // 1: far-to-near ordered pushes (7 in-group shuffles); 0: nearest followed, rest pushed in lane order
#endif
#ifndef T8_INNER_REPEAT
#define T8_INNER_REPEAT 1        // inner-node steps per outer iteration before leaves are serviced ("while-while" when > 1)
#endif

// Src: uint fetch(uint i, float3& o, float3& d, float& tmin, float& tmax) -> user tag (e.g. path index); called by all 8 lanes of a group
// Dst: void commit(uint tag, const HitInfo& h) ; called by the group leader only (closest: best hit or prim == ~0; any-hit: prim != ~0 when occluded)
template <bool ANYHIT, bool COUNT, class Src, class Dst>
__device__ __forceinline__ void traverse8_persistent(const DeviceScene& sc, uint count, uint2* stackBase, Src fetch, Dst commit, Traverse8Counters& ctr, uint* overflowFlag) {
    const uint lane = threadIdx.x & 63u, j = lane & 7u, gl = lane & ~7u;
    const uint grp = threadIdx.x >> 3;
    uint2* stack = stackBase + grp * BVH8_STACK_STRIDE;
    const uint wavesPerBlock = T8_BLOCK / 64u;
    const uint waveId = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6), numWaves = gridDim.x * wavesPerBlock;
    const char* nodes = reinterpret_cast<const char*>(sc.nodes8);
    const char* tris = reinterpret_cast<const char*>(sc.tris);
    const float INF = __uint_as_float(0x7F800000u);

    uint chunk = waveId;
    uint chunkPos = chunk * T8_CHUNK, chunkEnd = (chunkPos + T8_CHUNK < count) ? chunkPos + T8_CHUNK : count;
    bool exhausted = (chunkPos >= count) || !sc.rootIsValid;
    if (!sc.rootIsValid && waveId == 0 && count) {            // empty scene: every ray misses
        for (uint i = lane; i < count; i += 64u) { float3 o, d; float a, b; uint tag = fetch(i, o, d, a, b); HitInfo h; h.t = b; h.prim = 0xFFFFFFFFu; h.u = h.v = 0.f; commit(tag, h); }
    }
    bool active = false;
    float3 o = make_float3(0.f), d = make_float3(0.f), id = make_float3(0.f);
    float tmin = 0.f, tmax = 0.f;
    HitInfo best; best.t = 0.f; best.prim = 0xFFFFFFFFu; best.u = best.v = 0.f;
    uint cur = 0, sp = 0, tag = 0;

    // pop the next node whose entry distance can still matter; finishes the ray when the stack is empty
    auto pop = [&]() {
        while (true) {
            if (sp == 0u) { if (j == 0u) { if (ANYHIT) { best.prim = 0xFFFFFFFFu; } commit(tag, best); } active = false; break; }
            sp--;
            uint2 e = stack[sp];
            if (ANYHIT || __uint_as_float(e.y) <= best.t) { cur = e.x; break; }
        }
    };

    while (true) {
        // ---- refill idle groups from the wave's current chunk
        bool need = !active && !exhausted;
        unsigned long long needMask = __ballot(need && j == 0u);
        if (needMask) {
            if (chunkPos >= chunkEnd) {
                chunk += numWaves; chunkPos = chunk * T8_CHUNK; chunkEnd = (chunkPos + T8_CHUNK < count) ? chunkPos + T8_CHUNK : count;
                if (chunkPos >= count) { chunkPos = chunkEnd = count; }
            }
            uint avail = chunkEnd - chunkPos;
            if (avail == 0u) { if (need) exhausted = true; }
            else {
                uint rank = (uint)__popcll(needMask & ((1ull << gl) - 1ull));        // rank of my group among the needing groups
                uint n = (uint)__popcll(needMask);
                if (need && rank < avail) {
                    tag = fetch(chunkPos + rank, o, d, tmin, tmax);
                    id = make_float3(t8_rcp_dir(d.x), t8_rcp_dir(d.y), t8_rcp_dir(d.z));
                    best.t = tmax; best.prim = 0xFFFFFFFFu; best.u = 0.f; best.v = 0.f;
                    cur = 0u; sp = 0u; active = true;
                }
                chunkPos += (n < avail) ? n : avail;
            }
        }
        if (__ballot(active) == 0ull) { if (__ballot(!exhausted) == 0ull) break; else continue; }

        // ---- inner nodes: lane j tests child j
#pragma unroll 1
        for (int rep = 0; rep < T8_INNER_REPEAT; rep++) {
            bool inner = active && !(cur & BVH_LEAF_BIT);
            if (T8_INNER_REPEAT > 1 && __ballot(inner) == 0ull) break;
            if (inner) {
                const char* np = nodes + (size_t)cur * 128u;
                const u32x4 hdr = *reinterpret_cast<const u32x4*>(np);
                const uint cref = *reinterpret_cast<const uint*>(np + 16u + 4u * j);
                const u32x2 q = *reinterpret_cast<const u32x2*>(np + 48u + 8u * j);
                if (COUNT && j == 0u) ctr.nodeVisits++;
                const float sx = __uint_as_float((hdr.w & 0xFFu) << 23), sy = __uint_as_float(((hdr.w >> 8) & 0xFFu) << 23), sz = __uint_as_float(((hdr.w >> 16) & 0xFFu) << 23);
                const float ox = __uint_as_float(hdr.x), oy = __uint_as_float(hdr.y), oz = __uint_as_float(hdr.z);
                const float lox = ox + (float)(q.x & 0xFFu) * sx, loy = oy + (float)((q.x >> 8) & 0xFFu) * sy, loz = oz + (float)((q.x >> 16) & 0xFFu) * sz;
                const float hix = ox + (float)(q.x >> 24) * sx, hiy = oy + (float)(q.y & 0xFFu) * sy, hiz = oz + (float)((q.y >> 8) & 0xFFu) * sz;
                float tx1 = (lox - o.x) * id.x, tx2 = (hix - o.x) * id.x;
                float ty1 = (loy - o.y) * id.y, ty2 = (hiy - o.y) * id.y;
                float tz1 = (loz - o.z) * id.z, tz2 = (hiz - o.z) * id.z;
                float tn = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), tmin));
                float tf = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fminf(fmaxf(tz1, tz2), best.t));
                bool hit = (cref != BVH_EMPTY) && (tn <= tf * 1.0000005f);
                float key = hit ? tn : INF;
                uint hitBits = group_bits(__ballot(hit), gl);
                uint nhit = (uint)__popc(hitBits);
                if (nhit == 0u) pop();
                else if (T8_ORDERED_PUSH) {
                    // rank of my child among the hit children (ties to the lower lane): compare against lanes j^1 .. j^7
                    const float hm = dpp_f<DPP_HALF_MIRROR>(key);
                    const float k1 = dpp_f<DPP_QP_XOR1>(key), k2 = dpp_f<DPP_QP_XOR2>(key), k3 = dpp_f<DPP_QP_XOR3>(key);
                    const float k4 = dpp_f<DPP_QP_XOR3>(hm), k5 = dpp_f<DPP_QP_XOR2>(hm), k6 = dpp_f<DPP_QP_XOR1>(hm), k7 = hm;
                    uint rank = 0;
                    rank += ((k1 < key) || (k1 == key && (j ^ 1u) < j)) ? 1u : 0u;
                    rank += ((k2 < key) || (k2 == key && (j ^ 2u) < j)) ? 1u : 0u;
                    rank += ((k3 < key) || (k3 == key && (j ^ 3u) < j)) ? 1u : 0u;
                    rank += ((k4 < key) || (k4 == key && (j ^ 4u) < j)) ? 1u : 0u;
                    rank += ((k5 < key) || (k5 == key && (j ^ 5u) < j)) ? 1u : 0u;
                    rank += ((k6 < key) || (k6 == key && (j ^ 6u) < j)) ? 1u : 0u;
                    rank += ((k7 < key) || (k7 == key && (j ^ 7u) < j)) ? 1u : 0u;
                    // the nearest child's reference reaches every lane through a 3-step butterfly (rank 0 holds it)
                    uint next = (hit && rank == 0u) ? cref : 0u;
                    next |= dpp_u<DPP_QP_XOR1>(next); next |= dpp_u<DPP_QP_XOR2>(next); next |= dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(next));
                    if (nhit > 1u) {
                        if (sp + nhit - 1u > BVH8_STACK) { if (j == 0u) atomicOr(overflowFlag, 1u); }
                        else { if (hit && rank > 0u) stack[sp + (nhit - 1u - rank)] = make_uint2(cref, __float_as_uint(tn)); sp += nhit - 1u; }
                    }
                    cur = next;
                } else {
                    float mk = fminf(key, dpp_f<DPP_QP_XOR1>(key)); mk = fminf(mk, dpp_f<DPP_QP_XOR2>(mk)); mk = fminf(mk, dpp_f<DPP_QP_XOR3>(dpp_f<DPP_HALF_MIRROR>(mk)));
                    uint nearBits = group_bits(__ballot(hit && key == mk), gl);
                    uint nearLane = (uint)__ffs((int)nearBits) - 1u;
                    uint next = (j == nearLane) ? cref : 0u;
                    next |= dpp_u<DPP_QP_XOR1>(next); next |= dpp_u<DPP_QP_XOR2>(next); next |= dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(next));
                    uint pushBits = hitBits & ~(1u << nearLane);
                    uint npush = nhit - 1u;
                    if (npush) {
                        if (sp + npush > BVH8_STACK) { if (j == 0u) atomicOr(overflowFlag, 1u); }
                        else { if ((pushBits >> j) & 1u) stack[sp + (uint)__popc(pushBits & ((1u << j) - 1u))] = make_uint2(cref, __float_as_uint(tn)); sp += npush; }
                    }
                    cur = next;
                }
            }
        }

        // ---- leaves: lane j tests triangle j
        if (active && (cur & BVH_LEAF_BIT)) {
            const uint first = (cur & 0x7FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
            bool cand = false; float t = 0.f, u = 0.f, v = 0.f; uint prim = 0xFFFFFFFFu;
            if (j < cnt) {
                const char* tp = tris + (size_t)(first + j) * 48u;
                const f32x4 a = *reinterpret_cast<const f32x4*>(tp), b = *reinterpret_cast<const f32x4*>(tp + 16), c = *reinterpret_cast<const f32x4*>(tp + 32);
                TriRecord tr; tr.v0 = make_float3(a.x, a.y, a.z); tr.prim = __float_as_uint(a.w);
                tr.e1 = make_float3(b.x, b.y, b.z); tr.flags = __float_as_uint(b.w); tr.e2 = make_float3(c.x, c.y, c.z);
                if (COUNT) ctr.triTests++;
                if (intersect_tri(tr, o, d, tmin, tmax, t, u, v)) {
                    prim = tr.prim;
                    if (ANYHIT) {
                        cand = true;
                        if (tr.flags & 1u) cand = !(tr.flags & 2u) && alpha_test(sc, prim, u, v);      // AlphaTestVisibilityRay (BridgeDonut:981-989)
                    } else {
                        cand = (t < best.t) || (t == best.t && prim < best.prim);
                        if (cand && (tr.flags & 1u)) cand = alpha_test(sc, prim, u, v);
                    }
                }
            }
            uint candBits = group_bits(__ballot(cand), gl);
            bool finished = false;
            if (candBits) {
                if (ANYHIT) {
                    uint wl = (uint)__ffs((int)candBits) - 1u;
                    float wt = (j == wl) ? t : 0.f; uint wp = (j == wl) ? prim : 0u;      // butterfly-OR broadcast from the first candidate lane
                    uint wtb = __float_as_uint(wt);
                    wtb |= dpp_u<DPP_QP_XOR1>(wtb); wtb |= dpp_u<DPP_QP_XOR2>(wtb); wtb |= dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(wtb));
                    wp |= dpp_u<DPP_QP_XOR1>(wp); wp |= dpp_u<DPP_QP_XOR2>(wp); wp |= dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(wp));
                    best.t = __uint_as_float(wtb); best.prim = wp;
                    if (j == 0u) commit(tag, best);
                    active = false; finished = true;
                } else {
                    // lexicographic min of (t, prim) over the group, carrying (u, v): 3 butterfly steps, branch-free
                    float tk = cand ? t : INF; uint pk = cand ? prim : 0xFFFFFFFFu; float uk = u, vk = v;
                    {   float ot = dpp_f<DPP_QP_XOR1>(tk); uint op = dpp_u<DPP_QP_XOR1>(pk); float ou = dpp_f<DPP_QP_XOR1>(uk), ov = dpp_f<DPP_QP_XOR1>(vk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; uk = take ? ou : uk; vk = take ? ov : vk; }
                    {   float ot = dpp_f<DPP_QP_XOR2>(tk); uint op = dpp_u<DPP_QP_XOR2>(pk); float ou = dpp_f<DPP_QP_XOR2>(uk), ov = dpp_f<DPP_QP_XOR2>(vk);
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; uk = take ? ou : uk; vk = take ? ov : vk; }
                    {   float ot = dpp_f<DPP_QP_XOR3>(dpp_f<DPP_HALF_MIRROR>(tk)); uint op = dpp_u<DPP_QP_XOR3>(dpp_u<DPP_HALF_MIRROR>(pk));
                        float ou = dpp_f<DPP_QP_XOR3>(dpp_f<DPP_HALF_MIRROR>(uk)), ov = dpp_f<DPP_QP_XOR3>(dpp_f<DPP_HALF_MIRROR>(vk));
                        bool take = (ot < tk) || (ot == tk && op < pk); tk = take ? ot : tk; pk = take ? op : pk; uk = take ? ou : uk; vk = take ? ov : vk; }
                    best.t = tk; best.prim = pk; best.u = uk; best.v = vk;
                }
            }
            if (!finished) pop();
        }
    }
}

} // namespace ptk
