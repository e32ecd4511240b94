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
    const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);

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
        if (!active) continue;

        bool doPop = false;
        if (!(cur & BVH_LEAF_BIT)) {
            // ---- inner node: lane j tests child j
            const uint4 hdr = *reinterpret_cast<const uint4*>(nodes + (size_t)cur * 128u);
            const uint* cp = reinterpret_cast<const uint*>(nodes + (size_t)cur * 128u + 16u + 12u * j);
            const uint cref = cp[0], q0 = cp[1], q1 = cp[2];
            if (COUNT && j == 0u) ctr.nodeVisits++;
            const float sx = __uint_as_float((hdr.w & 0xFFu) << 23), sy = __uint_as_float(((hdr.w >> 8) & 0xFFu) << 23), sz = __uint_as_float(((hdr.w >> 16) & 0xFFu) << 23);
            const float ox = __uint_as_float(hdr.x), oy = __uint_as_float(hdr.y), oz = __uint_as_float(hdr.z);
            const float lox = ox + (float)(q0 & 0xFFu) * sx, loy = oy + (float)((q0 >> 8) & 0xFFu) * sy, loz = oz + (float)((q0 >> 16) & 0xFFu) * sz;
            const float hix = ox + (float)(q0 >> 24) * sx, hiy = oy + (float)(q1 & 0xFFu) * sy, hiz = oz + (float)((q1 >> 8) & 0xFFu) * sz;
            float tx1 = (lox - o.x) * id.x, tx2 = (hix - o.x) * id.x;
            float ty1 = (loy - o.y) * id.y, ty2 = (hiy - o.y) * id.y;
            float tz1 = (loz - o.z) * id.z, tz2 = (hiz - o.z) * id.z;
            float tn = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), tmin));
            float tf = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fminf(fmaxf(tz1, tz2), best.t));
            bool hit = (cref != BVH_EMPTY) && (tn <= tf * 1.0000005f);
            float key = hit ? tn : __uint_as_float(0x7F800000u);
            uint hitBits = group_bits(__ballot(hit), gl);
            uint nhit = (uint)__popc(hitBits);
            if (nhit == 0u) doPop = true;
            else {
                uint rank = 0;
#pragma unroll
                for (uint k = 1; k < 8u; k++) {
                    uint oj = (j + k) & 7u;
                    float ok = __shfl(key, (int)(gl + oj));
                    rank += ((ok < key) || (ok == key && oj < j)) ? 1u : 0u;
                }
                uint nearBits = group_bits(__ballot(hit && rank == 0u), gl);
                uint nearLane = (uint)__ffs((int)nearBits) - 1u;
                uint next = (uint)__shfl((int)cref, (int)(gl + nearLane));
                if (nhit > 1u) {
                    if (sp + nhit - 1u > BVH8_STACK) { if (j == 0u) atomicOr(overflowFlag, 1u); }
                    else if (hit && rank > 0u) stack[sp + (nhit - 1u - rank)] = make_uint2(cref, __float_as_uint(tn));
                    if (sp + nhit - 1u <= BVH8_STACK) sp += nhit - 1u;
                }
                cur = next;
            }
        } else {
            // ---- leaf: lane j tests triangle j
            const uint first = (cur & 0x7FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
            bool cand = false; float t = 0.f, u = 0.f, v = 0.f; uint prim = 0xFFFFFFFFu;
            if (j < cnt) {
                const float4 a = tris4[(first + j) * 3u + 0u], b = tris4[(first + j) * 3u + 1u], c = tris4[(first + j) * 3u + 2u];
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
            if (candBits) {
                if (ANYHIT) {
                    uint wl = (uint)__ffs((int)candBits) - 1u;
                    best.t = __shfl(t, (int)(gl + wl)); best.prim = (uint)__shfl((int)prim, (int)(gl + wl));
                    if (j == 0u) commit(tag, best);
                    active = false;
                } else {
                    float tk = cand ? t : __uint_as_float(0x7F800000u); uint pk = cand ? prim : 0xFFFFFFFFu;
#pragma unroll
                    for (int m = 1; m < 8; m <<= 1) {
                        float ot = __shfl_xor(tk, m); uint op = (uint)__shfl_xor((int)pk, m);
                        if (ot < tk || (ot == tk && op < pk)) { tk = ot; pk = op; }
                    }
                    uint winBits = group_bits(__ballot(cand && prim == pk), gl);
                    uint wl = (uint)__ffs((int)winBits) - 1u;
                    best.t = tk; best.prim = pk; best.u = __shfl(u, (int)(gl + wl)); best.v = __shfl(v, (int)(gl + wl));
                }
            }
            doPop = active;
        }
        if (doPop) {
            while (true) {
                if (sp == 0u) { if (j == 0u) { if (ANYHIT) { best.prim = 0xFFFFFFFFu; } commit(tag, best); } active = false; break; }
                sp--;
                uint2 e = stack[sp];
                if (ANYHIT || __uint_as_float(e.y) <= best.t) { cur = e.x; break; }
            }
        }
    }
}

} // namespace ptk
