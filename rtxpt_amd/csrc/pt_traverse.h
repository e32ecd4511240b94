// mi355pt — BVH2 traversal for wave64: one ray per lane, per-lane node stack staged in LDS (column layout:
// entry e of thread t lives at stack[e * blockDim + t], so the 64 lanes of a wave always touch 64 consecutive banks).
// Replaces RayQuery::TraceRayInline (closest hit, BridgeDonut:1029-1055) and the ACCEPT_FIRST_HIT visibility query
// (BridgeDonut:993-1027). Results are traversal-order independent: the closest hit is min(t) with ties broken towards the
// lower global primitive id, exactly like the CPU oracle, so any exact BVH over the same triangles returns the same record.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_scene.h"

namespace ptk {

struct TraverseCounters { uint nodeVisits, triTests; };

// conservative slab test; returns entry distance in tEntry. The boxes are padded at build time; the (1 + 2^-21) factor on the
// exit distance keeps the test conservative under rounding (Ize, "Robust BVH ray traversal", JCGT 2013).
__device__ __forceinline__ bool slab_test(float3 bmin, float3 bmax, float3 o, float3 id, float tmin, float tmax, float& tEntry) {
    float tx1 = (bmin.x - o.x) * id.x, tx2 = (bmax.x - o.x) * id.x;
    float ty1 = (bmin.y - o.y) * id.y, ty2 = (bmax.y - o.y) * id.y;
    float tz1 = (bmin.z - o.z) * id.z, tz2 = (bmax.z - o.z) * id.z;
    float tn = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), tmin));
    float tf = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fminf(fmaxf(tz1, tz2), tmax));
    tEntry = tn;
    return tn <= tf * 1.0000005f;
}
__device__ __forceinline__ float safe_rcp_dir(float d) {
    // avoid inf * 0 = NaN on the slab planes: clamp tiny components to +-2^-100 (keeps sign)
    float a = fabsf(d);
    float s = (a < 7.888609e-31f) ? 7.888609e-31f : a;
    return 1.0f / ((d < 0.0f) ? -s : s);
}

template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ HitInfo traverse(const DeviceScene& sc, float3 o, float3 d, float tmin, float tmax, uint* stack, uint stride, TraverseCounters& ctr) {
    HitInfo h; h.t = tmax; h.prim = 0xFFFFFFFFu; h.u = 0.f; h.v = 0.f;
    if (!sc.rootIsValid) return h;
    float3 id = make_float3(safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z));
    uint cur = 0;
    int sp = 0;
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
    while (true) {
        if (!(cur & BVH_LEAF_BIT)) {
            const float4 n0 = nodes4[cur * 4u + 0], n1 = nodes4[cur * 4u + 1], n2 = nodes4[cur * 4u + 2], n3 = nodes4[cur * 4u + 3];
            if (COUNT) ctr.nodeVisits++;
            uint left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
            float tl, tr;
            bool hl = (left != BVH_EMPTY) && slab_test(make_float3(n0.x, n0.y, n0.z), make_float3(n0.w, n1.x, n1.y), o, id, tmin, h.t, tl);
            bool hr = (right != BVH_EMPTY) && slab_test(make_float3(n1.z, n1.w, n2.x), make_float3(n2.y, n2.z, n2.w), o, id, tmin, h.t, tr);
            if (hl && hr) {
                bool leftNear = tl <= tr;
                stack[(uint)sp * stride] = leftNear ? right : left;
                sp++;
                cur = leftNear ? left : right;
                continue;
            }
            if (hl) { cur = left; continue; }
            if (hr) { cur = right; continue; }
        } else {
            uint first = (cur & 0x7FFFFFFFu) >> 3, count = (cur & 7u) + 1u;
            for (uint i = 0; i < count; i++) {
                const float4 a = tris4[(first + i) * 3u + 0], b = tris4[(first + i) * 3u + 1], c = tris4[(first + i) * 3u + 2];
                TriRecord tr; tr.v0 = make_float3(a.x, a.y, a.z); tr.prim = __float_as_uint(a.w);
                tr.e1 = make_float3(b.x, b.y, b.z); tr.flags = __float_as_uint(b.w); tr.e2 = make_float3(c.x, c.y, c.z);
                if (COUNT) ctr.triTests++;
                float t, u, v;
                if (!intersect_tri(tr, o, d, tmin, tmax, t, u, v)) continue;
                if (ANYHIT) {
                    if (tr.flags & 1u) {                    // AlphaTestVisibilityRay (BridgeDonut:981-989)
                        if (tr.flags & 2u) continue;
                        if (!alpha_test(sc, tr.prim, u, v)) continue;
                    }
                    h.t = t; h.prim = tr.prim; h.u = u; h.v = v;
                    return h;
                } else {
                    if (!(t < h.t || (t == h.t && tr.prim < h.prim))) continue;
                    if ((tr.flags & 1u) && !alpha_test(sc, tr.prim, u, v)) continue;
                    h.t = t; h.prim = tr.prim; h.u = u; h.v = v;
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stack[(uint)sp * stride];
    }
    return h;
}

} // namespace ptk
