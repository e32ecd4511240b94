// mi355pt — cooperative BVH8 traversal for wave64: what the traversal kernel (pt_traverse8p.h: two lanes per ray, 32 rays per wave) shares with its callers — counters, tuning
// constants, the candidate's box test, DPP helpers and the description of the template interface. Replaces RayQuery::TraceRayInline / the DXR any-hit visibility query
// (PathTracerBridgeDonut.hlsli:993-1055). Results are traversal-order free (min t, ties to the lower primitive id).
//
// How the shape was arrived at on MI355X (profiles/r01a_*, r01b_*, r03s_*):
//  * one ray per lane over BVH2 was bound by divergent 16-byte gathers through the per-CU texture-address path (33 % L2 misses but only
//    ~0.4 TB/s of HBM traffic): 4 line lookups per lane per node. Cooperative 128 B nodes cut line lookups per ray by ~6-9x.
//  * 8 lanes per ray / 1 child per lane then turned out VALU-issue bound: every per-ray scalar (addresses, stack, control) is replicated over the lanes of its
//    group, so the cure is fewer lanes per ray: 4 lanes x 2 children per lane (rounds 1-3), then 2 lanes x 4 children (round 3 on; the four-lane kernel is in the history).
#pragma once
#include <hip/hip_runtime.h>
#include "pt_scene.h"

namespace ptk {

struct Traverse8Counters { uint nodeVisits, triTests, leafVisits, iters, leafBlocks; uint ev[8]; unsigned long long cyc[4]; unsigned long long* rayIterHist; uint* longRayCount; float* longRays; };

#define T8_EVENT(k, cond) do { if (COUNT) { unsigned long long m_ = t8_ballot(cond); if (m_ && lane == (uint)__ffsll((long long)m_) - 1u) ctr.ev[k]++; } } while (0)
static const uint T8_RAY_STRIDE = 13, T8_TASK_STRIDE = 15;      // origin, shear + axes, tag, interval (fixed-range launches: the slab test's byte selectors) | best hit, (task: best primitive, start node,) the three reciprocals                                              // per-wave LDS parking lot for a chunk's rays / tasks (odd strides)
static const uint T8_RAYBUF_WORDS = (T8_BLOCK / 64u) * T8_CHUNK * T8_RAY_STRIDE, T8_TASKBUF_WORDS = (T8_BLOCK / 64u) * T8_CHUNK * T8_TASK_STRIDE;
static const uint T8_LEAF_ROUNDS = (BVH_MAX_LEAF + T8_LANES - 1u) / T8_LANES;

#ifndef T8_EXTEND_MIN_BLOCKS
#define T8_EXTEND_MIN_BLOCKS 7   // waves per SIMD the register allocator must leave room for in k_extend (64 VGPRs, 2 spilled outside the loop): 6 -> 7 -> 8 waves were
                                  // 1161 -> 1187 -> 1239 Mrays/s in round 2 (profiles/r02n_occupancy_ab.txt). At 8 waves the loop is VALU-issue bound (round 3: extra v_nop
                                  // slots lengthen it one for one, profiles/r03i_valu_bound_probe.txt): from here on instructions per ray count, not waves in flight
#endif
#ifndef T8_SHADOW_MIN_WAVES
#define T8_SHADOW_MIN_WAVES 7    // the same for k_shadow
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
#define T8_LEAF_QUEUE 3        // postponed leaves a ray may hold (1..3) before it has to wait for the wave's next leaf block (A/B: within noise on extend, -4 % on shadow)
#endif

// the ray's reciprocal direction is the correctly rounded one of the hit definition (pt_scene.h tri_box_accepts): inner nodes and the triangle's own
// box are then tested with the same arithmetic, which is what makes the closest hit independent of the tree (three divisions per ray, not per node)
__device__ __forceinline__ float t8_rcp_dir(float d) { return ray_safe_rcp(d); }
// tri_box_accepts (pt_scene.h) with the hardware's min/max (v_min3/v_max3 instead of compare + select pairs), on the leaf block's operands: g0 / g1 / g2 are the record's 16-byte
// groups of the axes x / y / z (three coordinates each, one per vertex), o* and i* the ray's origin and reciprocal direction. The operands are finite here —
// the triangle test has already accepted the triangle, so neither the ray nor the vertices hold a NaN — and on finite operands minNum / maxNum differ from `(a < b) ? a : b` only
// in the sign of a zero, which no comparison below can see; the maximum / minimum over the three axes does not depend on their order: same boolean as tri_box_accepts, bit for bit.
__device__ __forceinline__ bool t8_tri_box_accepts(f32x4 g0, f32x4 g1, f32x4 g2, float pad, float okx, float oky, float okz, float ikx, float iky, float ikz, float t) {
    const float mnx = fminf(g0.x, fminf(g0.y, g0.z)) - pad, mny = fminf(g1.x, fminf(g1.y, g1.z)) - pad, mnz = fminf(g2.x, fminf(g2.y, g2.z)) - pad;
    const float mxx = fmaxf(g0.x, fmaxf(g0.y, g0.z)) + pad, mxy = fmaxf(g1.x, fmaxf(g1.y, g1.z)) + pad, mxz = fmaxf(g2.x, fmaxf(g2.y, g2.z)) + pad;
    const float ax = (mnx - okx) * ikx, bx = (mxx - okx) * ikx, ay = (mny - oky) * iky, by = (mxy - oky) * iky, az = (mnz - okz) * ikz, bz = (mxz - okz) * ikz;
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
// (the traversal itself: pt_traverse8p.h, two lanes per ray; the four-lane kernel of rounds 1-3 is in the history: commit af4c2b2)

} // namespace ptk
