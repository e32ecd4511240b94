// mi355pt — device-side helpers shared by the wavefront kernels (pt_wavefront.hip) and the tail kernel (pt_tail.hip): wave-level queue appends,
// the path pool's SoA load / store, traversal counter set-up.
#pragma once
#include "pt_wavefront.h"
#include "pt_traverse8.h"

namespace ptk {

__device__ __forceinline__ uint lane_id() { return __lane_id(); }
// one atomic per wave: returns this lane's slot if `pred`, garbage otherwise
__device__ __forceinline__ uint wave_append(bool pred, uint* counter) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    uint lane = lane_id();
    uint leader = (uint)__ffsll((long long)mask) - 1u;
    uint base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint)__popcll(mask));
    base = __shfl(base, (int)leader);
    return base + (uint)__popcll(mask & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ void wave_add64(unsigned long long v, unsigned long long* counter) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane_id() == 0 && v) atomicAdd(counter, v);
}

__device__ __forceinline__ void store_path(const PathPool& pool, uint i, const PathState& p) {
    pool.s0[i] = make_uint4(asuint(p.origin.x), asuint(p.origin.y), asuint(p.origin.z), p.id);
    pool.s1[i] = make_uint4(asuint(p.dir.x), asuint(p.dir.y), asuint(p.dir.z), asuint(p.sceneLength));
    pool.s2[i] = make_uint4(p.pack23[0], p.pack23[1], p.pack45[0], p.pack45[1]);
    pool.s3[i] = make_uint4(p.interiorList.slots[0], p.interiorList.slots[1], p.packedCounters, p.rayCone.widthSpreadAngleFP16);
    pool.s4[i] = make_uint4(p.pack0, p.pack1, p.flagsAndVertexIndex, p.sampleIndex);
}
__device__ __forceinline__ PathState load_path(const PathPool& pool, uint i) {
    PathState p;
    uint4 a = pool.s0[i], b = pool.s1[i], c = pool.s2[i], d = pool.s3[i], e = pool.s4[i];
    p.origin = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)); p.id = a.w;
    p.dir = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z)); p.sceneLength = asfloat(b.w);
    p.pack23[0] = c.x; p.pack23[1] = c.y; p.pack45[0] = c.z; p.pack45[1] = c.w;
    p.interiorList.slots[0] = d.x; p.interiorList.slots[1] = d.y; p.packedCounters = d.z; p.rayCone.widthSpreadAngleFP16 = d.w;
    p.pack0 = e.x; p.pack1 = e.y; p.flagsAndVertexIndex = e.z; p.sampleIndex = e.w;
    return p;
}

// k_shade's view of a path (PathKernelContext::HandleHit's IO policy, pt_path.h): the words loadSurface needs come first (direction | length, interior list |
// counters | ray cone), the rest after the surface is loaded; the scattered path's first four groups are stored before the light sampling, the last group at
// the end
struct PathPoolIO {
    static constexpr bool streams = true;
    PathPool pool; uint i;
#ifdef PT_SHADE_PHASE_PROBE      // developer build: cycle stamps at HandleHit's phase boundaries (k_shade adds the differences up in WaveCounters::eventsExt)
    mutable uint tk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    __device__ __forceinline__ void mark(int k) const { tk[k] = (uint)__builtin_readcyclecounter(); }
#else
    __device__ __forceinline__ void mark(int) const {}
#endif
    __device__ __forceinline__ PathState load_first() const {
        PathState p; uint4 b = pool.s1[i], d = pool.s3[i];
        p.dir = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z)); p.sceneLength = asfloat(b.w);
        p.interiorList.slots[0] = d.x; p.interiorList.slots[1] = d.y; p.packedCounters = d.z; p.rayCone.widthSpreadAngleFP16 = d.w;
        // (PF_terminateAtNextBounce decides how much of the surface is loaded; load_rest brings the word again with its group)
        p.flagsAndVertexIndex = reinterpret_cast<const uint*>(pool.s4)[4u * (size_t)i + 2u];
        return p;
    }
    __device__ __forceinline__ void load_rest(PathState& p) const {
        asm volatile("" ::: "memory");      // (not before the surface is loaded: these twelve registers are what the load order is about)
        uint4 a = pool.s0[i], c = pool.s2[i], e = pool.s4[i];
        p.origin = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)); p.id = a.w;
        p.pack23[0] = c.x; p.pack23[1] = c.y; p.pack45[0] = c.z; p.pack45[1] = c.w;
        p.pack0 = e.x; p.pack1 = e.y; p.flagsAndVertexIndex = e.z; p.sampleIndex = e.w;
    }
    __device__ __forceinline__ void store_front(const PathState& p) const {
        pool.s0[i] = make_uint4(asuint(p.origin.x), asuint(p.origin.y), asuint(p.origin.z), p.id);
        pool.s1[i] = make_uint4(asuint(p.dir.x), asuint(p.dir.y), asuint(p.dir.z), asuint(p.sceneLength));
        pool.s2[i] = make_uint4(p.pack23[0], p.pack23[1], p.pack45[0], p.pack45[1]);
        pool.s3[i] = make_uint4(p.interiorList.slots[0], p.interiorList.slots[1], p.packedCounters, p.rayCone.widthSpreadAngleFP16);
    }
    __device__ __forceinline__ void store_back(const PathState& p) const { pool.s4[i] = make_uint4(p.pack0, p.pack1, p.flagsAndVertexIndex, p.sampleIndex); }
    __device__ __forceinline__ void store_all(const PathState& p) const { store_path(pool, i, p); }
};

// ... and of a compacted pool (PathPool::home): the same load order from the position's word groups and the home slot's throughput | radiance; nothing is stored from inside HandleHit —
// the surviving path's position is only known once its block has counted its survivors (k_shade stores it there)
struct PathCompactIO {
    static constexpr bool streams = true;
    PathPool pool; uint i, hp;
    __device__ __forceinline__ void mark(int) const {}
    __device__ __forceinline__ PathState load_first() const {
        PathState p; uint4 b = pool.s1[i], d = pool.s3[i];
        p.dir = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z)); p.sceneLength = asfloat(b.w);
        p.interiorList.slots[0] = d.x; p.interiorList.slots[1] = d.y; p.packedCounters = d.z; p.rayCone.widthSpreadAngleFP16 = d.w;
        p.flagsAndVertexIndex = reinterpret_cast<const uint*>(pool.s4)[4u * (size_t)i + 2u];
        return p;
    }
    __device__ __forceinline__ void load_rest(PathState& p) const {
        asm volatile("" ::: "memory");
        uint4 a = pool.s0[i], c = pool.s2[hp], e = pool.s4[i];
        p.origin = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)); p.id = a.w;
        p.pack23[0] = c.x; p.pack23[1] = c.y; p.pack45[0] = c.z; p.pack45[1] = c.w;
        p.pack0 = e.x; p.pack1 = e.y; p.flagsAndVertexIndex = e.z; p.sampleIndex = e.w;
    }
    __device__ __forceinline__ PathState load_all() const { PathState p = load_first(); load_rest(p); return p; }
    __device__ __forceinline__ void store_front(const PathState&) const {}
    __device__ __forceinline__ void store_back(const PathState&) const {}
    __device__ __forceinline__ void store_all(const PathState&) const {}
};

__device__ __forceinline__ void t8_counters_init(Traverse8Counters& ctr) {
    ctr.nodeVisits = 0; ctr.triTests = 0; ctr.leafVisits = 0; ctr.iters = 0; ctr.leafBlocks = 0; for (int q = 0; q < 8; q++) ctr.ev[q] = 0u; ctr.cyc[0] = ctr.cyc[1] = ctr.cyc[2] = ctr.cyc[3] = 0ull;
    ctr.rayIterHist = nullptr; ctr.longRayCount = nullptr; ctr.longRays = nullptr;
}
// t > 0: bits order like the value
__device__ __forceinline__ unsigned long long t8_hit_key(float t, uint prim) { return ((unsigned long long)__float_as_uint(t) << 32) | prim; }


} // namespace ptk
