// mi355pt — the realtime mode's pre-pass on the device: stable planes, stable radiance and the guide buffers of a frame (second half).
// Reference: PathTracerSample.hlsl:200-250 compiled with PATH_TRACER_MODE_BUILD_STABLE_PLANES (Sample.cpp:2456-2473 dispatches it once per frame before the noisy passes).
// Wavefront form: one pool slot per pixel; a slot walks the pixel's delta tree plane after plane exactly as the reference's raygen loop does (postProcessHit hands the slot the
// next enqueued branch), so a pass of the loop below is one vertex of every pixel that is still exploring. The closest-hit queries are the reference mode's own traversal
// launches (launch_extend: k_extend + straggler rounds); k_sp_build_shade is the pass's hit / miss shader. Only delta paths are followed: no random numbers beyond the camera ray,
// no NEE, no shadow rays, no Russian roulette. The pass is a few rays per pixel; its kernels are written for clarity, not tuned.
#include "pt_stableplanes_launch.h"

namespace ptk {

__device__ __forceinline__ void sp_store_path(const PathPool& pool, uint i, const PathState& p) {
    pool.s0[i] = make_uint4(asuint(p.origin.x), asuint(p.origin.y), asuint(p.origin.z), p.id);
    pool.s1[i] = make_uint4(asuint(p.dir.x), asuint(p.dir.y), asuint(p.dir.z), asuint(p.sceneLength));
    pool.s2[i] = make_uint4(p.pack23[0], p.pack23[1], p.pack45[0], p.pack45[1]);
    pool.s3[i] = make_uint4(p.interiorList.slots[0], p.interiorList.slots[1], p.packedCounters, p.rayCone.widthSpreadAngleFP16);
    pool.s4[i] = make_uint4(p.pack0, p.pack1, p.flagsAndVertexIndex, p.sampleIndex);
}
__device__ __forceinline__ PathState sp_load_path(const PathPool& pool, uint i) {
    PathState p;
    uint4 a = pool.s0[i], b = pool.s1[i], c = pool.s2[i], d = pool.s3[i], e = pool.s4[i];
    p.origin = make_float3(asfloat(a.x), asfloat(a.y), asfloat(a.z)); p.id = a.w;
    p.dir = make_float3(asfloat(b.x), asfloat(b.y), asfloat(b.z)); p.sceneLength = asfloat(b.w);
    p.pack23[0] = c.x; p.pack23[1] = c.y; p.pack45[0] = c.z; p.pack45[1] = c.w;
    p.interiorList.slots[0] = d.x; p.interiorList.slots[1] = d.y; p.packedCounters = d.z; p.rayCone.widthSpreadAngleFP16 = d.w;
    p.pack0 = e.x; p.pack1 = e.y; p.flagsAndVertexIndex = e.z; p.sampleIndex = e.w;
    return p;
}

template <class PKC>
__global__ void __launch_bounds__(256) k_sp_generate(PKC k, StablePlanesContext sp, PathPool pool, const uint* __restrict__ ownedPixels, uint numOwned, uint sampleIndex, uint* __restrict__ queue) {
    const uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= numOwned) return;
    const uint px = ownedPixels[i];
    const StablePlanesBuilder<PKC> b{k, sp, sampleIndex};
    PathState p = b.generate(px >> 16, px & 0xFFFFu);
    sp_store_path(pool, i, p);
    queue[i] = i;
}

// the pass's closest-hit / miss shader + postProcessHit; the slots that still explore are appended to the next queue (one atomic per wave)
template <class PKC>
__global__ void __launch_bounds__(256) k_sp_build_shade(PKC k, StablePlanesContext sp, PathPool pool, const uint* __restrict__ queueIn, const uint* __restrict__ countInPtr,
                                                        uint* __restrict__ queueOut, uint* countOutPtr, uint sampleIndex, WaveCounters* wc) {
    const uint count = *countInPtr;
    const uint i = blockIdx.x * 256u + threadIdx.x;
    bool alive = false, isHit = false; uint p = 0;
    if (i < count) {
        p = queueIn[i];
        PathState path = sp_load_path(pool, p);
        const uint4 hr = pool.hit[p];
        const float3 rayOrigin = path.origin, rayDir = path.dir;
        const StablePlanesBuilder<PKC> b{k, sp, sampleIndex};
        if (hr.y == 0xFFFFFFFFu) b.HandleMiss(path, rayOrigin, rayDir, kMaxRayTravel);
        else { isHit = true; b.HandleHit(path, rayOrigin, rayDir, hr.y, asfloat(hr.x), asfloat(hr.z), asfloat(hr.w)); }
        b.postProcessHit(path);
        sp_store_path(pool, p, path);
        alive = path.isActive();
    }
    const unsigned long long mAlive = __builtin_amdgcn_ballot_w64(alive), mHit = __builtin_amdgcn_ballot_w64(isHit);
    const uint lane = threadIdx.x & 63u;
    uint base = 0;
    if (lane == 0u) { if (mAlive) base = atomicAdd(countOutPtr, (uint)__popcll(mAlive)); if (mHit) atomicAdd(&wc->hits, (unsigned long long)__popcll(mHit)); }
    base = __shfl(base, 0);
    if (alive) queueOut[base + (uint)__popcll(mAlive & ((1ull << lane) - 1ull))] = p;
}

void launch_sp_generate(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleIndex, uint* queue, hipStream_t st) {
    const dim3 g((numOwned + 255u) / 256u), b(256);
    if (k.S.useFp16Types) { PathKernelContextT<true> k16; __builtin_memcpy(&k16, &k, sizeof(k16)); hipLaunchKernelGGL((k_sp_generate<PathKernelContextT<true>>), g, b, 0, st, k16, sp, pool, ownedPixels, numOwned, sampleIndex, queue); }
    else hipLaunchKernelGGL((k_sp_generate<PathKernelContext>), g, b, 0, st, k, sp, pool, ownedPixels, numOwned, sampleIndex, queue);
}
void launch_sp_build_shade(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr,
                           uint sampleIndex, WaveCounters* wc, hipStream_t st) {
    const dim3 g((countIn + 255u) / 256u), b(256);
    if (k.S.useFp16Types) { PathKernelContextT<true> k16; __builtin_memcpy(&k16, &k, sizeof(k16));
        hipLaunchKernelGGL((k_sp_build_shade<PathKernelContextT<true>>), g, b, 0, st, k16, sp, pool, queueIn, countInPtr, queueOut, countOutPtr, sampleIndex, wc); }
    else hipLaunchKernelGGL((k_sp_build_shade<PathKernelContext>), g, b, 0, st, k, sp, pool, queueIn, countInPtr, queueOut, countOutPtr, sampleIndex, wc);
}

} // namespace ptk
