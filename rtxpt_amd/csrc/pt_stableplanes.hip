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
    __shared__ uint sCnt[4][2]; __shared__ uint sBase;      // one atomic per block and counter (see k_sp_fill_shade)
    const unsigned long long mAlive = __builtin_amdgcn_ballot_w64(alive), mHit = __builtin_amdgcn_ballot_w64(isHit);
    const uint wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0u) { sCnt[wave][0] = (uint)__popcll(mAlive); sCnt[wave][1] = (uint)__popcll(mHit); }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint tot = 0; for (uint w = 0; w < 4u; w++) { const uint c = sCnt[w][threadIdx.x]; sCnt[w][threadIdx.x] = tot; tot += c; }
        if (threadIdx.x == 0u) sBase = tot ? atomicAdd(countOutPtr, tot) : 0u; else if (tot) atomicAdd(&wc->hits, (unsigned long long)tot);
    }
    __syncthreads();
    if (alive) queueOut[sBase + sCnt[wave][0] + (uint)__popcll(mAlive & ((1ull << lane) - 1ull))] = p;
}

// ---- the noisy (fill) passes: PathTracerSample.hlsl:200-250 with PATH_TRACER_MODE_FILL_STABLE_PLANES, one sub-sample per call of pt_fill_stable_planes.
// A pass of the host loop is one vertex of every live path, as in reference mode: launch_extend, k_sp_fill_shade (hit / miss shader up to the light sample), launch_shadow, k_sp_fill_resolve.
// The visibility rays go through the reference mode's own shadow launches, unchanged: their "visible" action adds the entry's radiance to the L words of pool.s2 — here they are handed a
// scratch copy of that stream and a unit radiance, so a visible entry leaves a mark that k_sp_fill_resolve turns into the float4 increment (total + specular average) this pass needs.
template <class PKC>
__global__ void __launch_bounds__(1024) k_sp_fill_generate(PKC k, StablePlanesContext sp, PathPool pool, const uint* __restrict__ ownedPixels, uint numOwned, uint sampleIndex, uint* __restrict__ queue, uint* countPtr) {
    const uint i = blockIdx.x * 1024u + threadIdx.x;
    bool alive = false;
    if (i < numOwned) {
        const uint px = ownedPixels[i];
        const StablePlanesFiller<PKC> f{k, sp, sampleIndex};
        PathState p = f.generate(px >> 16, px & 0xFFFFu);
        sp_store_path(pool, i, p);
        alive = p.isActive();
        if (alive) { float t0, t1; f.firstHitInterval(px >> 16, px & 0xFFFFu, t0, t1); pool.hit[i] = make_uint4(asuint(t0), asuint(t1), 0u, 0u); }      // the interval of the pass's first traversal launch (launch_extend(..., ranged)): the hit record is free until that launch fills it
    }
    // one atomic per 1024-thread block: with one per wave the 130 000 waves of a 4K frame queue up on the counter's L2 line for 1.3 ms — longer than everything else the kernel does
    // (k_shade's lesson once more; profiles/r04w_fill_kernel_stats_before.csv: 1.36 ms per pass)
    __shared__ uint sCnt[16]; __shared__ uint sBase;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
    const uint lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u) sCnt[wave] = (uint)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0u) { uint tot = 0; for (uint w = 0; w < 16u; w++) { const uint c = sCnt[w]; sCnt[w] = tot; tot += c; } sBase = tot ? atomicAdd(countPtr, tot) : 0u; }
    __syncthreads();
    if (alive) queue[sBase + sCnt[wave] + (uint)__popcll(m & ((1ull << lane) - 1ull))] = i;
}
template <class PKC>
__global__ void __launch_bounds__(256) k_sp_fill_shade(PKC k, StablePlanesContext sp, PathPool pool, const uint* __restrict__ queueIn, const uint* __restrict__ countInPtr, uint* __restrict__ queueOut, uint* countOutPtr,
                                                       ShadowQueue sq, float4* __restrict__ newL, uint sampleIndex, WaveCounters* wc, const uint* __restrict__ classCount) {
    const uint count = *countInPtr;
    const uint i = blockIdx.x * 256u + threadIdx.x;
    bool alive = false, isHit = false; uint p = 0;
    SPNeeRequest req; req.valid = false;
    if (i < count) {
        if (classCount) {                                     // queueIn = k_classify's arrays (pt_wavefront.hip): thread i takes the i-th path of the order {continuing hit, terminating hit, miss}
            const uint nGo = classCount[0], nEnd = classCount[1];
            p = queueIn[i < nGo ? i : (i < nGo + nEnd ? count - 1u - (i - nGo) : count + (i - nGo - nEnd))];
        } else p = queueIn[i];
        PathState path = sp_load_path(pool, p);
        const uint4 hr = pool.hit[p];
        const float3 rayOrigin = path.origin, rayDir = path.dir;
        const StablePlanesFiller<PKC> f{k, sp, sampleIndex};
        if (hr.y == 0xFFFFFFFFu) f.HandleMiss(path, rayDir, kMaxRayTravel);
        else { isHit = true; f.HandleHit(path, rayOrigin, rayDir, hr.y, asfloat(hr.x), asfloat(hr.z), asfloat(hr.w), req); }
        sp_store_path(pool, p, path);
        alive = path.isActive();
    }
    // queue appends with one atomic per BLOCK and counter (the four waves' counts meet in LDS): same-address atomics of every wave of the GPU serialise in the L2 (k_shade's lesson)
    __shared__ uint sCnt[4][3]; __shared__ uint sBase[3];
    const unsigned long long mAlive = __builtin_amdgcn_ballot_w64(alive), mHit = __builtin_amdgcn_ballot_w64(isHit), mReq = __builtin_amdgcn_ballot_w64(req.valid);
    const uint wave = threadIdx.x >> 6, lane = threadIdx.x & 63u; const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0u) { sCnt[wave][0] = (uint)__popcll(mAlive); sCnt[wave][1] = (uint)__popcll(mReq); sCnt[wave][2] = (uint)__popcll(mHit); }
    __syncthreads();
    if (threadIdx.x < 3u) {
        uint tot = 0; for (uint w = 0; w < 4u; w++) { const uint c = sCnt[w][threadIdx.x]; sCnt[w][threadIdx.x] = tot; tot += c; }
        uint b = 0;
        if (tot) { if (threadIdx.x == 0u) b = atomicAdd(countOutPtr, tot); else if (threadIdx.x == 1u) b = atomicAdd(&wc->shadowCount, tot); else atomicAdd(&wc->hits, (unsigned long long)tot); }
        sBase[threadIdx.x] = b;
    }
    __syncthreads();
    if (alive) queueOut[sBase[0] + sCnt[wave][0] + (uint)__popcll(mAlive & below)] = p;
    if (req.valid) {
        const uint s = sBase[1] + sCnt[wave][1] + (uint)__popcll(mReq & below);
        sq.q0[s] = make_float4(req.origin.x, req.origin.y, req.origin.z, req.tmax);
        sq.q1[s] = make_float4(req.dir.x, req.dir.y, req.dir.z, asfloat(p));
        sq.q2[s] = make_float4(1.0f, 0.f, 0.f, 0.f);      // the mark a visible entry leaves in the scratch L
        if (sq.q3) sq.q3[s] = make_float4(req.fbWeight, req.fbRandom, asfloat(req.fbLight), asfloat(req.rrFix));      // NEE-AT feedback of the visible case: applied by the shadow kernels (shadow_visible)
        newL[s] = req.newL;
    }
}
// AccumulatePathRadiance of the vertex's light sample, for the entries the shadow launches found visible; the marks are cleared for the next pass
__global__ void __launch_bounds__(256) k_sp_fill_resolve(PathPool pool, uint4* __restrict__ mark, ShadowQueue sq, const float4* __restrict__ newL, const uint* __restrict__ countPtr) {
    const uint count = *countPtr;
    for (uint i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint p = asuint(sq.q1[i].w);
        const uint4 m = mark[p];
        if ((m.z | m.w) == 0u) continue;
        mark[p] = make_uint4(0u, 0u, 0u, 0u);
        uint4 c = pool.s2[p];
        PathState t; t.pack45[0] = c.z; t.pack45[1] = c.w;
        SPNeeRequest req; req.newL = newL[i];
        t.SetL(t.GetL() + req.newL);
        c.z = t.pack45[0]; c.w = t.pack45[1];
        pool.s2[p] = c;
    }
}
// CommitPixel -> CommitDenoiserRadiance for every pixel of the pass
template <class PKC>
__global__ void __launch_bounds__(256) k_sp_fill_commit(PKC k, StablePlanesContext sp, PathPool pool, uint numOwned, uint sampleIndex) {
    const uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= numOwned) return;
    PathState path = sp_load_path(pool, i);
    const StablePlanesFiller<PKC> f{k, sp, sampleIndex};
    f.CommitDenoiserRadiance(path);
}

#define SP_LAUNCH_B(KERNEL, G, B, ...) do { if (k.S.useFp16Types) { PathKernelContextT<true> k16; __builtin_memcpy(&k16, &k, sizeof(k16)); hipLaunchKernelGGL((KERNEL<PathKernelContextT<true>>), G, dim3(B), 0, st, k16, __VA_ARGS__); } \
                                             else hipLaunchKernelGGL((KERNEL<PathKernelContext>), G, dim3(B), 0, st, k, __VA_ARGS__); } while (0)
#define SP_LAUNCH(KERNEL, G, ...) SP_LAUNCH_B(KERNEL, G, 256, __VA_ARGS__)
void launch_sp_fill_generate(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleIndex, uint* queue, uint* countPtr, hipStream_t st) {
    SP_LAUNCH_B(k_sp_fill_generate, dim3((numOwned + 1023u) / 1024u), 1024, sp, pool, ownedPixels, numOwned, sampleIndex, queue, countPtr);
}
void launch_sp_fill_shade(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr, ShadowQueue sq, float4* newL,
                          uint sampleIndex, WaveCounters* wc, uint* classScratch, uint* classCount, hipStream_t st) {
    // class-ordered shading as in reference mode (k_classify): a shading wave lives as long as its longest lane, misses and hits that end after their emission leave early when they sit together
    if (classScratch) { launch_classify(pool, queueIn, countInPtr, countIn, classScratch, classCount, st); queueIn = classScratch; } else classCount = nullptr;
    SP_LAUNCH(k_sp_fill_shade, dim3((countIn + 255u) / 256u), sp, pool, queueIn, countInPtr, queueOut, countOutPtr, sq, newL, sampleIndex, wc, (const uint*)classCount);
}
void launch_sp_fill_resolve(PathPool pool, uint4* mark, ShadowQueue sq, const float4* newL, const uint* countPtr, uint count, hipStream_t st) {
    uint g = (count + 255u) / 256u; if (g > 4096u) g = 4096u; if (g < 1u) g = 1u;
    hipLaunchKernelGGL(k_sp_fill_resolve, dim3(g), dim3(256), 0, st, pool, mark, sq, newL, countPtr);
}
void launch_sp_fill_commit(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, uint numOwned, uint sampleIndex, hipStream_t st) {
    SP_LAUNCH(k_sp_fill_commit, dim3((numOwned + 255u) / 256u), sp, pool, numOwned, sampleIndex);
}
#undef SP_LAUNCH
#undef SP_LAUNCH_B

// DenoisingGuidesBaker.hlsl DenoiseSpecHitT: one thread per pixel, src -> dst
__global__ void __launch_bounds__(256) k_sp_denoise_spec_hit_t(const float* __restrict__ src, const float* __restrict__ depth, float* __restrict__ dst, uint width, uint height) {
    const uint i = blockIdx.x * 256u + threadIdx.x;
    if (i < width * height) dst[i] = SpecHitTNeighbourhood(src, depth, width, height, (int)(i % width), (int)(i / width));
}
void launch_sp_denoise_spec_hit_t(float* specHitT, const float* depth, float* scratch, uint width, uint height, hipStream_t st) {      // ping (main -> scratch), pong (scratch -> main): DenoisingGuidesBaker.cpp:62-84
    const dim3 g((width * height + 255u) / 256u);
    hipLaunchKernelGGL(k_sp_denoise_spec_hit_t, g, dim3(256), 0, st, specHitT, depth, scratch, width, height);
    hipLaunchKernelGGL(k_sp_denoise_spec_hit_t, g, dim3(256), 0, st, scratch, depth, specHitT, width, height);
}
// PostProcess.hlsl NO_DENOISER_FINAL_MERGE: output colour = (GetAllRadiance, 1)
__global__ void __launch_bounds__(256) k_sp_merge(StablePlanesContext sp, const uint* __restrict__ ownedPixels, uint numOwned, float4* __restrict__ out) {
    const uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= numOwned) return;
    const uint px = ownedPixels[i] >> 16, py = ownedPixels[i] & 0xFFFFu;
    out[(size_t)py * sp.C.imageWidth + px] = make_float4(sp.GetAllRadiance(px, py), 1.0f);
}
void launch_sp_merge(const StablePlanesContext& sp, const uint* ownedPixels, uint numOwned, float4* out, hipStream_t st) {
    hipLaunchKernelGGL(k_sp_merge, dim3((numOwned + 255u) / 256u), dim3(256), 0, st, sp, ownedPixels, numOwned, out);
}
// The plane buffers of a list of pixels as flat records (tile-sharded frames: what a rank sends to the rank that denoises / shows the frame): SP_SHARD_WORDS words per pixel —
// the four header words, the three 80-byte StablePlane records (from their tiled-swizzled addresses), stable radiance, depth, specular hit distance, motion vectors, throughput.
// One wave-friendly layout: thread = (pixel, word), so both sides of the copy are coalesced along the record.
__global__ void __launch_bounds__(256) k_sp_pack(StablePlanesContext sp, const uint* __restrict__ pixels, uint num, uint* __restrict__ dst, uint unpack, uint firstWord, uint numWords) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)num * numWords) return;
    const uint k = (uint)(i / numWords), wd = firstWord + (uint)(i - (size_t)k * numWords);      // (the whole record: 0, SP_SHARD_WORDS; the guides the light baker reads: SP_GUIDE_FIRST, SP_GUIDE_WORDS)
    const uint px = pixels[k] >> 16, py = pixels[k] & 0xFFFFu;
    const size_t pix = (size_t)py * sp.C.imageWidth + px, plane = (size_t)sp.C.imageWidth * sp.C.imageHeight;
    uint* p;
    if (wd < 4u) p = sp.B.Header + wd * plane + pix;
    else if (wd < 64u) { const uint pl = (wd - 4u) / 20u, w = (wd - 4u) % 20u; p = reinterpret_cast<uint*>(sp.B.Planes + sp.PixelToAddress(px, py, pl)) + w; }
    else if (wd < 66u) p = reinterpret_cast<uint*>(sp.B.StableRadiance + pix) + (wd - 64u);
    else if (wd == 66u) p = reinterpret_cast<uint*>(sp.B.Depth + pix);
    else if (wd == 67u) p = reinterpret_cast<uint*>(sp.B.SpecularHitT + pix);
    else if (wd < 70u) p = reinterpret_cast<uint*>(sp.B.MotionVectors + pix) + (wd - 68u);
    else p = sp.B.Throughput + pix;
    if (unpack) *p = dst[i]; else dst[i] = *p;
}
void launch_sp_pack(const StablePlanesContext& sp, const uint* pixels, uint num, uint* buf, bool unpack, hipStream_t st, uint firstWord, uint numWords) {
    const size_t n = (size_t)num * numWords;
    if (n) hipLaunchKernelGGL(k_sp_pack, dim3((uint)((n + 255u) / 256u)), dim3(256), 0, st, sp, pixels, num, buf, unpack ? 1u : 0u, firstWord, numWords);
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
