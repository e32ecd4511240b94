// mi355pt — launches of the stable-plane build pass (pt_stableplanes.hip), called by pt_build_stable_planes (pt_api.hip)
#pragma once
#include "pt_wavefront.h"
#include "pt_stableplanes_device.h"

namespace ptk {
void launch_sp_generate(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleIndex, uint* queue, hipStream_t st);
void launch_sp_build_shade(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr,
                           uint sampleIndex, WaveCounters* wc, hipStream_t st);
// the fill passes (one sub-sample): generate from plane 0, the pass's shader, the float4 resolve of the visible light samples (mark: a scratch copy of pool.s2 the shadow launches write to), the final commit
void launch_sp_fill_generate(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleIndex, uint* queue, uint* countPtr, hipStream_t st);
void launch_sp_fill_shade(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr, ShadowQueue sq, float4* newL,
                          uint sampleIndex, WaveCounters* wc, uint* classScratch /* 2 x countIn words or null */, uint* classCount /* 3 words, zero on entry */, hipStream_t st);
void launch_sp_fill_resolve(PathPool pool, uint4* mark, ShadowQueue sq, const float4* newL, const uint* countPtr, uint count, hipStream_t st);
void launch_sp_fill_commit(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, uint numOwned, uint sampleIndex, hipStream_t st);
void launch_sp_denoise_spec_hit_t(float* specHitT, const float* depth, float* scratch, uint width, uint height, hipStream_t st);
void launch_sp_merge(const StablePlanesContext& sp, const uint* ownedPixels, uint numOwned, float4* out, hipStream_t st);
// the plane buffers of `num` pixels <-> flat records of SP_SHARD_WORDS words each (pt_pack_stable_planes / pt_unpack_stable_planes / pt_gather_stable_planes)
static const uint SP_SHARD_WORDS = 71u;      // header 4 + planes 3 x 20 + stable radiance 2 + depth 1 + specular hit distance 1 + motion vectors 2 + throughput 1
static const uint SP_GUIDE_FIRST = 66u, SP_GUIDE_WORDS = 4u;      // depth, specular hit distance, motion vectors: what LightsBaker::UpdateEnd reads of a frame (depth + motion; the word in between rides along)
void launch_sp_pack(const StablePlanesContext& sp, const uint* pixels, uint num, uint* buf, bool unpack, hipStream_t st, uint firstWord = 0u, uint numWords = SP_SHARD_WORDS);
} // namespace ptk
