// mi355pt — launches of the stable-plane build pass (pt_stableplanes.hip), called by pt_build_stable_planes (pt_api.hip)
#pragma once
#include "pt_wavefront.h"
#include "pt_stableplanes_device.h"

namespace ptk {
void launch_sp_generate(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* ownedPixels, uint numOwned, uint sampleIndex, uint* queue, hipStream_t st);
void launch_sp_build_shade(const PathKernelContext& k, const StablePlanesContext& sp, PathPool pool, const uint* queueIn, const uint* countInPtr, uint countIn, uint* queueOut, uint* countOutPtr,
                           uint sampleIndex, WaveCounters* wc, hipStream_t st);
} // namespace ptk
