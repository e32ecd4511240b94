// mi355pt — the device side's view of pt_stableplanes.h: what the stable-plane passes take from the wavefront path tracer (PathKernelContextT, pt_path.h).
#pragma once
#include "pt_path.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

template <class PT> struct SPTraits;
template <bool L> struct SPTraits<PathKernelContextT<L>> { typedef LPOps<L> LP; };
template <bool L> static inline void SP_camera_ray(const PathKernelContextT<L>& pt, uint px, uint py, uint sampleIndex, float3& o, float3& d) { pt.computeCameraRay(px, py, sampleIndex, o, d); }
template <bool L> static inline bool SP_env_enabled(const PathKernelContextT<L>& pt) { return pt.sc.envEnabled != 0; }
template <bool L> static inline float3 SP_env_eval(const PathKernelContextT<L>& pt, float3 rayDir, float mipLevel) {      // EnvMap::ToLocal + EvalLocal, as HandleMiss does it (pt_path.h)
    return env_eval_local(pt.sc, mul_vec_mat3(rayDir, pt.sc.envToLocal), mipLevel);
}
template <bool L> static inline uint SP_material_flags(const PathKernelContextT<L>& pt, uint materialID) { return pt.sc.materials[materialID].Flags; }
template <bool L> static inline bool SP_analytic_proxy(const PathKernelContextT<L>& pt, uint id, uint lightIndex, float3 rayOrigin, float3 rayDir, float3& add) {
    LightSampler lightSampler = LightSampler::make(pt.sc.lights, id >> 16, id & 0xFFFFu, false);
    return lightSampler.ComputeAnalyticLightProxyContribution(lightIndex, 0.0f, rayOrigin, rayDir, 0u, 0u, add);
}

#define SP_BRANCH_FIELD sampleIndex
#include "pt_stableplanes.h"

#pragma clang force_cuda_host_device end
} // namespace ptk
