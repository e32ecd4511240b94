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

template <bool L> static inline float3 SP_env_to_local(const PathKernelContextT<L>& pt, float3 rayDir) { return mul_vec_mat3(rayDir, pt.sc.envToLocal); }
template <bool L> static inline float3 SP_env_eval_local(const PathKernelContextT<L>& pt, float3 localDir, float mipLevel) { return env_eval_local(pt.sc, localDir, mipLevel); }
template <bool L> static inline LightSampler SP_light_sampler(const PathKernelContextT<L>& pt, uint id, bool ssc) { return LightSampler::make(pt.sc.lights, id >> 16, id & 0xFFFFu, ssc); }
template <bool L> static inline bool SP_ssc_heuristic(const PathKernelContextT<L>& pt, float rayConeWidth, float totalPathLength) { return LightSampler::IsScreenSpaceCoherentHeuristic(pt.sc.lights, rayConeWidth, totalPathLength); }
template <bool L> static inline float3 SP_firefly_filter(const PathKernelContextT<L>&, float3 signal, float threshold, float k) { return FireflyFilter<LPOps<L>>(signal, threshold, k); }
template <bool L> static inline float SP_new_scatter_ffk(const PathKernelContextT<L>&, float currentK, float bouncePDF, float lobeP) { return ComputeNewScatterFireflyFilterK<LPOps<L>>(currentK, bouncePDF, lobeP); }
template <bool L> static inline float SP_ray_cone_expansion(const PathKernelContextT<L>&, float pdf) { return ComputeRayConeSpreadAngleExpansionByScatterPDF(pdf, 0.3f); }
static inline bool SP_isfinite(float v) { return (asuint(v) & 0x7F800000u) != 0x7F800000u; }

#define SP_BRANCH_FIELD sampleIndex
#include "pt_stableplanes.h"

#pragma clang force_cuda_host_device end
} // namespace ptk
