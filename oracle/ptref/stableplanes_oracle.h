// ORACLE (test infrastructure only) — the oracle's view of stableplanes.h: what the stable-plane passes take from ptref::PathTracer, and the per-pixel loops of the
// two passes as the reference's raygen shader runs them (PathTracerSample.hlsl:200-250 with PATH_TRACER_MODE_BUILD_STABLE_PLANES / _FILL_STABLE_PLANES).
#pragma once
#include "pathtracer.h"

namespace ptref {

template <class PT> struct SPTraits;
template <> struct SPTraits<PathTracer> { typedef ::ptref::LP LP; };
static inline void SP_camera_ray(const PathTracer& pt, uint px, uint py, uint, float3& o, float3& d) { pt.computeCameraRay(px, py, o, d); }      // (the tracer was made for that sample index)
static inline bool SP_env_enabled(const PathTracer& pt) { return pt.sc.env.enabled; }
static inline float3 SP_env_eval(const PathTracer& pt, float3 rayDir, float mipLevel) { return pt.sc.env.EvalLocal(pt.sc.env.ToLocal(rayDir), mipLevel); }
static inline uint SP_material_flags(const PathTracer& pt, uint materialID) { return pt.sc.materials[materialID].Flags; }
static inline bool SP_analytic_proxy(const PathTracer& pt, uint id, uint lightIndex, float3 rayOrigin, float3 rayDir, float3& add) {
    const LightSampler lightSampler = pt.CreateLightSampler(id, false);
    return lightSampler.ComputeAnalyticLightProxyContribution(lightIndex, 0.0f, rayOrigin, rayDir, 0u, 0u, add);
}

static inline float3 SP_env_to_local(const PathTracer& pt, float3 rayDir) { return pt.sc.env.ToLocal(rayDir); }
static inline float3 SP_env_eval_local(const PathTracer& pt, float3 localDir, float mipLevel) { return pt.sc.env.EvalLocal(localDir, mipLevel); }
static inline LightSampler SP_light_sampler(const PathTracer& pt, uint id, bool ssc) { return pt.CreateLightSampler(id, ssc); }
static inline bool SP_ssc_heuristic(const PathTracer& pt, float rayConeWidth, float totalPathLength) { return LightSampler::IsScreenSpaceCoherentHeuristic(pt.sc.lightTable, rayConeWidth, totalPathLength); }
static inline float3 SP_firefly_filter(const PathTracer&, float3 signal, float threshold, float k) { return FireflyFilter(signal, threshold, k); }
static inline float SP_new_scatter_ffk(const PathTracer&, float currentK, float bouncePDF, float lobeP) { return ComputeNewScatterFireflyFilterK(currentK, bouncePDF, lobeP); }
static inline float SP_ray_cone_expansion(const PathTracer&, float pdf) { return ComputeRayConeSpreadAngleExpansionByScatterPDF(pdf); }
static inline bool SP_isfinite(float v) { return (asuint(v) & 0x7F800000u) != 0x7F800000u; }

#define SP_BRANCH_FIELD stableBranchID
#include "stableplanes.h"

// the BUILD pass of one pixel: delta paths only, one plane after the other (PathTracerSample.hlsl:216-230, nextHit :114-169, postProcessHit :96-112)
static inline void sp_build_pixel(const StablePlanesBuilder<PathTracer>& b, uint px, uint py) {
    PathState path = b.generate(px, py);
    RayCounters* counters = b.pt.counters;
    while (path.isActive()) {
        float3 o = path.origin, d = path.dir;
        if (counters) counters->extendRays++;
        HitInfo h = trace_closest(b.pt.sc, o, d, 0.0f, kMaxRayTravel, counters ? &counters->nodeVisitsExt : 0, counters ? &counters->triTestsExt : 0);
        if (h.prim == 0xFFFFFFFFu) b.HandleMiss(path, o, d, kMaxRayTravel);
        else { if (counters) counters->hits++; b.HandleHit(path, o, d, h.prim, h.t, h.u, h.v); }
        b.postProcessHit(path);
    }
}

// one sub-sample of the FILL pass of one pixel (PathTracerSample.hlsl:200-250: FirstHitFromVBuffer, the loop, CommitPixel)
static inline void sp_fill_pixel(const StablePlanesFiller<PathTracer>& f, uint px, uint py) {
    PathState path = f.generate(px, py);
    RayCounters* counters = f.pt.counters;
    while (path.isActive()) {
        float3 o = path.origin, d = path.dir;
        if (counters) counters->extendRays++;
        HitInfo h = trace_closest(f.pt.sc, o, d, 0.0f, kMaxRayTravel, counters ? &counters->nodeVisitsExt : 0, counters ? &counters->triTestsExt : 0);
        if (h.prim == 0xFFFFFFFFu) f.HandleMiss(path, d, kMaxRayTravel);
        else {
            if (counters) counters->hits++;
            SPNeeRequest req; f.HandleHit(path, o, d, h.prim, h.t, h.u, h.v, req);
            if (req.valid) {
                if (counters) counters->shadowRays++;
                if (trace_visibility(f.pt.sc, req.origin, req.dir, 0.0f, req.tmax, counters ? &counters->nodeVisitsSh : 0, counters ? &counters->triTestsSh : 0)) StablePlanesFiller<PathTracer>::ApplyVisibleLight(path, req, f.pt.fbTotalWeight, f.pt.fbCandidates, f.pt.fbWidth);
            }
        }
    }
    f.CommitDenoiserRadiance(path);      // CommitPixel
}

} // namespace ptref
