// ORACLE (test infrastructure only) — the procedural sky the environment cube can be baked from, restated. Twin of rtxpt_amd/csrc/pt_sky.h;
// pinned to the reference's own text by oracle/refpin (hlsl_tu.py: namespace sky). Restates, function by function and in the reference's operation order:
//   Rtxpt/Lighting/Distant/precomputed_sky.hlsli   (Bruneton & Neyret's precomputed atmospheric scattering as Q2RTX ships it): ClampRadius, RayIntersectsGround,
//       DistanceToTopAtmosphereBoundary, GetTransmittanceUV, GetTransmittanceToTopAtmosphereBoundary, GetTransmittance, GetScatteringUVWZ, GetMieFromfloat4,
//       Sample4D, RayleighPhaseFunction, MiePhaseFunction, GetParameters, GetSkyRadiance, CorrectViewRay, GetSkyRadianceToPoint, GetIrradiance(UV), GetSkyIrradiance
//   Rtxpt/Lighting/Distant/SampleProceduralSky.hlsli: intersectSphere, GetDensity, HenyeyGreenstein, ComputeSunTransmittanceAtPos, FineRaymarching,
//       RaymarchClouds, ProceduralSkyLowRes (the half-resolution cloud pre-pass), ProceduralSky (sky + sun disc + 3x3-filtered clouds)
// The four look-up textures (transmittance 2-D, in-scatter 3-D, irradiance 2-D, clouds 3-D; the reference also binds a noise texture that no function samples) are
// inputs: the host hands them over as RGBA float texels (pt_set_procedural_sky). SampleLevel(linear, wrap) is restated as the bilinear / trilinear fetch below (texel
// centres at (i + 0.5) / size, wrap addressing) — texture filtering has no reference source here, as for every other texture of the path.
#pragma once
#include "dmath.h"
#include "vec.h"
#include "rng.h"
#include "envcube.h"

namespace ptref {

struct AtmosphereParameters {               // precomputed_sky.hlsli:23-35
    float3 StarIrradiance; float StarAngularDiameter;
    float3 RayleightScatteringRGB; float PlanetSurfaceRadius;
    float3 MieScatteringRGB; float PlanetAtmosphereRadius;
    float MieHenyeyGreensteinG, SqDistanceToHorizontalBoundary, AtmosphereHeight, reserved;
};
struct ProceduralSkyConstants {             // SampleProceduralSky.hlsli:18-46
    AtmosphereParameters SkyParams;
    float3 FinalRadianceMultiplier; float _padding3;
    float3 SunDir; float CloudsTime;
    float3 GroundAlbedo; float SunAngularDiameter;
    float _padding0, _padding1, sun_solid_angle, _padding2;
    float3 physical_sky_ground_radiance; float cloud_density_offset;
    float sky_transmittance, sky_phase_g, sky_amb_phase_g, sky_scattering;
};
static_assert(sizeof(AtmosphereParameters) == 64 && sizeof(ProceduralSkyConstants) == 160, "procedural sky constants layout");
struct SkyTexture { const float4* texels; uint w, h, d, _pad; };      // d = 1: a 2-D texture
struct ProceduralSkyContext { ProceduralSkyConstants Consts; SkyTexture Transmittance, Scatter, Irradiance, Clouds; };      // ProceduralSkyWorkingContext

static inline int sky_wrap(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static inline float4 sky_lerp4(float4 a, float4 b, float t) { return make_float4(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)); }
// Texture2D::SampleLevel(linear wrap, uv, 0)
static inline float4 sky_sample2d(const SkyTexture& t, float u, float v) {
    const float fx = u * (float)t.w - 0.5f, fy = v * (float)t.h - 0.5f, flx = floorf(fx), fly = floorf(fy), ax = fx - flx, ay = fy - fly;
    const int x0 = sky_wrap((int)flx, (int)t.w), x1 = sky_wrap((int)flx + 1, (int)t.w), y0 = sky_wrap((int)fly, (int)t.h), y1 = sky_wrap((int)fly + 1, (int)t.h);
    const float4* r0 = t.texels + (size_t)y0 * t.w; const float4* r1 = t.texels + (size_t)y1 * t.w;
    return sky_lerp4(sky_lerp4(r0[x0], r0[x1], ax), sky_lerp4(r1[x0], r1[x1], ax), ay);
}
// Texture3D::SampleLevel(linear wrap, uvw, 0)
static inline float4 sky_sample3d(const SkyTexture& t, float3 uvw) {
    const float fz = uvw.z * (float)t.d - 0.5f, flz = floorf(fz), az = fz - flz;
    const int z0 = sky_wrap((int)flz, (int)t.d), z1 = sky_wrap((int)flz + 1, (int)t.d);
    SkyTexture s0 = t, s1 = t; s0.texels = t.texels + (size_t)z0 * t.w * t.h; s1.texels = t.texels + (size_t)z1 * t.w * t.h;
    return sky_lerp4(sky_sample2d(s0, uvw.x, uvw.y), sky_sample2d(s1, uvw.x, uvw.y), az);
}

// ---- precomputed_sky.hlsli
static const float SM_PI = 3.1415926535897932384626433832795f;
static const float SKY_LUM_SCALE = 0.001f, SUN_LUM_SCALE = 0.00001f;
static inline float3 sky_spectral_ratio() {      // SUN_SPECTRAL_RADIANCE_TO_LUMINANCE / SKY_SPECTRAL_RADIANCE_TO_LUMINANCE (precomputed_sky.hlsli:41-52)
    const float sky = 683.000000f * SKY_LUM_SCALE;
    return make_float3((98242.786222f * SUN_LUM_SCALE) / sky, (69954.398112f * SUN_LUM_SCALE) / sky, (66475.012354f * SUN_LUM_SCALE) / sky);
}
static const float TRANSMITTANCE_TEXTURE_WIDTH = 256.0f, TRANSMITTANCE_TEXTURE_HEIGHT = 64.0f, SCATTERING_TEXTURE_R_SIZE = 32.0f, SCATTERING_TEXTURE_MU_SIZE = 128.0f,
                   SCATTERING_TEXTURE_MU_S_SIZE = 32.0f, SCATTERING_TEXTURE_NU_SIZE = 8.0f, IRRADIANCE_TEXTURE_WIDTH = 64.0f, IRRADIANCE_TEXTURE_HEIGHT = 16.0f,
                   SCATTERING_TEXTURE_MU_SIZE_HALF = 64.0f;
static inline float sky_ranged(float val, float size) { return (val) * (size - 1) / (size) + (0.5f / size); }      // the RANGED_* macros
static const float SKY_IRRADIANCE_TO_RADIANCE = 0.5f / SM_PI;

static inline float ClampRadius(const AtmosphereParameters& atmosphere, float PointHeight) { return clampf(PointHeight, atmosphere.PlanetSurfaceRadius, atmosphere.PlanetAtmosphereRadius); }
static inline bool RayIntersectsGround(const AtmosphereParameters& atmosphere, float PointHeight, float ViewAngleCos) {
    return ViewAngleCos < 0.0f && PointHeight * PointHeight * (ViewAngleCos * ViewAngleCos - 1.0f) + atmosphere.PlanetSurfaceRadius * atmosphere.PlanetSurfaceRadius >= 0.0f;
}
static inline float DistanceToTopAtmosphereBoundary(const AtmosphereParameters& atmosphere, float PlanetRadius, float ViewAngleCos) {
    float D = PlanetRadius * PlanetRadius * (ViewAngleCos * ViewAngleCos - 1.0f) + atmosphere.PlanetAtmosphereRadius * atmosphere.PlanetAtmosphereRadius;
    return fmaxf_(0.0f, -PlanetRadius * ViewAngleCos + sqrtf_(fmaxf_(0.0f, D)));
}
static inline float2 GetTransmittanceUV(const AtmosphereParameters& atmosphere, float PointHeight, float ViewAngleCos) {
    float X0 = sqrtf_(atmosphere.SqDistanceToHorizontalBoundary);
    float dh = sqrtf_(fmaxf_(0.0f, (PointHeight * PointHeight - atmosphere.PlanetSurfaceRadius * atmosphere.PlanetSurfaceRadius)));
    float dH = DistanceToTopAtmosphereBoundary(atmosphere, PointHeight, ViewAngleCos);
    float Xtop = atmosphere.PlanetAtmosphereRadius - PointHeight;
    float XH = dh + X0;
    float U = (dH - Xtop) / (XH - Xtop);
    float V = dh / X0;
    return make_float2(sky_ranged(U, TRANSMITTANCE_TEXTURE_WIDTH), sky_ranged(V, TRANSMITTANCE_TEXTURE_HEIGHT));
}
static inline float3 GetTransmittanceToTopAtmosphereBoundary(const AtmosphereParameters& atmosphere, const SkyTexture& transmittance_texture, float PointHeight, float ViewAngleCos) {
    float2 uv = GetTransmittanceUV(atmosphere, PointHeight, ViewAngleCos);
    return xyz(sky_sample2d(transmittance_texture, uv.x, uv.y));
}
static inline float3 GetTransmittance(const AtmosphereParameters& atmosphere, const SkyTexture& transmittance_texture, float PointHeight, float ViewAngleCos, float Destination, bool IntersectsGround) {
    float DestinationHeight = ClampRadius(atmosphere, sqrtf_(Destination * Destination + 2.0f * PointHeight * ViewAngleCos * Destination + PointHeight * PointHeight));
    float DestinationViewAngleCos = clampf((PointHeight * ViewAngleCos + Destination) / DestinationHeight, -1.0f, 1.0f);
    if (IntersectsGround)
        return min3v(GetTransmittanceToTopAtmosphereBoundary(atmosphere, transmittance_texture, DestinationHeight, -DestinationViewAngleCos) / GetTransmittanceToTopAtmosphereBoundary(atmosphere, transmittance_texture, PointHeight, -ViewAngleCos),
                     make_float3(1.0f, 1.0f, 1.0f));
    else
        return min3v(GetTransmittanceToTopAtmosphereBoundary(atmosphere, transmittance_texture, PointHeight, ViewAngleCos) / GetTransmittanceToTopAtmosphereBoundary(atmosphere, transmittance_texture, DestinationHeight, DestinationViewAngleCos),
                     make_float3(1.0f, 1.0f, 1.0f));
}
static inline float4 GetScatteringUVWZ(const AtmosphereParameters& atmosphere, float PointHeight, float ViewAngleCos, float SunZenithAngleCos, float SunViewAngleCos, bool IntersectsGround) {
    float SquareHeight = PointHeight * PointHeight;
    float SquareViewAngleSin = 1.0f - ViewAngleCos * ViewAngleCos;
    float H = sqrtf_(atmosphere.SqDistanceToHorizontalBoundary);
    float HorizonDistance = sqrtf_(fmaxf_(0.0f, (SquareHeight - atmosphere.PlanetSurfaceRadius * atmosphere.PlanetSurfaceRadius)));
    float u_Height = sky_ranged((HorizonDistance / H), SCATTERING_TEXTURE_R_SIZE);
    float discriminant = -SquareHeight * SquareViewAngleSin + atmosphere.PlanetSurfaceRadius * atmosphere.PlanetSurfaceRadius;
    float u_ViewToZeinthCos;
    if (IntersectsGround) {
        float d = -PointHeight * ViewAngleCos - sqrtf_(fmaxf_(0.0f, discriminant));
        float d_min = PointHeight - atmosphere.PlanetSurfaceRadius;
        float d_max = HorizonDistance;
        float du = d_max == d_min ? 0.0f : (d - d_min) / (d_max - d_min);
        du = sky_ranged(du, SCATTERING_TEXTURE_MU_SIZE_HALF);
        u_ViewToZeinthCos = 0.5f - 0.5f * du;
    } else {
        float d = -PointHeight * ViewAngleCos + sqrtf_(fmaxf_(0.0f, discriminant + H * H));
        float d_min = atmosphere.PlanetAtmosphereRadius - PointHeight;
        float d_max = HorizonDistance + H;
        float du = (d - d_min) / (d_max - d_min);
        du = sky_ranged(du, SCATTERING_TEXTURE_MU_SIZE_HALF);
        u_ViewToZeinthCos = 0.5f + 0.5f * du;
    }
    float d = DistanceToTopAtmosphereBoundary(atmosphere, atmosphere.PlanetSurfaceRadius, SunZenithAngleCos);
    float d_min = atmosphere.AtmosphereHeight;
    float d_max = H;
    float a = (d - d_min) / (d_max - d_min);
    float A = 0.41582f * atmosphere.PlanetSurfaceRadius / (d_max - d_min);
    float dy = fmaxf_(1.0f - a / A, 0.0f) / (1.0f + a);
    float u_SunZenithAngleCos = sky_ranged(dy, SCATTERING_TEXTURE_MU_S_SIZE);
    float u_SunViewAngleCos = (SunViewAngleCos + 1.0f) / 2.0f;
    return make_float4(u_SunViewAngleCos, u_SunZenithAngleCos, u_ViewToZeinthCos, u_Height);
}
static inline float3 GetMieFromfloat4(const AtmosphereParameters& atmosphere, float4 C) {
    if (C.x == 0.0f) return make_float3(0.f, 0.f, 0.f);
    return ((xyz(C) * C.w) / C.x) * (atmosphere.RayleightScatteringRGB.x / atmosphere.MieScatteringRGB.x) * (atmosphere.MieScatteringRGB / atmosphere.RayleightScatteringRGB);
}
static inline float3 Sample4D(const AtmosphereParameters& atmosphere, const SkyTexture& scattering_texture, float PointHeight, float ViewAngleCos, float SunZenithAngleCos, float SunViewAngleCos,
                              bool IntersectsGround, float3& OutMieScattering) {
    float4 uvwz = GetScatteringUVWZ(atmosphere, PointHeight, ViewAngleCos, SunZenithAngleCos, SunViewAngleCos, IntersectsGround);
    float ux = uvwz.x * (float)(SCATTERING_TEXTURE_NU_SIZE - 1);
    float offset = floorf(ux);
    float lerp = ux - floorf(ux);
    float3 uvw0 = make_float3((offset + uvwz.y) / (float)(SCATTERING_TEXTURE_NU_SIZE), uvwz.z, uvwz.w);
    float3 uvw1 = make_float3((offset + 1.0f + uvwz.y) / (float)(SCATTERING_TEXTURE_NU_SIZE), uvwz.z, uvwz.w);
    float4 InterpolatedScattering = sky_sample3d(scattering_texture, uvw0) * (1.0f - lerp) + sky_sample3d(scattering_texture, uvw1) * lerp;
    OutMieScattering = GetMieFromfloat4(atmosphere, InterpolatedScattering);
    return xyz(InterpolatedScattering);
}
static inline float RayleighPhaseFunction(float nu) { float k = 3.0f / (16.0f * SM_PI); return k * (1.0f + nu * nu); }
static inline float MiePhaseFunction(float g, float nu) {
    float k = 3.0f / (8.0f * SM_PI) * (1.0f - g * g) / (2.0f + g * g);
    return k * (1.0f + nu * nu) / dm_pow(1.0f + g * g - 2.0f * g * nu, 1.5f);
}
static inline void GetParameters(const AtmosphereParameters& atmosphere, float3 view_ray, float3 camera, float& PointHeight, float& DotViewAngleCos, bool& bIntersectsAtmoshpere) {
    PointHeight = length(camera);
    DotViewAngleCos = dot(camera, view_ray);
    float IntersectsAtmoshpere = -DotViewAngleCos - sqrtf_(DotViewAngleCos * DotViewAngleCos - PointHeight * PointHeight + atmosphere.PlanetAtmosphereRadius * atmosphere.PlanetAtmosphereRadius);
    if (IntersectsAtmoshpere > 0.0f) {          // (the reference also moves its by-value copy of `camera` to the boundary: no caller sees that)
        PointHeight = atmosphere.PlanetAtmosphereRadius;
        DotViewAngleCos += IntersectsAtmoshpere;
        bIntersectsAtmoshpere = true;
    } else bIntersectsAtmoshpere = false;
}
static inline float3 sky_to_radiance(const AtmosphereParameters& atmosphere, float3 result) {      // `result /= StarIrradiance * (SUN / SKY); result *= SKY_IRRADIANCE_TO_RADIANCE`
    result = result / (atmosphere.StarIrradiance * sky_spectral_ratio());
    return result * SKY_IRRADIANCE_TO_RADIANCE;
}
static inline float3 GetSkyRadiance(const AtmosphereParameters& atmosphere, const SkyTexture& transmittance_texture, const SkyTexture& scattering_texture, float3 camera, float3 view_ray, float3 sun_direction,
                                    float3& transmittance) {
    transmittance = make_float3(1.0f, 1.0f, 1.0f);
    float PointHeight, DotViewAngleCos; bool IntersectsAtmoshpere;
    GetParameters(atmosphere, view_ray, camera, PointHeight, DotViewAngleCos, IntersectsAtmoshpere);
    if (!IntersectsAtmoshpere && PointHeight > atmosphere.PlanetAtmosphereRadius) return make_float3(0.f, 0.f, 0.f);
    float ViewAngleCos = DotViewAngleCos / PointHeight;
    float SunZenithAngleCos = dot(camera, sun_direction) / PointHeight;
    float SunViewAngleCos = dot(view_ray, sun_direction);
    bool IntersectsGround = RayIntersectsGround(atmosphere, PointHeight, ViewAngleCos);
    transmittance = IntersectsGround ? make_float3(0.f, 0.f, 0.f) : GetTransmittanceToTopAtmosphereBoundary(atmosphere, transmittance_texture, PointHeight, ViewAngleCos);
    float3 single_mie_scattering;
    float3 scattering = Sample4D(atmosphere, scattering_texture, PointHeight, ViewAngleCos, SunZenithAngleCos, SunViewAngleCos, IntersectsGround, single_mie_scattering);
    float3 result = scattering * RayleighPhaseFunction(SunViewAngleCos) + single_mie_scattering * MiePhaseFunction(atmosphere.MieHenyeyGreensteinG, SunViewAngleCos);
    return sky_to_radiance(atmosphere, result);
}
static inline float3 CorrectViewRay(float3 view_ray, float3 sun_direction) {
    if (sun_direction.z == 1.0f) return view_ray;
    float3 dir_axis = normalize(make_float3(sun_direction.x, sun_direction.y, 0.0f));
    float3 ortho_axis = make_float3(dir_axis.y, -dir_axis.x, 0.0f);
    float vx = dot(view_ray, dir_axis), vy = dot(view_ray, ortho_axis);
    vx = vx * 0.75f - 0.25f;
    return vx * dir_axis + vy * ortho_axis + make_float3(0.0f, 0.0f, view_ray.z);
}
static inline float sky_smoothstep(float a, float b, float x) { float t = saturate((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }
static inline float3 GetSkyRadianceToPoint(const AtmosphereParameters& atmosphere, const SkyTexture& transmittance_texture, const SkyTexture& scattering_texture, float3 camera, float3 spoint, float3 sun_direction,
                                           float3& transmittance) {
    float3 view_ray = normalize(spoint - camera);
    view_ray = CorrectViewRay(view_ray, sun_direction);
    float PointHeight, DotViewAngleCos; bool IntersectsAtmoshpere;
    GetParameters(atmosphere, view_ray, camera, PointHeight, DotViewAngleCos, IntersectsAtmoshpere);
    float ViewAngleCos = DotViewAngleCos / PointHeight;
    float SunZenithCos = dot(camera, sun_direction) / PointHeight;
    float ViewSunCos = dot(view_ray, sun_direction);
    float DistanceToPoint = length(spoint - camera);
    bool IntersectsGround = RayIntersectsGround(atmosphere, PointHeight, ViewAngleCos);
    float ViewAngleCos1 = 0.02f, ViewAngleCos2 = -0.06f;
    float3 single_mie_scattering, single_mie_scattering_p, scattering, scattering_p;
    // one evaluation at a given view angle: transmittance to the point, scattering at the eye and at the point
    auto at = [&](float vac, bool ground, float3& T, float3& S, float3& M, float3& Sp, float3& Mp) {
        T = GetTransmittance(atmosphere, transmittance_texture, PointHeight, vac, DistanceToPoint, ground);
        S = Sample4D(atmosphere, scattering_texture, PointHeight, vac, SunZenithCos, ViewSunCos, ground, M);
        float PointHeight_p = ClampRadius(atmosphere, sqrtf_(DistanceToPoint * DistanceToPoint + 2.0f * PointHeight * vac * DistanceToPoint + PointHeight * PointHeight));
        float ViewAngle_p = (PointHeight * vac + DistanceToPoint) / PointHeight_p;
        float SunZenithCos_p = (PointHeight * SunZenithCos + DistanceToPoint * ViewSunCos) / PointHeight_p;
        Sp = Sample4D(atmosphere, scattering_texture, PointHeight_p, ViewAngle_p, SunZenithCos_p, ViewSunCos, ground, Mp);
    };
    if (ViewAngleCos > ViewAngleCos1 || ViewAngleCos < ViewAngleCos2) at(ViewAngleCos, IntersectsGround, transmittance, scattering, single_mie_scattering, scattering_p, single_mie_scattering_p);
    else {      // near the horizon: interpolate between two fixed view angles (precomputed_sky.hlsli:410-452)
        float3 t1, s1, m1, sp1, mp1, t2, s2, m2, sp2, mp2;
        at(ViewAngleCos1, RayIntersectsGround(atmosphere, PointHeight, ViewAngleCos1), t1, s1, m1, sp1, mp1);
        at(ViewAngleCos2, RayIntersectsGround(atmosphere, PointHeight, ViewAngleCos2), t2, s2, m2, sp2, mp2);
        float lerpK = (ViewAngleCos1 - ViewAngleCos) / (ViewAngleCos1 - ViewAngleCos2);
        transmittance = lerp3(t1, t2, lerpK); scattering = lerp3(s1, s2, lerpK); single_mie_scattering = lerp3(m1, m2, lerpK);
        single_mie_scattering_p = lerp3(mp1, mp2, lerpK); scattering_p = lerp3(sp1, sp2, lerpK);
    }
    scattering = scattering - transmittance * scattering_p;
    single_mie_scattering = single_mie_scattering - transmittance * single_mie_scattering_p;
    single_mie_scattering = single_mie_scattering * sky_smoothstep(0.0f, 0.01f, SunZenithCos);
    float3 result = scattering * RayleighPhaseFunction(ViewSunCos) + single_mie_scattering * MiePhaseFunction(atmosphere.MieHenyeyGreensteinG, ViewSunCos);
    return sky_to_radiance(atmosphere, result);
}
static inline float3 GetSkyIrradiance(const AtmosphereParameters& atmosphere, const SkyTexture& irradiance_texture, float3 spoint, float3 sun_direction) {
    float PointHeight = length(spoint);
    float SunZenithCos = dot(spoint, sun_direction) / PointHeight;
    float uHeight = (PointHeight - atmosphere.PlanetSurfaceRadius) / atmosphere.AtmosphereHeight;          // GetIrradianceUV
    float vViewAngle = SunZenithCos * 0.5f + 0.5f;
    float3 sky_irradiance = xyz(sky_sample2d(irradiance_texture, sky_ranged(vViewAngle, IRRADIANCE_TEXTURE_WIDTH), sky_ranged(uHeight, IRRADIANCE_TEXTURE_HEIGHT)));
    return sky_to_radiance(atmosphere, sky_irradiance);
}

// ---- SampleProceduralSky.hlsli
static const float SKY_PI = 3.1415926535897932384626433832795f;
static const float CLOUD_START = 2.0f, CLOUD_HEIGHT = 1.4f, HORIZONFADE = 0.2f;
static const int CLOUDS_FINE_COUNT = 24, CLOUDS_SKY_SUN_COUNT = 1;
static inline float intersectSphere(float3 origin, float3 dir, float3 spherePos, float sphereRad) {
    float3 oc = origin - spherePos;
    float b = 2.0f * dot(dir, oc);
    float c = dot(oc, oc) - sphereRad * sphereRad;
    float disc = b * b - 4.0f * c;
    if (disc < 0.0f) return -1.0f;
    float q = (-b + ((b < 0.0f) ? -sqrtf_(disc) : sqrtf_(disc))) / 2.0f;
    float t0 = q, t1 = c / q;
    if (t0 > t1) { float temp = t0; t0 = t1; t1 = temp; }
    if (t1 < 0.0f) return -1.0f;
    return (t0 < 0.0f) ? t1 : t0;
}
static inline float GetDensity(const ProceduralSkyContext& workingContext, float3 step, float raylen, float current_ray) {
    float w = saturate(current_ray / raylen);
    float sizeScale = 0.5f;
    float3 uvw1 = make_float3(step.x * 0.1f * sizeScale, step.y * 0.1f * sizeScale, w);
    float3 uvw2 = make_float3(step.x * sizeScale, step.y * sizeScale, w);
    float ox = workingContext.Consts.CloudsTime * 0.707f * 0.01f, oy = workingContext.Consts.CloudsTime * 0.707f * 0.01f;
    float4 cloud1 = sky_sample3d(workingContext.Clouds, uvw1 + make_float3(ox, oy, 0.0f));
    float4 cloud2 = sky_sample3d(workingContext.Clouds, uvw2 + make_float3(ox, oy, 0.0f));
    return cloud1.x + (cloud2.y - 0.5f) * 0.1f;
}
static inline float HenyeyGreenstein(float mu, float inG) { return (1.f - inG * inG) / (dm_pow(1.f + inG * inG - 2.0f * inG * mu, 1.5f) * 4.0f * SKY_PI); }
static inline float sky_cloud_density(const ProceduralSkyContext& workingContext, float3 step, float raylen, float cray) {
    float Density = GetDensity(workingContext, step, raylen, cray);
    return fmaxf_(0.0f, Density - workingContext.Consts.cloud_density_offset) / (1.001f - workingContext.Consts.cloud_density_offset);
}
static inline float ComputeSunTransmittanceAtPos(const ProceduralSkyContext& workingContext, float3 camera, float3 view_dir, float raylen, float current_ray, float raypart, int stepCount) {
    float3 step = camera, delta = view_dir * raypart;
    float cray = current_ray, Transmittance = 1.0f;
    for (int i = 0; i < stepCount; i++) {
        float Density = sky_cloud_density(workingContext, step, raylen, cray);
        if (Density > 0.001f) Transmittance *= dm_exp(-workingContext.Consts.sky_transmittance * raypart * Density);
        cray += raypart;
        step = step + delta;
        if (cray > raylen) break;
    }
    return clampf(Transmittance, 0.0f, 1.0f);
}
static inline float4 FineRaymarching(const ProceduralSkyContext& workingContext, float3 camera, float3 view_dir, float raylen, float current_ray, float raypart, int stepCount, uint randHash) {
    const ProceduralSkyConstants& procSkyConsts = workingContext.Consts;
    const float rndScale = 0.3f; float rndSample = (Hash32ToFloat(randHash) * rndScale - rndScale * 0.5f);
    float cray = current_ray + rndSample * raypart;
    float3 delta = view_dir * raypart;
    float3 step = camera + rndSample * delta;
    float3 sun_transmittance;
    (void)GetSkyRadiance(procSkyConsts.SkyParams, workingContext.Transmittance, workingContext.Scatter, camera, procSkyConsts.SunDir, procSkyConsts.SunDir, sun_transmittance);
    float3 sun_direct_radiance = sun_transmittance;
    float3 sky_irradiance = GetSkyIrradiance(procSkyConsts.SkyParams, workingContext.Irradiance, camera, procSkyConsts.SunDir);
    float PhaseFunc = HenyeyGreenstein(dot(procSkyConsts.SunDir, view_dir), procSkyConsts.sky_phase_g);
    float AmbientPhaseFunc = HenyeyGreenstein(dot(procSkyConsts.SunDir, view_dir), procSkyConsts.sky_amb_phase_g);
    float Transmittance = 1.0f;
    float3 Scattering = make_float3(0.f, 0.f, 0.f);
    const float stepSize = 1.0f / (float)stepCount;
    const float SUN_RAY_LENGTH = CLOUD_HEIGHT / (float)(CLOUDS_SKY_SUN_COUNT * 4);
    for (int i = 0; i < stepCount; i++) {
        float Density = sky_cloud_density(workingContext, step, raylen, cray);
        const float fadeRange = 0.04f;
        float fade = saturate((Density) / fadeRange);
        if (fade > 0.0f) {
            Transmittance *= dm_exp(-procSkyConsts.sky_transmittance * raypart * Density * fade);
            float SunTransmittance = ComputeSunTransmittanceAtPos(workingContext, step, procSkyConsts.SunDir, raylen, cray, SUN_RAY_LENGTH, CLOUDS_SKY_SUN_COUNT);
            float3 S = procSkyConsts.sky_scattering * stepSize * (PhaseFunc * sun_direct_radiance * SunTransmittance + AmbientPhaseFunc * sky_irradiance);
            Scattering = Scattering + S * Transmittance;
        }
        cray += raypart;
        step = step + delta;
        if (cray > raylen) break;
    }
    return make_float4(Scattering, Transmittance);
}
static inline float3 sky_sun_and_atmosphere(const ProceduralSkyContext& workingContext, float3 camera, float3 eyeVec) {      // the opening of ProceduralSky / ProceduralSkyLowRes
    const ProceduralSkyConstants& C = workingContext.Consts;
    float3 sun_transmittance = make_float3(0.f, 0.f, 0.f);
    float3 radiance = GetSkyRadiance(C.SkyParams, workingContext.Transmittance, workingContext.Scatter, camera, eyeVec, C.SunDir, sun_transmittance);
    float3 sun_direct_radiance = sun_transmittance;
    sun_direct_radiance = sun_direct_radiance / C.sun_solid_angle;
    float angl = dm_acos(saturate(dot(eyeVec, C.SunDir)));
    angl /= C.SunAngularDiameter * 0.5f;
    sun_direct_radiance = sun_direct_radiance * dm_pow(saturate((1.0f - angl) * 5.0f + 0.875f), 10.0f);
    return radiance + sun_direct_radiance;
}
static inline uint sky_texel_hash(uint x, uint y, uint face) { return Hash32Combine(Hash32Combine(Hash32(x), y), face); }
// LowResPrePassLayerCS's texel: (cloud in-scatter, cloud transmittance), or zero below the horizon fade
static inline float4 ProceduralSkyLowRes(uint x, uint y, uint face, float3 viewDirection, const ProceduralSkyContext& workingContext) {
    uint randHash = sky_texel_hash(x, y, face);
    float3 eyeVec = viewDirection, camera = make_float3(0.0f, 0.0f, 6360.1f);
    float CloudsVisible = saturate((dot(normalize(camera), eyeVec) + 0.05f) * 10.0f);
    if (CloudsVisible > 0.0f) {
        const float ATM_START = 6360.1f + CLOUD_START, ATM_END = ATM_START + CLOUD_HEIGHT;
        float3 fadeDir = normalize(eyeVec + make_float3(0.0f, 0.0f, HORIZONFADE));
        float distToAtmStart = intersectSphere(camera, fadeDir, make_float3(0.0f, 0.0f, 0.0f), ATM_START);
        float distToAtmEnd = intersectSphere(camera, fadeDir, make_float3(0.0f, 0.0f, 0.0f), ATM_END);
        float raylen = distToAtmEnd - distToAtmStart;          // RaymarchClouds
        return FineRaymarching(workingContext, camera + distToAtmStart * eyeVec, eyeVec, raylen, 0.0f, raylen / (float)CLOUDS_FINE_COUNT, CLOUDS_FINE_COUNT, randHash);
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);
}
// the sky's term of GenerateTexel (EnvMapBaker.hlsl:228-236): atmosphere + sun disc, the clouds of the half-resolution pre-pass cube through a 3x3 filter
static inline float3 ProceduralSky(float3 viewDirection, const ProceduralSkyContext& workingContext, const EnvCube& lowResPrePassCube, float3 cubeDir, float3 cubeDirRight, float3 cubeDirBottom) {
    const ProceduralSkyConstants& C = workingContext.Consts;
    float3 eyeVec = viewDirection, camera = make_float3(0.0f, 0.0f, 6360.1f);
    float3 radiance = sky_sun_and_atmosphere(workingContext, camera, eyeVec);
    float CloudsVisible = saturate((dot(normalize(camera), eyeVec) + 0.02f) * 5.0f);
    if (CloudsVisible > 0.0f) {
        const float ATM_START = 6360.1f + CLOUD_START;
        float distToAtmStart = intersectSphere(camera, normalize(eyeVec + make_float3(0.0f, 0.0f, HORIZONFADE)), make_float3(0.0f, 0.0f, 0.0f), ATM_START);
        float3 spoint = camera + distToAtmStart * eyeVec;
        float4 color = make_float4(0.f, 0.f, 0.f, 0.f);
        int counter = 0;
        const int steps = 1; const float scale = 2.1f;
        for (int x = -steps; x <= steps; x++)
            for (int y = -steps; y <= steps; y++) {
                color = color + env_cube_sample_level(lowResPrePassCube, normalize(cubeDir + scale * cubeDirRight * (float)x + scale * cubeDirBottom * (float)y), 0.0f);
                counter++;
            }
        color = make_float4(color.x / (float)counter, color.y / (float)counter, color.z / (float)counter, color.w / (float)counter);
        color.w = 1.0f - ((1.0f - color.w) * CloudsVisible);
        float3 ground_transmittance = make_float3(0.f, 0.f, 0.f);
        float3 radiance_to_point = GetSkyRadianceToPoint(C.SkyParams, workingContext.Transmittance, workingContext.Scatter, camera, spoint, C.SunDir, ground_transmittance);
        float3 col = xyz(color) * ground_transmittance + radiance_to_point * 0.9f;
        radiance = lerp3(col, radiance, color.w);
    }
    return radiance * C.FinalRadianceMultiplier;
}

} // namespace ptref
