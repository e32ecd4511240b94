// Host side of the procedural sky: SampleProceduralSky::Update (Rtxpt/Lighting/Distant/SampleProceduralSky.cpp:67-153) with the members it reads
// (SampleProceduralSky.h:70-86) — scene time and preset name -> the constant block of SampleProceduralSky.hlsli that pt_set_procedural_sky takes. Restated statement by
// statement, including the two assignments whose right-hand sides are comma expressions (GroundAlbedo keeps 0.3 in all channels, physical_sky_ground_radiance the last
// of its three numbers: neither is read by the shader). Donut's dm::rotation(euler) and affine3 product are not vendored under /root/reference: restated from Donut's
// published affine.h (row vectors, v * M; rotation about x, then y, then z) — unpinned.
#include "../../include/mi355pt.h"
#include <cmath>
#include <cstring>
#include <cfloat>

namespace {
const float PI_f = 3.141592654f;          // donut/core/math/basics.h
inline float lerpf(float a, float b, float t) { return a + (b - a) * t; }
inline float TimeIndependentLerpF(float deltaTime, float lerpRate) { return 1.0f - expf(-fabsf(deltaTime * lerpRate)); }      // SampleProceduralSky.cpp:62-65
// v * R with R = rotation((ax, 0, 0)) * rotation((0, ay, 0)) * rotation((0, 0, az)): Donut's rows for the three axes
void rotate_xyz(float v[3], float ax, float ay, float az) {
    float s = sinf(ax), c = cosf(ax); { float y = v[1] * c + v[2] * -s, z = v[1] * s + v[2] * c; v[1] = y; v[2] = z; }                  // rows (1,0,0) (0,c,s) (0,-s,c)
    s = sinf(ay); c = cosf(ay);       { float x = v[0] * c + v[2] * s, z = v[0] * -s + v[2] * c; v[0] = x; v[2] = z; }                  // rows (c,0,-s) (0,1,0) (s,0,c)
    s = sinf(az); c = cosf(az);       { float x = v[0] * c + v[1] * -s, y = v[0] * s + v[1] * c; v[0] = x; v[1] = y; }                  // rows (c,s,0) (-s,c,0) (0,0,1)
}
} // namespace

extern "C" void pt_procedural_sky_default_params(PtProceduralSkyParams* p) {          // SampleProceduralSky.h:70-82
    if (!p) return;
    p->colorTint[0] = 1.45f; p->colorTint[1] = 1.29f; p->colorTint[2] = 1.27f; p->brightness = 1.0f; p->sunBrightness = 5.0f; p->cloudsMovementSpeed = 0.8f;
    p->timeOfDayMovementSpeed = 300.0f; p->sunTimeOfDayOffset = -0.4f; p->sunEastWestRotation = 0.0f; p->sunAngularDiameterDeg = 0.5332f;
    p->cloudDensityOffset = 0.75f; p->cloudTransmittance = 2.5f; p->cloudScattering = 2.0f;
}
extern "C" int32_t pt_procedural_sky_update(PtProceduralSkyState* st, const PtProceduralSkyParams* params, double sceneTime, const char* preset, int32_t forceInstantUpdate, PtProceduralSkyConstants* out) {
    if (!st || !out) return -PT_ERROR_INVALID_ARGUMENT;
    PtProceduralSkyParams P; if (params) P = *params; else pt_procedural_sky_default_params(&P);
    const char* base = "==PROCEDURAL_SKY==";
    if (!preset) preset = base;
    if (strncmp(preset, base, 12) != 0) return -PT_ERROR_INVALID_ARGUMENT;               // IsProceduralSky (SampleCommon.h:78): the first 12 characters
    PtProceduralSkyConstants& o = *out; memset(&o, 0, sizeof(o));
    for (int i = 0; i < 3; i++) o.FinalRadianceMultiplier[i] = P.brightness * P.colorTint[i] * P.sunBrightness;
    const float cloudsLoopLength = 60 * 60 * 24;
    o.CloudsTime = (float)fmod(sceneTime * P.cloudsMovementSpeed, cloudsLoopLength);
    o.GroundAlbedo[0] = o.GroundAlbedo[1] = o.GroundAlbedo[2] = 0.3f;                     // `= 0.3f, 0.15f, 0.14f;`
    o.SunAngularDiameter = P.sunAngularDiameterDeg / 180.0f * PI_f;
    const float star[3] = {1.47399998f, 1.85039997f, 1.91198003f};
    for (int i = 0; i < 3; i++) o.SkyParams.StarIrradiance[i] = star[i] * P.sunBrightness;
    o.SkyParams.StarAngularDiameter = o.SunAngularDiameter;
    o.SkyParams.RayleightScatteringRGB[0] = 0.00580233941f; o.SkyParams.RayleightScatteringRGB[1] = 0.0135577619f; o.SkyParams.RayleightScatteringRGB[2] = 0.0331000052f;
    o.SkyParams.PlanetSurfaceRadius = 6360.00000f;
    o.SkyParams.MieScatteringRGB[0] = o.SkyParams.MieScatteringRGB[1] = o.SkyParams.MieScatteringRGB[2] = 0.00149850000f;
    o.SkyParams.PlanetAtmosphereRadius = 6420.00000f; o.SkyParams.MieHenyeyGreensteinG = 0.8f; o.SkyParams.SqDistanceToHorizontalBoundary = 766800.000f;
    o.SkyParams.AtmosphereHeight = 60.0f; o.SkyParams.reserved = 0.0f;
    o.sun_solid_angle = 2 * PI_f * (float)(1.0 - cos(0.5 * o.SunAngularDiameter));
    o.cloud_density_offset = P.cloudDensityOffset; o.sky_transmittance = P.cloudTransmittance; o.sky_phase_g = 0.9f; o.sky_amb_phase_g = 0.3f; o.sky_scattering = P.cloudScattering;
    o.physical_sky_ground_radiance[0] = o.physical_sky_ground_radiance[1] = o.physical_sky_ground_radiance[2] = 0.00655480893f;      // `= (0.177055925f, 0.0584776886f, 0.00655480893f);`

    float timeOfTheDay = (float)fmod((sceneTime * P.timeOfDayMovementSpeed) / float(60 * 60 * 24) + P.sunTimeOfDayOffset + 1.0f, 2.0f) - 1.0f;
    if (strcmp(preset, base) != 0) {
        double dt = sceneTime - st->lastSceneTime; if (dt < 0.0) dt = 0.0; if (dt > 0.3) dt = 0.3;
        const float deltaTime = (float)dt;
        float timeOfDayTarget = -FLT_MAX;
        if (!strcmp(preset, "==PROCEDURAL_SKY_MORNING==")) timeOfDayTarget = -0.25f;
        else if (!strcmp(preset, "==PROCEDURAL_SKY_MIDDAY==")) timeOfDayTarget = 0.1f;
        else if (!strcmp(preset, "==PROCEDURAL_SKY_EVENING==")) timeOfDayTarget = 0.51f;
        else if (!strcmp(preset, "==PROCEDURAL_SKY_DAWN==")) timeOfDayTarget = 0.63f;
        else if (!strcmp(preset, "==PROCEDURAL_SKY_PITCHBLACK==")) { timeOfDayTarget = 1.0f; o.FinalRadianceMultiplier[0] = o.FinalRadianceMultiplier[1] = o.FinalRadianceMultiplier[2] = 0.0f; }
        if (timeOfDayTarget == -FLT_MAX) return -PT_ERROR_INVALID_ARGUMENT;               // (the reference asserts)
        float lerpK = TimeIndependentLerpF(deltaTime, 0.1f);
        if (forceInstantUpdate) lerpK = 1.0f;
        st->timeOfDayL1 = lerpf(st->timeOfDayL1, timeOfDayTarget, lerpK);
        st->timeOfDayL2 = lerpf(st->timeOfDayL2, st->timeOfDayL1, lerpK);
        timeOfTheDay = (fabsf(timeOfDayTarget - st->timeOfDayL2) < 1e-4f) ? timeOfDayTarget : st->timeOfDayL2;
    } else st->timeOfDayL1 = st->timeOfDayL2 = timeOfTheDay;
    // (m_lastSceneTime is declared, initialised to 0 and never written in the reference: deltaTime is clamp(sceneTime, 0, 0.3) there; the state keeps the field, unwritten, for the same result)

    float sunDir[3] = {cosf(timeOfTheDay * PI_f), 0.f, sinf(timeOfTheDay * PI_f)};
    { const float l = sqrtf(sunDir[0] * sunDir[0] + sunDir[1] * sunDir[1] + sunDir[2] * sunDir[2]); for (int i = 0; i < 3; i++) sunDir[i] /= l; }
    rotate_xyz(sunDir, -0.8f, -1.1f, P.sunEastWestRotation * (PI_f / 180.f));
    memcpy(o.SunDir, sunDir, sizeof(sunDir));
    const int changes = memcmp(&o, &st->lastConstants, sizeof(o)) != 0;
    st->lastConstants = o;
    return changes;
}
