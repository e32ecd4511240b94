// ORACLE pin (test infrastructure only): stand-ins for what Rtxpt/Lighting/Distant/EnvMapImportanceSamplingBaker.hlsl binds, so that its
// BuildMIPDescentImportanceMapCS text compiles as C++ and runs over the oracle's baked environment cube. Included inside namespace hl::emisb by hlsl_tu.py
// --integrator, after struct EnvMapImportanceSamplingBakerConstants.
//   * t_EnvMapCube + s_LinearWrap: the cube fetch restated in oracle/ptref/envcube.h (hardware behaviour, no reference text)
//   * u_ImportanceMap: R32_FLOAT, u_RadianceMap: RGBA16_FLOAT (EnvMapImportanceSamplingBaker.cpp:159, 170) — the store to the latter rounds to binary16
static EnvMapImportanceSamplingBakerConstants g_BuilderConsts;
struct PinCube { const ptref::EnvCube* cube = nullptr;
    float4 SampleLevel(SamplerState, float3 dir, float lod) const { ptref::float4 c = ptref::env_cube_sample_level(*cube, ptref::make_float3(dir.x, dir.y, dir.z), lod); return float4(c.x, c.y, c.z, c.w); } };
struct PinR32 { float* p = nullptr; uint w = 0; float& operator[](uint2 c) { return p[(size_t)c.y * w + c.x]; } };
struct PinRGBA16F { ptref::float4* p = nullptr; uint w = 0;
    struct Ref { ptref::float4* q; void operator=(float4 v) { *q = ptref::env_round_rgba16f(ptref::make_float4(v.x, v.y, v.z, v.w)); } };
    Ref operator[](uint2 c) { return Ref{p + (size_t)c.y * w + c.x}; } };
static PinCube t_EnvMapCube; static PinR32 u_ImportanceMap; static PinRGBA16F u_RadianceMap; static SamplerState s_PointClamp, s_LinearWrap;
