// ORACLE pin (test infrastructure only): what Rtxpt/Lighting/Distant/EnvMapBaker.hlsl binds, as stand-ins, so that its BaseLayerCS / MIPReduceCS text
// (and GenerateTexel, SampleSource, ComputeLightContribution, CubemapGetDirectionFor, CubemapTexelSolidAngle4 under them) compiles as C++ and runs
// over the oracle's environment source image. Included inside namespace hl::embake by hlsl_tu.py --integrator, after struct EMB_DirectionalLight.
//   * the constant buffer with the procedural-sky block (SampleProceduralSky.hlsli + precomputed_sky.hlsli are compiled ahead of this file: hlsl_tu.py); the sky's look-up
//     textures are the oracle's SkyTexture arrays behind Texture2D / Texture3D stand-ins, the half-resolution pre-pass cube is read through the oracle's cube fetch
//   * t_SrcEquirectangularEnvMap + s_EquiRectSampler (linear, wrap u / clamp v: EnvMapBaker.cpp:92-98): the oracle's bilinear fetch with the v clamp
//   * t_SrcCubemapEnvMap (a cube map as the source image, BackgroundSourceType 2): the oracle's cube fetch over the image cube
//   * the RGBA16_FLOAT cube UAVs: a store rounds to binary16 (round-to-nearest-even), a load widens
struct PinEMBConsts { EMB_DirectionalLight DirectionalLights[EMB_MAXDIRLIGHTS]; ProceduralSkyConstants ProcSkyConsts; float3 ScaleColor; uint DirectionalLightCount, CubeDim, CubeDimLowRes, ProcSkyEnabled, BackgroundSourceType; };
static PinEMBConsts g_Const;
struct PinEquirect { const ptref::Texture* tex = nullptr;
    float4 SampleLevel(SamplerState, float2 uv, float) const { if (!tex || !tex->w) return float4(0.f, 0.f, 0.f, 0.f);
        ptref::float2 q = ptref::make_float2(uv.x, uv.y); const float mh = (float)tex->h;
        q.y = ptref::clampf(q.y, 0.5f / mh, 1.0f - 0.5f / mh);
        ptref::float4 c = ptref::sample_bilinear(*tex, 0, q); return float4(c.x, c.y, c.z, c.w); } };
struct PinCubeUAV { ptref::uint2* texels = nullptr; uint dim = 0;
    struct Ref { ptref::uint2* p;
        void operator=(float4 v) { *p = ptref::env_pack_rgba16f(ptref::make_float4(v.x, v.y, v.z, v.w)); }
        operator float4() const { ptref::float4 c = ptref::env_unpack_rgba16f(*p); return float4(c.x, c.y, c.z, c.w); }
        float4 operator*(float w) const { return float4(*this) * w; } };
    Ref operator[](uint3 c) { return Ref{texels + ((size_t)c.z * dim + c.y) * dim + c.x}; }
    void GetDimensions(uint& w, uint& h, uint& e) const { w = dim; h = dim; e = 6; } };
static PinCubeUAV u_EnvMapCubeFacesDst0, u_EnvMapCubeFacesDst1, u_EnvMapCubeFacesDst, u_EnvMapCubeFacesSrc;
static PinEquirect t_SrcEquirectangularEnvMap; static TextureCube<float4> t_SrcCubemapEnvMap, t_LowResPrePassCube;
static Texture2D<float4> t_ProcSkyTransmittance, t_ProcSkyIrradiance, t_ProcSkyNoise; static Texture3D t_ProcSkyScatter, t_ProcSkyClouds;
static SamplerState s_Point, s_Linear, s_EquiRectSampler;
