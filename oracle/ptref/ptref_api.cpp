// ORACLE (test infrastructure only) — C entry points for ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
// Scene finalisation (sub-instances, world-space triangles, BVH), light baking and the frame loop.
// Reference anchors:
//   Rtxpt/Materials/MaterialsBaker.cpp:960-1017         SubInstanceData fill (alpha cutoff quantised to 8 bit at :990)
//   Rtxpt/Lighting/LightsBaker.cpp:663-827               light order: env quads, analytic lights, emissive triangles per sub-instance
//   Rtxpt/Lighting/LightsBaker.hlsl:167-198,262-467      env quad-tree (base 4x4, 24 subdivisions, 20 boost subdivisions per node)
//   Rtxpt/Lighting/LightsBaker.hlsl:544-716              BakeEmissiveTriangles
//   Rtxpt/Lighting/LightsBaker.hlsl:738-751,880-948      ComputeWeight (flux^0.8) and ComputeProxyCounts
//   Rtxpt/Lighting/Distant/EnvMapImportanceSamplingBaker.hlsl:57-90   importance/radiance map (1024^2, 16 taps per texel)
//   Rtxpt/ProcessingPasses/AccumulationPass.hlsl:36-66 + Rtxpt/Sample.cpp:2770-2778   accumulation lerp, weight 1/(n+1)
#include "pathtracer.h"
#include "tonemap.h"
#include "neeat.h"
#include "stableplanes_oracle.h"
#include "../refpin/pin_fns.h"
#include <cstdio>
#include <cstdlib>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace ptref;

namespace ptref {

static const uint RTXPT_LIGHTING_MAX_LIGHTS = 512 * 1024, RTXPT_LIGHTING_SAMPLING_PROXY_RATIO = 12;
static const uint RTXPT_LIGHTING_MAX_SAMPLING_PROXIES_PER_LIGHT = 256 * 1024;
static const float RTXPT_LIGHTING_MIN_WEIGHT_THRESHOLD = 1e-8f;
static const uint QT_BASE_RES = 4, QT_SUBDIV = 24, QT_UNBOOSTED = QT_BASE_RES * QT_BASE_RES + 3 * QT_SUBDIV, QT_BOOST_DPT = 3, QT_BOOST_SUBDIV = 20,
                  QT_BOOST_MULT = QT_BOOST_SUBDIV * 3 + 1, QT_TOTAL = QT_UNBOOSTED * QT_BOOST_MULT;
static const uint EMISB_IMPORTANCE_MAP_DIM = 1024, EMISB_SAMPLES = 16;

void finalize_geometry(Scene& sc) {
    sc.instFirstSubInstance.clear(); sc.subInstToInstGeom.clear(); sc.subInstances.clear(); sc.tris.clear();
    uint running = 0;
    for (size_t i = 0; i < sc.instances.size(); i++) {
        const MeshDesc& m = sc.meshes[sc.instances[i].meshIndex];
        sc.instFirstSubInstance.push_back(running);
        for (uint g = 0; g < m.numGeometries; g++) {
            uint gi = m.firstGeometry + g;
            const GeometryDesc& gd = sc.geometries[gi];
            const PTMaterialData& mat = sc.materials[gd.materialIndex];
            SubInstanceData si; memset(&si, 0, sizeof(si));
            bool alphaTested = (gd.geomFlags & GEOMF_ALPHA_TESTED) && (mat.Flags & PTMaterialFlags_UseBaseOrDiffuseTexture) && (gd.flags & GEOM_HAS_UV);
            if (alphaTested) {
                si.FlagsAndAlphaInfo |= SubInstanceData::Flags_AlphaTested;
                uint cutoff = (uint)(saturate(mat.AlphaCutoff) * 255.0f);                    // MaterialsBaker.cpp:990
                si.FlagsAndAlphaInfo |= (cutoff & 0xFFu) << SubInstanceData::Flags_AlphaOffsetOffset;
                si.FlagsAndAlphaInfo |= (mat.BaseOrDiffuseTextureIndex & 0xFFFFu);
            }
            if (gd.geomFlags & GEOMF_EXCLUDE_FROM_NEE) si.FlagsAndAlphaInfo |= SubInstanceData::Flags_ExcludeFromNEE;
            si.GlobalGeometryIndex_PTMaterialDataIndex = (gi << 16) | (gd.materialIndex & 0xFFFFu);
            si.EmissiveLightMappingOffset = 0xFFFFFFFFu; si.AnalyticProxyLightIndex = 0xFFFFFFFFu;
            si.IndexOffset = gd.indexOffset; si.TexCoord1Offset = gd.vertexOffset;
            uint2 ig = {(uint)i, gi};
            sc.subInstToInstGeom.push_back(ig);
            sc.subInstances.push_back(si);
            uint subInst = running + g;
            const float3x4& M = sc.instances[i].transform;
            // non-opaque if alpha tested or excluded from NEE (AccelerationStructureUtil.h:60-70)
            uint triFlags = (alphaTested ? 1u : 0u) | ((gd.geomFlags & GEOMF_EXCLUDE_FROM_NEE) ? 3u : 0u);
            for (uint t = 0; t < gd.numIndices / 3; t++) {
                const uint* idx = &sc.indices[gd.indexOffset + 3 * t];
                float3 p0 = xform_point(M, sc.positions[gd.vertexOffset + idx[0]]);
                float3 p1 = xform_point(M, sc.positions[gd.vertexOffset + idx[1]]);
                float3 p2 = xform_point(M, sc.positions[gd.vertexOffset + idx[2]]);
                Triangle tr; tr.v0 = p0; tr.v1 = p1; tr.v2 = p2; tr.subInstance = subInst; tr.triIndex = t; tr.flags = triFlags; tr.pad = 0.f;
                sc.tris.push_back(tr);
            }
        }
        running += m.numGeometries;
    }
    // the triangles' own padded boxes (scene.h tri_box_accepts): pad from the scene bounds
    float3 smn = make_float3(3.0e38f), smx = make_float3(-3.0e38f);
    for (const Triangle& t : sc.tris) { smn = min3v(smn, min3v(t.v0, min3v(t.v1, t.v2))); smx = max3v(smx, max3v(t.v0, max3v(t.v1, t.v2))); }
    const float scenePad = sc.tris.empty() ? 0.f : scene_pad(smn, smx);
    for (Triangle& t : sc.tris) t.pad = tri_pad(min3v(t.v0, min3v(t.v1, t.v2)), max3v(t.v0, max3v(t.v1, t.v2)), scenePad);
}

// ---- binned SAH BVH2
struct BuildPrim { float3 bmin, bmax, c; };
static void tri_bounds(const Triangle& t, float3& mn, float3& mx) {      // the PADDED box of the hit definition: every node box contains it
    mn = min3v(t.v0, min3v(t.v1, t.v2)) - make_float3(t.pad); mx = max3v(t.v0, max3v(t.v1, t.v2)) + make_float3(t.pad);
}
static float half_area(float3 mn, float3 mx) { float3 e = mx - mn; return e.x * e.y + e.y * e.z + e.z * e.x; }
static void subdivide(Scene& sc, std::vector<BuildPrim>& prims, uint nodeIdx, uint first, uint count) {
    float3 mn = make_float3(1e30f), mx = make_float3(-1e30f), cmn = mn, cmx = mx;
    for (uint i = first; i < first + count; i++) {
        const BuildPrim& p = prims[sc.triOrder[i]];
        mn = min3v(mn, p.bmin); mx = max3v(mx, p.bmax); cmn = min3v(cmn, p.c); cmx = max3v(cmx, p.c);
    }
    // pad boxes a little (relative + absolute) so conservative culling never loses an acceptable hit
    float3 ext = mx - mn; float pad = 1e-5f * fmaxf_(ext.x, fmaxf_(ext.y, ext.z)) + 1e-7f;
    sc.nodes[nodeIdx].bmin = mn - make_float3(pad); sc.nodes[nodeIdx].bmax = mx + make_float3(pad);
    if (count <= 4) { sc.nodes[nodeIdx].leftFirst = first; sc.nodes[nodeIdx].count = count; return; }
    const int NB = 16; int bestAxis = -1, bestSplit = 0; float bestCost = 1e30f;
    for (int a = 0; a < 3; a++) {
        float lo = (&cmn.x)[a], hi = (&cmx.x)[a];
        if (!(hi > lo)) continue;
        float3 bmn[NB], bmx[NB]; uint bc[NB];
        for (int b = 0; b < NB; b++) { bmn[b] = make_float3(1e30f); bmx[b] = make_float3(-1e30f); bc[b] = 0; }
        float scale = (float)NB / (hi - lo);
        for (uint i = first; i < first + count; i++) {
            const BuildPrim& p = prims[sc.triOrder[i]];
            int b = (int)(((&p.c.x)[a] - lo) * scale); if (b > NB - 1) b = NB - 1; if (b < 0) b = 0;
            bmn[b] = min3v(bmn[b], p.bmin); bmx[b] = max3v(bmx[b], p.bmax); bc[b]++;
        }
        float la[NB], ra[NB]; uint lc[NB], rc[NB];
        float3 amn = make_float3(1e30f), amx = make_float3(-1e30f); uint cnt = 0;
        for (int b = 0; b < NB - 1; b++) { amn = min3v(amn, bmn[b]); amx = max3v(amx, bmx[b]); cnt += bc[b]; la[b] = cnt ? half_area(amn, amx) : 0; lc[b] = cnt; }
        amn = make_float3(1e30f); amx = make_float3(-1e30f); cnt = 0;
        for (int b = NB - 1; b > 0; b--) { amn = min3v(amn, bmn[b]); amx = max3v(amx, bmx[b]); cnt += bc[b]; ra[b - 1] = cnt ? half_area(amn, amx) : 0; rc[b - 1] = cnt; }
        for (int b = 0; b < NB - 1; b++) {
            if (!lc[b] || !rc[b]) continue;
            float cost = la[b] * (float)lc[b] + ra[b] * (float)rc[b];
            if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
        }
    }
    uint mid;
    if (bestAxis < 0) mid = first + count / 2;
    else {
        float lo = (&cmn.x)[bestAxis], hi = (&cmx.x)[bestAxis]; float scale = (float)NB / (hi - lo);
        uint* b = &sc.triOrder[first]; uint* e = b + count;
        uint* m = std::partition(b, e, [&](uint pi) {
            int bin = (int)(((&prims[pi].c.x)[bestAxis] - lo) * scale); if (bin > NB - 1) bin = NB - 1; if (bin < 0) bin = 0;
            return bin <= bestSplit; });
        mid = first + (uint)(m - b);
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    uint left = (uint)sc.nodes.size();
    sc.nodes.push_back(Scene::Node()); sc.nodes.push_back(Scene::Node());
    sc.nodes[nodeIdx].leftFirst = left; sc.nodes[nodeIdx].count = 0;
    subdivide(sc, prims, left, first, mid - first);
    subdivide(sc, prims, left + 1, mid, first + count - mid);
}
void build_bvh(Scene& sc) {
    sc.nodes.clear(); sc.triOrder.clear();
    uint n = (uint)sc.tris.size();
    if (!n) return;
    std::vector<BuildPrim> prims(n);
    sc.triOrder.resize(n);
    for (uint i = 0; i < n; i++) { tri_bounds(sc.tris[i], prims[i].bmin, prims[i].bmax); prims[i].c = (prims[i].bmin + prims[i].bmax) * 0.5f; sc.triOrder[i] = i; }
    sc.nodes.reserve(2 * n);
    sc.nodes.push_back(Scene::Node());
    subdivide(sc, prims, 0, 0, n);
}

// ---- the environment cube (EnvMapBaker: BaseLayerCS + MIPReduceCS, EnvMapBaker.hlsl:194-246, 268-371; EnvMapBaker.cpp:298-343, 540-620)
static float4 env_generate_texel(const EnvMap& e, uint px, uint py, uint face, uint dim) {      // GenerateTexel (:194-246), equirectangular source, no procedural sky
    float3 envCol = e.SampleSource(CubemapGetDirectionFor(face, make_float2(((float)px + 0.0f + 0.5f) / (float)dim, ((float)py + 0.0f + 0.5f) / (float)dim)));
    for (const EnvDirectionalLight& l : e.dirLights) envCol = envCol + EnvComputeLightContribution(px, py, face, l, dim);
    if (e.skyEnabled) {                                       // g_Const.ProcSkyEnabled (EnvMapBaker.hlsl:224-236): toLocal swaps y and z
        const float3 cubeDir = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f) / (float)dim, ((float)py + 0.5f) / (float)dim));
        const float3 cubeDirRight = CubemapGetDirectionFor(face, make_float2((((float)px + 1.0f) + 0.5f) / (float)dim, ((float)py + 0.5f) / (float)dim)) - cubeDir;
        const float3 cubeDirBottom = CubemapGetDirectionFor(face, make_float2(((float)px + 0.5f) / (float)dim, (((float)py + 1.0f) + 0.5f) / (float)dim)) - cubeDir;
        envCol = envCol + ProceduralSky(make_float3(cubeDir.x, cubeDir.z, cubeDir.y), e.sky, e.skyLowRes, cubeDir, cubeDirRight, cubeDirBottom);
    }
    envCol = envCol * kEnvMapRadianceScale;
    envCol = clamp3(envCol, 0.0f, HLF_MAX);
    return make_float4(envCol.x, envCol.y, envCol.z, 1.0f);
}
static void bake_env_cube(EnvMap& e) {
    const uint dim = e.cubeDim, levels = env_cube_mip_levels(dim);
    EnvCube& c = e.cube; memset(&c, 0, sizeof(c)); c.dim = dim; c.mipLevels = levels;
    size_t total = 0; for (uint l = 0; l < levels; l++) { c.mipOffset[l] = (uint)total; total += 6ull * (dim >> l) * (dim >> l); }
    e.cubeTexels.assign(total, make_uint2(0, 0)); c.texels = e.cubeTexels.data();
    uint2* T = e.cubeTexels.data();
    auto at = [&](uint mip, uint face, uint x, uint y) -> uint2& { uint d = dim >> mip; return T[c.mipOffset[mip] + ((size_t)face * d + y) * d + x]; };
    auto reduce = [](float4 e00, float4 e01, float4 e10, float4 e11, float4 wsa) {      // solid-angle weighted 2x2 average, summation order of the shader
        float wsum = wsa.x + wsa.y + wsa.z + wsa.w;
        float4 s = (e00 * wsa.x + e01 * wsa.y) + e10 * wsa.z + e11 * wsa.w;
        return make_float4(s.x / wsum, s.y / wsum, s.z / wsum, s.w / wsum);
    };
    const uint h = dim / 2;
    if (e.skyEnabled) {                                                                      // LowResPrePassLayerCS (EnvMapBaker.hlsl:247-265): the clouds at half resolution, RGBA16F
        for (int i = 0; i < 4; i++) (i == 0 ? e.sky.Transmittance : i == 1 ? e.sky.Scatter : i == 2 ? e.sky.Irradiance : e.sky.Clouds).texels = e.skyTex[i].data();
        e.skyLowResTexels.assign(6ull * h * h, make_uint2(0, 0));
        memset(&e.skyLowRes, 0, sizeof(e.skyLowRes)); e.skyLowRes.texels = e.skyLowResTexels.data(); e.skyLowRes.dim = h; e.skyLowRes.mipLevels = 1;
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
        for (int face = 0; face < 6; face++) for (int y = 0; y < (int)h; y++) for (uint x = 0; x < h; x++) {
            const float3 d = CubemapGetDirectionFor((uint)face, make_float2(((float)x + 0.5f) / (float)h, ((float)y + 0.5f) / (float)h));
            e.skyLowResTexels[((size_t)face * h + (uint)y) * h + x] = env_pack_rgba16f(ProceduralSkyLowRes(x, (uint)y, (uint)face, make_float3(d.x, d.z, d.y), e.sky));
        }
    }
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
    for (int face = 0; face < 6; face++) for (int y = 0; y < (int)h; y++) for (uint x = 0; x < h; x++) {          // BaseLayerCS: 4 texels of mip 0 + their mip-1 texel (from the unrounded values)
        float4 e00 = env_generate_texel(e, 2 * x, 2 * y, face, dim), e01 = env_generate_texel(e, 2 * x, 2 * y + 1, face, dim),
               e10 = env_generate_texel(e, 2 * x + 1, 2 * y, face, dim), e11 = env_generate_texel(e, 2 * x + 1, 2 * y + 1, face, dim);
        at(0, face, 2 * x, 2 * y) = env_pack_rgba16f(e00); at(0, face, 2 * x, 2 * y + 1) = env_pack_rgba16f(e01);
        at(0, face, 2 * x + 1, 2 * y) = env_pack_rgba16f(e10); at(0, face, 2 * x + 1, 2 * y + 1) = env_pack_rgba16f(e11);
        if (levels > 1) at(1, face, x, y) = env_pack_rgba16f(reduce(e00, e01, e10, e11, CubemapTexelSolidAngle4((float)dim, 2 * x, 2 * y)));
    }
    for (uint l = 2; l < levels; l++) {                                                      // MIPReduceCS: from the stored (fp16) texels of the level above
        const uint d = dim >> l;
#pragma omp parallel for schedule(static) collapse(2)
        for (int face = 0; face < 6; face++) for (int y = 0; y < (int)d; y++) for (uint x = 0; x < d; x++)
            at(l, face, x, y) = env_pack_rgba16f(reduce(env_unpack_rgba16f(at(l - 1, face, 2 * x, 2 * y)), env_unpack_rgba16f(at(l - 1, face, 2 * x, 2 * y + 1)),
                                                       env_unpack_rgba16f(at(l - 1, face, 2 * x + 1, 2 * y)), env_unpack_rgba16f(at(l - 1, face, 2 * x + 1, 2 * y + 1)),
                                                       CubemapTexelSolidAngle4((float)(d * 2), 2 * x, 2 * y)));
    }
    e.cubeSource = c; e.cubeTexelsSource.clear();
    if (e.cubeCompression) {                                                                 // the compressed cube the path tracer samples; the uncompressed one stays for the importance map
        e.cubeTexelsSource = e.cubeTexels; e.cubeSource.texels = e.cubeTexelsSource.data();
        for (uint l = 0; l < levels; l++) {
            const uint d = dim >> l, nb = d / 4u; uint2* level = T + c.mipOffset[l];
#pragma omp parallel for schedule(static) collapse(2)
            for (int face = 0; face < 6; face++) for (int by = 0; by < (int)nb; by++) for (uint bx = 0; bx < nb; bx++) env_cube_bc6_round_trip_block(level, d, (uint)face, bx, (uint)by, e.cubeCompression);
        }
    }
    e.cubeDirty = false;
}

// ---- environment importance map + quad tree (host restatement of the baker compute passes)
struct EnvImportance { uint dim, mipCount; std::vector<std::vector<float4> > mips; };   // rgb = mean radiance, w = mean (lum+avg)/2
static void build_env_importance(const Scene& sc, EnvImportance& im, uint dim = EMISB_IMPORTANCE_MAP_DIM, uint mipCount = 11) {
    im.dim = dim; im.mipCount = mipCount; im.mips.resize(im.mipCount);
    const uint sx = 4, sy = EMISB_SAMPLES / 4; const uint dimS = im.dim * sx;
    im.mips[0].resize((size_t)im.dim * im.dim);
    const float invSamples = 1.f / (float)(sx * sy);
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < (int)im.dim; y++) for (uint x = 0; x < im.dim; x++) {
        float L = 0.f; float3 R = make_float3(0.f);
        for (uint j = 0; j < sy; j++) for (uint i = 0; i < sx; i++) {
            float2 p = make_float2(((float)(x * sx + i) + 0.5f) / (float)dimS, ((float)((uint)y * sy + j) + 0.5f) / (float)(im.dim * sy));
            float3 dir = oct_to_ndir_equal_area_unorm(p);
            float3 radiance = xyz(env_cube_sample_level(sc.env.cubeSource, dir, 0.f));          // t_EnvMapCube.SampleLevel(s_LinearWrap, dir, 0) (EnvMapImportanceSamplingBaker.hlsl:77)
            L += (Luminance(radiance) + Average(radiance)) * 0.5f;
            R += radiance;
        }
        // u_RadianceMap is an RGBA16_FLOAT texture (EnvMapImportanceSamplingBaker.cpp:170): the store rounds to binary16
        im.mips[0][(size_t)y * im.dim + x] = env_round_rgba16f(make_float4(R.x * invSamples, R.y * invSamples, R.z * invSamples, L * invSamples));
    }
    for (uint l = 1; l < im.mipCount; l++) {
        uint pd = im.dim >> (l - 1), d = im.dim >> l;
        im.mips[l].resize((size_t)d * d);
        for (uint y = 0; y < d; y++) for (uint x = 0; x < d; x++) {
            const std::vector<float4>& p = im.mips[l - 1];
            float4 s = (p[(size_t)(2 * y) * pd + 2 * x] + p[(size_t)(2 * y) * pd + 2 * x + 1]) + (p[(size_t)(2 * y + 1) * pd + 2 * x] + p[(size_t)(2 * y + 1) * pd + 2 * x + 1]);
            im.mips[l][(size_t)y * d + x] = env_round_rgba16f(s * 0.25f);      // MipMapGenPass MODE_COLOR (Donut, not vendored: restated as the 2x2 mean of the stored texels, stored as binary16)
        }
    }
}
static uint firstbithigh(uint v) { uint r = 0; while (v >>= 1) r++; return r; }
static uint qt_weight(const EnvImportance& im, uint dim, uint x, uint y, uint lightIndex, uint depthLimit) {   // EnvironmentComputeWeightForQTBuild
    uint mipLevel = im.mipCount - firstbithigh(dim) - 1;
    float areaMul = (float)(1u << (mipLevel * 2));
    float radiance = im.mips[mipLevel][(size_t)y * dim + x].w;
    float ret = areaMul * radiance;
    ret = fmaxf_(sq(1.0f / 100.0f) * (float)mipLevel, ret);
    ret *= (mipLevel > depthLimit) ? 1.0f : 0.0f;
    uint v = (uint)(FastSqrt(ret) * 100 + 0.5f); if (v > 0x000FFFFFu) v = 0x000FFFFFu;
    return (v << 12) | lightIndex;
}
static float4 env_radiance_and_weight(const EnvImportance& im, float3 colorMultiplier, float distantVsLocal, uint dim, uint x, uint y) {   // EnvironmentComputeRadianceAndWeight (LightsBaker.hlsl:167-178)
    uint mipLevel = im.mipCount - firstbithigh(dim) - 1;
    float areaMul = (float)(1u << (mipLevel * 2));
    float4 value = im.mips[mipLevel][(size_t)y * dim + x];
    float weight = areaMul * fmaxf_(0.f, value.w * Average(colorMultiplier) * distantVsLocal);
    return make_float4(xyz(value) * colorMultiplier, weight);
}
static float light_weight(const PolymorphicLightInfoFull& lf) {                        // ComputeWeight (LightsBaker.hlsl:738-751)
    float flux = PolymorphicLight_GetPower(lf);
    float wt = dm_pow(flux, 0.8f);
    if (wt < RTXPT_LIGHTING_MIN_WEIGHT_THRESHOLD) wt = 0;
    return wt;
}
struct QTNode { uint dim, x, y; };
static void qt_subdivide(const EnvImportance& im, std::vector<QTNode>& nodes, std::vector<uint>& packed, uint subdivisions, uint depthLimit) {
    for (uint si = 0; si < subdivisions; si++) {
        uint best = 0; for (size_t i = 0; i < packed.size(); i++) best = std::max(best, packed[i]);
        uint gi = best & 0xFFFu;
        QTNode n = nodes[gi];
        for (uint k = 0; k < 4; k++) {
            QTNode c; c.dim = n.dim * 2; c.x = n.x * 2 + (k % 2); c.y = n.y * 2 + (k / 2);
            uint ni = (k == 0) ? gi : (uint)nodes.size();
            if (k == 0) { nodes[gi] = c; packed[gi] = qt_weight(im, c.dim, c.x, c.y, ni, depthLimit); }
            else { nodes.push_back(c); packed.push_back(qt_weight(im, c.dim, c.x, c.y, ni, depthLimit)); }
        }
    }
}
static void bake_env_quads(Scene& sc) {
    EnvImportance im; build_env_importance(sc, im);
    std::vector<QTNode> base; std::vector<uint> packed;
    for (uint li = 0; li < QT_BASE_RES * QT_BASE_RES; li++) {
        QTNode n; n.dim = QT_BASE_RES; n.x = li / QT_BASE_RES; n.y = li % QT_BASE_RES;
        base.push_back(n); packed.push_back(qt_weight(im, n.dim, n.x, n.y, li, QT_BOOST_DPT));
    }
    qt_subdivide(im, base, packed, QT_SUBDIV, QT_BOOST_DPT);
    sc.envLookupDim = im.dim; sc.envLookup.assign((size_t)im.dim * im.dim, 0);
    const float distantVsLocal = 1.0f * 0.0002f;                                  // LightsBaker.cpp:1029-1030
    for (uint g = 0; g < QT_UNBOOSTED; g++) {
        std::vector<QTNode> nodes(1, base[g]); std::vector<uint> pk(1, qt_weight(im, base[g].dim, base[g].x, base[g].y, 0, 0));
        qt_subdivide(im, nodes, pk, QT_BOOST_SUBDIV, 0);
        for (uint li = 0; li < QT_BOOST_MULT; li++) {
            EnvironmentQuadLight e; e.NodeDim = nodes[li].dim; e.NodeX = nodes[li].x; e.NodeY = nodes[li].y;
            float4 rw = env_radiance_and_weight(im, sc.env.colorMultiplier, distantVsLocal, e.NodeDim, e.NodeX, e.NodeY);
            e.Weight = rw.w; e.Radiance = xyz(rw);
            uint uniqueID = 0;
            PolymorphicLightInfoFull lf = e.Store(uniqueID);
            float2 sub = make_float2(((float)e.NodeX + 0.5f) / (float)e.NodeDim, ((float)e.NodeY + 0.5f) / (float)e.NodeDim);
            lf.Base.Center = mul_vec_mat3(oct_to_ndir_equal_area_unorm(sub), sc.env.toWorld) * DISTANT_LIGHT_DISTANCE;
            uint out = g * QT_BOOST_MULT + li;
            sc.lights[out] = lf.Base; sc.lightsEx[out] = lf.Extended;
            uint dimScale = im.dim / e.NodeDim;                                    // EnvLightsFillLookupMap
            for (uint yy = 0; yy < dimScale; yy++) for (uint xx = 0; xx < dimScale; xx++)
                sc.envLookup[(size_t)(e.NodeY * dimScale + yy) * im.dim + (e.NodeX * dimScale + xx)] = out;
        }
    }
}

void bind_light_table(Scene& sc);
void build_light_proxies(Scene& sc, uint importanceSamplingType, const std::vector<float>& w, const uint* usage, uint totalMaxFeedbackCount, float globalFeedbackUseWeight);
void bake_lights(Scene& sc, bool neeEnabled, uint importanceSamplingType) {
    sc.lights.clear(); sc.lightsEx.clear(); sc.proxyCounters.clear(); sc.proxyIndices.clear(); sc.envLookup.clear(); sc.envLookupDim = 0;
    for (size_t i = 0; i < sc.subInstances.size(); i++) { sc.subInstances[i].EmissiveLightMappingOffset = 0xFFFFFFFFu; sc.subInstances[i].AnalyticProxyLightIndex = 0xFFFFFFFFu; }
    if (neeEnabled) {
        if (sc.env.enabled) { sc.lights.resize(QT_TOTAL); sc.lightsEx.resize(QT_TOTAL); bake_env_quads(sc); }
        const uint analyticBase = (uint)sc.lights.size();
        for (size_t i = 0; i < sc.analyticLights.size(); i++) { sc.lights.push_back(sc.analyticLights[i].Base); sc.lightsEx.push_back(sc.analyticLights[i].Extended); }
        for (size_t s = 0; s < sc.subInstances.size(); s++) {          // analytic light proxies (LightsBaker.cpp:718-753)
            const uint proxy = sc.instances[sc.subInstToInstGeom[s].x].analyticProxyLight;
            if (proxy && proxy <= sc.analyticLights.size() && (sc.materials[sc.subInstances[s].GlobalGeometryIndex_PTMaterialDataIndex & 0xFFFFu].Flags & PTMaterialFlags_EnableAsAnalyticLightProxy))
                sc.subInstances[s].AnalyticProxyLightIndex = analyticBase + proxy - 1u;
        }
        // emissive triangles, in sub-instance order (LightsBaker.cpp:663-827 + LightsBaker.hlsl:544-716)
        for (size_t s = 0; s < sc.subInstances.size(); s++) {
            SubInstanceData& si = sc.subInstances[s];
            const GeometryDesc& g = sc.geometries[si.GlobalGeometryIndex_PTMaterialDataIndex >> 16];
            const PTMaterialData& mat = sc.materials[g.materialIndex];
            bool isEmissive = any_gt0(mat.EmissiveColor);                          // PTMaterial::IsEmissive (MaterialsBaker.cpp:511-514)
            uint ntri = g.numIndices / 3;
            if (!isEmissive || sc.lights.size() + ntri >= RTXPT_LIGHTING_MAX_LIGHTS) continue;
            si.EmissiveLightMappingOffset = (uint)sc.lights.size();
            const InstanceDesc& inst = sc.instances[sc.subInstToInstGeom[s].x];
            bool isFlipped = det3(inst.transform) < 0.f;
            for (uint t = 0; t < ntri; t++) {
                const uint* idx = &sc.indices[g.indexOffset + 3 * t];
                float3 p0 = xform_point(inst.transform, sc.positions[g.vertexOffset + idx[0]]);
                float3 p1 = xform_point(inst.transform, sc.positions[g.vertexOffset + idx[1]]);
                float3 p2 = xform_point(inst.transform, sc.positions[g.vertexOffset + idx[2]]);
                float3 radiance = mat.EmissiveColor;
                if ((mat.Flags & PTMaterialFlags_UseEmissiveTexture) && (g.flags & GEOM_HAS_UV)) {
                    // reference: anisotropic SampleGrad at the centroid (LightsBaker.hlsl:591-650): scene.h sample_grad_anisotropic
                    float2 uv0 = sc.uvs[g.vertexOffset + idx[0]], uv1 = sc.uvs[g.vertexOffset + idx[1]], uv2 = sc.uvs[g.vertexOffset + idx[2]];
                    float2 e0 = uv1 - uv0, e1 = uv2 - uv1, e2 = uv0 - uv2;
                    float l0 = length(e0), l1 = length(e1), l2 = length(e2);
                    float2 shortE, longE1, longE2;
                    if (l0 < l1 && l0 < l2) { shortE = e0; longE1 = e1; longE2 = e2; } else if (l1 < l2) { shortE = e1; longE1 = e2; longE2 = e0; } else { shortE = e2; longE1 = e0; longE2 = e1; }
                    float2 sg = shortE * (2.0f / 3.0f); float2 lg = (longE1 + longE2) * (1.0f / 3.0f);
                    const Texture& tex = sc.textures[mat.EmissiveTextureIndex & 0xFFFFu];
                    float2 c = (uv0 + uv1 + uv2) * (1.0f / 3.0f);
                    radiance = radiance * xyz(sample_grad_anisotropic(tex, c, sg, lg));
                }
                radiance = max3v(radiance, make_float3(0.f));
                TriangleLight tl; tl.base = p0;
                if (!isFlipped) { tl.edge1 = p1 - p0; tl.edge2 = p2 - p0; } else { tl.edge1 = p2 - p0; tl.edge2 = p1 - p0; }
                if (fmaxf_(radiance.x, fmaxf_(radiance.y, radiance.z)) < 1e-7f) radiance = make_float3(0.f);
                tl.radiance = radiance; tl.normal = make_float3(0.f); tl.surfaceArea = 0;
                PolymorphicLightInfoFull lf = tl.Store(0);
                sc.lights.push_back(lf.Base); sc.lightsEx.push_back(lf.Extended);
            }
        }
        // weights (LightsBaker.hlsl:738-751, 836-878, 118-136) — the proxies follow in build_light_proxies
        uint N = (uint)sc.lights.size();
        sc.lightWeights.assign(N, 0.f);
        for (uint i = 0; i < N; i++) {
            PolymorphicLightInfoFull lf; lf.Base = sc.lights[i]; lf.Extended = sc.lightsEx[i];
            float wt = light_weight(lf);
            if (!(wt == wt)) wt = 0;                    // (a NaN flux cannot pass `lightWeight > 0` in ComputeProxyCounts either)
            sc.lightWeights[i] = light_importance_frustum_boost(sc.lightBoost, lf, wt);      // ImportanceBooster, frustum term (off unless a view-projection matrix was supplied)
        }
        build_light_proxies(sc, importanceSamplingType, sc.lightWeights, nullptr, 0, 0.f);
    } else bind_light_table(sc);
}
// ComputeProxyCounts + the proxy fill (LightsBaker.hlsl:880-948, 1009-1060). usage != null (NEE-AT with last frame's feedback): the weights are pulled towards the counts P0 took
void build_light_proxies(Scene& sc, uint importanceSamplingType, const std::vector<float>& w, const uint* usage, uint totalMaxFeedbackCount, float globalFeedbackUseWeight) {
    const uint N = (uint)sc.lights.size();
    float weightSum = 0.f;
    for (uint i = 0; i < N; i++) weightSum += w[i];
    uint budget = RTXPT_LIGHTING_SAMPLING_PROXY_RATIO * std::max(N, RTXPT_LIGHTING_MAX_LIGHTS / 10);
    sc.proxyCounters.assign(N, 0); sc.proxyIndices.clear();
    for (uint i = 0; i < N; i++) {
        float lightWeight = w[i];
        if (usage) lightWeight = neeat_feedback_light_weight(lightWeight, usage[i], weightSum, totalMaxFeedbackCount, usage[N], globalFeedbackUseWeight);
        uint c = 0;
        if (lightWeight > 0) c = (importanceSamplingType == 0) ? 1u : (uint)ceilf(((float)(budget - N) * lightWeight) / weightSum);   // LightsBaker.hlsl:920-923 (type 0 = uniform: 1 proxy per light)
        c = std::min(c, RTXPT_LIGHTING_MAX_SAMPLING_PROXIES_PER_LIGHT - 1);
        sc.proxyCounters[i] = c;
        for (uint k = 0; k < c; k++) sc.proxyIndices.push_back(i);
    }
    bind_light_table(sc);
}
void bind_light_table(Scene& sc) {
    LightTable& T = sc.lightTable;
    T.Lights = sc.lights.data(); T.LightsEx = sc.lightsEx.data(); T.ProxyCounters = sc.proxyCounters.data(); T.ProxyIndices = sc.proxyIndices.data();
    T.TotalLightCount = (uint)sc.lights.size(); T.SamplingProxyCount = (uint)sc.proxyIndices.size();
    T.EnvLookupMap = sc.envLookup.data(); T.EnvLookupDim = sc.envLookupDim; T.EnvToWorld = sc.env.toWorld; T.WorldToEnv = sc.env.toLocal;
    sc.bindLocalSampling();
}

// NEE-AT run state: what LightsBaker keeps between frames (LightsBaker.h:225-260) and the textures / buffers its feedback passes bind
struct NeeAtState {
    bool enabled = false; float globalFeedbackWeight = 0.75f, localRatio = 0.65f, sscThreshold = 0.3f, dropoff = 0.005f, intensityDeltaMul = 64.0f; bool preFilter = true;      // SampleUI.h:158-159, LightsBaker.h:240-253
    uint updateCounter = 0; float jitterF[2] = {0, 0}; uint jitter[2] = {0, 0}, prevJitter[2] = {0, 0};
    bool feedbackFilled = false, lastFeedbackAvailable = false; uint historicTotalLightCount = 0;
    bool frameOpen = false, frameFeedbackAvailable = false, frameLocalAvailable = false; uint framePrevLightCount = 0;      // between UpdateBegin and UpdateEnd of a frame (realtime mode: the build pass runs in between)
    uint W = 0, H = 0; std::vector<float> fbW, scW, blW, histWeights, curWeights, depth, histDepth; std::vector<uint> fbC, scC, blC, local, counters;      // depth: the last traced frame's export; histDepth: the one before
    void reset() { updateCounter = 0; jitterF[0] = jitterF[1] = 0; jitter[0] = jitter[1] = prevJitter[0] = prevJitter[1] = 0; feedbackFilled = lastFeedbackAvailable = false; frameOpen = false; historicTotalLightCount = 0; W = H = 0; histWeights.clear(); }
};
// the passes as the oracle restates them (neeat.h), one call per pixel / low-resolution pixel / tile in the order a dispatch would enumerate them (the order does not matter:
// every pass reads what the previous one wrote and writes only its own slot). The reference-text harness (refpin/hlsl_pt_wrappers.inc) supplies the same interface.
struct OracleNeeAtPasses {
    void prefilter(const NeeAtFrame& F) {
        std::vector<float> sw(F.fbW, F.fbW + (size_t)F.W * F.H); std::vector<uint> sc(F.fbC, F.fbC + (size_t)F.W * F.H);
        for (uint y = 0; y < F.H; y++) for (uint x = 0; x < F.W; x++) neeat_prefilter_pixel(F, sw.data(), sc.data(), (int)x, (int)y);
    }
    void p0(const NeeAtFrame& F, uint totalThreads) {
        for (uint y = 0; y < F.H; y++) for (uint x = 0; x < F.W; x++) F.perLightCounters[neeat_p0_pixel(F, x, y)]++;
        F.perLightCounters[F.totalLightCount] += totalThreads - F.W * F.H;      // the threads beyond the frame count as "no valid feedback" (LightsBaker.hlsl:1283-1305)
    }
    void p1a(const NeeAtFrame& F) { for (uint y = 0; y < F.BH; y++) for (uint x = 0; x < F.BW; x++) neeat_p1a_pixel(F, x, y); }
    void p1b(const NeeAtFrame& F) { for (uint y = 0; y < F.H; y++) for (uint x = 0; x < F.W; x++) neeat_p1b_pixel(F, x, y); }
    void p2(const NeeAtFrame& F) { for (uint y = 0; y < F.tilesY; y++) for (uint x = 0; x < F.tilesX; x++) neeat_fill_tile(F, x, y); }
    void p3(const NeeAtFrame& F) { for (uint t = 0; t < F.tilesX * F.tilesY; t++) neeat_sort_tile(F.local + (size_t)t * RTXPT_LIGHTING_LOCAL_PROXY_COUNT); }
    void clear(const NeeAtFrame& F) { for (uint y = 0; y < F.H; y++) for (uint x = 0; x < F.W; x++) neeat_clear_pixel(F, x, y); }
};

struct Context {
    Scene sc; PtSettings S; PathTracerCameraData cam; uint w, h; std::vector<float4> accum; uint accumCount; RayCounters ctr;
    bool geomDirty, lightsDirty; NeeAtState neeat;
    std::vector<float> fbWeight; std::vector<uint> fbCand; uint fbSamples = 0;      // NEE-AT feedback reservoirs of the last render call: one plane of w x h slots per sample
    void beginFeedback(uint n) {      // LightFeedbackReservoir::Clear for every slot
        fbSamples = 0; if (!sc.feedbackRequired) return;
        fbWeight.assign((size_t)w * h * n, 0.0f); fbCand.assign((size_t)w * h * n, 0xFFFFFFFFu); fbSamples = n;
    }
};

} // namespace ptref

extern "C" {

void* ptref_create() {
    Context* c = new Context();
    memset(&c->S, 0, sizeof(c->S)); memset(&c->cam, 0, sizeof(c->cam)); memset(&c->ctr, 0, sizeof(c->ctr));
    c->w = c->h = 0; c->accumCount = 0; c->geomDirty = c->lightsDirty = true;
    c->sc.env.enabled = false; c->sc.envLookupDim = 0;
    float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; memcpy(c->sc.env.toWorld.m, I, 48); memcpy(c->sc.env.toLocal.m, I, 48);
    c->sc.env.colorMultiplier = make_float3(1.f);
    return c;
}
void ptref_destroy(void* h) { delete (Context*)h; }

void ptref_set_geometry(void* h, const uint32_t* indices, uint32_t nIdx, const float* positions, const float* uvs, const uint32_t* normals, const uint32_t* tangents,
                        uint32_t nVerts, const GeometryDesc* geoms, uint32_t nGeoms, const MeshDesc* meshes, uint32_t nMeshes) {
    Context* c = (Context*)h; Scene& sc = c->sc;
    sc.indices.assign(indices, indices + nIdx);
    sc.positions.resize(nVerts); memcpy(sc.positions.data(), positions, (size_t)nVerts * 12);
    sc.uvs.resize(nVerts); if (uvs) memcpy(sc.uvs.data(), uvs, (size_t)nVerts * 8); else memset(sc.uvs.data(), 0, (size_t)nVerts * 8);
    sc.normals.resize(nVerts); if (normals) memcpy(sc.normals.data(), normals, (size_t)nVerts * 4); else memset(sc.normals.data(), 0, (size_t)nVerts * 4);
    sc.tangents.resize(nVerts); if (tangents) memcpy(sc.tangents.data(), tangents, (size_t)nVerts * 4); else memset(sc.tangents.data(), 0, (size_t)nVerts * 4);
    sc.geometries.assign(geoms, geoms + nGeoms); sc.meshes.assign(meshes, meshes + nMeshes);
    c->geomDirty = true;
}
void ptref_set_instances(void* h, const InstanceDesc* inst, uint32_t n) { Context* c = (Context*)h; c->sc.instances.assign(inst, inst + n); c->geomDirty = true; }
// the previous frame's instance transforms and vertex positions (same counts as the scene's; NULL / 0 = did not move): Bridge::loadSurface's prevPosW in the stable-plane build pass
void ptref_set_previous_pose(void* h, const InstanceDesc* inst, uint32_t nInst, const float* positions, uint32_t nVerts) {
    Context* c = (Context*)h; Scene& sc = c->sc;
    if (inst && nInst) sc.prevInstances.assign(inst, inst + nInst); else sc.prevInstances.clear();
    if (positions && nVerts) { sc.prevPositions.resize(nVerts); memcpy(sc.prevPositions.data(), positions, (size_t)nVerts * 12); } else sc.prevPositions.clear();
}
void ptref_set_materials(void* h, const PTMaterialData* m, uint32_t n) { Context* c = (Context*)h; c->sc.materials.assign(m, m + n); c->geomDirty = true; }
// format: 0 = RGBA8 UNORM, 1 = RGBA8 sRGB (rgb decoded to linear at load, like an _SRGB view), 2 = RGBA32F
void ptref_add_texture(void* h, uint32_t w, uint32_t hgt, uint32_t format, const void* pixels) {
    Context* c = (Context*)h; Texture t; t.w = w; t.h = hgt; t.mips.resize(1); t.mips[0].resize((size_t)w * hgt);
    for (size_t i = 0; i < (size_t)w * hgt; i++) {
        float4 v;
        if (format == 2) { const float* p = (const float*)pixels + 4 * i; v = make_float4(p[0], p[1], p[2], p[3]); }
        else {
            const uint8_t* p = (const uint8_t*)pixels + 4 * i;
            v = make_float4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
            if (format == 1) { v.x = srgb_to_linear(v.x); v.y = srgb_to_linear(v.y); v.z = srgb_to_linear(v.z); }
        }
        t.mips[0][i] = v;
    }
    build_mips(t);
    c->sc.textures.push_back(t); c->geomDirty = true;
}
void ptref_clear_textures(void* h) { ((Context*)h)->sc.textures.clear(); }
// lat-long float RGB, row 0 at +Y; transform: 12 floats local->world (row major 3x4), colorMultiplier rgb; w==0 disables
// the image as a cube map (rtxpt_amd: pt_set_environment_cube): six faces of dim x dim RGBA floats, D3D face order; kept as RGBA16F texels. One image source at a time.
void ptref_set_environment_cube(void* h, const float* rgbaFaces, uint32_t dim, const float* toWorld, const float* colorMul) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    e.tex.w = e.tex.h = 0; e.tex.mips.clear();
    e.enabled = (dim != 0) || e.skyEnabled;
    e.imageCubeDim = dim; e.imageCube.resize(6ull * dim * dim);
    for (size_t i = 0; i < e.imageCube.size(); i++) e.imageCube[i] = env_pack_rgba16f(make_float4(rgbaFaces[4 * i], rgbaFaces[4 * i + 1], rgbaFaces[4 * i + 2], rgbaFaces[4 * i + 3]));
    if (toWorld) {
        memcpy(e.toWorld.m, toWorld, 48);
        float3x4 inv; memset(&inv, 0, sizeof(inv));
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) inv.m[r * 4 + k] = e.toWorld.m[k * 4 + r];
        e.toLocal = inv;
    }
    if (colorMul) e.colorMultiplier = make_float3(colorMul[0], colorMul[1], colorMul[2]);
    e.cubeDirty = true; c->lightsDirty = true;
}
void ptref_set_environment(void* h, const float* rgb, uint32_t w, uint32_t hgt, const float* toWorld, const float* colorMul) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    e.imageCubeDim = 0; e.imageCube.clear();
    e.enabled = (w != 0) || e.skyEnabled;
    if (!w) { e.tex.w = e.tex.h = 0; e.tex.mips.clear(); }
    if (w) {
        e.tex.w = w; e.tex.h = hgt; e.tex.mips.clear(); e.tex.mips.resize(1); e.tex.mips[0].resize((size_t)w * hgt);
        for (size_t i = 0; i < (size_t)w * hgt; i++) e.tex.mips[0][i] = make_float4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 1.f);
        build_mips(e.tex);
    }
    if (toWorld) {
        memcpy(e.toWorld.m, toWorld, 48);
        // inverse of a rotation = transpose (EnvMapSceneParams.InvTransform)
        float3x4 inv; memset(&inv, 0, sizeof(inv));
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) inv.m[r * 4 + k] = e.toWorld.m[k * 4 + r];
        e.toLocal = inv;
    }
    if (colorMul) e.colorMultiplier = make_float3(colorMul[0], colorMul[1], colorMul[2]);
    e.cubeDirty = true; c->lightsDirty = true;
}
// the procedural sky (rtxpt_amd: pt_set_procedural_sky): 40 floats of constants (SampleProceduralSky.hlsli:18-46), four RGBA float textures (dims: w, h, d each); consts == null switches it off
void ptref_set_procedural_sky(void* h, const float* consts, const float* const* rgba, const uint32_t* dims) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    if (!consts) { if (e.skyEnabled) { e.skyEnabled = false; if (!e.hasImage()) e.enabled = false; } e.cubeDirty = true; c->lightsDirty = true; return; }
    memcpy(&e.sky.Consts, consts, sizeof(ProceduralSkyConstants));
    SkyTexture* dst[4] = {&e.sky.Transmittance, &e.sky.Scatter, &e.sky.Irradiance, &e.sky.Clouds};
    if (rgba) for (int i = 0; i < 4; i++) {
        const size_t n = (size_t)dims[3 * i] * dims[3 * i + 1] * dims[3 * i + 2];
        e.skyTex[i].resize(n); memcpy(e.skyTex[i].data(), rgba[i], n * sizeof(float4));
        dst[i]->texels = e.skyTex[i].data(); dst[i]->w = dims[3 * i]; dst[i]->h = dims[3 * i + 1]; dst[i]->d = dims[3 * i + 2]; dst[i]->_pad = 0;
    }
    if (!e.enabled) { if (!e.hasImage()) { const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; memcpy(e.toWorld.m, I, 48); memcpy(e.toLocal.m, I, 48); e.colorMultiplier = make_float3(1.0f / kEnvMapRadianceScale); } e.enabled = true; }
    e.skyEnabled = true; e.cubeDirty = true; c->lightsDirty = true;
}
// the sky's two texel functions on their own (fixtures, the reference-text pin): mode 0 = ProceduralSkyLowRes -> 4 floats, mode 1 = ProceduralSky's atmosphere + sun
// opening (sky_sun_and_atmosphere) -> 3 floats, mode 2 = GetSkyRadianceToPoint -> radiance, transmittance (6 floats); in: n x (x, y, face as floats, then a direction / point xyz)
void ptref_sky_eval(void* h, uint32_t mode, uint32_t n, const float* in, float* out) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    for (int i = 0; i < 4; i++) (i == 0 ? e.sky.Transmittance : i == 1 ? e.sky.Scatter : i == 2 ? e.sky.Irradiance : e.sky.Clouds).texels = e.skyTex[i].data();
    const float3 camera = make_float3(0.0f, 0.0f, 6360.1f);
#pragma omp parallel for schedule(dynamic, 16)
    for (int k = 0; k < (int)n; k++) {
        const float* a = in + 6 * (size_t)k; const float3 d = make_float3(a[3], a[4], a[5]);
        if (mode == 0) { float4 r = ProceduralSkyLowRes((uint)a[0], (uint)a[1], (uint)a[2], d, e.sky); float* o = out + 4 * (size_t)k; o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
        else if (mode == 1) { float3 r = sky_sun_and_atmosphere(e.sky, camera, d); float* o = out + 3 * (size_t)k; o[0] = r.x; o[1] = r.y; o[2] = r.z; }
        else { float3 t; float3 r = GetSkyRadianceToPoint(e.sky.Consts.SkyParams, e.sky.Transmittance, e.sky.Scatter, camera, d, e.sky.Consts.SunDir, t); float* o = out + 6 * (size_t)k; o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = t.x; o[4] = t.y; o[5] = t.z; }
    }
}
// cube resolution (EnvMapBaker::m_targetResolution: 2048 for an image source) and the directional lights baked into it (Sample::UpdateLighting, Sample.cpp:1361-1388)
void ptref_set_environment_compression(void* h, uint32_t quality) { Context* c = (Context*)h; c->sc.env.cubeCompression = quality > 2u ? 2u : quality; c->sc.env.cubeDirty = true; c->lightsDirty = true; }
void ptref_set_environment_bake(void* h, uint32_t cubeDim, const EnvDirectionalLight* lights, uint32_t n) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    if (cubeDim) e.cubeDim = cubeDim;
    e.dirLights.assign(lights, lights + (lights ? n : 0));
    e.cubeDirty = true; c->lightsDirty = true;
}
// the baked cube (runs the bake if it is due): 2 words per RGBA16F texel, mips one after the other, face-major within a mip. Returns the texel count.
uint32_t ptref_get_env_cube(void* h, uint32_t* out, uint32_t capacity, uint32_t* dim, uint32_t* mipLevels) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    if (!e.enabled) return 0;
    if (e.cubeDirty) { bake_env_cube(e); c->lightsDirty = true; }
    if (dim) *dim = e.cube.dim; if (mipLevels) *mipLevels = e.cube.mipLevels;
    if (out && capacity >= e.cubeTexels.size()) memcpy(out, e.cubeTexels.data(), e.cubeTexels.size() * sizeof(uint2));
    return (uint32_t)e.cubeTexels.size();
}
// the cube compressor: BC6UCompress.hlsl's EncodeP1 restated (envcube.h) on n blocks of 16 RGB texels -> 4 words each; and the mode-11 decode -> 16 x 3 half bit patterns
void ptref_bc6_encode(const float* texels, uint32_t n, uint32_t* out) {
    for (uint32_t k = 0; k < n; k++) { float3 t[16]; for (int i = 0; i < 16; i++) t[i] = make_float3(texels[48 * k + 3 * i], texels[48 * k + 3 * i + 1], texels[48 * k + 3 * i + 2]); bc6_encode_p1(t, out + 4 * k); }
}
void ptref_bc6_encode_quality(const float* texels, uint32_t n, uint32_t* out) {      // QUALITY 1: EncodeP1 + the best two-region partition
    for (uint32_t k = 0; k < n; k++) { float3 t[16]; for (int i = 0; i < 16; i++) t[i] = make_float3(texels[48 * k + 3 * i], texels[48 * k + 3 * i + 1], texels[48 * k + 3 * i + 2]); bc6_encode_quality(t, out + 4 * k); }
}
void ptref_bc6_decode(const uint32_t* blocks, uint32_t n, uint32_t* halfBits) {
    for (uint32_t k = 0; k < n; k++) { uint hb[16][3]; bc6_decode(blocks + 4 * k, hb); for (int i = 0; i < 16; i++) for (int c = 0; c < 3; c++) halfBits[48 * k + 3 * i + c] = hb[i][c]; }
}
// level 0 of the radiance / importance map the environment quad tree is built from (dim x dim float4; the light baker uses EMISB_IMPORTANCE_MAP_DIM = 1024)
void ptref_get_env_importance(void* h, uint32_t dim, float* out) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    if (!e.enabled) return;
    if (e.cubeDirty) { bake_env_cube(e); c->lightsDirty = true; }
    EnvImportance im; build_env_importance(c->sc, im, dim, 1);
    memcpy(out, im.mips[0].data(), sizeof(float4) * (size_t)dim * dim);
}
// EnvMap::EvalLocal on the baked cube: rows (localDir.xyz, lod) -> rgb (twin of the product's pt_probe kind 9)
void ptref_env_eval(void* h, const float* in, uint32_t n, float* out) {
    Context* c = (Context*)h; EnvMap& e = c->sc.env;
    if (e.enabled && e.cubeDirty) { bake_env_cube(e); c->lightsDirty = true; }
    for (uint32_t i = 0; i < n; i++) {
        float3 r = e.enabled ? e.EvalLocal(make_float3(in[4 * i], in[4 * i + 1], in[4 * i + 2]), in[4 * i + 3]) : make_float3(0.f);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void ptref_set_lights(void* h, const PolymorphicLightInfo* base, const PolymorphicLightInfoEx* ex, uint32_t n) {
    Context* c = (Context*)h; c->sc.analyticLights.clear();
    for (uint32_t i = 0; i < n; i++) { PolymorphicLightInfoFull f; f.Base = base[i]; if (ex) f.Extended = ex[i]; else memset(&f.Extended, 0, 16); c->sc.analyticLights.push_back(f); }
    c->lightsDirty = true;
}
void ptref_set_camera(void* h, const PathTracerCameraData* cam) { ((Context*)h)->cam = *cam; }
void ptref_set_settings(void* h, const PtSettings* s) { Context* c = (Context*)h; if (c->S.NEEEnabled != s->NEEEnabled || c->S.NEEType != s->NEEType) c->lightsDirty = true; c->S = *s; }
void ptref_resize(void* h, uint32_t w, uint32_t hgt) { Context* c = (Context*)h; c->w = w; c->h = hgt; c->accum.assign((size_t)w * hgt, make_float4(0, 0, 0, 0)); c->accumCount = 0; }
void ptref_set_brute_force(void* h, int enable) { ((Context*)h)->sc.bruteForce = enable; }      // diagnostics: O(triangles) per ray
void ptref_reset_accumulation(void* h) { Context* c = (Context*)h; std::fill(c->accum.begin(), c->accum.end(), make_float4(0, 0, 0, 0)); c->accumCount = 0; memset(&c->ctr, 0, sizeof(c->ctr)); }

static void prepare(Context* c) {
    if (c->geomDirty) { finalize_geometry(c->sc); build_bvh(c->sc); c->geomDirty = false; c->lightsDirty = true; }
    if (c->sc.env.enabled && c->sc.env.cubeDirty) { bake_env_cube(c->sc.env); c->lightsDirty = true; }      // the light baker's importance map is made from the cube
    if (c->lightsDirty) { bake_lights(c->sc, c->S.NEEEnabled != 0, c->S.NEEType); c->lightsDirty = false; }
}
void ptref_prepare(void* h) { prepare((Context*)h); }

// One frame of LightsBaker::UpdateBegin + UpdateEnd for the NEE-AT layer (LightsBaker.cpp:943-962, 985-1075, 1186-1213, 1335-1420), ahead of the frame's path tracing:
// the light set is already baked (static between bakes); what changes from frame to frame is the global proxy table (usage feedback), the tile tables and the jitter.
extern "C++" {
// phases: NEEAT_BEGIN = LightsBaker::UpdateBegin (before the frame's G-buffer), NEEAT_END = UpdateEnd (after it, on the frame's depth and motion vectors: Sample.cpp:2491-2494),
// NEEAT_BOTH = reference mode, where nothing happens in between and UpdateEnd reads what the last traced frame exported (depth == nullptr) with zero motion vectors
enum { NEEAT_BEGIN = 1, NEEAT_END = 2, NEEAT_BOTH = 3 };
template <class Passes> static void neeat_frame(Context* c, Passes& P, int phases = NEEAT_BOTH, const float* depth = nullptr, const uint32_t* motion = nullptr) {
    NeeAtState& st = c->neeat; Scene& sc = c->sc;
    const uint N = (uint)sc.lights.size();
    if (!N || sc.proxyIndices.empty()) {            // nothing to sample: NEE does not run, the frame is traced without a local layer
        sc.localTable.clear(); sc.localResX = sc.localResY = 0; sc.localRatio = 0.f; sc.feedbackRequired = false; st.feedbackFilled = st.lastFeedbackAvailable = false; st.frameOpen = false; sc.bindLocalSampling(); return;
    }
    if (phases & NEEAT_BEGIN) {
        if (st.historicTotalLightCount && st.historicTotalLightCount != N) st.feedbackFilled = st.lastFeedbackAvailable = false;      // another light set: its indices mean nothing to the old reservoirs and tiles
        if (st.W != c->w || st.H != c->h) {             // (re)create the textures: LightsBaker::CreateRenderPasses (LightsBaker.cpp:300-345)
            st.W = c->w; st.H = c->h; const size_t px = (size_t)st.W * st.H, bpx = (size_t)((st.W + 1) / 2) * ((st.H + 1) / 2), tiles = (size_t)((st.W + 7) / 8 + 1) * ((st.H + 7) / 8 + 1);
            st.fbW.assign(px, 0.f); st.fbC.assign(px, 0xFFFFFFFFu); st.scW.assign(px, 0.f); st.scC.assign(px, 0xFFFFFFFFu); st.blW.assign(bpx, 0.f); st.blC.assign(bpx, 0xFFFFFFFFu);
            st.local.assign(tiles * RTXPT_LIGHTING_LOCAL_PROXY_COUNT, 0u); st.feedbackFilled = false; st.lastFeedbackAvailable = false; st.depth.assign(px, 0.f); st.histDepth.assign(px, 0.f);
        }
        st.prevJitter[0] = st.jitter[0]; st.prevJitter[1] = st.jitter[1];
        neeat_advance_jitter(st.updateCounter, st.jitterF, st.jitter);
        st.updateCounter++;
        st.frameLocalAvailable = st.lastFeedbackAvailable;      // "if last frame had temporal feedback, it will have had built local (tile) sampling"
        st.frameFeedbackAvailable = st.feedbackFilled;
        st.framePrevLightCount = st.historicTotalLightCount; st.historicTotalLightCount = N;
        st.frameOpen = true;
    }
    if (!st.frameOpen) return;                      // (UpdateEnd without UpdateBegin)
    const bool lastFrameLocalSamplesAvailable = st.frameLocalAvailable, lastFrameFeedbackAvailable = st.frameFeedbackAvailable;
    NeeAtFrame F; memset(&F, 0, sizeof(F));
    F.W = st.W; F.H = st.H; F.BW = (st.W + 1) / 2; F.BH = (st.H + 1) / 2; F.tilesX = (st.W + 7) / 8 + 1; F.tilesY = (st.H + 7) / 8 + 1;
    F.jitterX = st.jitter[0]; F.jitterY = st.jitter[1]; F.jitterPrevX = st.prevJitter[0]; F.jitterPrevY = st.prevJitter[1];
    F.updateCounter = st.updateCounter; F.dropoff = st.dropoff; F.totalLightCount = N; F.historicTotalLightCount = st.framePrevLightCount;
    F.lastFrameFeedbackAvailable = lastFrameFeedbackAvailable ? 1u : 0u; F.lastFrameLocalSamplesAvailable = (lastFrameLocalSamplesAvailable && lastFrameFeedbackAvailable) ? 1u : 0u;
    F.fbW = st.fbW.data(); F.fbC = st.fbC.data(); F.scW = st.scW.data(); F.scC = st.scC.data(); F.blW = st.blW.data(); F.blC = st.blC.data(); F.local = st.local.data();
    F.depth = depth ? depth : st.depth.data(); F.motion = (const uint2*)motion; F.historyDepth = st.histDepth.data(); F.depthDisocclusionThreshold = 1.5f;
    const uint totalMaxFeedbackCount = lastFrameFeedbackAvailable ? ((st.W + 7) / 8) * ((st.H + 7) / 8) * 64u : 0u;
    if (phases & NEEAT_BEGIN) {
        st.counters.assign(N + 1, 0u); F.perLightCounters = st.counters.data();                                    // ResetLightProxyCounters
        if (lastFrameFeedbackAvailable) { if (st.preFilter) P.prefilter(F); P.p0(F, totalMaxFeedbackCount); }
        st.curWeights = sc.lightWeights;                                                                            // ComputeWeights: the baked weight, boosted where a light got brighter
        if (lastFrameFeedbackAvailable && st.intensityDeltaMul > 0)
            for (uint i = 0; i < N; i++) st.curWeights[i] = neeat_intensity_delta_boost(st.curWeights[i], i < st.histWeights.size() ? st.histWeights[i] : 0.f, st.intensityDeltaMul);
        build_light_proxies(sc, c->S.NEEType, st.curWeights, lastFrameFeedbackAvailable ? st.counters.data() : nullptr, totalMaxFeedbackCount, lastFrameFeedbackAvailable ? st.globalFeedbackWeight : 0.f);
        st.histWeights = st.curWeights;
        st.lastFeedbackAvailable = lastFrameFeedbackAvailable;
    }
    if (!(phases & NEEAT_END)) return;
    F.perLightCounters = st.counters.data();
    F.samplingProxyCount = (uint)sc.proxyIndices.size(); F.proxies = sc.proxyIndices.data();
    // ---- UpdateEnd
    P.p1a(F); P.p1b(F); P.p2(F); P.p3(F); P.clear(F);
    if (!depth) std::fill(st.depth.begin(), st.depth.end(), 0.f);             // reference mode: Bridge::ExportSurfaceInit of every pixel of the frame about to be traced
    sc.depthExport = (sc.haveClip && !depth) ? st.depth.data() : nullptr; sc.depthWidth = st.W;      // (the fill passes of realtime mode export nothing: the build pass wrote this frame's depth)
    st.feedbackFilled = true; st.frameOpen = false;
    // what the path tracer binds this frame (LightingControlData: ratio 0 until feedback exists)
    sc.localTable = st.local; sc.localResX = F.tilesX; sc.localResY = F.tilesY; sc.localJitterX = st.jitter[0]; sc.localJitterY = st.jitter[1];
    sc.localRatio = lastFrameFeedbackAvailable ? st.localRatio : 0.f; sc.sscThreshold = st.sscThreshold; sc.feedbackRequired = true;
    sc.bindLocalSampling();
}
} // extern "C++"

// Sample::Render for n accumulated frames starting at sample index `first` (Sample.cpp:1416-1450, 2770-2778).
// Restricting to a pixel rectangle is an oracle-only convenience for bounded-time checks.
void ptref_render_rect(void* h, uint32_t first, uint32_t n, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    Context* c = (Context*)h; prepare(c);
    if (c->neeat.enabled) c->sc.feedbackRequired = true;
    c->beginFeedback(n);
    for (uint32_t s = 0; s < n; s++) {
        uint32_t sampleIndex = first + s;
        if (c->neeat.enabled) { OracleNeeAtPasses passes; neeat_frame(c, passes); }      // every sample is a frame: baker, then the path tracer
        float blend = 1.0f / (float)(c->accumCount + 1);
        RayCounters total; memset(&total, 0, sizeof(total));
#pragma omp parallel
        {
            RayCounters local; memset(&local, 0, sizeof(local));
            PathTracer pt(c->sc, c->S, c->cam, sampleIndex, &local);
            if (c->neeat.enabled) { pt.fbTotalWeight = c->neeat.fbW.data(); pt.fbCandidates = c->neeat.fbC.data(); pt.fbWidth = c->w; }      // the run's own reservoirs: they carry the 0.5 % the Clear pass kept
            else if (c->fbSamples) { pt.fbTotalWeight = c->fbWeight.data() + (size_t)c->w * c->h * s; pt.fbCandidates = c->fbCand.data() + (size_t)c->w * c->h * s; pt.fbWidth = c->w; }
#pragma omp for schedule(dynamic, 1) nowait
            for (int y = (int)y0; y < (int)y1; y++) for (uint32_t x = x0; x < x1; x++) {
                float4 col = pt.tracePixel(x, (uint32_t)y);
                float4& a = c->accum[(size_t)y * c->w + x];
                a = (blend < 1.f) ? lerp4(a, col, blend) : col;
            }
#pragma omp critical
            { total.extendRays += local.extendRays; total.shadowRays += local.shadowRays; total.hits += local.hits; total.nodeVisitsExt += local.nodeVisitsExt;
              total.triTestsExt += local.triTestsExt; total.nodeVisitsSh += local.nodeVisitsSh; total.triTestsSh += local.triTestsSh; }
        }
        c->ctr.extendRays += total.extendRays; c->ctr.shadowRays += total.shadowRays; c->ctr.hits += total.hits; c->ctr.nodeVisitsExt += total.nodeVisitsExt;
        c->ctr.triTestsExt += total.triTestsExt; c->ctr.nodeVisitsSh += total.nodeVisitsSh; c->ctr.triTestsSh += total.triTestsSh;
        c->accumCount++;
        if (c->neeat.enabled && c->fbSamples && c->neeat.fbW.size() == (size_t)c->w * c->h) { const size_t plane = (size_t)c->w * c->h;      // a copy per frame, for the tests
            memcpy(c->fbWeight.data() + plane * s, c->neeat.fbW.data(), 4 * plane); memcpy(c->fbCand.data() + plane * s, c->neeat.fbC.data(), 4 * plane); }
    }
}
// NEE-AT with the baker in the loop: every sample of a render call is one frame (LightsBaker::UpdateBegin / UpdateEnd, then the path tracer); enable = 0 leaves whatever
// ptref_set_local_light_sampling set. reset: forget the history (a new run).
void ptref_set_neeat(void* h, int enable, float globalFeedbackWeight, float localToGlobalRatio, float sscThreshold, int preFilter) {
    Context* c = (Context*)h; NeeAtState& st = c->neeat;
    if (st.enabled && !enable) { Scene& sc = c->sc; sc.localTable.clear(); sc.localResX = sc.localResY = 0; sc.localRatio = 0; sc.feedbackRequired = false; sc.bindLocalSampling(); c->lightsDirty = true; }
    st.enabled = enable != 0; st.globalFeedbackWeight = globalFeedbackWeight; st.localRatio = localToGlobalRatio; st.sscThreshold = sscThreshold; st.preFilter = preFilter != 0;
}
// ImportanceBooster's frustum term: the host's view-projection matrix (row-vector convention, 16 floats row-major; null: off), LightsBaker.h:247-249 defaults mul 8, fade 5
void ptref_set_light_importance_boost(void* h, const float* viewProj16, float mul, float fadeDistance) {
    Context* c = (Context*)h; LightFrustumBoost b; memset(&b, 0, sizeof(b));
    if (viewProj16 && mul > 0) { light_frustum_planes_from_viewproj(viewProj16, b.planes); b.mul = mul; b.fadeDistance = fadeDistance; }
    c->sc.lightBoost = b; c->lightsDirty = true;
}
// probes for the pins: ImportanceBooster (frustum term, then intensity delta) on n lights, and UpdateFrustumConsts' planes
void ptref_importance_boost(uint32_t n, const uint32_t* lights12, const float* planes20, float mul, float fade, float intensityDeltaMul, const float* hist, const float* unboosted, float* out) {
    LightFrustumBoost b; memcpy(b.planes, planes20, sizeof(b.planes)); b.mul = mul; b.fadeDistance = fade;
    for (uint32_t i = 0; i < n; i++) {
        PolymorphicLightInfoFull lf; memcpy(&lf.Base, lights12 + 12 * i, 32); memcpy(&lf.Extended, lights12 + 12 * i + 8, 16);
        float wt = light_importance_frustum_boost(b, lf, unboosted[i]);
        if (hist && intensityDeltaMul > 0) wt = neeat_intensity_delta_boost(wt, hist[i], intensityDeltaMul);
        out[i] = wt;
    }
}
// ComputeProxyCounts as bake_lights / neeat_frame run it (build_light_proxies): counts out, returns SamplingProxyCount; weightSumOut: the sum the counts were made with (light order)
uint32_t ptref_proxy_counts(uint32_t n, const float* weights, float* weightSumOut, const uint32_t* usage, uint32_t totalMaxFeedbackCount, float globalFeedbackUseWeight, uint32_t importanceSamplingType, uint32_t* counts) {
    Scene sc; sc.lights.resize(n); sc.lightsEx.resize(n); sc.envLookupDim = 0;
    std::vector<float> w(weights, weights + n);
    build_light_proxies(sc, importanceSamplingType, w, usage, totalMaxFeedbackCount, globalFeedbackUseWeight);
    memcpy(counts, sc.proxyCounters.data(), 4 * (size_t)n);
    float s = 0.f; for (uint32_t i = 0; i < n; i++) s += weights[i]; if (weightSumOut) *weightSumOut = s;
    return (uint32_t)sc.proxyIndices.size();
}
void ptref_frustum_planes(const float* m16, float* out20) { float p[5][4]; light_frustum_planes_from_viewproj(m16, p); memcpy(out20, p, sizeof(p)); }
void ptref_neeat_reset(void* h) { ((Context*)h)->neeat.reset(); }
// realtime mode (Sample.cpp:2438-2516): LightsBaker::UpdateBegin before the build pass, UpdateEnd after it on the frame's depth (float per pixel) and screen-space motion vectors
// (RGBA16F as the build pass stores them), then the fill passes sample the tiles and fill the reservoirs (ptref_fill_stable_planes). The run's own reservoirs, as in ptref_render.
void ptref_neeat_update_begin(void* h) { Context* c = (Context*)h; prepare(c); OracleNeeAtPasses passes; neeat_frame(c, passes, NEEAT_BEGIN); }
void ptref_neeat_update_end(void* h, const float* depth, const uint32_t* motionVectors) { Context* c = (Context*)h; prepare(c); OracleNeeAtPasses passes; neeat_frame(c, passes, NEEAT_END, depth, motionVectors); }
// PlanarViewConstants::matWorldToClip (row vectors, 16 floats row-major; null: no export): what the reference-mode guide-buffer dump projects the path's last vertex with
void ptref_set_view_projection(void* h, const float* m16) {
    Scene& sc = ((Context*)h)->sc;
    if (m16) { memcpy(sc.worldToClip, m16, 64); for (int r = 0; r < 4; r++) { sc.clipZ[r] = m16[4 * r + 2]; sc.clipW[r] = m16[4 * r + 3]; } sc.haveClip = true; }
    else { memset(sc.worldToClip, 0, 64); memset(sc.clipZ, 0, 16); memset(sc.clipW, 0, 16); sc.haveClip = false; sc.depthExport = nullptr; }
    sc.bindLocalSampling();
}
// the tile tables and the jitter the last frame was traced with, and the global proxy counters
// the run's own reservoirs as they stand (after a fill pass: what the next frame's UpdateBegin reads)
int ptref_neeat_get_feedback(void* h, float* totalWeight, uint32_t* candidates) {
    Context* c = (Context*)h; const NeeAtState& st = c->neeat; if (!st.W || st.fbW.size() != (size_t)st.W * st.H) return 0;
    memcpy(totalWeight, st.fbW.data(), 4 * st.fbW.size()); memcpy(candidates, st.fbC.data(), 4 * st.fbC.size()); return 1;
}
int ptref_neeat_get_tables(void* h, uint32_t* tilesXY, uint32_t* jitterXY, uint32_t* table, uint32_t* proxyCounters) {
    Context* c = (Context*)h; const NeeAtState& st = c->neeat; if (!st.W) return 0;
    if (tilesXY) { tilesXY[0] = (st.W + 7) / 8 + 1; tilesXY[1] = (st.H + 7) / 8 + 1; }
    if (jitterXY) { jitterXY[0] = st.jitter[0]; jitterXY[1] = st.jitter[1]; }
    if (table) memcpy(table, st.local.data(), 4 * st.local.size());
    if (proxyCounters) memcpy(proxyCounters, c->sc.proxyCounters.data(), 4 * c->sc.proxyCounters.size());
    return 1;
}
// NEE-AT, the path tracer's side (LightSampler.hlsli; the table is what LightsBaker's feedback passes write: tiles of 8 x 8 pixels, 128 packed entries each, sorted by light index)
void ptref_set_local_light_sampling(void* h, const uint32_t* table, uint32_t resX, uint32_t resY, uint32_t jitterX, uint32_t jitterY, float ratio, float sscThreshold, int feedback) {
    Context* c = (Context*)h; Scene& sc = c->sc;
    if (table) { sc.localTable.assign(table, table + (size_t)resX * resY * RTXPT_LIGHTING_LOCAL_PROXY_COUNT); sc.localResX = resX; sc.localResY = resY; sc.localJitterX = jitterX; sc.localJitterY = jitterY; }
    else { sc.localTable.clear(); sc.localResX = sc.localResY = sc.localJitterX = sc.localJitterY = 0; }
    sc.localRatio = ratio; sc.sscThreshold = sscThreshold; sc.feedbackRequired = feedback != 0;
    sc.bindLocalSampling();
}
int ptref_get_light_feedback(void* h, uint32_t sample, float* totalWeight, uint32_t* candidates) {
    Context* c = (Context*)h; if (sample >= c->fbSamples) return 0;
    const size_t plane = (size_t)c->w * c->h;
    memcpy(totalWeight, c->fbWeight.data() + plane * sample, 4 * plane); memcpy(candidates, c->fbCand.data() + plane * sample, 4 * plane);
    return 1;
}
// the realtime mode's pre-pass (Sample.cpp:2456-2473, PATH_TRACER_MODE_BUILD_STABLE_PLANES): stable planes, stable radiance and the guide buffers of one frame, into caller-owned arrays
// (header 4 x w x h words, planes 3 x plane-stride records of 80 bytes in tiled-swizzled order, stable radiance / motion vectors 2 words per pixel (RGBA16F), depth / specHitT floats, throughput R11G11B10)
uint32_t ptref_stable_planes_plane_stride(uint32_t w, uint32_t hgt) { return GenericTSComputePlaneStride(w, hgt); }
void ptref_build_stable_planes(void* h, uint32_t sampleIndex, const StablePlanesParams* params, uint32_t* header, void* planes, uint32_t* stableRadiance, float* depth, float* specHitT, uint32_t* motionVectors, uint32_t* throughput) {
    Context* c = (Context*)h; prepare(c);
    StablePlanesContext ctx; ctx.C = SP_make_consts(*params, c->w, c->h, c->S.bounceCount);
    ctx.B.Header = header; ctx.B.Planes = (StablePlane*)planes; ctx.B.StableRadiance = (uint2*)stableRadiance; ctx.B.Depth = depth; ctx.B.SpecularHitT = specHitT; ctx.B.MotionVectors = (uint2*)motionVectors; ctx.B.Throughput = throughput;
    RayCounters total; memset(&total, 0, sizeof(total));
#pragma omp parallel
    {
        RayCounters local; memset(&local, 0, sizeof(local));
        PathTracer pt(c->sc, c->S, c->cam, sampleIndex, &local);
        StablePlanesBuilder<PathTracer> b{pt, ctx, sampleIndex};
#pragma omp for schedule(dynamic, 1) nowait
        for (int y = 0; y < (int)c->h; y++) for (uint32_t x = 0; x < c->w; x++) sp_build_pixel(b, x, (uint32_t)y);
#pragma omp critical
        { total.extendRays += local.extendRays; total.hits += local.hits; total.nodeVisitsExt += local.nodeVisitsExt; total.triTestsExt += local.triTestsExt; }
    }
    c->ctr.extendRays += total.extendRays; c->ctr.hits += total.hits; c->ctr.nodeVisitsExt += total.nodeVisitsExt; c->ctr.triTestsExt += total.triTestsExt;
}
// one sub-sample of the realtime mode's noisy pass (Sample.cpp:2497-2516, PATH_TRACER_MODE_FILL_STABLE_PLANES) over the buffers a build pass left: the planes' noisy radiance and the specular hit distance are updated in place
void ptref_fill_stable_planes(void* h, uint32_t sampleIndex, const StablePlanesParams* params, uint32_t* header, void* planes, float* specHitT) {
    Context* c = (Context*)h; prepare(c);
    StablePlanesContext ctx; ctx.C = SP_make_consts(*params, c->w, c->h, c->S.bounceCount);
    memset(&ctx.B, 0, sizeof(ctx.B)); ctx.B.Header = header; ctx.B.Planes = (StablePlane*)planes; ctx.B.SpecularHitT = specHitT;
    RayCounters total; memset(&total, 0, sizeof(total));
#pragma omp parallel
    {
        RayCounters local; memset(&local, 0, sizeof(local));
        PathTracer pt(c->sc, c->S, c->cam, sampleIndex, &local);
        if (c->neeat.enabled && c->neeat.fbW.size() == (size_t)c->w * c->h) { pt.fbTotalWeight = c->neeat.fbW.data(); pt.fbCandidates = c->neeat.fbC.data(); pt.fbWidth = c->w; }      // realtime mode with the baker in the loop: the run's own reservoirs
        StablePlanesFiller<PathTracer> f{pt, ctx, sampleIndex};
#pragma omp for schedule(dynamic, 1) nowait
        for (int y = 0; y < (int)c->h; y++) for (uint32_t x = 0; x < c->w; x++) sp_fill_pixel(f, x, (uint32_t)y);
#pragma omp critical
        { total.extendRays += local.extendRays; total.shadowRays += local.shadowRays; total.hits += local.hits; total.nodeVisitsExt += local.nodeVisitsExt; total.triTestsExt += local.triTestsExt;
          total.nodeVisitsSh += local.nodeVisitsSh; total.triTestsSh += local.triTestsSh; }
    }
    c->ctr.extendRays += total.extendRays; c->ctr.shadowRays += total.shadowRays; c->ctr.hits += total.hits; c->ctr.nodeVisitsExt += total.nodeVisitsExt; c->ctr.triTestsExt += total.triTestsExt;
    c->ctr.nodeVisitsSh += total.nodeVisitsSh; c->ctr.triTestsSh += total.triTestsSh;
}
// DenoisingGuidesBaker::DenoiseSpecHitT (DenoisingGuidesBaker.cpp:62-84): ping into a scratch plane, pong back
void ptref_denoise_spec_hit_t(uint32_t w, uint32_t hgt, const float* depth, float* specHitT) {
    std::vector<float> scratch((size_t)w * hgt);
    for (uint32_t y = 0; y < hgt; y++) for (uint32_t x = 0; x < w; x++) scratch[(size_t)y * w + x] = SpecHitTNeighbourhood(specHitT, depth, w, hgt, (int)x, (int)y);
    for (uint32_t y = 0; y < hgt; y++) for (uint32_t x = 0; x < w; x++) specHitT[(size_t)y * w + x] = SpecHitTNeighbourhood(scratch.data(), depth, w, hgt, (int)x, (int)y);
}
// PostProcess.hlsl NO_DENOISER_FINAL_MERGE (Sample.cpp:2764-2765): the realtime frame without a denoiser, RGBA32F
void ptref_stable_planes_merge(uint32_t w, uint32_t hgt, const uint32_t* header, const void* planes, const uint32_t* stableRadiance, float* rgba) {
    StablePlanesParams p; memset(&p, 0, sizeof(p)); p.activeStablePlaneCount = 3;
    StablePlanesContext ctx; ctx.C = SP_make_consts(p, w, hgt, 0); memset(&ctx.B, 0, sizeof(ctx.B));
    ctx.B.Header = (uint32_t*)header; ctx.B.Planes = (StablePlane*)planes; ctx.B.StableRadiance = (uint2*)stableRadiance;
    for (uint32_t y = 0; y < hgt; y++) for (uint32_t x = 0; x < w; x++) { const float3 c = ctx.GetAllRadiance(x, y); float* o = rgba + 4 * ((size_t)y * w + x); o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = 1.0f; }
}
void ptref_render(void* h, uint32_t first, uint32_t n) { Context* c = (Context*)h; ptref_render_rect(h, first, n, 0, 0, c->w, c->h); }
const float* ptref_radiance(void* h) { return (const float*)((Context*)h)->accum.data(); }
void ptref_get_counters(void* h, uint64_t* out7) { memcpy(out7, &((Context*)h)->ctr, sizeof(RayCounters)); }
int ptref_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---- probes used by parity / known-answer tests
uint32_t ptref_num_tris(void* h) { Context* c = (Context*)h; prepare(c); return (uint32_t)c->sc.tris.size(); }
// rays: n x 8 floats (o.xyz, tmin, d.xyz, tmax); out: n x 4 (t, prim as uint bits, u, v). mode 0 = BVH, 1 = brute force
void ptref_trace_closest(void* h, const float* rays, uint32_t n, float* out, int mode) {
    Context* c = (Context*)h; prepare(c);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < (int)n; i++) {
        const float* r = rays + 8 * (size_t)i;
        float3 o = make_float3(r[0], r[1], r[2]), d = make_float3(r[4], r[5], r[6]);
        HitInfo hi = mode ? trace_closest_bruteforce(c->sc, o, d, r[3], r[7]) : trace_closest(c->sc, o, d, r[3], r[7]);
        float* q = out + 4 * (size_t)i; q[0] = hi.t; q[1] = asfloat(hi.prim); q[2] = hi.u; q[3] = hi.v;
    }
}
void ptref_trace_visibility(void* h, const float* rays, uint32_t n, uint32_t* outVisible) {
    Context* c = (Context*)h; prepare(c);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < (int)n; i++) {
        const float* r = rays + 8 * (size_t)i;
        outVisible[i] = trace_visibility(c->sc, make_float3(r[0], r[1], r[2]), make_float3(r[4], r[5], r[6]), r[3], r[7]) ? 1u : 0u;
    }
}
// light table read-back: returns counts; copies when pointers are non-null
void ptref_get_lights(void* h, uint32_t* nLights, uint32_t* nProxies, void* lights32, void* lightsEx16, uint32_t* proxyCounters, uint32_t* proxyIndices, uint32_t* envLookup, uint32_t* envLookupDim) {
    Context* c = (Context*)h; prepare(c); Scene& sc = c->sc;
    if (nLights) *nLights = (uint32_t)sc.lights.size(); if (nProxies) *nProxies = (uint32_t)sc.proxyIndices.size(); if (envLookupDim) *envLookupDim = sc.envLookupDim;
    if (lights32) memcpy(lights32, sc.lights.data(), sc.lights.size() * 32); if (lightsEx16) memcpy(lightsEx16, sc.lightsEx.data(), sc.lightsEx.size() * 16);
    if (proxyCounters) memcpy(proxyCounters, sc.proxyCounters.data(), sc.proxyCounters.size() * 4);
    if (proxyIndices) memcpy(proxyIndices, sc.proxyIndices.data(), sc.proxyIndices.size() * 4);
    if (envLookup) memcpy(envLookup, sc.envLookup.data(), sc.envLookup.size() * 4);
}
void ptref_get_subinstances(void* h, uint32_t* n, void* out32) {
    Context* c = (Context*)h; prepare(c);
    if (n) *n = (uint32_t)c->sc.subInstances.size(); if (out32) memcpy(out32, c->sc.subInstances.data(), c->sc.subInstances.size() * 32);
}

// ---- scalar known-answer probes
uint32_t ptref_hash32(uint32_t x) { return Hash32(x); }
uint32_t ptref_hash32_combine(uint32_t s, uint32_t v) { return Hash32Combine(s, v); }
uint32_t ptref_sobol(uint32_t index, uint32_t dim) { return bhos_sobol(index, dim); }
uint32_t ptref_owen_scramble(uint32_t x, uint32_t seed) { return bhos_owen_scramble(x, seed); }
float ptref_hash32_to_float(uint32_t x) { return Hash32ToFloat(x); }
uint32_t ptref_f32tof16(float f) { return f32tof16(f); }
float ptref_f16tof32(uint32_t h) { return f16tof32(h); }
// sample streams: kind 0 = SampleSequenceGenerator::Generate (LD), 1 = UniformSampleSequenceGenerator::Generate,
// 2 = UniformSampleSequenceGenerator::make + n x Next (floats), 3 = SampleSequenceGenerator::make(lowDiscrepancy=false) + Next
void ptref_sample_stream(uint32_t packedPixel, uint32_t vertexIndex, uint32_t sampleIndex, uint32_t effectSeed, int kind, uint32_t n, float* out) {
    SampleGeneratorVertexBase vb = SampleGeneratorVertexBase::make(packedPixel, vertexIndex, sampleIndex);
    if (kind == 0) { float4 v = SampleSequenceGenerator::Generate(n, vb, effectSeed); memcpy(out, &v, 4 * (n > 4 ? 4 : n)); }
    else if (kind == 1) { float4 v = UniformSampleSequenceGenerator::Generate(n, vb, effectSeed); memcpy(out, &v, 4 * (n > 4 ? 4 : n)); }
    else if (kind == 2) { UniformSampleSequenceGenerator g = UniformSampleSequenceGenerator::make(vb, effectSeed); for (uint32_t i = 0; i < n; i++) out[i] = sampleNext1D(g); }
    else { SampleSequenceGenerator g = SampleSequenceGenerator::make(vb, effectSeed, kind == 4); for (uint32_t i = 0; i < n; i++) out[i] = sampleNext1D(g); }
}
void ptref_dmath(int fn, const float* x, const float* y, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = dm_sin(x[i]); break; case 1: out[i] = dm_cos(x[i]); break; case 2: out[i] = dm_exp2(x[i]); break; case 3: out[i] = dm_log2(x[i]); break;
        case 4: out[i] = dm_atan2(y[i], x[i]); break; case 5: out[i] = dm_pow(x[i], y[i]); break; case 6: out[i] = FastACos(x[i]); break; default: out[i] = FastSqrt(x[i]); break;
        }
    }
}
// BSDF probe: material parameters -> eval / pdf / sample in the local frame N=(0,0,1), T=(1,0,0), B=(0,1,0).
// in: params[14] = diffuse rgb, specular rgb, roughness, metallic, transmission rgb, diffTrans, specTrans, eta ; thin, diffuseModel
// mode 0: eval(wi, wo) -> out[0..3], pdf -> out[4]; mode 1: sample(u.xyz) -> out[0..2]=wo, [3]=pdf, [4..6]=weight, [7]=lobe, [8]=lobeP, [9]=valid
void ptref_bsdf_probe(const float* params, int thin, int diffuseModel, const float* wi, const float* wo_or_u, int mode, float* out) {
    ShadingData sd; memset(&sd, 0, sizeof(sd));
    sd.N = make_float3(0, 0, 1); sd.T = make_float3(1, 0, 0); sd.B = make_float3(0, 1, 0); sd.V = make_float3(wi[0], wi[1], wi[2]);
    sd.faceNCorrected = sd.N; sd.vertexN = sd.N; sd.frontFacing = true; sd.mtl = MaterialHeader::make(); sd.mtl.setActiveLobes(Lobe_All); sd.mtl.setThinSurface(thin != 0);
    StandardBSDF b; b.diffuseModel = diffuseModel;
    b.data.diffuse = make_float3(params[0], params[1], params[2]); b.data.specular = make_float3(params[3], params[4], params[5]);
    b.data.roughness = params[6]; b.data.metallic = params[7]; b.data.transmission = make_float3(params[8], params[9], params[10]);
    b.data.diffuseTransmission = params[11]; b.data.specularTransmission = params[12]; b.data.eta = params[13];
    if (mode == 0) {
        float3 wo = make_float3(wo_or_u[0], wo_or_u[1], wo_or_u[2]);
        float4 e = b.eval(sd, wo); out[0] = e.x; out[1] = e.y; out[2] = e.z; out[3] = e.w; out[4] = b.evalPdf(sd, wo); out[5] = (float)b.getLobes();
    } else {
        BSDFSample s; memset(&s, 0, sizeof(s));
        bool v = b.sample(sd, make_float4(wo_or_u[0], wo_or_u[1], wo_or_u[2], 0), s);
        out[0] = s.wo.x; out[1] = s.wo.y; out[2] = s.wo.z; out[3] = s.pdf; out[4] = s.weight.x; out[5] = s.weight.y; out[6] = s.weight.z; out[7] = (float)s.lobe; out[8] = s.lobeP; out[9] = v ? 1.f : 0.f;
    }
}
// camera probe
void ptref_camera_ray(void* h, uint32_t px, uint32_t py, uint32_t sampleIndex, float* out6) {
    Context* c = (Context*)h; PathTracer pt(c->sc, c->S, c->cam, sampleIndex, 0); float3 o, d; pt.computeCameraRay(px, py, o, d);
    out6[0] = o.x; out6[1] = o.y; out6[2] = o.z; out6[3] = d.x; out6[4] = d.y; out6[5] = d.z;
}

// display path: n RGBA32F pixels -> n packed sRGB RGBA8 (tonemap.h)
void ptref_tonemap(const float* rgba, uint32_t n, const ToneMapParams* p, uint32_t* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = tm_pixel(*p, make_float4(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]));
}

// the oracle's restatement of the functions pinned against reference text (oracle/refpin/pin_fns.h; tests/test_oracle_refpin_hlsl.py)
// loadSurface of THIS library's build (fp32 or lp16) as 45 words per hit — the layout of refpt_surface_probe (oracle/refpin/hlsl_pt_wrappers.inc): the device's
// loadSurface is compared with it (tests/test_gpu_parity.py). rows: [u, v, dir.xyz, coneWidth, coneSpread]
void ptref_surface_probe(void* h, uint32_t n, const uint32_t* prims, const float* uvDirCone, uint32_t* out) {
    Context* c = (Context*)h; prepare(c);
    PathTracer svc(c->sc, c->S, c->cam, 0, nullptr);
    for (uint32_t k = 0; k < n; k++) {
        const float* a = uvDirCone + 7 * k;
        RayCone rc = RayCone::make(a[5], a[6]);
        SurfaceData q = svc.loadSurface(prims[k], a[0], a[1], make_float3(a[2], a[3], a[4]), rc);
        uint32_t* o = out + 45 * k; const ShadingData& s = q.shadingData; const StandardBSDFData& b = q.bsdf.data;
        auto put3 = [&](float3 v) { *o++ = asuint(v.x); *o++ = asuint(v.y); *o++ = asuint(v.z); };
        put3(s.posW); put3(s.faceNCorrected); put3(s.V); put3(s.N); put3(s.T); put3(s.B); put3(s.vertexN);
        *o++ = s.frontFacing; *o++ = s.mtl.packedData; *o++ = s.materialID; *o++ = asuint(s.IoR); *o++ = asuint(s.shadowNoLFadeout); put3(s.emission);
        put3(b.diffuse); *o++ = asuint(b.roughness); put3(b.specular); *o++ = asuint(b.metallic); put3(b.transmission);
        *o++ = asuint(b.diffuseTransmission); *o++ = asuint(b.specularTransmission); *o++ = asuint(b.eta); *o++ = asuint(q.interiorIoR); *o++ = q.neeTriangleLightIndex;
    }
}
void ptref_pin_call(int fn, const float* in, unsigned n, float* out) {
    const int ni = kPinArity[fn][0], no = kPinArity[fn][1];
    for (unsigned k = 0; k < n; k++) {
        const float* a = in + (size_t)k * ni; float* o = out + (size_t)k * no;
        switch (fn) {
        case PIN_evalFresnelSchlick: o[0] = evalFresnelSchlick(a[0], a[1], a[2]); break;
        case PIN_evalFresnelSchlick3: { float3 r = evalFresnelSchlick(make_float3(a[0], a[1], a[2]), a[3], a[4]); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case PIN_evalFresnelDielectric: { float ct = 0.f; o[0] = evalFresnelDielectric(a[0], a[1], ct); o[1] = ct; } break;
        case PIN_evalNdfGGX: o[0] = evalNdfGGX(a[0], a[1]); break;
        case PIN_evalPdfGGX_BVNDF: o[0] = evalPdfGGX_BVNDF(a[0], make_float3(a[1], a[2], a[3]), make_float3(a[4], a[5], a[6])); break;
        case PIN_sampleGGX_BVNDF: { float3 r = sampleGGX_BVNDF(a[0], make_float3(a[1], a[2], a[3]), make_float2(a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case PIN_evalLambdaGGX: o[0] = evalLambdaGGX(a[0], a[1]); break;
        case PIN_evalMaskingSmithGGXCorrelated: o[0] = evalMaskingSmithGGXCorrelated(a[0], a[1], a[2]); break;
        case PIN_ndir_to_oct_equal_area_unorm: { float2 r = ndir_to_oct_equal_area_unorm(make_float3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; } break;
        case PIN_oct_to_ndir_equal_area_unorm: { float3 r = oct_to_ndir_equal_area_unorm(make_float2(a[0], a[1])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case PIN_sample_disk: { float2 r = sample_disk(make_float2(a[0], a[1])); o[0] = r.x; o[1] = r.y; } break;
        case PIN_sample_disk_concentric: { float2 r = sample_disk_concentric(make_float2(a[0], a[1])); o[0] = r.x; o[1] = r.y; } break;
        case PIN_sample_cosine_hemisphere_concentric: { float pdf = 0.f; float3 r = sample_cosine_hemisphere_concentric(make_float2(a[0], a[1]), pdf); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = pdf; } break;
        case PIN_perp_stark: { float3 r = perp_stark(make_float3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case PIN_ComputeRayOrigin: { float3 r = ComputeRayOrigin(make_float3(a[0], a[1], a[2]), make_float3(a[3], a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case PIN_FastSqrt: o[0] = FastSqrt(a[0]); break;
        case PIN_FastACos: o[0] = FastACos(a[0]); break;
        case PIN_ComputeRayConeSpreadAngleExpansionByScatterPDF: o[0] = ComputeRayConeSpreadAngleExpansionByScatterPDF(a[0], a[1]); break;
        case PIN_ComputeNewScatterFireflyFilterK: o[0] = ComputeNewScatterFireflyFilterK(a[0], a[1], a[2]); break;
        case PIN_FireflyFilter: { float3 r = FireflyFilter(make_float3(a[0], a[1], a[2]), a[3], a[4]); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case PIN_FireflyFilterShort: o[0] = FireflyFilterShort(a[0], a[1], a[2]); break;
        case PIN_ComputeLowGrazingAngleFalloff: o[0] = ComputeLowGrazingAngleFalloff(make_float3(a[0], a[1], a[2]), make_float3(a[3], a[4], a[5]), a[6], a[7]); break;
        default: break;
        }
    }
}

// lights: the oracle's side of refhlsl_light_probe (oracle/refpin/hlsl_wrappers.inc documents the kinds); environment transform = identity
static PolymorphicLightInfoFull pin_info(const uint32_t* w) {
    PolymorphicLightInfoFull li; memset(&li, 0, sizeof(li));
    li.Base.Center = make_float3(asfloat(w[0]), asfloat(w[1]), asfloat(w[2])); li.Base.ColorTypeAndFlags = w[3]; li.Base.Direction1 = w[4]; li.Base.Direction2 = w[5]; li.Base.Scalars = w[6]; li.Base.LogRadiance = w[7];
    li.Extended.IesProfileIndex = w[8]; li.Extended.PrimaryAxis = w[9]; li.Extended.CosConeAngleAndSoftness = w[10]; li.Extended.UniqueID = w[11];
    return li;
}
void ptref_light_probe(int kind, const uint32_t* in, unsigned n, uint32_t* out) {
    static const int NI[5] = {3, 12, 17, 18, 3}, NO[5] = {5, 12, 12, 1, 4};
    const float3x4 I = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
    for (unsigned k = 0; k < n; k++) {
        const uint32_t* a = in + (size_t)k * NI[kind]; uint32_t* o = out + (size_t)k * NO[kind];
        if (kind == 0) {
            PolymorphicLightInfo b; memset(&b, 0, sizeof(b));
            PackLightColor(make_float3(asfloat(a[0]), asfloat(a[1]), asfloat(a[2])), b);
            float3 c = UnpackLightColor(b);
            o[0] = b.ColorTypeAndFlags; o[1] = b.LogRadiance; o[2] = asuint(c.x); o[3] = asuint(c.y); o[4] = asuint(c.z);
        } else if (kind == 1) {
            TriangleLight t; t.base = make_float3(asfloat(a[0]), asfloat(a[1]), asfloat(a[2])); t.edge1 = make_float3(asfloat(a[3]), asfloat(a[4]), asfloat(a[5])); t.edge2 = make_float3(asfloat(a[6]), asfloat(a[7]), asfloat(a[8]));
            t.radiance = make_float3(asfloat(a[9]), asfloat(a[10]), asfloat(a[11])); t.normal = make_float3(0.f); t.surfaceArea = 0;
            PolymorphicLightInfoFull li = t.Store(7u);
            o[0] = asuint(li.Base.Center.x); o[1] = asuint(li.Base.Center.y); o[2] = asuint(li.Base.Center.z); o[3] = li.Base.ColorTypeAndFlags; o[4] = li.Base.Direction1; o[5] = li.Base.Direction2; o[6] = li.Base.Scalars; o[7] = li.Base.LogRadiance;
            o[8] = li.Extended.IesProfileIndex; o[9] = li.Extended.PrimaryAxis; o[10] = li.Extended.CosConeAngleAndSoftness; o[11] = li.Extended.UniqueID;
        } else if (kind == 2) {
            PolymorphicLightInfoFull li = pin_info(a);
            PolymorphicLightSample s = PolymorphicLight_CalcSample(li, make_float2(asfloat(a[12]), asfloat(a[13])), make_float3(asfloat(a[14]), asfloat(a[15]), asfloat(a[16])), I);
            o[0] = asuint(s.Position.x); o[1] = asuint(s.Position.y); o[2] = asuint(s.Position.z); o[3] = asuint(s.Normal.x); o[4] = asuint(s.Normal.y); o[5] = asuint(s.Normal.z);
            o[6] = asuint(s.Radiance.x); o[7] = asuint(s.Radiance.y); o[8] = asuint(s.Radiance.z); o[9] = asuint(s.SolidAnglePdf); o[10] = s.LightSampleableByBSDF ? 1u : 0u;
            o[11] = asuint(PolymorphicLight_GetPower(li));
        } else if (kind == 3) {
            TriangleLight t = TriangleLight::Create(pin_info(a));
            o[0] = asuint(t.CalcSolidAnglePdfForMIS(make_float3(asfloat(a[12]), asfloat(a[13]), asfloat(a[14])), make_float3(asfloat(a[15]), asfloat(a[16]), asfloat(a[17]))));
        } else {
            uint p = NDirToOctUnorm32(make_float3(asfloat(a[0]), asfloat(a[1]), asfloat(a[2])));
            float3 d = OctToNDirUnorm32(p);
            o[0] = p; o[1] = asuint(d.x); o[2] = asuint(d.y); o[3] = asuint(d.z);
        }
    }
}

// display path before the SRGBA8 store: tm_apply (tonemap.h) as floats, the oracle's side of refhlsl_tonemap
void ptref_tonemap_linear(const float* rgba, uint32_t n, const ToneMapParams* p, float* out) {
    for (uint32_t i = 0; i < n; i++) { float3 c = tm_apply(*p, make_float3(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2])); out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = rgba[4 * i + 3]; }
}

// light baking: the oracle's side of refhlsl_lightbake_probe. mips: pyramid of float4 (rgb mean radiance, w importance), mip l has dims[l]^2 texels.
// kind 0: rows [12-word light record] -> ComputeWeight. kind 1: rows [dim, x, y, lightIndex, depthLimit] -> [QT weight word, radiance rgb, weight] (5 words)
void ptref_lightbake_probe(int kind, const uint32_t* in, unsigned n, uint32_t* out, const float* const* mips, const uint32_t* dims, uint32_t mipCount, const float* colorMul, float distantVsLocal) {
    EnvImportance im; im.mipCount = mipCount; im.dim = dims ? dims[0] : 0;
    if (kind == 1) { im.mips.resize(mipCount); for (uint32_t l = 0; l < mipCount; l++) { im.mips[l].resize((size_t)dims[l] * dims[l]); memcpy(im.mips[l].data(), mips[l], im.mips[l].size() * 16); } }
    for (unsigned k = 0; k < n; k++) {
        if (kind == 0) out[k] = asuint(light_weight(pin_info(in + 12 * k)));
        else {
            const uint32_t* a = in + 5 * k; uint32_t* o = out + 5 * k;
            o[0] = qt_weight(im, a[0], a[1], a[2], a[3], a[4]);
            float4 rw = env_radiance_and_weight(im, make_float3(colorMul[0], colorMul[1], colorMul[2]), distantVsLocal, a[0], a[1], a[2]);
            o[1] = asuint(rw.x); o[2] = asuint(rw.y); o[3] = asuint(rw.z); o[4] = asuint(rw.w);
        }
    }
}

} // extern "C"




