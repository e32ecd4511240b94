// ORACLE (test infrastructure only) — NEE-AT, the light baker's feedback passes: last frame's per-pixel feedback reservoirs -> this frame's
// screen-tile local samplers and the usage counts that re-weight the global sampler. CPU restatement of LightsBaker.hlsl:1062-1855, :753-830, :880-948, :118-162,
// LightsBaker.cpp:943-962, 985-1075, 1335-1420, MicroRng.hlsli and LightingTypes.hlsli:184-320; pinned against that text by tests/test_neeat_baker.py
// (oracle/refpin/hlsl_lbfb_stubs.h runs the reference's passes thread by thread). See rtxpt_amd/csrc/pt_neeat.h for the order of a frame and the three stated differences
// (PreFilter from a snapshot, identity light remap; reference-mode reprojection — zero motion vectors, depth test on the exported path-end depth — is restated).
#pragma once
#include "lights.h"

namespace ptref {

// Shaders/Libraries/MicroRng.hlsli:12-62
struct MicroRng {
    uint N;
    static MicroRng make(uint x, uint y, uint seedValueA, uint seedValueB) {
        MicroRng r; r.N = ((x << 16) | y) ^ 0x9e3779b9u;
        r.N = r.N ^ (seedValueA + (r.N << 6) + (r.N >> 2));
        r.N = r.N ^ (seedValueB + (r.N << 6) + (r.N >> 2));
        return r;
    }
    uint Next() { N ^= N >> 16; N *= 0x21f0aaadu; N ^= N >> 15; N *= 0xf35a2d97u; N ^= N >> 15; return N; }
    float NextFloat() { return (float)(Next() >> 8) / 16777216.0f; }
};

// the bindings of the passes (LightsBaker.cpp FillBindings) and the per-frame constants (:1004-1075)
struct NeeAtFrame {
    uint W, H;                          // FeedbackResolution (= the frame)
    uint BW, BH;                        // BlendedFeedbackResolution = div_ceil(W, 2) x div_ceil(H, 2)   (RTXPT_NEEAT_EARLY_FEEDBACK_TILE_SIZE 2, LightsBaker.cpp:324)
    uint tilesX, tilesY;                // LocalSamplingResolution = div_ceil(W, 8) + 1 x div_ceil(H, 8) + 1 (LightsBaker.cpp:338-341)
    uint jitterX, jitterY, jitterPrevX, jitterPrevY;
    uint updateCounter; float dropoff;  // BakerConstants.UpdateCounter, ReservoirHistoryDropoff (0.005)
    uint totalLightCount, historicTotalLightCount, samplingProxyCount;
    uint lastFrameFeedbackAvailable, lastFrameLocalSamplesAvailable;
    float* fbW; uint* fbC;              // u_feedbackTotalWeight / u_feedbackCandidates (W x H)
    float* scW; uint* scC;              // ...Scratch (W x H)
    float* blW; uint* blC;              // ...Blended (BW x BH)
    uint* local;                        // u_localSamplingBuffer (tilesX x tilesY x 128)
    const uint* proxies;                // u_lightSamplingProxies (the global sampler, "only for filling in the gaps")
    uint* perLightCounters;             // u_perLightProxyCounters: totalLightCount + 1 words, the last one counts the pixels without a valid candidate
    const float* depth; float* historyDepth; float depthDisocclusionThreshold;      // t_depthBuffer (reference mode: what the last traced frame exported; realtime: the build pass's depth of this frame), u_historyDepth (the frame before), 1.5 (LightsBaker.h:255)
    const uint2* motion;                // t_motionVectors as the build pass stores them (RGBA16F, .xy = screen-space motion in pixels); nullptr = zero (reference mode, Sample.cpp:2494)
};
static const uint NEEAT_EARLY_FEEDBACK_TILE_SIZE = 2, NEEAT_WINDOW_SIZE = 8, NEEAT_TOP_UP_SAMPLES = RTXPT_LIGHTING_LOCAL_PROXY_COUNT - NEEAT_WINDOW_SIZE * NEEAT_WINDOW_SIZE;

// LightFeedbackReservoir on a (weight, candidate) slot pair (LightingTypes.hlsli:184-320)
static inline void lfr_clear(float& w, uint& c) { w = 0.0f; c = RTXPT_INVALID_LIGHT_INDEX; }                       // Clear: SetTotalWeight(0); SetCandidate(invalid, false)
static inline void lfr_clone_from(float& w, uint& c, float otherW, uint otherC, float scale) {                        // :199-209
    if (otherW > 0) { w = fminf_(LFR_MAX_WEIGHT, otherW * scale); c = otherC; } else lfr_clear(w, c);
}
static inline void lfr_merge(float& w, uint& c, float randomValue, float otherW, uint otherC, float otherScale) {     // :299-312
    float otherTotalWeight = fminf_(LFR_MAX_WEIGHT, otherW * otherScale);
    if (otherTotalWeight > 0) {
        uint lightIndex = otherC;
        if (lightIndex != RTXPT_INVALID_LIGHT_INDEX) {
            bool ssc = (lightIndex & LFR_SCREEN_SPACE_COHERENT_FLAG) != 0;
            lightIndex &= ~LFR_SCREEN_SPACE_COHERENT_FLAG;
            LightFeedbackReservoir_Add(w, c, randomValue, lightIndex, otherTotalWeight, ssc);
        }
    }
}
static inline uint neeat_remap_past_to_current(const NeeAtFrame& F, uint historicLightIndex) {                        // LightsBaker.hlsl:1062-1090 with an identity table
    uint lightIndex = RTXPT_INVALID_LIGHT_INDEX;
    if (historicLightIndex != RTXPT_INVALID_LIGHT_INDEX) {
        lightIndex = (historicLightIndex < F.historicTotalLightCount) ? historicLightIndex : RTXPT_INVALID_LIGHT_INDEX;
        if (lightIndex != RTXPT_INVALID_LIGHT_INDEX && lightIndex >= F.totalLightCount) lightIndex = RTXPT_INVALID_LIGHT_INDEX;
    }
    return lightIndex;
}
static inline uint neeat_sample_light_global(const NeeAtFrame& F, MicroRng& rng) {                                    // :1314-1321
    float rnd = rng.NextFloat();
    uint total = F.samplingProxyCount;
    uint idx = (uint)(rnd * (float)total);
    if (idx > total - 1u) idx = total - 1u;
    return F.proxies[idx];
}
static inline int neeat_mirror(int c, int maxRes) {                                                                   // MirrorCoord, :1323-1328 (one axis)
    int r = c >= 0 ? c : -c;
    r = r < maxRes ? r : 2 * maxRes - 2 - r;
    return r < 0 ? 0 : (r > maxRes - 1 ? maxRes - 1 : r);
}
// Reproject (LightsBaker.hlsl:1348-1375) with ConvertMotionVectorToPixelSpace (:765-772, PrevOverCurrentViewportSize = 1): where this pixel was last frame — pixel + motion
// vector, rounded — and whether what was there then is what is here now (the depth of the history frame at the old position against the current depth, 1.5 x apart = disoccluded;
// off-screen = disoccluded). A disoccluded pixel reads its history at its own position with weight 0.
//   * REFERENCE mode: the motion vectors are zero (Sample.cpp:2494; F.motion == nullptr), history is read at the same pixel, and the depths are what the path tracer itself
//     exported in the last two frames — the clip depth of each path's LAST vertex (PathTracer.hlsli:487, 684), so the test fires wherever two consecutive paths ended at depths
//     more than 1.5 x apart; without a world-to-clip matrix both depths are 0, 0 / 0 compares false, and every pixel is valid;
//   * REALTIME mode (pt_realtime_frame): F.motion / F.depth are the build pass's ScreenMotionVectors (RGBA16F: .xy in pixels) and Depth of THIS frame (Sample.cpp:2491-2494),
//     F.historyDepth the depth the Clear pass of the previous frame kept.
// Returns false if "disoccluded".
static inline bool neeat_reproject(const NeeAtFrame& F, uint x, uint y, int& hx, int& hy) {
    float mx = 0.f, my = 0.f;
    if (F.motion) { const float2 m = Fp16ToFp32(F.motion[y * F.W + x].x); mx = m.x; my = m.y; }
    const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
    const float px = (cx + mx) * 1.0f, py = (cy + my) * 1.0f;
    mx = px - cx; my = py - cy;
    hx = (int)((float)x + mx + 0.5f); hy = (int)((float)y + my + 0.5f);
    bool disocclusion = false;
    if (!(hx >= 0 && hy >= 0 && hx < (int)F.W && hy < (int)F.H)) disocclusion = true;
    else {
        const float historicDepth = F.historyDepth[(uint)hy * F.W + (uint)hx], currentDepth = F.depth[y * F.W + x];
        disocclusion = fmaxf_(historicDepth / currentDepth, currentDepth / historicDepth) > F.depthDisocclusionThreshold;
    }
    if (disocclusion) { hx = (int)x; hy = (int)y; }
    return !disocclusion;
}
static inline uint neeat_lsb_address(const NeeAtFrame& F, uint tileX, uint tileY, uint index) { return LLSB_ComputeBaseAddress(tileX, tileY, F.tilesX) + index; }

// ProcessFeedbackHistoryPreFilter (:1129-1180) for one pixel; srcW / srcC: the feedback as it was before the pass
static inline void neeat_prefilter_pixel(const NeeAtFrame& F, const float* srcW, const uint* srcC, int x, int y) {
    const float kCenterMultiplier = 48, kLikenessMultiplier = 128;
    auto load = [&](int px, int py, uint& idx, float& wgt) {                                                         // LocalReservoir::LoadWithBoundsCheck
        px = px < 0 ? 0 : (px > (int)F.W - 1 ? (int)F.W - 1 : px); py = py < 0 ? 0 : (py > (int)F.H - 1 ? (int)F.H - 1 : py);
        idx = srcC[(uint)py * F.W + (uint)px]; wgt = srcW[(uint)py * F.W + (uint)px];
        if (idx == RTXPT_INVALID_LIGHT_INDEX) wgt = 0;
    };
    auto isSSC = [](uint idx) { return idx != RTXPT_INVALID_LIGHT_INDEX && (idx & LFR_SCREEN_SPACE_COHERENT_FLAG) != 0; };
    uint cIdx; float cW; load(x, y, cIdx, cW);
    const bool centerIsSSC = isSSC(cIdx), centerIsNotEmpty = cIdx != RTXPT_INVALID_LIGHT_INDEX;
    uint kIdx[9]; float kW[9], cdf[9]; uint n = 0; float totalWeightSum = 0;
    for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        load(x + dx, y + dy, kIdx[n], kW[n]);
        float weightMul = (dx == 0 && dy == 0) ? kCenterMultiplier : 1.0f;
        weightMul *= (centerIsSSC == isSSC(kIdx[n]) && centerIsNotEmpty) ? kLikenessMultiplier : 1.0f;
        totalWeightSum += kW[n] * weightMul;
        cdf[n] = totalWeightSum; n++;
    }
    MicroRng rng = MicroRng::make((uint)x, (uint)y, F.updateCounter, 7);
    float rnd = rng.NextFloat();
    int pick = 8;
    for (int i = 0; i < 8; i++) if (rnd < (cdf[i] / totalWeightSum)) { pick = i; break; }
    F.fbC[(uint)y * F.W + (uint)x] = kIdx[pick]; F.fbW[(uint)y * F.W + (uint)x] = fminf_(LFR_MAX_WEIGHT, kW[pick]);       // LocalReservoir::Store
}
// ProcessFeedbackHistoryP0 (:1185-1311) for one pixel: returns the slot of u_perLightProxyCounters this pixel counts in (the caller adds 1 atomically)
static inline uint neeat_p0_pixel(const NeeAtFrame& F, uint x, uint y) {
    uint lightIndexAll = RTXPT_INVALID_LIGHT_INDEX;
    float& w = F.fbW[y * F.W + x]; uint& c = F.fbC[y * F.W + x];
    if (w != 0) {                                                     // !IsEmpty()
        uint candidateIndex = c; bool ssc = false;                    // GetCandidate
        if (candidateIndex != RTXPT_INVALID_LIGHT_INDEX) { ssc = (candidateIndex & LFR_SCREEN_SPACE_COHERENT_FLAG) != 0; candidateIndex &= ~LFR_SCREEN_SPACE_COHERENT_FLAG; }
        candidateIndex = neeat_remap_past_to_current(F, candidateIndex);
        lightIndexAll = candidateIndex;
        if (!ssc) candidateIndex = RTXPT_INVALID_LIGHT_INDEX;         // world-space coherent candidates are stripped from the reservoir
        c = candidateIndex | (ssc ? LFR_SCREEN_SPACE_COHERENT_FLAG : 0u);
        if (candidateIndex == RTXPT_INVALID_LIGHT_INDEX) lfr_clear(w, c);
    }
    return lightIndexAll < F.totalLightCount ? lightIndexAll : F.totalLightCount;
}
// ProcessFeedbackHistoryP1a (:1379-1452) for one low-resolution pixel
static inline void neeat_p1a_pixel(const NeeAtFrame& F, uint lx, uint ly) {
    MicroRng rng = MicroRng::make(lx, ly, F.updateCounter, 3);
    float& w = F.blW[ly * F.BW + lx]; uint& c = F.blC[ly * F.BW + lx];
    lfr_clear(w, c);
    if (F.lastFrameFeedbackAvailable) {
        const int expandMargin = 1, T = (int)NEEAT_EARLY_FEEDBACK_TILE_SIZE;
        for (int x = -expandMargin; x < T + expandMargin; x++) for (int y = -expandMargin; y < T + expandMargin; y++) {
            int px = (int)lx * T + x, py = (int)ly * T + y;
            px = px < 0 ? 0 : (px > (int)F.W - 1 ? (int)F.W - 1 : px); py = py < 0 ? 0 : (py > (int)F.H - 1 ? (int)F.H - 1 : py);
            float baseWeight = 1.0f;
            if (x < 0 || y < 0 || x >= T || y >= T) baseWeight = F.dropoff;
            int hx, hy;
            if (!neeat_reproject(F, (uint)px, (uint)py, hx, hy)) continue;
            const float sw = F.fbW[(uint)hy * F.W + (uint)hx]; const uint sc = F.fbC[(uint)hy * F.W + (uint)hx];
            if (sw != 0) { float rnd = rng.NextFloat(); lfr_merge(w, c, rnd, sw, sc, baseWeight); }
        }
    }
    if (c == RTXPT_INVALID_LIGHT_INDEX) c = neeat_sample_light_global(F, rng);      // "always has valid light indices even when it's empty"
}
// ProcessFeedbackHistoryP1b (:1455-1528) for one pixel
static inline void neeat_p1b_pixel(const NeeAtFrame& F, uint x, uint y) {
    MicroRng rng = MicroRng::make(x, y, F.updateCounter, 4);
    float& w = F.scW[y * F.W + x]; uint& c = F.scC[y * F.W + x];
    int hx, hy;
    const bool reprojectionValid = neeat_reproject(F, x, y, hx, hy);
    if (F.lastFrameFeedbackAvailable) lfr_clone_from(w, c, F.fbW[(uint)hy * F.W + (uint)hx], F.fbC[(uint)hy * F.W + (uint)hx], reprojectionValid ? 1.0f : 0.0f);
    else { lfr_clear(w, c); c = neeat_sample_light_global(F, rng); return; }
    const uint lx = x / NEEAT_EARLY_FEEDBACK_TILE_SIZE, ly = y / NEEAT_EARLY_FEEDBACK_TILE_SIZE;
    const float bw = F.blW[ly * F.BW + lx]; const uint bc = F.blC[ly * F.BW + lx];
    if (bw != 0) { float rnd = rng.NextFloat(); lfr_merge(w, c, rnd, bw, bc, F.dropoff); }
    if (c == RTXPT_INVALID_LIGHT_INDEX) {
        uint res = RTXPT_INVALID_LIGHT_INDEX;
        if (reprojectionValid && F.lastFrameLocalSamplesAvailable) {                                     // SampleLightLocalHistoric (:1332-1345); "no point ... if reprojection isn't valid"
            const uint tx = ((uint)hx + F.jitterPrevX) / RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE, ty = ((uint)hy + F.jitterPrevY) / RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE;
            const uint idx = rng.Next() % RTXPT_LIGHTING_LOCAL_PROXY_COUNT;
            res = neeat_remap_past_to_current(F, UnpackMiniListLight(F.local[neeat_lsb_address(F, tx, ty, idx)]));
        }
        if (res == RTXPT_INVALID_LIGHT_INDEX) res = neeat_sample_light_global(F, rng);
        c = res;
    }
}
// FillTile (:1531-1598): the tile's 128 unsorted entries
static inline void neeat_fill_tile(const NeeAtFrame& F, uint tx, uint ty) {
    const int margin = ((int)NEEAT_WINDOW_SIZE - (int)RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE) / 2;
    const int cellX = (int)(tx * RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE) - (int)F.jitterX, cellY = (int)(ty * RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE) - (int)F.jitterY;
    uint n = 0;
    for (int x = 0; x < (int)NEEAT_WINDOW_SIZE; x++) for (int y = 0; y < (int)NEEAT_WINDOW_SIZE; y++) {
        const int sx = neeat_mirror(cellX - margin + x, (int)F.W), sy = neeat_mirror(cellY - margin + y, (int)F.H);
        F.local[neeat_lsb_address(F, tx, ty, n++)] = PackMiniListLightAndCount(F.scC[(uint)sy * F.W + (uint)sx], 1);
    }
    MicroRng rng = MicroRng::make(tx, ty, F.updateCounter, 5);
    const float centerX = (float)cellX + (float)RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE * 0.5f, centerY = (float)cellY + (float)RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE * 0.5f;
    const float radius = (float)NEEAT_WINDOW_SIZE * 4.0f;
    for (uint i = 0; i < NEEAT_TOP_UP_SAMPLES; i++) {
        const float r0 = rng.NextFloat(), r1 = rng.NextFloat();                                          // NextFloat2: x first
        const float ox = (r0 - 0.5f) * radius, oy = (r1 - 0.5f) * radius;
        const int sx = neeat_mirror((int)((centerX + ox) + 0.5f), (int)F.W), sy = neeat_mirror((int)((centerY + oy) + 0.5f), (int)F.H);
        const uint lx = (uint)sx / NEEAT_EARLY_FEEDBACK_TILE_SIZE, ly = (uint)sy / NEEAT_EARLY_FEEDBACK_TILE_SIZE;
        F.local[neeat_lsb_address(F, tx, ty, n++)] = PackMiniListLightAndCount(F.blC[ly * F.BW + lx], 1);
    }
}
// ProcessFeedbackHistoryP3 (:1774-1854) for one tile, serially: sort the light indices, then every entry carries its light's number of entries. (The reference sorts with a
// 64-thread bitonic network and counts runs with a pointer-jumping scan; any correct sort gives this output.)
static inline void neeat_sort_tile(uint* tile) {
    const uint N = RTXPT_LIGHTING_LOCAL_PROXY_COUNT;
    uint key[RTXPT_LIGHTING_LOCAL_PROXY_COUNT];
    for (uint i = 0; i < N; i++) key[i] = UnpackMiniListLight(tile[i]);
    for (uint i = 1; i < N; i++) { uint k = key[i]; int j = (int)i - 1; while (j >= 0 && key[j] > k) { key[j + 1] = key[j]; j--; } key[j + 1] = k; }
    for (uint i = 0; i < N;) { uint j = i; while (j < N && key[j] == key[i]) j++; for (uint k = i; k < j; k++) tile[k] = PackMiniListLightAndCount(key[i], j - i); i = j; }
}
// ClearFeedbackHistory (:774-830) for one pixel
static inline void neeat_clear_pixel(const NeeAtFrame& F, uint x, uint y) {
    float& w = F.fbW[y * F.W + x]; uint& c = F.fbC[y * F.W + x];
    F.historyDepth[y * F.W + x] = F.depth[y * F.W + x];
    if (F.lastFrameFeedbackAvailable) {
        const float dropOffFactor = F.dropoff;
        lfr_clone_from(w, c, F.scW[y * F.W + x], F.scC[y * F.W + x], dropOffFactor);
        const int offs[4][2] = {{-1, 0}, {+1, 0}, {0, -1}, {0, +1}};
        MicroRng rng = MicroRng::make(x, y, F.updateCounter, 6);
        for (int i = 0; i < 4; i++) {
            int sx = (int)x + offs[i][0], sy = (int)y + offs[i][1];
            sx = sx < 0 ? 0 : (sx > (int)F.W - 1 ? (int)F.W - 1 : sx); sy = sy < 0 ? 0 : (sy > (int)F.H - 1 ? (int)F.H - 1 : sy);
            const float sw = F.scW[(uint)sy * F.W + (uint)sx]; const uint sc = F.scC[(uint)sy * F.W + (uint)sx];
            if (sw != 0) { float rnd = rng.NextFloat(); lfr_merge(w, c, rnd, sw, sc, dropOffFactor * dropOffFactor); }
        }
        if (w < 1e-12f) lfr_clear(w, c);
    } else lfr_clear(w, c);
}
// ComputeProxyCounts' weight (:898-915): the baked weight pulled towards what last frame's paths asked for
static inline float neeat_feedback_light_weight(float lightWeight, uint usageCount, float weightSum, uint totalMaxFeedbackCount, uint invalidCount, float globalFeedbackUseWeight) {
    uint validFeedbackCount = totalMaxFeedbackCount - invalidCount;
    float denom = (float)validFeedbackCount; if (!(denom > 1.0f)) denom = 1.0f;                     // (float)max(1.0, validFeedbackCount)
    float feedbackWeight = (float)usageCount * weightSum / denom;
    return lightWeight + (feedbackWeight - lightWeight) * globalFeedbackUseWeight;                   // lerp
}
// ImportanceBooster's first term (LightsBaker.hlsl:108-136; LightsBaker.h:247-249: on by default, mul 8, fade distance 5) for EVERY NEEType: a light inside the camera
// frustum weighs 1 + mul times as much, one within the fade distance of it proportionally less; environment lights have no position and get half the boost. The five
// planes (left, right, top, bottom, near; normalised, inside = dot(p, n) - w > 0) come from the host's view-projection matrix (light_frustum_planes_from_viewproj).
struct LightFrustumBoost { float planes[5][4]; float mul, fadeDistance; };      // mul 0: off
static inline float light_distance_from_frustum(const LightFrustumBoost& B, float3 position) {
    float distMin = 0;
    for (int i = 0; i < 5; i++) {
        float dist = dot(position, make_float3(B.planes[i][0], B.planes[i][1], B.planes[i][2])) - B.planes[i][3];
        distMin = fminf_(distMin, dist);
    }
    return fmaxf_(0.f, -distMin);
}
static inline float light_importance_frustum_boost(const LightFrustumBoost& B, const PolymorphicLightInfoFull& light, float unboostedWeight) {
    float boostedWeight = unboostedWeight;
    if (B.mul > 0) {
        float boostK = 0;
        const uint type = DecodeLightType(light.Base);
        if (type == kEnvironmentQuad || type == kEnvironment || type == kDirectional) boostK = 0.5f;
        else boostK = saturate(1 - light_distance_from_frustum(B, light.Base.Center) / fmaxf_(1e-5f, B.fadeDistance));
        boostedWeight *= 1 + B.mul * boostK;
    }
    return boostedWeight;
}
// LightsBaker::UpdateFrustumConsts (LightsBaker.cpp:884-925): the clip planes of a row-vector view-projection matrix M (clip = p * M, Donut's convention; vp(row, col) =
// M[row][col], m = 16 floats row-major), each scaled by 1 / sqrt(|n|^2)
static inline void light_frustum_planes_from_viewproj(const float* m, float planes[5][4]) {
    auto vp = [&](int row, int col) { return m[4 * row + col]; };
    const int sign[5] = {+1, -1, -1, +1, -1}, col[5] = {0, 0, 1, 1, 2};
    for (int i = 0; i < 5; i++) {
        float p[4];
        if (sign[i] > 0) { p[0] = vp(0, 3) + vp(0, col[i]); p[1] = vp(1, 3) + vp(1, col[i]); p[2] = vp(2, 3) + vp(2, col[i]); p[3] = -(vp(3, 3) + vp(3, col[i])); }
        else { p[0] = vp(0, 3) - vp(0, col[i]); p[1] = vp(1, 3) - vp(1, col[i]); p[2] = vp(2, 3) - vp(2, col[i]); p[3] = -(vp(3, 3) - vp(3, col[i])); }
        float lengthSq = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
        float scale = lengthSq > 0.f ? (1.0f / sqrtf_(lengthSq)) : 0.f;
        for (int k = 0; k < 4; k++) planes[i][k] = p[k] * scale;
    }
}
// ImportanceBooster's second term (:137-147) with ImportanceBoostIntensityDelta = 64 (LightsBaker.h:245-246): a light that got brighter than 1.1 x its last weight is boosted
static inline float neeat_intensity_delta_boost(float boostedWeight, float historicWeight, float intensityDeltaMul) {
    float delta = boostedWeight - historicWeight * 1.1f;
    if (delta > 0) boostedWeight += intensityDeltaMul * delta;
    return boostedWeight;
}
// LightsBaker::UpdateLocalJitter (LightsBaker.cpp:943-962): the R2 sequence, restarted every 1024 updates
static inline void neeat_advance_jitter(uint updateCounter, float jitterF[2], uint jitter[2]) {
    if ((updateCounter % 1024u) == 0u) { jitterF[0] = 0.f; jitterF[1] = 0.f; }
    const float g = 1.32471795724474602596f, a1 = 1.0f / g, a2 = 1.0f / (g * g);
    jitterF[0] = jitterF[0] + a1; jitterF[0] = jitterF[0] - (float)(int)jitterF[0];                 // fmodf(v, 1) for v >= 0
    jitterF[1] = jitterF[1] + a2; jitterF[1] = jitterF[1] - (float)(int)jitterF[1];
    for (int k = 0; k < 2; k++) { uint v = (uint)(jitterF[k] * (float)RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE); jitter[k] = v > RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE - 1u ? RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE - 1u : v; }
}

} // namespace ptref
