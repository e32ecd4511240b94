// Host side of the display path: the colour transform ToneMappingPass puts in its constant buffer — UpdateWhiteBalanceTransform + UpdateColorTransform
// (Rtxpt/ToneMapper/ToneMappingPasses.cpp:392-401, 428-441) with calculateWhiteBalanceTransformRGB_Rec709 / colorTemperatureToXYZ
// (Rtxpt/ToneMapper/ColorUtils.h:128-197) and the copy into ToneMappingConstants::colorTransform (ToneMappingPasses.cpp:344-347).
// Matrix semantics are Donut's (donut/core/math/matrix.h, not vendored in the reference tree): row-major storage, nine-scalar constructor in row
// order, M * v = sum_j M[i][j] v[j], (A * B)[i][j] = sum_k A[i][k] B[k][j] accumulated from zero in k order, all in fp32. The reference fills its
// constants with the (column-major) Falcor tables as they stand, so "RGBtoXYZ" below holds the transposed matrix; that is reproduced, not corrected.
#include "../../include/mi355pt.h"
#include <cmath>
#include <cstring>

namespace {
struct V3 { float v[3]; };
struct M3 { float m[3][3]; };
M3 mul(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0.f; for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; } return r; }
V3 mul(const M3& a, const V3& b) { V3 r; for (int i = 0; i < 3; i++) { float s = 0.f; for (int j = 0; j < 3; j++) s += a.m[i][j] * b.v[j]; r.v[i] = s; } return r; }
// ColorUtils.h:60-101, values in constructor (row) order
const M3 kRGBtoXYZ_Rec709 = {{{0.4123907992659595f, 0.2126390058715104f, 0.0193308187155918f}, {0.3575843393838780f, 0.7151686787677559f, 0.1191947797946259f}, {0.1804807884018343f, 0.0721923153607337f, 0.9505321522496608f}}};
const M3 kXYZtoRGB_Rec709 = {{{3.2409699419045213f, -0.9692436362808798f, 0.0556300796969936f}, {-1.5373831775700935f, 1.8759675015077206f, -0.2039769588889765f}, {-0.4986107602930033f, 0.0415550574071756f, 1.0569715142428784f}}};
const M3 kXYZtoLMS_CAT02 = {{{0.7328f, -0.7036f, 0.0030f}, {0.4296f, 1.6975f, 0.0136f}, {-0.1624f, 0.0061f, 0.9834f}}};
const M3 kLMStoXYZ_CAT02 = {{{1.096123820835514f, 0.454369041975359f, -0.009627608738429f}, {-0.278869000218287f, 0.473533154307412f, -0.005698031216113f}, {0.182745179382773f, 0.072097803717229f, 1.015325639954543f}}};
// ColorUtils.h:128-170 (Kang et al. 2002): chromaticity in double, then xyYtoXYZ :116-119 in fp32 with Y = 1
V3 colorTemperatureToXYZ(float T) {
    if (T < 1667.f || T > 25000.f) return V3{{0.f, 0.f, 0.f}};
    double t = T, t2 = t * t, t3 = t * t * t;
    double xc = (T < 4000.f) ? -0.2661239e9 / t3 - 0.2343580e6 / t2 + 0.8776956e3 / t + 0.179910
                             : -3.0258469e9 / t3 + 2.1070379e6 / t2 + 0.2226347e3 / t + 0.240390;
    double x = xc, x2 = x * x, x3 = x * x * x;
    double yc = (T < 2222.f) ? -1.1063814 * x3 - 1.34811020 * x2 + 2.18555832 * x - 0.20219683
              : (T < 4000.f) ? -0.9549476 * x3 - 1.37418593 * x2 + 2.09137015 * x - 0.16748867
                             : +3.0817580 * x3 - 5.87338670 * x2 + 3.75112997 * x - 0.37001483;
    float fx = (float)xc, fy = (float)yc, Y = 1.f;
    return V3{{fx * Y / fy, Y, (1.f - fx - fy) * Y / fy}};
}
// ColorUtils.h:187-197
M3 whiteBalanceTransformRGB_Rec709(float T) {
    const M3 MA = mul(kXYZtoLMS_CAT02, kRGBtoXYZ_Rec709), invMA = mul(kXYZtoRGB_Rec709, kLMStoXYZ_CAT02);
    const V3 wd = mul(kXYZtoLMS_CAT02, colorTemperatureToXYZ(6500.f)), ws = mul(kXYZtoLMS_CAT02, colorTemperatureToXYZ(T));
    M3 D; memset(&D, 0, sizeof(D));
    for (int i = 0; i < 3; i++) D.m[i][i] = wd.v[i] / ws.v[i];
    return mul(mul(invMA, D), MA);
}
}

extern "C" int32_t pt_tonemap_color_transform(PtToneMapParams* params, uint32_t whiteBalance, float whitePoint, float exposureCompensation,
                                              float filmSpeed, float shutter, float fNumber) {
    if (!params) return PT_ERROR_INVALID_ARGUMENT;
    if (!params->autoExposure && (!(shutter > 0.f) || !(fNumber > 0.f))) return PT_ERROR_INVALID_ARGUMENT;
    M3 wb;
    if (whiteBalance) wb = whiteBalanceTransformRGB_Rec709(whitePoint);
    else { memset(&wb, 0, sizeof(wb)); wb.m[0][0] = wb.m[1][1] = wb.m[2][2] = 1.f; }
    float exposureScale = powf(2.f, exposureCompensation), manualExposureScale = 1.f;
    if (!params->autoExposure) manualExposureScale = ((1.f / 100.f) * filmSpeed) / (shutter * fNumber * fNumber);
    // constant-buffer row i = column i of m_ColorTransform; the shader's mul(color, M) then yields m_ColorTransform * color
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) params->colorTransform[i * 3 + j] = wb.m[j][i] * exposureScale * manualExposureScale;
    return PT_OK;
}

// ---- ToneMappingParameters (the UI block, ToneMappingPasses.h:36-53) -> the constants ToneMappingPass::Render uploads: PreRender :186-193 =
// SetParameters :373-390, UpdateExposureValue :402-427 (EV clamped to the range its inputs allow; in aperture priority — the default — the shutter is
// DERIVED from EV and fNumber, the UI's shutter value is overwritten), UpdateWhiteBalanceTransform, UpdateColorTransform; then the constant-buffer fill
// :316-348.
extern "C" int32_t pt_default_tone_mapping_parameters(PtToneMappingParameters* p) {
    if (!p) return PT_ERROR_INVALID_ARGUMENT;
    memset(p, 0, sizeof(*p));
    p->exposureMode = 0u; p->toneMapOperator = 5u; p->autoExposure = 0u; p->exposureCompensation = 0.f; p->exposureValue = 0.f; p->filmSpeed = 100.f; p->fNumber = 1.f; p->shutter = 1.f;
    p->whiteBalance = 0u; p->whitePoint = 6500.f; p->whiteMaxLuminance = 1.f; p->whiteScale = 5.1f; p->clamped = 1u; p->exposureValueMin = -16.f; p->exposureValueMax = 16.f;
    return PT_OK;
}
extern "C" int32_t pt_tonemap_from_parameters(const PtToneMappingParameters* ui, float avgLuminance, uint32_t enabled, PtToneMapParams* out) {
    if (!ui || !out || ui->exposureMode > 1u || ui->toneMapOperator > 5u) return PT_ERROR_INVALID_ARGUMENT;
    auto clampf = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
    const float kShutterMin = 0.001f, kShutterMax = 10000.f, kFNumberMin = 0.1f, kFNumberMax = 100.f;
    const float kExposureValueMin = log2f(kShutterMin * kFNumberMin * kFNumberMin), kExposureValueMax = log2f(kShutterMax * kFNumberMax * kFNumberMax);
    float ev = clampf(ui->exposureValue, kExposureValueMin, kExposureValueMax), shutter = ui->shutter, fNumber = ui->fNumber;
    if (ui->exposureMode == 0u) { shutter = powf(2.f, ev) / (fNumber * fNumber); shutter = clampf(shutter, kShutterMin, kShutterMax); }
    else { fNumber = sqrtf(powf(2.f, ev) / shutter); fNumber = clampf(fNumber, kFNumberMin, kFNumberMax); }
    memset(out, 0, sizeof(*out));
    out->whiteScale = ui->whiteScale; out->whiteMaxLuminance = ui->whiteMaxLuminance; out->clamped = ui->clamped ? 1u : 0u; out->toneMapOperator = ui->toneMapOperator;
    out->autoExposure = ui->autoExposure ? 1u : 0u; out->avgLuminance = avgLuminance;
    out->autoExposureLumValueMin = exp2f(ui->autoExposure ? ui->exposureValueMin : -16.0f); out->autoExposureLumValueMax = exp2f(ui->autoExposure ? ui->exposureValueMax : 16.0f);
    out->enabled = enabled ? 1u : 0u;
    M3 wb;
    if (ui->whiteBalance) wb = whiteBalanceTransformRGB_Rec709(ui->whitePoint);
    else { memset(&wb, 0, sizeof(wb)); wb.m[0][0] = wb.m[1][1] = wb.m[2][2] = 1.f; }
    float exposureScale = powf(2.f, ui->exposureCompensation), manualExposureScale = 1.f;
    if (!ui->autoExposure) manualExposureScale = ((1.f / 100.f) * ui->filmSpeed) / (shutter * fNumber * fNumber);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out->colorTransform[i * 3 + j] = wb.m[j][i] * exposureScale * manualExposureScale;
    return PT_OK;
}
