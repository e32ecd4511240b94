// ORACLE (test infrastructure only) — deterministic elementary functions.
//
// The reference (HLSL on a GPU) evaluates sin/cos/exp2/log2/pow/atan2 with implementation-defined, few-ulp
// hardware approximations, so any <=2-3 ulp implementation is an equally valid restatement.
// These versions use only +,-,*,/ and explicit fmaf() so that they produce bit-identical results on the host CPU
// and on gfx950; the HIP product path carries its own copy (rtxpt_amd/csrc/pt_dmath.h) written to the same contract.
// Polynomial coefficients are the classic Cephes single-precision minimax sets.
#pragma once
#include "vec.h"

namespace ptref {

static const float K_PI     = 3.14159265358979323846f;
static const float K_2PI    = 6.28318530717958647692f;
static const float K_PI_2   = 1.57079632679489661923f;
static const float K_PI_4   = 0.785398163397448309616f;
static const float K_1_PI   = 0.318309886183790671538f;
static const float K_1_2PI  = 0.159154943091895335769f;
static const float K_2_PI   = 0.636619772367581343076f;
static const float FLT_MAX_ = 3.402823466e+38f;
static const float FLT_MIN_ = 1.175494351e-38f;

// floor/round helpers that do not depend on libm rounding-mode behaviour
static inline float dm_floor(float x) { return floorf(x); }     // exact in IEEE, identical everywhere

// ---- sin / cos : quadrant reduction (Cody-Waite, 3 constants) + Cephes polynomials. Valid for |x| < ~8000.
static inline void dm_sincos(float x, float& s, float& c) {
    float ax = fabsf(x);
    float q = dm_floor(fmaf(ax, 0.636619772367581343f, 0.5f));   // nearest quadrant index
    int   qi = (int)q;
    // r = ax - q*pi/2 with pi/2 split in three parts
    float r = fmaf(q, -1.5703125f, ax);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    float z = r * r;
    // sin(r), r in [-pi/4, pi/4]
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float sr = fmaf(ps * z, r, r);
    // cos(r)
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float cr = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
    float ss, cc;
    switch (qi & 3) {
    case 0: ss = sr;  cc = cr;  break;
    case 1: ss = cr;  cc = -sr; break;
    case 2: ss = -sr; cc = -cr; break;
    default: ss = -cr; cc = sr; break;
    }
    s = (x < 0.f) ? -ss : ss;
    c = cc;
}
static inline float dm_sin(float x) { float s, c; dm_sincos(x, s, c); return s; }
static inline float dm_cos(float x) { float s, c; dm_sincos(x, s, c); return c; }

// ---- exp2: x = n + f, f in [-0.5, 0.5]; 2^f by degree-6 polynomial; scale by constructing 2^n.
static inline float dm_exp2(float x) {
    if (!(x < 128.0f)) return (x != x) ? x : asfloat(0x7f800000u);
    if (x < -150.0f) return 0.0f;
    float n = dm_floor(x + 0.5f);
    float f = x - n;
    float p = 1.535336188319500e-4f;
    p = fmaf(p, f, 1.339887440266574e-3f);
    p = fmaf(p, f, 9.618437357674640e-3f);
    p = fmaf(p, f, 5.550332471162809e-2f);
    p = fmaf(p, f, 2.402264791363012e-1f);
    p = fmaf(p, f, 6.931472028550421e-1f);
    p = fmaf(p, f, 1.0f);
    int ni = (int)n;
    // split the scale in two so that subnormal results are produced by a real multiply
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = asfloat((uint)(n1 + 127) << 23);
    float s2 = asfloat((uint)(n2 + 127) << 23);
    return (p * s1) * s2;
}
// ---- log2: x = m * 2^e, m in [sqrt(1/2), sqrt(2)); Cephes logf polynomial on (m-1).
static inline float dm_log2(float x) {
    if (!(x > 0.0f)) return (x == 0.0f) ? -asfloat(0x7f800000u) : asfloat(0x7fc00000u);
    if (x == asfloat(0x7f800000u)) return x;
    uint ux = asuint(x);
    int e = 0;
    if (ux < 0x00800000u) { x = x * 8388608.0f; ux = asuint(x); e = -23; }    // subnormal
    e += (int)(ux >> 23) - 127;
    float m = asfloat((ux & 0x007fffffu) | 0x3f800000u);                       // [1,2)
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float t = m - 1.0f;
    float z = t * t;
    float p = 7.0376836292e-2f;
    p = fmaf(p, t, -1.1514610310e-1f);
    p = fmaf(p, t, 1.1676998740e-1f);
    p = fmaf(p, t, -1.2420140846e-1f);
    p = fmaf(p, t, 1.4249322787e-1f);
    p = fmaf(p, t, -1.6668057665e-1f);
    p = fmaf(p, t, 2.0000714765e-1f);
    p = fmaf(p, t, -2.4999993993e-1f);
    p = fmaf(p, t, 3.3333331174e-1f);
    float ln = fmaf(p * t, z, fmaf(z, -0.5f, t));            // ln(m)
    return fmaf(ln, 1.44269504088896341f, (float)e);
}
static inline float dm_exp(float x) { return dm_exp2(x * 1.44269504088896341f); }
static inline float dm_log(float x) { return dm_log2(x) * 0.693147180559945309f; }
// HLSL pow(x,y) = exp2(y*log2(x)) for x > 0; pow(0,y>0) = 0
static inline float dm_pow(float x, float y) {
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : asfloat(0x7f800000u));
    return dm_exp2(y * dm_log2(x));
}
// x^5 used by the Schlick Fresnel term (Fresnel.hlsli:30-37: pow(max(1-cosTheta,0),5)); exact products
static inline float dm_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// ---- atan / atan2 (Cephes atanf)
static inline float dm_atan_pos(float x) {   // x >= 0
    float y0;
    if (x > 2.414213562373095f) { y0 = K_PI_2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = K_PI_4; x = (x - 1.0f) / (x + 1.0f); }
    else y0 = 0.0f;
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, x, x);
}
static inline float dm_atan2(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax == 0.0f) a = K_PI_2;
    else a = dm_atan_pos(ay / ax);
    if (x < 0.0f) a = K_PI - a;
    return (y < 0.0f) ? -a : a;
}
// acos through atan2: stays inside the deterministic function set (EnvMapBaker.hlsl, world_to_latlong_map)
static inline float dm_acos(float x) { return dm_atan2(sqrtf_(fmaxf_(0.0f, 1.0f - x * x)), x); }

// Utils.hlsli:486-499 — bit-trick approximations that are PART of the reference maths (ray cone / firefly K)
static inline float FastSqrt(float x) { return asfloat(0x1fbd1df5 + (asint(x) >> 1)); }
static inline float FastACos(float inX) {
    const float PI = 3.141593f, HALF_PI = 1.570796f;
    float x = fabsf(inX);
    float res = -0.156583f * x + HALF_PI;
    res *= FastSqrt(1.0f - x);
    return (inX >= 0.f) ? res : PI - res;
}

} // namespace ptref
