// ORACLE (test infrastructure only) — CPU restatement helpers: HLSL-like vector types and scalar helpers.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Arithmetic contract (shared with the HIP product path so that parity can be bit-exact):
//   * every operation is a single IEEE-754 binary32 operation evaluated in the order written;
//     compile with -ffp-contract=off; fused multiply-adds appear only as explicit fmaf().
//   * dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z ; normalize(v) = v * (1/sqrt(dot(v,v))).
//   * fp16 conversions are round-to-nearest-even, software (HLSL f32tof16/f16tof32 semantics,
//     reference: Rtxpt/Shaders/PathTracer/Utils/Packing.hlsli:206-238).
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

namespace ptref {

typedef uint32_t uint;

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { uint x, y; };
struct uint3 { uint x, y, z; };
struct uint4 { uint x, y, z, w; };

static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float3 make_float3(float s) { float3 r = {s, s, s}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline float4 make_float4(float3 v, float w) { float4 r = {v.x, v.y, v.z, w}; return r; }
static inline float3 xyz(float4 v) { return make_float3(v.x, v.y, v.z); }

static inline uint4 make_uint4(uint x, uint y, uint z, uint w) { uint4 r = {x, y, z, w}; return r; }
static inline uint2 make_uint2(uint x, uint y) { uint2 r = {x, y}; return r; }
static inline uint asuint(float f) { uint u; memcpy(&u, &f, 4); return u; }
static inline int asint(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float asfloat(uint u) { float f; memcpy(&f, &u, 4); return f; }
static inline float asfloat(int u) { float f; memcpy(&f, &u, 4); return f; }

// ---- scalar helpers
static inline float fminf_(float a, float b) { return (a < b) ? a : b; }   // NaN-unaware on purpose (a<b ? a : b)
static inline float fmaxf_(float a, float b) { return (a > b) ? a : b; }
static inline float clampf(float v, float lo, float hi) { return fminf_(fmaxf_(v, lo), hi); }
static inline float saturate(float v) { return clampf(v, 0.f, 1.f); }
static inline float lerpf(float a, float b, float t) { return a + (b - a) * t; }   // HLSL lerp: x + s*(y-x)
static inline float sq(float v) { return v * v; }
static inline float signf_(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
static inline float sqrtf_(float v) { return sqrtf(v); }                           // IEEE correctly rounded
static inline float rcpf_(float v) { return 1.0f / v; }

// ---- float2
static inline float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
static inline float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
static inline float2 operator*(float2 a, float b) { return make_float2(a.x * b, a.y * b); }
static inline float2 operator*(float a, float2 b) { return make_float2(a * b.x, a * b.y); }
static inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static inline float length(float2 a) { return sqrtf_(dot(a, a)); }

// ---- float3
static inline float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float3 operator-(float3 a) { return make_float3(-a.x, -a.y, -a.z); }
static inline float3 operator*(float3 a, float3 b) { return make_float3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float3 operator*(float3 a, float b) { return make_float3(a.x * b, a.y * b, a.z * b); }
static inline float3 operator*(float a, float3 b) { return make_float3(a * b.x, a * b.y, a * b.z); }
static inline float3 operator/(float3 a, float b) { return make_float3(a.x / b, a.y / b, a.z / b); }
static inline float3 operator/(float3 a, float3 b) { return make_float3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
static inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
static inline float3& operator*=(float3& a, float b) { a = a * b; return a; }
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length(float3 a) { return sqrtf_(dot(a, a)); }
static inline float3 normalize(float3 a) { float il = 1.0f / sqrtf_(dot(a, a)); return a * il; }
static inline float3 lerp3(float3 a, float3 b, float t) { return a + (b - a) * t; }
static inline float3 abs3(float3 a) { return make_float3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline float3 max3v(float3 a, float3 b) { return make_float3(fmaxf_(a.x, b.x), fmaxf_(a.y, b.y), fmaxf_(a.z, b.z)); }
static inline float3 min3v(float3 a, float3 b) { return make_float3(fminf_(a.x, b.x), fminf_(a.y, b.y), fminf_(a.z, b.z)); }
static inline float3 clamp3(float3 a, float lo, float hi) { return make_float3(clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)); }
static inline float3 saturate3(float3 a) { return clamp3(a, 0.f, 1.f); }
static inline float max3(float3 a) { return fmaxf_(a.x, fmaxf_(a.y, a.z)); }    // HLSL max3 helper used by EvalSampleWeight
static inline bool any_gt0(float3 a) { return a.x > 0.f || a.y > 0.f || a.z > 0.f; }
static inline float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline float4 operator*(float4 a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
static inline float4 lerp4(float4 a, float4 b, float t) {
    return make_float4(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t));
}

// SafeNormalize as used by Donut-side helpers (BridgeDonut:196,216,240): zero vector stays zero.
static inline float3 SafeNormalize(float3 v) {
    float l2 = dot(v, v);
    if (l2 > 0.f) return v * (1.0f / sqrtf_(l2));
    return make_float3(0.f);
}

// Utils.hlsli:28-48
static inline float Luminance(float3 rgb) { return dot(rgb, make_float3(0.2126f, 0.7152f, 0.0722f)); }
static inline float Average(float3 rgb) { return (rgb.x + rgb.y + rgb.z) / 3.0f; }

// row-major 3x4 affine transform (Donut InstanceData::transform): p' = M * float4(p,1)
struct float3x4 { float m[12]; };
static inline float3 xform_point(const float3x4& M, float3 p) {
    return make_float3(((M.m[0] * p.x + M.m[1] * p.y) + M.m[2] * p.z) + M.m[3],
                       ((M.m[4] * p.x + M.m[5] * p.y) + M.m[6] * p.z) + M.m[7],
                       ((M.m[8] * p.x + M.m[9] * p.y) + M.m[10] * p.z) + M.m[11]);
}
// mul(M, float4(v, 0.0)).xyz as the reference writes it for normals and tangents (BridgeDonut:230, 246, 251): the fourth term is M[i][3] * 0, which
// turns a -0 component into +0 (and is the only difference to xform_vector)
static inline float3 xform_direction4(const float3x4& M, float3 v) {
    return make_float3(((M.m[0] * v.x + M.m[1] * v.y) + M.m[2] * v.z) + M.m[3] * 0.0f,
                       ((M.m[4] * v.x + M.m[5] * v.y) + M.m[6] * v.z) + M.m[7] * 0.0f,
                       ((M.m[8] * v.x + M.m[9] * v.y) + M.m[10] * v.z) + M.m[11] * 0.0f);
}
static inline float3 xform_vector(const float3x4& M, float3 v) {
    return make_float3((M.m[0] * v.x + M.m[1] * v.y) + M.m[2] * v.z,
                       (M.m[4] * v.x + M.m[5] * v.y) + M.m[6] * v.z,
                       (M.m[8] * v.x + M.m[9] * v.y) + M.m[10] * v.z);
}
// mul(v, (float3x3)M)  (row-vector times matrix; EnvMap.hlsli:71-81)
static inline float3 mul_vec_mat3(float3 v, const float3x4& M) {
    return make_float3((v.x * M.m[0] + v.y * M.m[4]) + v.z * M.m[8],
                       (v.x * M.m[1] + v.y * M.m[5]) + v.z * M.m[9],
                       (v.x * M.m[2] + v.y * M.m[6]) + v.z * M.m[10]);
}
static inline float det3(const float3x4& M) {
    float3 r0 = make_float3(M.m[0], M.m[1], M.m[2]);
    float3 r1 = make_float3(M.m[4], M.m[5], M.m[6]);
    float3 r2 = make_float3(M.m[8], M.m[9], M.m[10]);
    return dot(r0, cross(r1, r2));
}

// ---- fp16 (binary16) <-> fp32, round-to-nearest-even, denormals preserved
static inline uint f32tof16(float f) {
    uint x = asuint(f);
    uint sign = (x >> 16) & 0x8000u;
    uint ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {                       // inf / nan
        return sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u);
    }
    if (ax >= 0x477ff000u) {                       // >= 65520 rounds to inf
        return sign | 0x7c00u;
    }
    if (ax < 0x38800000u) {                        // result is subnormal (or zero)
        if (ax < 0x33000000u) return sign;         // < 2^-25 -> 0 (exact half 2^-25 ties to even = 0)
        uint e = ax >> 23;                         // biased exponent (>= 102)
        uint m = (ax & 0x7fffffu) | 0x800000u;     // 24-bit significand
        uint shift = 126u - e;                     // 14..24 ; value = m * 2^(e-150); half subnormal unit 2^-24
        uint r = m >> shift;
        uint rem = m & ((1u << shift) - 1u);
        uint half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return sign | r;
    }
    uint e = (ax >> 23) - 112u;                    // rebias 127 -> 15
    uint m = ax & 0x7fffffu;
    uint r = (e << 10) | (m >> 13);
    uint rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;   // carry may bump exponent: correct behaviour
    return sign | r;
}
static inline float f16tof32(uint h) {
    h &= 0xffffu;
    uint sign = (h & 0x8000u) << 16;
    uint e = (h >> 10) & 0x1fu;
    uint m = h & 0x3ffu;
    if (e == 0u) {
        if (m == 0u) return asfloat(sign);
        float v = (float)m * 5.9604644775390625e-8f;     // m * 2^-24 (exact)
        return asfloat(sign | asuint(v));
    }
    if (e == 31u) return asfloat(sign | 0x7f800000u | (m << 13));
    return asfloat(sign | ((e + 112u) << 23) | (m << 13));
}
static const float HLF_MAX = 65504.0f;
// Packing.hlsli:206-232
static inline uint Fp32ToFp16(float2 v) {
    uint rx = f32tof16(clampf(v.x, -HLF_MAX, HLF_MAX)), ry = f32tof16(clampf(v.y, -HLF_MAX, HLF_MAX));
    return (ry << 16) | (rx & 0xffffu);
}
static inline uint Fp32ToFp16NoClamp(float2 v) { return (f32tof16(v.y) << 16) | (f32tof16(v.x) & 0xffffu); }
static inline float2 Fp16ToFp32(uint r) { return make_float2(f16tof32(r & 0xffffu), f16tof32(r >> 16)); }
// ---- "lp" types of the reference's 16-bit build (Utils.hlsli:28-48: lpfloat = float16_t when RTXPT_LP_TYPES_USE_16BIT_PRECISION, the reference's default —
// SampleUI.h:182 UseFp16Types = true, Sample.cpp:1035). A value of an lp type is kept in a float that holds a binary16 number; LPOps<true>::r() is the
// conversion lpfloat(x), and the arithmetic helpers are the half-typed operators (one rounding to nearest even after every operation, as float16_t
// arithmetic under -enable-16bit-types; an exactly computed fp32 result rounded once to binary16 equals the native half operation for + - * /, since
// 24 >= 2 * 11 + 2). LPOps<false> is the fp32 build: every helper is the plain float expression it stands for.
template <bool LP16> struct LPOps {
    static inline float r(float x) { return LP16 ? f16tof32(f32tof16(x)) : x; }
    static inline float add(float a, float b) { return r(a + b); }
    static inline float sub(float a, float b) { return r(a - b); }
    static inline float mul(float a, float b) { return r(a * b); }
    static inline float div(float a, float b) { return r(a / b); }       // fp32 quotient (correctly rounded), one rounding to binary16
    static inline float3 r3(float3 v) { return make_float3(r(v.x), r(v.y), r(v.z)); }
    static inline float3 mul3(float3 a, float3 b) { return make_float3(mul(a.x, b.x), mul(a.y, b.y), mul(a.z, b.z)); }
    static inline float3 mul3(float3 a, float b) { return make_float3(mul(a.x, b), mul(a.y, b), mul(a.z, b)); }
    static inline float3 div3(float3 a, float b) { return make_float3(div(a.x, b), div(a.y, b), div(a.z, b)); }
    static inline float lerp(float a, float b, float t) { return add(a, mul(sub(b, a), t)); }                       // HLSL lerp: x + s*(y-x), every operation in half
    static inline float3 lerp3(float3 a, float3 b, float t) { return make_float3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
    static inline float average3(float3 v) { return div(add(add(v.x, v.y), v.z), 3.0f); }       // the lpfloat3 overload of Average (Utils.hlsli:62-67): (x + y + z) / 3.0 in half
};


// Packing.hlsli:17-51, 127-167
static inline uint Pack_R8_UFLOAT(float r, float d = 0.5f) { return (uint)floorf(r * 255.0f + d) & 255u; }
static inline float Unpack_R8_UFLOAT(uint r) { return (float)(r & 255u) / 255.0f; }
static inline uint Pack_R8G8B8_UFLOAT(float3 rgb) {
    return Pack_R8_UFLOAT(rgb.x) | (Pack_R8_UFLOAT(rgb.y) << 8) | (Pack_R8_UFLOAT(rgb.z) << 16);
}
static inline float3 Unpack_R8G8B8_UFLOAT(uint rgb) {
    return make_float3(Unpack_R8_UFLOAT(rgb), Unpack_R8_UFLOAT(rgb >> 8), Unpack_R8_UFLOAT(rgb >> 16));
}
static inline float Unpack_R8_SNORM(uint value) {
    int s = (int)(value << 24) >> 24;
    return clampf((float)s / 127.0f, -1.0f, 1.0f);
}
static inline uint Pack_R8_SNORM(float v) { return (uint)((int)(clampf(v, -1.0f, 1.0f) * 127.0f)) & 0xffu; }
static inline float3 Unpack_RGB8_SNORM(uint v) { return make_float3(Unpack_R8_SNORM(v), Unpack_R8_SNORM(v >> 8), Unpack_R8_SNORM(v >> 16)); }
static inline float4 Unpack_RGBA8_SNORM(uint v) {
    return make_float4(Unpack_R8_SNORM(v), Unpack_R8_SNORM(v >> 8), Unpack_R8_SNORM(v >> 16), Unpack_R8_SNORM(v >> 24));
}

} // namespace ptref
