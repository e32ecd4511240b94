// ORACLE pin (test infrastructure only): the HLSL vocabulary that lets g++ compile functions taken verbatim from the reference's .hlsli files
// (oracle/refpin/hlsl_tu.py streams them from /root/reference into the compiler; nothing is copied into this repo).
//
// Everything whose result HLSL leaves to the implementation is mapped to the oracle's arithmetic contract (oracle/ptref/vec.h, dmath.h):
// dot/normalize/length summation order, 1/sqrt for rsqrt, the dm_* transcendental functions, pow(x, 5) as repeated products, mad() unfused,
// software fp16. What the pin therefore checks is the *restatement*: operation order, constants, branches and clamps of every pinned function.
#pragma once
#include <cstdint>
#include <cmath>
#include <cfloat>
#include <type_traits>
#include "../ptref/vec.h"
#include "../ptref/dmath.h"

namespace hl {
typedef uint32_t uint;
namespace P = ptref;

struct half_t;
template <class S> struct is_num : std::integral_constant<bool, std::is_arithmetic<S>::value || std::is_same<S, half_t>::value> {};
template <class T> struct v2 { union { struct { T x, y; }; struct { T r, g; }; }; v2() : x(), y() {} template <class S, class = typename std::enable_if<is_num<S>::value && !std::is_same<S, T>::value>::type> v2(S s) : x((T)s), y((T)s) {}
    v2(T s) : x(s), y(s) {} v2(T a, T b) : x(a), y(b) {}
    template <class U, class = typename std::enable_if<!std::is_same<U, T>::value>::type> v2(const v2<U>& o) : x((T)o.x), y((T)o.y) {}
    v2& xy_() { return *this; } const v2& xy_() const { return *this; } const v2 yx_() const { return v2(y, x); } const v2 xx_() const { return v2(x, x); }
    T& operator[](uint i) { return (&x)[i]; } T operator[](uint i) const { return (&x)[i]; } };
template <class T> struct v3 { union { struct { T x, y, z; }; struct { T r, g, b; }; }; v3() : x(), y(), z() {} template <class S, class = typename std::enable_if<is_num<S>::value && !std::is_same<S, T>::value>::type> v3(S s) : x((T)s), y((T)s), z((T)s) {}
    v3(T s) : x(s), y(s), z(s) {} v3(T a, T b, T c) : x(a), y(b), z(c) {}
    v3(v2<T> a, T c) : x(a.x), y(a.y), z(c) {} v3(T a, v2<T> b) : x(a), y(b.x), z(b.y) {}
    template <class A, class B, class C, class = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value && std::is_arithmetic<C>::value && !(std::is_same<A, T>::value && std::is_same<B, T>::value && std::is_same<C, T>::value)>::type>
    v3(A a, B b, C c) : x((T)a), y((T)b), z((T)c) {}
    template <class U, class = typename std::enable_if<!std::is_same<U, T>::value>::type> v3(const v3<U>& o) : x((T)o.x), y((T)o.y), z((T)o.z) {}       // HLSL converts between component types implicitly
    v2<T>& xy_() { return *reinterpret_cast<v2<T>*>(this); } const v2<T> xy_() const { return v2<T>(x, y); } const v2<T> yx_() const { return v2<T>(y, x); } const v2<T> xz_() const { return v2<T>(x, z); } const v2<T> yz_() const { return v2<T>(y, z); }
    v2<T>& yz_() { return *reinterpret_cast<v2<T>*>(&y); }
    const v2<T> xy() const { return v2<T>(x, y); }          // Donut's dm::float3 spelling (host code)
    v3& xyz_() { return *this; } const v3& xyz_() const { return *this; } v3& rgb_() { return *this; } const v3& rgb_() const { return *this; }
    T& operator[](uint i) { return (&x)[i]; } T operator[](uint i) const { return (&x)[i]; } };
template <class T> struct v4 { union { struct { T x, y, z, w; }; struct { T r, g, b, a; }; }; v4() : x(), y(), z(), w() {} template <class S, class = typename std::enable_if<is_num<S>::value && !std::is_same<S, T>::value>::type> v4(S s) : x((T)s), y((T)s), z((T)s), w((T)s) {}
    v4(T s) : x(s), y(s), z(s), w(s) {} v4(T a, T b, T c, T d) : x(a), y(b), z(c), w(d) {}
    v4(v3<T> a, T d) : x(a.x), y(a.y), z(a.z), w(d) {} v4(v2<T> a, v2<T> b) : x(a.x), y(a.y), z(b.x), w(b.y) {} v4(v2<T> a, T c, T d) : x(a.x), y(a.y), z(c), w(d) {}
    template <class U, class = typename std::enable_if<!std::is_same<U, T>::value>::type> v4(const v4<U>& o) : x((T)o.x), y((T)o.y), z((T)o.z), w((T)o.w) {}
    v2<T>& zw_() { return *reinterpret_cast<v2<T>*>(&z); } v3<T>& yzw_() { return *reinterpret_cast<v3<T>*>(&y); }
    v4& rgba_() { return *this; } const v4& rgba_() const { return *this; }
    v4& xyzw_() { return *this; } const v4& xyzw_() const { return *this; } const v3<T> xyw_() const { return v3<T>(x, y, w); } const v3<T> xzw_() const { return v3<T>(x, z, w); } const v3<T> yzw_() const { return v3<T>(y, z, w); } const v4 wzyx_() const { return v4(w, z, y, x); }
    v3<T>& xyz_() { return *reinterpret_cast<v3<T>*>(this); } const v3<T> xyz_() const { return v3<T>(x, y, z); } v3<T>& rgb_() { return xyz_(); } const v3<T> rgb_() const { return xyz_(); }
    v2<T>& xy_() { return *reinterpret_cast<v2<T>*>(this); } const v2<T> xy_() const { return v2<T>(x, y); } const v2<T> zw_() const { return v2<T>(z, w); }
    T& operator[](uint i) { return (&x)[i]; } T operator[](uint i) const { return (&x)[i]; } };

typedef v2<float> float2; typedef v3<float> float3; typedef v4<float> float4;
typedef v2<int> int2; typedef v3<int> int3; typedef v4<int> int4;
typedef v2<uint> uint2; typedef v3<uint> uint3; typedef v4<uint> uint4;
typedef v2<bool> bool2; typedef v3<bool> bool3; typedef v4<bool> bool4;
#if defined(RTXPT_LP_TYPES_USE_16BIT_PRECISION) && RTXPT_LP_TYPES_USE_16BIT_PRECISION
// float16_t for the reference's default build (lp types in 16 bits): a float that is rounded to binary16 after every operation. Used only to MEASURE how
// far that build is from the fp32 build both sides of the parity fence restate (tools/lp16_deviation.py); half x float promotes to float as in HLSL.
struct half_t { float v; half_t() : v(0.f) {} half_t(float f) : v(P::f16tof32(P::f32tof16(f))) {} half_t(double f) : half_t((float)f) {} half_t(int i) : half_t((float)i) {} half_t(uint i) : half_t((float)i) {}
    half_t(bool b) : v(b ? 1.f : 0.f) {} operator float() const { return v; } };
// scalar typing rules of HLSL: half op half -> half; an int or an unsuffixed floating literal (a C++ double here) adapts to the half operand; a float
// operand — a float variable or an `f`-suffixed literal — promotes the half to float (exact-match overloads: nothing is left to -fpermissive)
#define HL_HALFOP(op) static inline half_t operator op(half_t a, half_t b) { return half_t(a.v op b.v); } \
    static inline half_t operator op(half_t a, int b) { return half_t(a.v op (float)b); } static inline half_t operator op(int a, half_t b) { return half_t((float)a op b.v); } \
    static inline half_t operator op(half_t a, uint b) { return half_t(a.v op (float)b); } static inline half_t operator op(uint a, half_t b) { return half_t((float)a op b.v); } \
    static inline half_t operator op(half_t a, double b) { return half_t(a.v op half_t(b).v); } static inline half_t operator op(double a, half_t b) { return half_t(half_t(a).v op b.v); } \
    static inline float operator op(half_t a, float b) { return a.v op b; } static inline float operator op(float a, half_t b) { return a op b.v; } \
    static inline half_t& operator op##=(half_t& a, half_t b) { a = half_t(a.v op b.v); return a; } static inline half_t& operator op##=(half_t& a, float b) { a = half_t(a.v op b); return a; }
HL_HALFOP(+) HL_HALFOP(-) HL_HALFOP(*) HL_HALFOP(/)
#undef HL_HALFOP
static inline half_t operator-(half_t a) { half_t r; r.v = -a.v; return r; }
} // namespace hl
namespace std { template <> struct common_type<hl::half_t, hl::half_t> { typedef hl::half_t type; }; template <> struct common_type<hl::half_t, float> { typedef float type; }; template <> struct common_type<float, hl::half_t> { typedef float type; }; }
namespace hl {
typedef half_t float16_t; typedef v2<half_t> float16_t2; typedef v3<half_t> float16_t3; typedef v4<half_t> float16_t4; typedef v2<uint16_t> uint16_t2; typedef v3<uint16_t> uint16_t3; typedef v4<uint16_t> uint16_t4;
typedef half_t lpfloat; typedef float16_t2 lpfloat2; typedef float16_t3 lpfloat3; typedef float16_t4 lpfloat4; typedef uint16_t lpuint;
struct float3x3; typedef float3x3 float16_t3x3;
static inline half_t min(half_t a, half_t b) { return (a.v < b.v) ? a : b; } static inline half_t max(half_t a, half_t b) { return (a.v > b.v) ? a : b; }
static inline half_t abs(half_t a) { half_t r; r.v = fabsf(a.v); return r; } static inline half_t saturate(half_t a) { return half_t(P::saturate(a.v)); }
static inline float min(float a, half_t b) { return P::fminf_(a, b.v); } static inline float min(half_t a, float b) { return P::fminf_(a.v, b); }
static inline float max(float a, half_t b) { return P::fmaxf_(a, b.v); } static inline float max(half_t a, float b) { return P::fmaxf_(a.v, b); }
#else
typedef float lpfloat; typedef float2 lpfloat2; typedef float3 lpfloat3; typedef float4 lpfloat4;      // RTXPT_LP_TYPES_USE_16BIT_PRECISION 0 (the build both sides of the parity fence restate)
typedef uint lpuint;
#endif

// scalar operand of a mixed vector/scalar expression: HLSL converts it to the vector's component type
template <class T, class S> using if_arith = typename std::enable_if<is_num<S>::value, T>::type;

template <class A, class B> using ctype = typename std::common_type<A, B>::type;
// vector (component T) with a scalar S: integers and literals adopt T (HLSL literal rules), a float scalar promotes a half vector to float
template <class T, class S> using stype = typename std::conditional<std::is_same<S, float>::value || std::is_same<S, double>::value, ctype<T, float>, T>::type;
#define HL_BINOP(op) \
    template <class A, class B> v2<ctype<A, B>> operator op(v2<A> a, v2<B> b) { typedef ctype<A, B> C; return v2<C>((C)a.x op (C)b.x, (C)a.y op (C)b.y); } \
    template <class A, class B> v3<ctype<A, B>> operator op(v3<A> a, v3<B> b) { typedef ctype<A, B> C; return v3<C>((C)a.x op (C)b.x, (C)a.y op (C)b.y, (C)a.z op (C)b.z); } \
    template <class A, class B> v4<ctype<A, B>> operator op(v4<A> a, v4<B> b) { typedef ctype<A, B> C; return v4<C>((C)a.x op (C)b.x, (C)a.y op (C)b.y, (C)a.z op (C)b.z, (C)a.w op (C)b.w); } \
    template <class T, class S> if_arith<v2<stype<T, S>>, S> operator op(v2<T> a, S b) { typedef stype<T, S> C; return v2<C>(a) op v2<C>((C)b); } \
    template <class T, class S> if_arith<v3<stype<T, S>>, S> operator op(v3<T> a, S b) { typedef stype<T, S> C; return v3<C>(a) op v3<C>((C)b); } \
    template <class T, class S> if_arith<v4<stype<T, S>>, S> operator op(v4<T> a, S b) { typedef stype<T, S> C; return v4<C>(a) op v4<C>((C)b); } \
    template <class T, class S> if_arith<v2<stype<T, S>>, S> operator op(S a, v2<T> b) { typedef stype<T, S> C; return v2<C>((C)a) op v2<C>(b); } \
    template <class T, class S> if_arith<v3<stype<T, S>>, S> operator op(S a, v3<T> b) { typedef stype<T, S> C; return v3<C>((C)a) op v3<C>(b); } \
    template <class T, class S> if_arith<v4<stype<T, S>>, S> operator op(S a, v4<T> b) { typedef stype<T, S> C; return v4<C>((C)a) op v4<C>(b); } \
    template <class T, class S> v2<T>& operator op##=(v2<T>& a, S b) { a = v2<T>(a op b); return a; } \
    template <class T, class S> v3<T>& operator op##=(v3<T>& a, S b) { a = v3<T>(a op b); return a; } \
    template <class T, class S> v4<T>& operator op##=(v4<T>& a, S b) { a = v4<T>(a op b); return a; }
HL_BINOP(+) HL_BINOP(-) HL_BINOP(*) HL_BINOP(/)
#undef HL_BINOP
template <class T> v2<T> operator+(v2<T> a) { return a; } template <class T> v3<T> operator+(v3<T> a) { return a; }
template <class T> v2<T> operator-(v2<T> a) { return v2<T>(-a.x, -a.y); }
template <class T> v3<T> operator-(v3<T> a) { return v3<T>(-a.x, -a.y, -a.z); }
template <class T> v4<T> operator-(v4<T> a) { return v4<T>(-a.x, -a.y, -a.z, -a.w); }
#define HL_CMP(op) \
    template <class T> bool2 operator op(v2<T> a, v2<T> b) { return bool2(a.x op b.x, a.y op b.y); } \
    template <class T> bool3 operator op(v3<T> a, v3<T> b) { return bool3(a.x op b.x, a.y op b.y, a.z op b.z); } \
    template <class T> bool4 operator op(v4<T> a, v4<T> b) { return bool4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); } \
    template <class T, class S> if_arith<bool4, S> operator op(v4<T> a, S b) { return a op v4<T>((T)b); } \
    template <class T, class S> if_arith<bool2, S> operator op(v2<T> a, S b) { return a op v2<T>((T)b); } \
    template <class T, class S> if_arith<bool3, S> operator op(v3<T> a, S b) { return a op v3<T>((T)b); }
HL_CMP(<) HL_CMP(>) HL_CMP(<=) HL_CMP(>=) HL_CMP(==) HL_CMP(!=)
#undef HL_CMP
#define HL_INTOP(op) \
    template <class T, class S> if_arith<v2<T>, S> operator op(v2<T> a, S b) { return v2<T>(a.x op (T)b, a.y op (T)b); } \
    template <class T, class S> if_arith<v3<T>, S> operator op(v3<T> a, S b) { return v3<T>(a.x op (T)b, a.y op (T)b, a.z op (T)b); } \
    template <class T, class S> if_arith<v4<T>, S> operator op(v4<T> a, S b) { return v4<T>(a.x op (T)b, a.y op (T)b, a.z op (T)b, a.w op (T)b); } \
    template <class T> v2<T> operator op(v2<T> a, v2<T> b) { return v2<T>(a.x op b.x, a.y op b.y); } \
    template <class T> v3<T> operator op(v3<T> a, v3<T> b) { return v3<T>(a.x op b.x, a.y op b.y, a.z op b.z); } \
    template <class T> v4<T> operator op(v4<T> a, v4<T> b) { return v4<T>(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); }
HL_INTOP(&) HL_INTOP(|) HL_INTOP(>>) HL_INTOP(<<)
#undef HL_INTOP
static inline bool any(bool2 b) { return b.x || b.y; } static inline bool any(bool3 b) { return b.x || b.y || b.z; }
static inline bool any(bool4 b) { return b.x || b.y || b.z || b.w; } static inline bool all(bool4 b) { return b.x && b.y && b.z && b.w; }
static inline bool all(bool2 b) { return b.x && b.y; } static inline bool all(bool3 b) { return b.x && b.y && b.z; }
static inline bool any(float3 v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f; }
template <class T> T select(bool c, T a, T b) { return c ? a : b; }
template <class T> v2<T> select(bool2 c, v2<T> a, v2<T> b) { return v2<T>(c.x ? a.x : b.x, c.y ? a.y : b.y); }
template <class S> if_arith<float2, S> select(bool2 c, S a, S b) { return float2(c.x ? (float)a : (float)b, c.y ? (float)a : (float)b); }
template <class T> v3<T> select(bool3 c, v3<T> a, v3<T> b) { return v3<T>(c.x ? a.x : b.x, c.y ? a.y : b.y, c.z ? a.z : b.z); }

// ---- component-wise lifting of scalar functions
#define HL_LIFT1(name) \
    static inline float2 name(float2 a) { return float2(name(a.x), name(a.y)); } \
    static inline float3 name(float3 a) { return float3(name(a.x), name(a.y), name(a.z)); } \
    static inline float4 name(float4 a) { return float4(name(a.x), name(a.y), name(a.z), name(a.w)); }
#define HL_LIFT2(name) \
    static inline float2 name(float2 a, float2 b) { return float2(name(a.x, b.x), name(a.y, b.y)); } \
    static inline float3 name(float3 a, float3 b) { return float3(name(a.x, b.x), name(a.y, b.y), name(a.z, b.z)); } \
    static inline float4 name(float4 a, float4 b) { return float4(name(a.x, b.x), name(a.y, b.y), name(a.z, b.z), name(a.w, b.w)); } \
    template <class S> if_arith<float2, S> name(float2 a, S b) { return name(a, float2((float)b)); } \
    template <class S> if_arith<float3, S> name(float3 a, S b) { return name(a, float3((float)b)); } \
    template <class S> if_arith<float3, S> name(S a, float3 b) { return name(float3((float)a), b); }

// min / max: (a < b) ? a : b like the oracle's fminf_/fmaxf_; integer pairs stay integer
template <class A, class B> typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value && !(std::is_integral<A>::value && std::is_integral<B>::value), float>::type
    min(A a, B b) { return P::fminf_((float)a, (float)b); }
template <class A, class B> typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value && !(std::is_integral<A>::value && std::is_integral<B>::value), float>::type
    max(A a, B b) { return P::fmaxf_((float)a, (float)b); }
template <class A, class B> typename std::enable_if<std::is_integral<A>::value && std::is_integral<B>::value, A>::type min(A a, B b) { return (a < (A)b) ? a : (A)b; }
template <class A, class B> typename std::enable_if<std::is_integral<A>::value && std::is_integral<B>::value, A>::type max(A a, B b) { return (a > (A)b) ? a : (A)b; }
HL_LIFT2(min) HL_LIFT2(max)
static inline float abs(float v) { return fabsf(v); } static inline int abs(int v) { return v < 0 ? -v : v; } HL_LIFT1(abs)
static inline float saturate(float v) { return P::saturate(v); } HL_LIFT1(saturate)
template <class A, class B, class C> typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value && std::is_arithmetic<C>::value, float>::type
    clamp(A v, B lo, C hi) { return P::clampf((float)v, (float)lo, (float)hi); }
template <class B, class C> if_arith<float2, B> clamp(float2 v, B lo, C hi) { return float2(clamp(v.x, lo, hi), clamp(v.y, lo, hi)); }
template <class B, class C> if_arith<float3, B> clamp(float3 v, B lo, C hi) { return float3(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi)); }
template <class B, class C> if_arith<float4, B> clamp(float4 v, B lo, C hi) { return float4(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi), clamp(v.w, lo, hi)); }
static inline float3 clamp(float3 v, float3 lo, float3 hi) { return float3(clamp(v.x, lo.x, hi.x), clamp(v.y, lo.y, hi.y), clamp(v.z, lo.z, hi.z)); }
static inline float4 clamp(float4 v, float4 lo, float4 hi) { return float4(clamp(v.x, lo.x, hi.x), clamp(v.y, lo.y, hi.y), clamp(v.z, lo.z, hi.z), clamp(v.w, lo.w, hi.w)); }
static inline float sign(float v) { return P::signf_(v); } HL_LIFT1(sign)
static inline float sqrt(float v) { return P::sqrtf_(v); } HL_LIFT1(sqrt)
static inline float rsqrt(float v) { return 1.0f / P::sqrtf_(v); }
static inline float rcp(float v) { return 1.0f / v; }
static inline float floor(float v) { return P::dm_floor(v); } HL_LIFT1(floor)
static inline float frac(float v) { return v - P::dm_floor(v); } HL_LIFT1(frac)
static inline float sin(float v) { return P::dm_sin(v); } static inline float cos(float v) { return P::dm_cos(v); }
static inline void sincos(float v, float& s, float& c) { P::dm_sincos(v, s, c); }
static inline float exp2(float v) { return P::dm_exp2(v); } static inline float log2(float v) { return P::dm_log2(v); }
static inline float3 exp2(float3 v) { return float3(exp2(v.x), exp2(v.y), exp2(v.z)); } static inline float3 log2(float3 v) { return float3(log2(v.x), log2(v.y), log2(v.z)); }
static inline float exp(float v) { return P::dm_exp(v); } static inline float log(float v) { return P::dm_log(v); } HL_LIFT1(exp) HL_LIFT1(log)
static inline float atan2(float y, float x) { return P::dm_atan2(y, x); } static inline float acos(float v) { return P::dm_acos(v); }
template <class E> float pow(float x, E e) { return ((float)e == 5.0f) ? P::dm_pow5(x) : P::dm_pow(x, (float)e); }      // pow(x, 5): the oracle's (x²·x²)·x
template <class E> if_arith<float3, E> pow(float3 x, E e) { return float3(pow(x.x, e), pow(x.y, e), pow(x.z, e)); }
static inline float3 pow(float3 x, float3 e) { return float3(pow(x.x, e.x), pow(x.y, e.y), pow(x.z, e.z)); }
template <class E> if_arith<float4, E> pow(float4 x, E e) { return float4(pow(x.x, e), pow(x.y, e), pow(x.z, e), pow(x.w, e)); }
static inline float mad(float a, float b, float c) { return a * b + c; }
static inline float3 mad(float3 a, float3 b, float3 c) { return a * b + c; }                                                // unfused (-ffp-contract=off)
static inline float smoothstep(float a, float b, float x) { float t = saturate((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }
static inline float lerp(float a, float b, float t) { return P::lerpf(a, b, t); }
static inline float3 lerp(float3 a, float3 b, float t) { return a + (b - a) * t; }
static inline float3 lerp(float3 a, float3 b, float3 t) { return a + (b - a) * t; }
#if defined(RTXPT_LP_TYPES_USE_16BIT_PRECISION) && RTXPT_LP_TYPES_USE_16BIT_PRECISION
template <class A, class B> v3<ctype<A, B>> lerp(v3<A> a, v3<B> b, half_t t) { typedef ctype<A, B> C; return v3<C>(a) + (v3<C>(b) - v3<C>(a)) * t; }
#endif
static inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline float3 cross(float3 a, float3 b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float length(float2 a) { return sqrt(dot(a, a)); } static inline float length(float3 a) { return sqrt(dot(a, a)); }
static inline float3 normalize(float3 a) { float il = 1.0f / sqrt(dot(a, a)); return a * il; }
static inline float2 normalize(float2 a) { float il = 1.0f / sqrt(dot(a, a)); return a * il; }
static inline bool isnan(float v) { return v != v; } static inline bool isinf(float v) { return std::isinf(v); }
static inline uint asuint(float f) { return P::asuint(f); } static inline uint asuint(uint u) { return u; } static inline uint asuint(int i) { return (uint)i; }
static inline int asint(float f) { return P::asint(f); } static inline int asint(uint u) { return (int)u; }
static inline float asfloat(uint u) { return P::asfloat(u); } static inline float asfloat(int i) { return P::asfloat(i); } static inline float asfloat(float f) { return f; }
static inline int3 asint(float3 v) { return int3(asint(v.x), asint(v.y), asint(v.z)); }
static inline uint3 asuint(float3 v) { return uint3(asuint(v.x), asuint(v.y), asuint(v.z)); }
static inline float2 asfloat(uint2 v) { return float2(asfloat(v.x), asfloat(v.y)); }
static inline float3 asfloat(int3 v) { return float3(asfloat(v.x), asfloat(v.y), asfloat(v.z)); }
static inline float3 asfloat(uint3 v) { return float3(asfloat(v.x), asfloat(v.y), asfloat(v.z)); }
static inline uint f32tof16(float f) { return P::f32tof16(f); } static inline float f16tof32(uint h) { return P::f16tof32(h); }
static inline uint2 f32tof16(float2 f) { return uint2(f32tof16(f.x), f32tof16(f.y)); }
static inline uint4 f32tof16(float4 f) { return uint4(f32tof16(f.x), f32tof16(f.y), f32tof16(f.z), f32tof16(f.w)); }
static inline float4 f16tof32(uint4 h) { return float4(f16tof32(h.x), f16tof32(h.y), f16tof32(h.z), f16tof32(h.w)); }
static inline uint3 f32tof16(float3 f) { return uint3(f32tof16(f.x), f32tof16(f.y), f32tof16(f.z)); }
static inline float3 f16tof32(uint3 h) { return float3(f16tof32(h.x), f16tof32(h.y), f16tof32(h.z)); }
static inline float2 f16tof32(uint2 h) { return float2(f16tof32(h.x), f16tof32(h.y)); }
static inline uint countbits(uint v) { return (uint)__builtin_popcount(v); }
static inline uint firstbithigh(uint v) { return v ? 31u - (uint)__builtin_clz(v) : 0xFFFFFFFFu; }
static inline uint reversebits(uint v) { uint r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
// ---- matrices (row-major, rows are vectors): only what the integrator's helpers touch
struct float3x4;
struct float3x3 { float3 r[3]; float3x3() {} float3x3(const float3x4& M); float3x3(float3 a, float3 b, float3 c) { r[0] = a; r[1] = b; r[2] = c; }
    float3x3(float a, float b, float c, float d, float e, float f, float g, float h, float i) { r[0] = float3(a, b, c); r[1] = float3(d, e, f); r[2] = float3(g, h, i); }
    float3& operator[](uint i) { return r[i]; } const float3& operator[](uint i) const { return r[i]; } };
struct float3x4 { float4 r[3]; float4& operator[](uint i) { return r[i]; } const float4& operator[](uint i) const { return r[i]; } };
struct float4x4 { float4 r[4]; float4& operator[](uint i) { return r[i]; } const float4& operator[](uint i) const { return r[i]; } };
struct float2x2 { float2 r[2]; float2x2() {} float2x2(float a, float b, float c, float d) { r[0] = float2(a, b); r[1] = float2(c, d); } float2& operator[](uint i) { return r[i]; } const float2& operator[](uint i) const { return r[i]; } };
struct float2x3 { float3 r[2]; float3& operator[](uint i) { return r[i]; } const float3& operator[](uint i) const { return r[i]; } };
inline float3x3::float3x3(const float3x4& M) { r[0] = M.r[0].xyz_(); r[1] = M.r[1].xyz_(); r[2] = M.r[2].xyz_(); }
static inline float2 mul(float2x2 M, float2 v) { return float2(dot(M.r[0], v), dot(M.r[1], v)); }
static inline float3 mul(float3x3 M, float3 v) { return float3(dot(M.r[0], v), dot(M.r[1], v), dot(M.r[2], v)); }
static inline float3 mul(float3 v, float3x3 M) { return float3((v.x * M.r[0].x + v.y * M.r[1].x) + v.z * M.r[2].x, (v.x * M.r[0].y + v.y * M.r[1].y) + v.z * M.r[2].y, (v.x * M.r[0].z + v.y * M.r[1].z) + v.z * M.r[2].z); }
static inline float3x3 mul(float3x3 A, float3x3 B) { float3x3 R; for (int i = 0; i < 3; i++) R.r[i] = mul(A.r[i], B); return R; }
static inline float3x3 transpose(float3x3 M) { return float3x3(float3(M.r[0].x, M.r[1].x, M.r[2].x), float3(M.r[0].y, M.r[1].y, M.r[2].y), float3(M.r[0].z, M.r[1].z, M.r[2].z)); }
static inline float determinant(float3x3 M) { return dot(M.r[0], cross(M.r[1], M.r[2])); }
template <class T> T ddx(T) { return T(); } template <class T> T ddy(T) { return T(); }       // pixel-shader derivatives: no meaning here, never executed
struct SamplerState {};
#undef HL_LIFT1
#undef HL_LIFT2
} // namespace hl
