// Small fixed-size linear algebra for the per-lane rigid-body code.
// Everything is force-inlined and indexed with compile-time constants so that
// hipcc keeps all values in VGPRs (no scratch, no dynamic indexing).
//
// Generic over the scalar type T: `float` (one leg per lane: the fused env step and the granular simulator calls, bg_dyn.h) or `f2`, two
// floats in one 64-bit register pair (both legs of an env in one lane: the packed ABA kernel, bg_dyn_pk.h, whose adds / multiplies / FMAs
// then issue as v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  V3, M3, ... are the float instances.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BG_HD __host__ __device__ __forceinline__
#else
#define BG_HD inline __attribute__((always_inline))
#endif

namespace bg {

// two floats side by side: component 0 = left leg, component 1 = right leg.  clang: ext_vector_type (HIP device code); g++ (the host test
// harness): vector_size.  Both give element-wise + - * and scalar-with-vector arithmetic; components are read and written with [0] / [1].
#if defined(__clang__)
typedef float f2 __attribute__((ext_vector_type(2)));
#else
typedef float f2 __attribute__((vector_size(8)));
#endif
BG_HD f2 mk2(float a, float b) { f2 r; r[0] = a; r[1] = b; return r; }
BG_HD f2 splat2(float a) { return mk2(a, a); }
template <class T> struct Splat;
template <> struct Splat<float> { static BG_HD float of(float a) { return a; } };
template <> struct Splat<f2> { static BG_HD f2 of(float a) { return splat2(a); } };
template <class T> BG_HD T splat(float a) { return Splat<T>::of(a); }

template <class T> struct V3T { T e[3]; };
template <class T> struct M3T { T e[3][3]; };
template <class T> struct S3T { T e[6]; };  // symmetric: xx yy zz xy xz yz
template <class T> struct SVT { V3T<T> a, l; };     // spatial vector: angular, linear (motion: w,v  force: n,f)
template <class T> struct SIT { S3T<T> A; M3T<T> H; S3T<T> M; };  // spatial inertia [A H; H^T M]
using V3 = V3T<float>; using M3 = M3T<float>; using S3 = S3T<float>; using SV = SVT<float>; using SI = SIT<float>;

// reciprocal / square root: the hardware approximations (1 ulp) on the GPU instead of the ~10-instruction IEEE division expansion
BG_HD float bg_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
BG_HD float bg_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
BG_HD uint32_t float_bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(x);
#else
    uint32_t u; __builtin_memcpy(&u, &x, 4); return u;
#endif
}
BG_HD void bg_sincos(float x, float* s, float* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    __sincosf(x, s, c);
#else
    *s = sinf(x);
    *c = cosf(x);
#endif
}
BG_HD float bg_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
BG_HD float bg_min(float a, float b) { return fminf(a, b); }
BG_HD float bg_max(float a, float b) { return fmaxf(a, b); }
BG_HD float bg_abs(float a) { return fabsf(a); }
// the two-wide forms: what has no packed instruction (transcendentals, min / max, selects) runs once per half
BG_HD f2 bg_rcp(f2 x) { return mk2(bg_rcp(x[0]), bg_rcp(x[1])); }
BG_HD f2 bg_sqrt(f2 x) { return mk2(bg_sqrt(x[0]), bg_sqrt(x[1])); }
BG_HD f2 bg_rsqrt(f2 x) { return mk2(bg_rsqrt(x[0]), bg_rsqrt(x[1])); }
BG_HD void bg_sincos(f2 x, f2* s, f2* c) {
    float s0, c0, s1, c1;
    bg_sincos(x[0], &s0, &c0); bg_sincos(x[1], &s1, &c1);
    *s = mk2(s0, s1); *c = mk2(c0, c1);
}
BG_HD f2 bg_min(f2 a, f2 b) { return mk2(fminf(a[0], b[0]), fminf(a[1], b[1])); }
BG_HD f2 bg_max(f2 a, f2 b) { return mk2(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])); }
BG_HD f2 bg_abs(f2 a) { return mk2(fabsf(a[0]), fabsf(a[1])); }

// comparisons and selects, per half for f2 (no packed compare / select exists: one v_cmp + v_cndmask per half)
struct B2 { bool x, y; };
BG_HD bool lt(float a, float b) { return a < b; }
BG_HD bool gt(float a, float b) { return a > b; }
BG_HD B2 lt(f2 a, f2 b) { B2 r; r.x = a[0] < b[0]; r.y = a[1] < b[1]; return r; }
BG_HD B2 gt(f2 a, f2 b) { B2 r; r.x = a[0] > b[0]; r.y = a[1] > b[1]; return r; }
BG_HD bool nonzero(float a) { return a != 0.f; }
BG_HD B2 nonzero(f2 a) { B2 r; r.x = a[0] != 0.f; r.y = a[1] != 0.f; return r; }
BG_HD bool both(bool a, bool b) { return a && b; }
BG_HD B2 both(B2 a, B2 b) { B2 r; r.x = a.x && b.x; r.y = a.y && b.y; return r; }
BG_HD bool any_of(bool a) { return a; }
BG_HD bool any_of(B2 a) { return a.x || a.y; }
BG_HD float sel(bool c, float a, float b) { return c ? a : b; }
BG_HD f2 sel(B2 c, f2 a, f2 b) { return mk2(c.x ? a[0] : b[0], c.y ? a[1] : b[1]); }

template <class T> BG_HD V3T<T> v3t(T x, T y, T z) { V3T<T> r; r.e[0] = x; r.e[1] = y; r.e[2] = z; return r; }
BG_HD V3 v3(float x, float y, float z) { return v3t<float>(x, y, z); }
template <class T> BG_HD V3T<T> v3_zero() { return v3t<T>(splat<T>(0.f), splat<T>(0.f), splat<T>(0.f)); }
template <class T> BG_HD V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3t<T>(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
template <class T> BG_HD V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3t<T>(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
template <class S, class T> BG_HD V3T<T> operator*(S s, V3T<T> a) { return v3t<T>(s * a.e[0], s * a.e[1], s * a.e[2]); }  // S = T or float
template <class T> BG_HD V3T<T> operator-(V3T<T> a) { return v3t<T>(-a.e[0], -a.e[1], -a.e[2]); }
template <class T> BG_HD T dot(V3T<T> a, V3T<T> b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
template <class T> BG_HD V3T<T> cross(V3T<T> a, V3T<T> b) {
    return v3t<T>(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
template <class T> BG_HD SVT<T> operator+(SVT<T> a, SVT<T> b) { SVT<T> r; r.a = a.a + b.a; r.l = a.l + b.l; return r; }
template <class T> BG_HD SVT<T> operator-(SVT<T> a, SVT<T> b) { SVT<T> r; r.a = a.a - b.a; r.l = a.l - b.l; return r; }
template <class S, class T> BG_HD SVT<T> operator*(S s, SVT<T> a) { SVT<T> r; r.a = s * a.a; r.l = s * a.l; return r; }
template <class T> BG_HD T dot(SVT<T> a, SVT<T> b) { return dot(a.a, b.a) + dot(a.l, b.l); }
template <class T> BG_HD SVT<T> svt_zero() { SVT<T> r; r.a = v3_zero<T>(); r.l = v3_zero<T>(); return r; }
BG_HD SV sv_zero() { return svt_zero<float>(); }

template <class T> BG_HD M3T<T> full(S3T<T> s) {
    M3T<T> m;
    m.e[0][0] = s.e[0]; m.e[1][1] = s.e[1]; m.e[2][2] = s.e[2];
    m.e[0][1] = m.e[1][0] = s.e[3]; m.e[0][2] = m.e[2][0] = s.e[4]; m.e[1][2] = m.e[2][1] = s.e[5];
    return m;
}
template <class T> BG_HD S3T<T> upper(M3T<T> m) {
    S3T<T> s;
    s.e[0] = m.e[0][0]; s.e[1] = m.e[1][1]; s.e[2] = m.e[2][2]; s.e[3] = m.e[0][1]; s.e[4] = m.e[0][2]; s.e[5] = m.e[1][2];
    return s;
}
template <class T> BG_HD S3T<T> s3t_zero() { S3T<T> s; for (int i = 0; i < 6; i++) s.e[i] = splat<T>(0.f); return s; }
template <class T> BG_HD M3T<T> m3t_zero() { M3T<T> m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.e[i][j] = splat<T>(0.f); return m; }
BG_HD S3 s3_zero() { return s3t_zero<float>(); }
BG_HD M3 m3_zero() { return m3t_zero<float>(); }
template <class T> BG_HD S3T<T> operator+(S3T<T> a, S3T<T> b) { S3T<T> r; for (int i = 0; i < 6; i++) r.e[i] = a.e[i] + b.e[i]; return r; }
template <class T> BG_HD M3T<T> operator+(M3T<T> a, M3T<T> b) { M3T<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i][j] + b.e[i][j]; return r; }
template <class T> BG_HD M3T<T> operator-(M3T<T> a, M3T<T> b) { M3T<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i][j] - b.e[i][j]; return r; }
template <class T> BG_HD V3T<T> mul(M3T<T> m, V3T<T> v) {
    return v3t<T>(m.e[0][0] * v.e[0] + m.e[0][1] * v.e[1] + m.e[0][2] * v.e[2], m.e[1][0] * v.e[0] + m.e[1][1] * v.e[1] + m.e[1][2] * v.e[2],
                  m.e[2][0] * v.e[0] + m.e[2][1] * v.e[1] + m.e[2][2] * v.e[2]);
}
template <class T> BG_HD V3T<T> mulT(M3T<T> m, V3T<T> v) {
    return v3t<T>(m.e[0][0] * v.e[0] + m.e[1][0] * v.e[1] + m.e[2][0] * v.e[2], m.e[0][1] * v.e[0] + m.e[1][1] * v.e[1] + m.e[2][1] * v.e[2],
                  m.e[0][2] * v.e[0] + m.e[1][2] * v.e[1] + m.e[2][2] * v.e[2]);
}
template <class T> BG_HD V3T<T> mul(S3T<T> s, V3T<T> v) {
    return v3t<T>(s.e[0] * v.e[0] + s.e[3] * v.e[1] + s.e[4] * v.e[2], s.e[3] * v.e[0] + s.e[1] * v.e[1] + s.e[5] * v.e[2],
                  s.e[4] * v.e[0] + s.e[5] * v.e[1] + s.e[2] * v.e[2]);
}
template <class T> BG_HD M3T<T> mul(M3T<T> a, M3T<T> b) {
    M3T<T> r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i][0] * b.e[0][j] + a.e[i][1] * b.e[1][j] + a.e[i][2] * b.e[2][j];
    return r;
}
template <class T> BG_HD M3T<T> transpose(M3T<T> a) { M3T<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[j][i]; return r; }
// r x M  (cross of r with every column)
template <class T> BG_HD M3T<T> cross_cols(V3T<T> r, M3T<T> m) {
    M3T<T> o;
    for (int j = 0; j < 3; j++) {
        V3T<T> c = cross(r, v3t<T>(m.e[0][j], m.e[1][j], m.e[2][j]));
        o.e[0][j] = c.e[0]; o.e[1][j] = c.e[1]; o.e[2][j] = c.e[2];
    }
    return o;
}
// M rx  (row i of result = row_i x ... : (M rx) v = M (r x v)  =>  row_i(M rx) = row_i(M) x r ... sign handled here)
template <class T> BG_HD M3T<T> mul_skew(M3T<T> m, V3T<T> r) {
    // (M rx)_{i,:} = -(r x row_i)^T  because  row_i . (r x v) = (row_i x r) . v
    M3T<T> o;
    for (int i = 0; i < 3; i++) {
        V3T<T> c = cross(v3t<T>(m.e[i][0], m.e[i][1], m.e[i][2]), r);
        o.e[i][0] = c.e[0]; o.e[i][1] = c.e[1]; o.e[i][2] = c.e[2];
    }
    return o;
}
template <class T> BG_HD M3T<T> skew(V3T<T> v) {
    M3T<T> s;
    s.e[0][0] = splat<T>(0.f); s.e[0][1] = -v.e[2]; s.e[0][2] = v.e[1];
    s.e[1][0] = v.e[2]; s.e[1][1] = splat<T>(0.f); s.e[1][2] = -v.e[0];
    s.e[2][0] = -v.e[1]; s.e[2][1] = v.e[0]; s.e[2][2] = splat<T>(0.f);
    return s;
}
template <class T> BG_HD M3T<T> outer(V3T<T> a, V3T<T> b) { M3T<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i] * b.e[j]; return r; }

// closed-form inverse of a symmetric positive definite 3x3
template <class T> BG_HD S3T<T> inv_sym(S3T<T> s) {
    T a = s.e[0], b = s.e[1], c = s.e[2], d = s.e[3], e = s.e[4], f = s.e[5];
    T c00 = b * c - f * f, c01 = e * f - d * c, c02 = d * f - b * e;
    T c11 = a * c - e * e, c12 = d * e - a * f, c22 = a * b - d * d;
    T idet = bg_rcp(a * c00 + d * c01 + e * c02);
    S3T<T> r;
    r.e[0] = c00 * idet; r.e[1] = c11 * idet; r.e[2] = c22 * idet; r.e[3] = c01 * idet; r.e[4] = c02 * idet; r.e[5] = c12 * idet;
    return r;
}

// --- rotations about a coordinate axis.  AX = 1 (x), 2 (y), 3 (z).  (J,K) is the rotated plane.
template <int AX> struct Plane { static constexpr int J = AX % 3, K = (AX + 1) % 3, A = AX - 1; };
// parent <- child :  R(axis, q) v
template <int AX, class T> BG_HD V3T<T> rot(T c, T s, V3T<T> v) {
    constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
    V3T<T> o = v;
    o.e[J] = c * v.e[J] - s * v.e[K];
    o.e[K] = s * v.e[J] + c * v.e[K];
    return o;
}
// child <- parent :  R^T v
template <int AX, class T> BG_HD V3T<T> rotT(T c, T s, V3T<T> v) { return rot<AX>(c, -s, v); }
// R m R^T
template <int AX, class T> BG_HD M3T<T> rot_conj(T c, T s, M3T<T> m) {
    constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
    M3T<T> t = m;
    for (int j = 0; j < 3; j++) {  // rows: t = R m
        t.e[J][j] = c * m.e[J][j] - s * m.e[K][j];
        t.e[K][j] = s * m.e[J][j] + c * m.e[K][j];
    }
    M3T<T> o = t;
    for (int i = 0; i < 3; i++) {  // columns: o = t R^T
        o.e[i][J] = c * t.e[i][J] - s * t.e[i][K];
        o.e[i][K] = s * t.e[i][J] + c * t.e[i][K];
    }
    return o;
}

// R s R^T for a SYMMETRIC s: the axis entry stays, the two entries that couple the axis with the rotated plane turn like a 2-vector, and
// the 2x2 block of the plane turns by the double angle -- 14 operations (+ 3 for c2, s2, shared by the caller) against 24 for a full matrix
BG_HD constexpr int s3_index(int i, int j) { return i == j ? i : (i + j == 1 ? 3 : (i + j == 2 ? 4 : 5)); }  // xx yy zz xy xz yz
template <int AX, class T> BG_HD S3T<T> rot_conj_sym(T c, T s, T c2, T s2, S3T<T> m) {
    constexpr int J = Plane<AX>::J, K = Plane<AX>::K, A = Plane<AX>::A;
    constexpr int JJ = s3_index(J, J), KK = s3_index(K, K), JK = s3_index(J, K), AJ = s3_index(A, J), AK = s3_index(A, K);
    S3T<T> o = m;
    o.e[AJ] = c * m.e[AJ] - s * m.e[AK];
    o.e[AK] = s * m.e[AJ] + c * m.e[AK];
    const T mean = 0.5f * (m.e[JJ] + m.e[KK]), dev = 0.5f * (m.e[JJ] - m.e[KK]);
    const T t = dev * c2 - m.e[JK] * s2;
    o.e[JJ] = mean + t;
    o.e[KK] = mean - t;
    o.e[JK] = dev * s2 + m.e[JK] * c2;
    return o;
}

// quaternion xyzw (Isaac Gym convention, t1.py:221) -> body-to-world rotation
BG_HD M3 quat_to_mat(const float q[4]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    M3 r;
    r.e[0][0] = 1.f - 2.f * (y * y + z * z); r.e[0][1] = 2.f * (x * y - z * w); r.e[0][2] = 2.f * (x * z + y * w);
    r.e[1][0] = 2.f * (x * y + z * w); r.e[1][1] = 1.f - 2.f * (x * x + z * z); r.e[1][2] = 2.f * (y * z - x * w);
    r.e[2][0] = 2.f * (x * z - y * w); r.e[2][1] = 2.f * (y * z + x * w); r.e[2][2] = 1.f - 2.f * (x * x + y * y);
    return r;
}
// rotation matrix -> unit quaternion xyzw with w >= 0 (Shepperd's branches)
BG_HD void mat_to_quat(const M3& R, float q[4]) {
    float tr = R.e[0][0] + R.e[1][1] + R.e[2][2], x, y, z, w;
    if (tr > 0.f) {
        float s = bg_sqrt(tr + 1.0f) * 2.0f, inv = bg_rcp(s);
        w = 0.25f * s; x = (R.e[2][1] - R.e[1][2]) * inv; y = (R.e[0][2] - R.e[2][0]) * inv; z = (R.e[1][0] - R.e[0][1]) * inv;
    } else if (R.e[0][0] > R.e[1][1] && R.e[0][0] > R.e[2][2]) {
        float s = bg_sqrt(1.0f + R.e[0][0] - R.e[1][1] - R.e[2][2]) * 2.0f, inv = bg_rcp(s);
        w = (R.e[2][1] - R.e[1][2]) * inv; x = 0.25f * s; y = (R.e[0][1] + R.e[1][0]) * inv; z = (R.e[0][2] + R.e[2][0]) * inv;
    } else if (R.e[1][1] > R.e[2][2]) {
        float s = bg_sqrt(1.0f + R.e[1][1] - R.e[0][0] - R.e[2][2]) * 2.0f, inv = bg_rcp(s);
        w = (R.e[0][2] - R.e[2][0]) * inv; x = (R.e[0][1] + R.e[1][0]) * inv; y = 0.25f * s; z = (R.e[1][2] + R.e[2][1]) * inv;
    } else {
        float s = bg_sqrt(1.0f + R.e[2][2] - R.e[0][0] - R.e[1][1]) * 2.0f, inv = bg_rcp(s);
        w = (R.e[1][0] - R.e[0][1]) * inv; x = (R.e[0][2] + R.e[2][0]) * inv; y = (R.e[1][2] + R.e[2][1]) * inv; z = 0.25f * s;
    }
    if (w < 0.f) { x = -x; y = -y; z = -z; w = -w; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

// float -> T (T = f2: the trunk's quantities as seen by both legs of a lane; T = float: the identity), and one half of a two-wide value
template <class T> BG_HD V3T<T> splat_v3(V3 v) { return v3t<T>(splat<T>(v.e[0]), splat<T>(v.e[1]), splat<T>(v.e[2])); }
template <class T> BG_HD M3T<T> splat_m3(const M3& m) { M3T<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = splat<T>(m.e[i][j]); return r; }
template <class T> BG_HD SVT<T> splat_sv(SV v) { SVT<T> r; r.a = splat_v3<T>(v.a); r.l = splat_v3<T>(v.l); return r; }
BG_HD V3 half(V3T<f2> v, int h) { return v3(v.e[0][h], v.e[1][h], v.e[2][h]); }
BG_HD M3 half(const M3T<f2>& m, int h) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = m.e[i][j][h]; return r; }
BG_HD SV half(SVT<f2> v, int h) { SV r; r.a = half(v.a, h); r.l = half(v.l, h); return r; }
BG_HD V3 sum_halves(V3T<f2> v) { return v3(v.e[0][0] + v.e[0][1], v.e[1][0] + v.e[1][1], v.e[2][0] + v.e[2][1]); }

}  // namespace bg
