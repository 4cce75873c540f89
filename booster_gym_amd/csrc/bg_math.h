// Small fixed-size linear algebra for the per-lane rigid-body code.
// Everything is force-inlined and indexed with compile-time constants so that
// hipcc keeps all values in VGPRs (no scratch, no dynamic indexing).
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

struct V3 { float e[3]; };
struct M3 { float e[3][3]; };
struct S3 { float e[6]; };  // symmetric: xx yy zz xy xz yz
struct SV { V3 a, l; };     // spatial vector: angular, linear (motion: w,v  force: n,f)
struct SI { S3 A; M3 H; S3 M; };  // spatial inertia [A H; H^T M]

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
BG_HD V3 v3(float x, float y, float z) { V3 r; r.e[0] = x; r.e[1] = y; r.e[2] = z; return r; }
BG_HD V3 operator+(V3 a, V3 b) { return v3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
BG_HD V3 operator-(V3 a, V3 b) { return v3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
BG_HD V3 operator*(float s, V3 a) { return v3(s * a.e[0], s * a.e[1], s * a.e[2]); }
BG_HD V3 operator-(V3 a) { return v3(-a.e[0], -a.e[1], -a.e[2]); }
BG_HD float dot(V3 a, V3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
BG_HD V3 cross(V3 a, V3 b) {
    return v3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
BG_HD SV operator+(SV a, SV b) { SV r; r.a = a.a + b.a; r.l = a.l + b.l; return r; }
BG_HD SV operator-(SV a, SV b) { SV r; r.a = a.a - b.a; r.l = a.l - b.l; return r; }
BG_HD SV operator*(float s, SV a) { SV r; r.a = s * a.a; r.l = s * a.l; return r; }
BG_HD float dot(SV a, SV b) { return dot(a.a, b.a) + dot(a.l, b.l); }
BG_HD SV sv_zero() { SV r; r.a = v3(0, 0, 0); r.l = v3(0, 0, 0); return r; }

BG_HD M3 full(S3 s) {
    M3 m;
    m.e[0][0] = s.e[0]; m.e[1][1] = s.e[1]; m.e[2][2] = s.e[2];
    m.e[0][1] = m.e[1][0] = s.e[3]; m.e[0][2] = m.e[2][0] = s.e[4]; m.e[1][2] = m.e[2][1] = s.e[5];
    return m;
}
BG_HD S3 upper(M3 m) {
    S3 s;
    s.e[0] = m.e[0][0]; s.e[1] = m.e[1][1]; s.e[2] = m.e[2][2]; s.e[3] = m.e[0][1]; s.e[4] = m.e[0][2]; s.e[5] = m.e[1][2];
    return s;
}
BG_HD S3 s3_zero() { S3 s; for (int i = 0; i < 6; i++) s.e[i] = 0.f; return s; }
BG_HD M3 m3_zero() { M3 m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.e[i][j] = 0.f; return m; }
BG_HD S3 operator+(S3 a, S3 b) { S3 r; for (int i = 0; i < 6; i++) r.e[i] = a.e[i] + b.e[i]; return r; }
BG_HD M3 operator+(M3 a, M3 b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i][j] + b.e[i][j]; return r; }
BG_HD M3 operator-(M3 a, M3 b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i][j] - b.e[i][j]; return r; }
BG_HD V3 mul(M3 m, V3 v) {
    return v3(m.e[0][0] * v.e[0] + m.e[0][1] * v.e[1] + m.e[0][2] * v.e[2], m.e[1][0] * v.e[0] + m.e[1][1] * v.e[1] + m.e[1][2] * v.e[2],
              m.e[2][0] * v.e[0] + m.e[2][1] * v.e[1] + m.e[2][2] * v.e[2]);
}
BG_HD V3 mulT(M3 m, V3 v) {
    return v3(m.e[0][0] * v.e[0] + m.e[1][0] * v.e[1] + m.e[2][0] * v.e[2], m.e[0][1] * v.e[0] + m.e[1][1] * v.e[1] + m.e[2][1] * v.e[2],
              m.e[0][2] * v.e[0] + m.e[1][2] * v.e[1] + m.e[2][2] * v.e[2]);
}
BG_HD V3 mul(S3 s, V3 v) {
    return v3(s.e[0] * v.e[0] + s.e[3] * v.e[1] + s.e[4] * v.e[2], s.e[3] * v.e[0] + s.e[1] * v.e[1] + s.e[5] * v.e[2],
              s.e[4] * v.e[0] + s.e[5] * v.e[1] + s.e[2] * v.e[2]);
}
BG_HD M3 mul(M3 a, M3 b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i][0] * b.e[0][j] + a.e[i][1] * b.e[1][j] + a.e[i][2] * b.e[2][j];
    return r;
}
BG_HD M3 transpose(M3 a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[j][i]; return r; }
// r x M  (cross of r with every column)
BG_HD M3 cross_cols(V3 r, M3 m) {
    M3 o;
    for (int j = 0; j < 3; j++) {
        V3 c = cross(r, v3(m.e[0][j], m.e[1][j], m.e[2][j]));
        o.e[0][j] = c.e[0]; o.e[1][j] = c.e[1]; o.e[2][j] = c.e[2];
    }
    return o;
}
// M rx  (row i of result = row_i x ... : (M rx) v = M (r x v)  =>  row_i(M rx) = row_i(M) x r ... sign handled here)
BG_HD M3 mul_skew(M3 m, V3 r) {
    // (M rx)_{i,:} = -(r x row_i)^T  because  row_i . (r x v) = (row_i x r) . v
    M3 o;
    for (int i = 0; i < 3; i++) {
        V3 c = cross(v3(m.e[i][0], m.e[i][1], m.e[i][2]), r);
        o.e[i][0] = c.e[0]; o.e[i][1] = c.e[1]; o.e[i][2] = c.e[2];
    }
    return o;
}
BG_HD M3 skew(V3 v) {
    M3 s;
    s.e[0][0] = 0.f; s.e[0][1] = -v.e[2]; s.e[0][2] = v.e[1];
    s.e[1][0] = v.e[2]; s.e[1][1] = 0.f; s.e[1][2] = -v.e[0];
    s.e[2][0] = -v.e[1]; s.e[2][1] = v.e[0]; s.e[2][2] = 0.f;
    return s;
}
BG_HD M3 outer(V3 a, V3 b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.e[i][j] = a.e[i] * b.e[j]; return r; }

// closed-form inverse of a symmetric positive definite 3x3
BG_HD S3 inv_sym(S3 s) {
    float a = s.e[0], b = s.e[1], c = s.e[2], d = s.e[3], e = s.e[4], f = s.e[5];
    float c00 = b * c - f * f, c01 = e * f - d * c, c02 = d * f - b * e;
    float c11 = a * c - e * e, c12 = d * e - a * f, c22 = a * b - d * d;
    float idet = bg_rcp(a * c00 + d * c01 + e * c02);
    S3 r;
    r.e[0] = c00 * idet; r.e[1] = c11 * idet; r.e[2] = c22 * idet; r.e[3] = c01 * idet; r.e[4] = c02 * idet; r.e[5] = c12 * idet;
    return r;
}

// --- rotations about a coordinate axis.  AX = 1 (x), 2 (y), 3 (z).  (J,K) is the rotated plane.
template <int AX> struct Plane { static constexpr int J = AX % 3, K = (AX + 1) % 3, A = AX - 1; };
// parent <- child :  R(axis, q) v
template <int AX> BG_HD V3 rot(float c, float s, V3 v) {
    constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
    V3 o = v;
    o.e[J] = c * v.e[J] - s * v.e[K];
    o.e[K] = s * v.e[J] + c * v.e[K];
    return o;
}
// child <- parent :  R^T v
template <int AX> BG_HD V3 rotT(float c, float s, V3 v) { return rot<AX>(c, -s, v); }
// R m R^T
template <int AX> BG_HD M3 rot_conj(float c, float s, M3 m) {
    constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
    M3 t = m;
    for (int j = 0; j < 3; j++) {  // rows: t = R m
        t.e[J][j] = c * m.e[J][j] - s * m.e[K][j];
        t.e[K][j] = s * m.e[J][j] + c * m.e[K][j];
    }
    M3 o = t;
    for (int i = 0; i < 3; i++) {  // columns: o = t R^T
        o.e[i][J] = c * t.e[i][J] - s * t.e[i][K];
        o.e[i][K] = s * t.e[i][J] + c * t.e[i][K];
    }
    return o;
}

// R s R^T for a SYMMETRIC s: the axis entry stays, the two entries that couple the axis with the rotated plane turn like a 2-vector, and
// the 2x2 block of the plane turns by the double angle -- 14 operations (+ 3 for c2, s2, shared by the caller) against 24 for a full matrix
BG_HD constexpr int s3_index(int i, int j) { return i == j ? i : (i + j == 1 ? 3 : (i + j == 2 ? 4 : 5)); }  // xx yy zz xy xz yz
template <int AX> BG_HD S3 rot_conj_sym(float c, float s, float c2, float s2, S3 m) {
    constexpr int J = Plane<AX>::J, K = Plane<AX>::K, A = Plane<AX>::A;
    constexpr int JJ = s3_index(J, J), KK = s3_index(K, K), JK = s3_index(J, K), AJ = s3_index(A, J), AK = s3_index(A, K);
    S3 o = m;
    o.e[AJ] = c * m.e[AJ] - s * m.e[AK];
    o.e[AK] = s * m.e[AJ] + c * m.e[AK];
    const float mean = 0.5f * (m.e[JJ] + m.e[KK]), dev = 0.5f * (m.e[JJ] - m.e[KK]);
    const float t = dev * c2 - m.e[JK] * s2;
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

}  // namespace bg
