// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the
// product (pbrt-rs_amd/); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it.
//
// CPU restatement of the scalar/vector/interval arithmetic of erxma/pbrt-rs
// (reference paths are relative to /root/reference/).  Float = f32, no
// contraction (build with -ffp-contract=off), fmaf only where the reference
// writes mul_add.
//
// Pinning: the reference is Rust and cannot be built here (no cargo/rustc);
// the functions below are pinned by the reference's own known-answer tests
// where they exist (see tests/test_oracle_golden.py) and are otherwise
// "parity unpinned" (SURVEY.md §8c).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace ora {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Rust f32::min / f32::max: if one operand is NaN the other is returned.
static inline float rmin(float a, float b) { return std::fmin(a, b); }
static inline float rmax(float a, float b) { return std::fmax(a, b); }

// src/core/float.rs:14-36
static inline float next_float_up(float v) {
    if ((std::isinf(v) && v > 0.0f) || std::isnan(v)) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t b = f2u(v);
    b = std::signbit(v) ? b - 1 : b + 1;
    return u2f(b);
}
// src/core/float.rs:40-60
static inline float next_float_down(float v) {
    if ((std::isinf(v) && v < 0.0f) || std::isnan(v)) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t b = f2u(v);
    b = std::signbit(v) ? b + 1 : b - 1;
    return u2f(b);
}

static const float MACHINE_EPSILON = 5.9604644775390625e-08f;     // 2^-24, constants.rs:19
static const float ONE_MINUS_EPSILON = 1.0f - 1.1920928955078125e-07f;  // 1-2^-23, constants.rs:34 (B19)
static const float SHADOW_EPSILON = 0.0001f;                      // constants.rs:35
static const float PI_F = 3.14159265358979323846f;
static const float FRAC_1_PI = 0.318309886183790671537767526745028724f;
static const float FRAC_PI_2 = 1.57079632679489661923132169163975144f;
static const float FRAC_PI_4 = 0.785398163397448309615660845819875721f;
static const float INV_4_PI = 0.07957747f;
static const float INV_2_PI = 0.15915494f;

// src/core/routines.rs:16-19
static inline float gamma_n(int n) {
    float nf = (float)n;
    return (MACHINE_EPSILON * nf) / (1.0f - nf * MACHINE_EPSILON);
}
// src/core/routines.rs:9-14 -- argument order (a, b, t)
static inline float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

static inline float safe_sqrt(float x) { return std::sqrt(rmax(x, 0.0f)); }   // routines.rs:38-45
static inline float clampf(float x, float lo, float hi) {                      // Rust f32::clamp
    float r = x;
    if (r < lo) r = lo;
    if (r > hi) r = hi;
    return r;
}
static inline float safe_acos(float x) { return std::acos(clampf(x, -1.0f, 1.0f)); }

// Transcendental functions of the per-sample path (sin, cos, atan2, atanh, cosh, exp).  Two modes:
//   0  FAITHFUL (default): glibc's f32 routines - what Rust's std f32::sin etc. call on Linux.
//   1  CORRECTLY ROUNDED: evaluated in f64 and rounded once to f32 - what the device kernels do (pm_shade.hpp cr_*), because
//      ROCm's f32 routines differ from glibc's in the last ulp.  In this mode the oracle and the GPU run the same arithmetic
//      (up to the ~2^-29 of inputs on which two f64 libms round differently), so samples compare bit for bit and a logic
//      bug can no longer hide inside a "libm flipped a discrete outcome" tolerance.
inline int& trans_mode() { static int mode = 0; return mode; }
static inline float t_sin(float x) { return trans_mode() ? (float)std::sin((double)x) : std::sin(x); }
static inline float t_cos(float x) { return trans_mode() ? (float)std::cos((double)x) : std::cos(x); }
static inline float t_atan2(float y, float x) { return trans_mode() ? (float)std::atan2((double)y, (double)x) : std::atan2(y, x); }
static inline float t_atanh(float x) { return trans_mode() ? (float)std::atanh((double)x) : std::atanh(x); }
static inline float t_cosh(float x) { return trans_mode() ? (float)std::cosh((double)x) : std::cosh(x); }
static inline float t_exp(float x) { return trans_mode() ? (float)std::exp((double)x) : std::exp(x); }
static inline float t_log(float x) { return trans_mode() ? (float)std::log((double)x) : std::log(x); }

// src/math.rs:78-87
static inline float difference_of_products(float a, float b, float c, float d) {
    float cd = c * d;
    float dop = std::fmaf(a, b, -cd);
    float err = std::fmaf(-c, d, cd);
    return dop + err;
}
// src/math.rs:42-76
static inline bool solve_quadratic(float a, float b, float c, float* t0, float* t1) {
    if (a == 0.0f) {
        if (b == 0.0f) return false;
        float t = -c / b;
        *t0 = t; *t1 = t;
        return true;
    }
    float discrim = difference_of_products(b, b, 4.0f * a, c);
    if (discrim < 0.0f) return false;
    float root = std::sqrt(discrim);
    float q = -0.5f * (b + std::copysign(root, b));
    float r0 = q / a;
    float r1 = c / q;
    if (r0 <= r1) { *t0 = r0; *t1 = r1; } else { *t0 = r1; *t1 = r0; }
    return true;
}
// src/math.rs:25-29
static inline float evaluate_polynomial(float t, const float* c, int n) {
    float sum = 0.0f;
    for (int i = n - 1; i >= 0; --i) sum = std::fmaf(sum, t, c[i]);
    return sum;
}

// ------------------------------------------------------------------ Interval
// src/core/interval.rs
struct Interval {
    float low, high;
    Interval() : low(0), high(0) {}
    Interval(float l, float h) : low(l), high(h) {}
    static Interval exact(float v) { return Interval(v, v); }
    static Interval with_err(float v, float err) {   // :23-29
        if (err == 0.0f) return exact(v);
        return Interval(next_float_down(v - err), next_float_up(v + err));
    }
    float midpoint() const { return (low + high) / 2.0f; }   // :39-41
    float width() const { return high - low; }
    Interval squared() const {                               // :55-76
        if (low >= 0.0f) return Interval(next_float_down(low * low), next_float_up(high * high));
        if (high <= 0.0f) return Interval(next_float_down(high * high), next_float_up(low * low));
        float la = rmax(std::fabs(low), std::fabs(high));
        return Interval(0.0f, next_float_up(la * la));
    }
    Interval sqrt() const {                                  // :78-83
        return Interval(rmax(next_float_down(std::sqrt(low)), 0.0f), next_float_up(std::sqrt(high)));
    }
};
static inline Interval operator+(Interval a, Interval b) {   // :121-130
    return Interval(next_float_down(a.low + b.low), next_float_up(a.high + b.high));
}
static inline Interval operator-(Interval a, Interval b) {   // :132-141
    return Interval(next_float_down(a.low - b.high), next_float_up(a.high - b.low));
}
static inline Interval operator*(Interval a, Interval b) {   // :143-159
    float p0 = a.low * b.low, p1 = a.low * b.high, p2 = a.high * b.low, p3 = a.high * b.high;
    float lo = rmin(rmin(rmin(p0, p1), p2), p3);
    float hi = rmax(rmax(rmax(p0, p1), p2), p3);
    return Interval(next_float_down(lo), next_float_up(hi));
}
static inline Interval operator*(Interval a, float f) {      // :161-183 (both orders)
    if (f > 0.0f) return Interval(next_float_down(a.low * f), next_float_up(a.high * f));
    return Interval(next_float_down(a.high * f), next_float_up(a.low * f));
}
static inline Interval operator*(float f, Interval a) {
    if (f > 0.0f) return Interval(next_float_down(f * a.low), next_float_up(f * a.high));
    return Interval(next_float_down(f * a.high), next_float_up(f * a.low));
}
static inline Interval operator/(Interval a, Interval b) {   // :201-224
    if (b.low < 0.0f && b.high > 0.0f)
        return Interval(-std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity());
    float q0 = a.low / b.low, q1 = a.low / b.high, q2 = a.high / b.low, q3 = a.high / b.high;
    float lo = rmin(rmin(rmin(q0, q1), q2), q3);
    float hi = rmax(rmax(rmax(q0, q1), q2), q3);
    return Interval(next_float_down(lo), next_float_up(hi));
}
static inline Interval operator/(Interval a, float f) {      // :226-249
    if (f < 0.0f) return Interval(next_float_down(a.high / f), next_float_up(a.low / f));
    if (f == 0.0f) {
        auto sgn = [](float x) { return std::isnan(x) ? x : (std::signbit(x) ? -1.0f : 1.0f); };
        if (sgn(a.low) == sgn(a.high)) return Interval::exact(a.low / f);
        return Interval(-std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity());
    }
    if (std::isnan(f)) return Interval(NAN, NAN);
    return Interval(next_float_down(a.low / f), next_float_up(a.high / f));
}
static inline bool operator==(Interval a, Interval b) { return a.low == b.low && a.high == b.high; }

// ------------------------------------------------------------------ Vec3 / Point3 / Normal3 (f32)
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, float f) { return V3(a.x * f, a.y * f, a.z * f); }
static inline V3 operator*(float f, V3 a) { return V3(a.x * f, a.y * f, a.z * f); }  // tuple.rs:166-173 (rhs * self)
static inline V3 operator/(V3 a, float f) { return V3(a.x / f, a.y / f, a.z / f); }
// src/core/vec.rs:689-691
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float absdot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
// src/core/vec.rs:700-712 -- computed in f64, narrowed once (B18)
static inline V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((float)(ay * bz - az * by), (float)(az * bx - ax * bz), (float)(ax * by - ay * bx));
}
static inline float length_squared(V3 a) { return dot(a, a); }
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalized(V3 a) { return a / length(a); }   // vec.rs:722-728
static inline V3 vabs(V3 a) { return V3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
static inline float max_component(V3 a) {                      // tuple.rs:63-74 (max_by keeps last max)
    float m = a.x;
    if (!(a.y < m)) m = a.y;
    if (!(a.z < m)) m = a.z;
    return m;
}
static inline V3 vlerp(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
static inline float distance(V3 a, V3 b) { return length(a - b); }
static inline float distance_squared(V3 a, V3 b) { return length_squared(a - b); }
// src/core/normal3.rs:82-88 (B16: is_sign_positive, so -0.0 flips)
static inline V3 face_forward(V3 n, V3 v) { return std::signbit(dot(n, v)) ? -n : n; }
// src/core/vec.rs:743-756
static inline void coordinate_system(V3 v1, V3* v2, V3* v3) {
    if (std::fabs(v1.x) > std::fabs(v1.y))
        *v2 = V3(-v1.z, 0.0f, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3(0.0f, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = cross(v1, *v2);
}

struct V3fi {   // Vec3fi / Point3fi
    Interval x, y, z;
    V3fi() {}
    V3fi(Interval a, Interval b, Interval c) : x(a), y(b), z(c) {}
    static V3fi exact(V3 v) { return V3fi(Interval::exact(v.x), Interval::exact(v.y), Interval::exact(v.z)); }
    static V3fi with_err(V3 v, V3 e) {
        return V3fi(Interval::with_err(v.x, e.x), Interval::with_err(v.y, e.y), Interval::with_err(v.z, e.z));
    }
    V3 midpoints() const { return V3(x.midpoint(), y.midpoint(), z.midpoint()); }
    V3 error() const { return V3(x.width() / 2.0f, y.width() / 2.0f, z.width() / 2.0f); }   // vec.rs:1194-1200
    bool is_exact() const { return x.width() == 0.0f && y.width() == 0.0f && z.width() == 0.0f; }
};

// ------------------------------------------------------------------ 4x4 transform (row-major m[r][c])
struct M4 { float m[4][4]; };
struct Xform { M4 m, inv; };

// src/core/transform.rs:320-335 (B11: divides only if w == 0, i.e. never normalises)
static inline V3 xf_point(const M4& t, V3 p) {
    const auto& m = t.m;
    float x = p.x * m[0][0] + p.y * m[0][1] + p.z * m[0][2] + m[0][3];
    float y = p.x * m[1][0] + p.y * m[1][1] + p.z * m[1][2] + m[1][3];
    float z = p.x * m[2][0] + p.y * m[2][1] + p.z * m[2][2] + m[2][3];
    float w = p.x * m[3][0] + p.y * m[3][1] + p.z * m[3][2] + m[3][3];
    if (w == 0.0f) { x /= w; y /= w; z /= w; }
    return V3(x, y, z);
}
// src/core/transform.rs:414-422
static inline V3 xf_vec(const M4& t, V3 v) {
    const auto& m = t.m;
    return V3(v.x * m[0][0] + v.y * m[0][1] + v.z * m[0][2],
              v.x * m[1][0] + v.y * m[1][1] + v.z * m[1][2],
              v.x * m[2][0] + v.y * m[2][1] + v.z * m[2][2]);
}
// src/core/transform.rs:436-443 -- uses the INVERSE matrix, transposed
static inline V3 xf_normal(const M4& tinv, V3 n) {
    const auto& mi = tinv.m;
    return V3(n.x * mi[0][0] + n.y * mi[1][0] + n.z * mi[2][0],
              n.x * mi[0][1] + n.y * mi[1][1] + n.z * mi[2][1],
              n.x * mi[0][2] + n.y * mi[1][2] + n.z * mi[2][2]);
}
// src/core/transform.rs:338-411
static inline V3fi xf_point_fi(const M4& t, const V3fi& p) {
    const auto& m = t.m;
    float x = p.x.midpoint(), y = p.y.midpoint(), z = p.z.midpoint();
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    float wp = (m[3][0] * x + m[3][1] * y) + (m[3][2] * z + m[3][3]);
    float g3 = gamma_n(3);
    V3 err;
    if (p.is_exact()) {
        err.x = g3 * (std::fabs(m[0][0] * x) + std::fabs(m[0][1] * y) + std::fabs(m[0][2] * z) + std::fabs(m[0][3]));
        err.y = g3 * (std::fabs(m[1][0] * x) + std::fabs(m[1][1] * y) + std::fabs(m[1][2] * z) + std::fabs(m[1][3]));
        err.z = g3 * (std::fabs(m[2][0] * x) + std::fabs(m[2][1] * y) + std::fabs(m[2][2] * z) + std::fabs(m[2][3]));
    } else {
        V3 pe = p.error();
        err.x = (g3 + 1.0f) * (std::fabs(m[0][0]) * pe.x + std::fabs(m[0][1]) * pe.y + std::fabs(m[0][2]) * pe.z) +
                g3 * (std::fabs(m[0][0] * x) + std::fabs(m[0][1] * y) + std::fabs(m[0][2] * z) + std::fabs(m[0][3]));
        err.y = (g3 + 1.0f) * (std::fabs(m[1][0]) * pe.x + std::fabs(m[1][1]) * pe.y + std::fabs(m[1][2]) * pe.z) +
                g3 * (std::fabs(m[1][0] * x) + std::fabs(m[1][1] * y) + std::fabs(m[1][2] * z) + std::fabs(m[1][3]));
        err.z = (g3 + 1.0f) * (std::fabs(m[2][0]) * pe.x + std::fabs(m[2][1]) * pe.y + std::fabs(m[2][2]) * pe.z) +
                g3 * (std::fabs(m[2][0] * x) + std::fabs(m[2][1] * y) + std::fabs(m[2][2] * z) + std::fabs(m[2][3]));
    }
    V3fi r = V3fi::with_err(V3(xp, yp, zp), err);
    if (wp == 1.0f) return r;
    return V3fi(r.x / wp, r.y / wp, r.z / wp);
}
// src/core/transform.rs:425-433
static inline V3fi xf_vec_fi(const M4& t, const V3fi& v) {
    const auto& m = t.m;
    return V3fi(v.x * m[0][0] + v.y * m[0][1] + v.z * m[0][2],
                v.x * m[1][0] + v.y * m[1][1] + v.z * m[1][2],
                v.x * m[2][0] + v.y * m[2][1] + v.z * m[2][2]);
}
// src/core/transform.rs:289-297
static inline bool swaps_handedness(const M4& t) {
    const auto& m = t.m;
    float det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
                m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    return det < 0.0f;
}

// ------------------------------------------------------------------ Bounds3f
struct B3 {
    V3 lo, hi;
    // src/core/bounds.rs:332-339
    static B3 empty() {
        float mx = std::numeric_limits<float>::max();
        B3 b; b.lo = V3(mx, mx, mx); b.hi = V3(-mx, -mx, -mx); return b;
    }
};
static inline B3 b3_from_points(V3 a, V3 b) {   // bounds.rs:345-350
    B3 r;
    r.lo = V3(rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z));
    r.hi = V3(rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z));
    return r;
}
static inline B3 b3_union(B3 a, B3 b) {         // bounds.rs:536-549
    B3 r;
    r.lo = V3(rmin(a.lo.x, b.lo.x), rmin(a.lo.y, b.lo.y), rmin(a.lo.z, b.lo.z));
    r.hi = V3(rmax(a.hi.x, b.hi.x), rmax(a.hi.y, b.hi.y), rmax(a.hi.z, b.hi.z));
    return r;
}
static inline B3 b3_union_point(B3 a, V3 p) {   // bounds.rs:555-568
    B3 r;
    r.lo = V3(rmin(a.lo.x, p.x), rmin(a.lo.y, p.y), rmin(a.lo.z, p.z));
    r.hi = V3(rmax(a.hi.x, p.x), rmax(a.hi.y, p.y), rmax(a.hi.z, p.z));
    return r;
}
static inline V3 b3_centroid(B3 b) { return 0.5f * b.lo + 0.5f * b.hi; }     // bounds.rs:600-602
static inline float b3_surface_area(B3 b) {                                   // bounds.rs:448-452
    V3 d = b.hi - b.lo;
    return (d.x * d.y + d.x * d.z + d.y * d.z) * 2.0f;
}
static inline int b3_max_extent(B3 b) {                                       // bounds.rs:461-470
    V3 d = b.hi - b.lo;
    if (d.x > d.y && d.x > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}
static inline bool b3_contains(B3 b, V3 p) {
    return p.x >= b.lo.x && p.x <= b.hi.x && p.y >= b.lo.y && p.y <= b.hi.y && p.z >= b.lo.z && p.z <= b.hi.z;
}
// src/core/bounds.rs:588-597
static inline void b3_bounding_sphere(B3 b, V3* c, float* r) {
    *c = (b.lo + b.hi) / 2.0f;
    *r = b3_contains(b, *c) ? distance(*c, b.hi) : 0.0f;
}
// src/core/bounds.rs:381-404 -- the slab test the aggregate uses (B8: 3 divides per node)
static inline bool b3_intersect_p(const B3& b, V3 o, V3 d, float t_max) {
    float t0 = 0.0f, t1 = t_max;
    for (int i = 0; i < 3; ++i) {
        float inv = 1.0f / d[i];
        float tmn = (b.lo[i] - o[i]) * inv;
        float tmx = (b.hi[i] - o[i]) * inv;
        float t_near = rmin(tmn, tmx);
        float t_far = rmax(tmn, tmx);
        t_far *= 1.0f + 2.0f * gamma_n(3);
        t0 = t_near > t0 ? t_near : t0;
        t1 = t_far < t1 ? t_far : t1;
        if (t0 > t1) return false;
    }
    return true;
}
// src/core/transform.rs:532-566
static inline B3 xf_bounds(const M4& t, B3 b) {
    const auto& m = t.m;
    B3 r; r.lo = V3(m[0][3], m[1][3], m[2][3]); r.hi = r.lo;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float a = m[i][j] * b.lo[j];
            float c = m[i][j] * b.hi[j];
            r.lo.at(i) += rmin(a, c);
            r.hi.at(i) += rmax(a, c);
        }
    return r;
}

// src/core/ray.rs:47-66
static inline V3 offset_ray_origin(const V3fi& pi, V3 n, V3 w) {
    float d = dot(vabs(n), pi.error());
    V3 off = d * n;
    if (dot(n, w) < 0.0f) off = -off;
    V3 po = pi.midpoints() + off;
    for (int i = 0; i < 3; ++i) {
        if (off[i] > 0.0f) po.at(i) = next_float_up(po[i]);
        else if (off[i] < 0.0f) po.at(i) = next_float_down(po[i]);
    }
    return po;
}

}  // namespace ora
