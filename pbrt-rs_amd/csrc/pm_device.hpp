// Device-side arithmetic and BVH traversal for gfx950 (wave64).  Compiled with -ffp-contract=off:
// every leaf test below must round exactly like the reference's Rust (which never fuses a*b+c and
// computes cross products in f64), so that hit ids / t / (u,v) are bit-identical to the CPU path.
//
// Reference arithmetic restated here:
//   ray/patch      shapes/bilinear_patch.rs:506-612, math.rs:42-87, core/square_matrix.rs:123-138
//   ray/sphere     shapes/sphere.rs:265-368, core/interval.rs, core/transform.rs:338-433
//   ray/box        core/bounds.rs:381-404 (same widening; applied to the de-quantised child boxes)
#pragma once
#include <hip/hip_runtime.h>

#include "pm_layout.h"

namespace pmd {

#define PMD __device__ __forceinline__
// ... also callable on the host: the few vector helpers with which pm_render.hip precomputes per-light constants in the very
// arithmetic the kernels (and the reference) use - IEEE operations only, contraction off on both sides
#define PMHD __host__ __device__ __forceinline__

struct f3 { float x, y, z; };
PMHD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PMHD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PMHD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PMHD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PMHD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PMHD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PMHD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PMHD f3 cross3(f3 a, f3 b) {   // f64 products, one narrowing (core/vec.rs:700-712)
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return mk3((float)(ay * bz - az * by), (float)(az * bx - ax * bz), (float)(ax * by - ay * bx));
}
PMD f3 lerp3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
PMD f3 abs3(f3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
PMD float maxc3(f3 a) { float m = a.x; if (!(a.y < m)) m = a.y; if (!(a.z < m)) m = a.z; return m; }
PMHD float len3(f3 a) { return sqrtf(dot3(a, a)); }
PMHD f3 norm3(f3 a) { return a / len3(a); }

PMD float fup(float v) {      // next_float_up, core/float.rs:14-36
    if ((isinf(v) && v > 0.0f) || isnan(v)) return v;
    if (v == 0.0f) v = 0.0f;
    uint32_t b = __float_as_uint(v);
    return __uint_as_float((b >> 31) ? b - 1 : b + 1);
}
PMD float fdn(float v) {      // next_float_down, core/float.rs:40-60
    if ((isinf(v) && v < 0.0f) || isnan(v)) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t b = __float_as_uint(v);
    return __uint_as_float((b >> 31) ? b + 1 : b - 1);
}
#define PMD_EPS 5.9604644775390625e-08f
PMD float gamma_n(int n) { float nf = (float)n; return (PMD_EPS * nf) / (1.0f - nf * PMD_EPS); }

PMD float dop(float a, float b, float c, float d) {   // math.rs:78-87
    float cd = c * d;
    float r = __fmaf_rn(a, b, -cd);
    float e = __fmaf_rn(-c, d, cd);
    return r + e;
}
PMD float det3(float m00, float m01, float m02, float m10, float m11, float m12, float m20, float m21, float m22) {
    float minor12 = dop(m11, m22, m12, m21);
    float minor02 = dop(m10, m22, m12, m20);
    float minor01 = dop(m10, m21, m11, m20);
    return __fmaf_rn(m02, minor01, dop(m00, minor12, m01, minor02));
}

// ---------------------------------------------------------------------------------- patch / triangle
PMD bool patch_root(f3 ro, f3 rd, f3 p00, f3 p10, f3 p01, f3 p11, float uu, float* vnum, float* tnum, float* p2o) {
    f3 uo = lerp3(p00, p10, uu);
    f3 ud = lerp3(p01, p11, uu) - uo;
    f3 dl = uo - ro;
    f3 perp = cross3(rd, ud);
    *p2o = dot3(perp, perp);
    *vnum = det3(dl.x, rd.x, perp.x, dl.y, rd.y, perp.y, dl.z, rd.z, perp.z);
    *tnum = det3(dl.x, ud.x, perp.x, dl.y, ud.y, perp.y, dl.z, ud.z, perp.z);
    return true;
}
// returns true and (t,u,v) iff the reference's intersect_bilinear_patch returns Some for this t_max
PMD bool intersect_patch(f3 ro, f3 rd, float t_max, f3 p00, f3 p10, f3 p01, f3 p11, float* to, float* uo, float* vo) {
    float a = dot3(cross3(p10 - p00, p01 - p11), rd);
    float c = dot3(cross3(p00 - ro, rd), p01 - p00);
    float b = dot3(cross3(p10 - ro, rd), p11 - p10) - (a + c);
    float u1, u2;
    if (a == 0.0f) {
        if (b == 0.0f) return false;   // (all-zero coefficients assert in the reference; a miss here)
        u1 = -c / b; u2 = u1;
    } else {
        float disc = dop(b, b, 4.0f * a, c);
        if (disc < 0.0f) return false;
        float rt = sqrtf(disc);
        float q = -0.5f * (b + copysignf(rt, b));
        float r0 = q / a, r1 = c / q;
        if (r0 <= r1) { u1 = r0; u2 = r1; } else { u1 = r1; u2 = r0; }
    }
    float eps = gamma_n(10) * (maxc3(abs3(ro)) + maxc3(abs3(rd)) + maxc3(abs3(p00)) + maxc3(abs3(p10)) +
                               maxc3(abs3(p01)) + maxc3(abs3(p11)));
    float t = t_max, u = 0.0f, v = 0.0f;
    if (u1 >= 0.0f && u1 <= 1.0f) {
        float vn, tn, p2;
        patch_root(ro, rd, p00, p10, p01, p11, u1, &vn, &tn, &p2);
        if (tn > p2 * eps && (vn >= 0.0f && vn <= p2)) { u = u1; v = vn / p2; t = tn / p2; }
    }
    if (u2 >= 0.0f && u2 <= 1.0f && u2 != u1) {
        float vn, tn, p2;
        patch_root(ro, rd, p00, p10, p01, p11, u2, &vn, &tn, &p2);
        float t2 = tn / p2;               // B10: divided before the epsilon compare
        if (t2 > p2 * eps && (vn >= 0.0f && vn <= p2) && t2 < t) { u = u2; v = vn / p2; t = t2; }
    }
    if (t < t_max) { *to = t; *uo = u; *vo = v; return true; }
    return false;
}

// ---------------------------------------------------------------------------------- intervals / sphere
struct Iv { float lo, hi; };
PMD Iv iv(float l, float h) { Iv r; r.lo = l; r.hi = h; return r; }
PMD Iv iv_exact(float v) { return iv(v, v); }
PMD Iv iv_err(float v, float e) { return e == 0.0f ? iv(v, v) : iv(fdn(v - e), fup(v + e)); }
PMD float iv_mid(Iv a) { return (a.lo + a.hi) / 2.0f; }
PMD Iv operator+(Iv a, Iv b) { return iv(fdn(a.lo + b.lo), fup(a.hi + b.hi)); }
PMD Iv operator-(Iv a, Iv b) { return iv(fdn(a.lo - b.hi), fup(a.hi - b.lo)); }
PMD Iv operator*(Iv a, Iv b) {
    float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return iv(fdn(fminf(fminf(fminf(p0, p1), p2), p3)), fup(fmaxf(fmaxf(fmaxf(p0, p1), p2), p3)));
}
PMD Iv operator*(Iv a, float f) { return f > 0.0f ? iv(fdn(a.lo * f), fup(a.hi * f)) : iv(fdn(a.hi * f), fup(a.lo * f)); }
PMD Iv operator*(float f, Iv a) { return f > 0.0f ? iv(fdn(f * a.lo), fup(f * a.hi)) : iv(fdn(f * a.hi), fup(f * a.lo)); }
PMD Iv operator/(Iv a, Iv b) {
    if (b.lo < 0.0f && b.hi > 0.0f) return iv(-INFINITY, INFINITY);
    float q0 = a.lo / b.lo, q1 = a.lo / b.hi, q2 = a.hi / b.lo, q3 = a.hi / b.hi;
    return iv(fdn(fminf(fminf(fminf(q0, q1), q2), q3)), fup(fmaxf(fmaxf(fmaxf(q0, q1), q2), q3)));
}
PMD Iv iv_sq(Iv a) {
    if (a.lo >= 0.0f) return iv(fdn(a.lo * a.lo), fup(a.hi * a.hi));
    if (a.hi <= 0.0f) return iv(fdn(a.hi * a.hi), fup(a.lo * a.lo));
    float m = fmaxf(fabsf(a.lo), fabsf(a.hi));
    return iv(0.0f, fup(m * m));
}
PMD Iv iv_sqrt(Iv a) { return iv(fmaxf(fdn(sqrtf(a.lo)), 0.0f), fup(sqrtf(a.hi))); }

struct SphereHit { float t; f3 p_obj; float phi; };

// Transform * Point3fi for an exact input point and an affine matrix (rows 0..2 in m[12]).
PMD void xf_point_exact(const float* m, f3 p, Iv* ox, Iv* oy, Iv* oz) {
    float g3 = gamma_n(3);
    float xp = (m[0] * p.x + m[1] * p.y) + (m[2] * p.z + m[3]);
    float yp = (m[4] * p.x + m[5] * p.y) + (m[6] * p.z + m[7]);
    float zp = (m[8] * p.x + m[9] * p.y) + (m[10] * p.z + m[11]);
    float ex = g3 * (fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]));
    float ey = g3 * (fabsf(m[4] * p.x) + fabsf(m[5] * p.y) + fabsf(m[6] * p.z) + fabsf(m[7]));
    float ez = g3 * (fabsf(m[8] * p.x) + fabsf(m[9] * p.y) + fabsf(m[10] * p.z) + fabsf(m[11]));
    *ox = iv_err(xp, ex); *oy = iv_err(yp, ey); *oz = iv_err(zp, ez);
}
PMD void xf_vec_exact(const float* m, f3 v, Iv* ox, Iv* oy, Iv* oz) {
    Iv x = iv_exact(v.x), y = iv_exact(v.y), z = iv_exact(v.z);
    *ox = x * m[0] + y * m[1] + z * m[2];
    *oy = x * m[4] + y * m[5] + z * m[6];
    *oz = x * m[8] + y * m[9] + z * m[10];
}

PMD void sphere_point_phi(f3 om, f3 dm, float tm, float radius, f3* p, float* phi) {
    f3 ph = om + dm * tm;
    ph = ph * (radius / len3(ph));
    if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * radius;
    float f = (float)atan2((double)ph.y, (double)ph.x);     // f64, rounded once: see pm_shade.hpp on transcendentals
    if (f <= 0.0f) f += 2.0f * 3.14159265358979323846f;
    *p = ph; *phi = f;
}

PMD bool intersect_sphere(f3 ro, f3 rd, float t_max, const float* ofr, float radius, float z_min, float z_max,
                          float phi_max, SphereHit* out) {
    Iv ox, oy, oz, dx, dy, dz;
    xf_point_exact(ofr, ro, &ox, &oy, &oz);
    xf_vec_exact(ofr, rd, &dx, &dy, &dz);
    Iv a = iv_sq(dx) + iv_sq(dy) + iv_sq(dz);
    Iv b = 2.0f * (dx * ox + dy * oy + dz * oz);
    Iv c = iv_sq(ox) + iv_sq(oy) + iv_sq(oz) - iv_sq(iv_exact(radius));
    Iv f = b / (2.0f * a);
    Iv vx = ox - dx * f, vy = oy - dy * f, vz = oz - dz * f;
    Iv len = iv_sqrt(iv_sq(vx) + iv_sq(vy) + iv_sq(vz));
    Iv disc = 4.0f * a * (iv_exact(radius) + len) * (iv_exact(radius) - len);
    if (disc.lo < 0.0f) return false;
    Iv rdisc = iv_sqrt(disc);
    Iv q = (iv_mid(b) < 0.0f) ? -0.5f * (b - rdisc) : -0.5f * (b + rdisc);
    Iv t0 = q / a, t1 = c / q;
    if (t0.lo > t1.lo) { Iv tmp = t0; t0 = t1; t1 = tmp; }
    if (t0.hi > t_max || t1.lo <= 0.0f) return false;
    Iv ts;
    bool is_t1 = false;
    if (t0.lo > 0.0f) ts = t0;
    else if (t1.hi <= t_max) { ts = t1; is_t1 = true; }
    else return false;
    f3 om = mk3(iv_mid(ox), iv_mid(oy), iv_mid(oz));
    f3 dm = mk3(iv_mid(dx), iv_mid(dy), iv_mid(dz));
    f3 ph; float phi;
    sphere_point_phi(om, dm, iv_mid(ts), radius, &ph, &phi);
    bool clip = (z_min > -radius && ph.z < z_min) || (z_max < radius && ph.z > z_max) || phi > phi_max;
    if (clip) {
        // sphere.rs:345 compares the intervals (t_shape_hit == t1), not which branch produced them
        if (is_t1 || (ts.lo == t1.lo && ts.hi == t1.hi) || t1.hi > t_max) return false;
        ts = t1;
        sphere_point_phi(om, dm, iv_mid(ts), radius, &ph, &phi);
        clip = (z_min > -radius && ph.z < z_min) || (z_max < radius && ph.z > z_max) || phi > phi_max;
        if (clip) return false;
    }
    out->t = iv_mid(ts); out->p_obj = ph; out->phi = phi;
    return true;
}

// ---------------------------------------------------------------------------------- traversal
struct Ray { f3 o, d; float tmax; };
struct HitRec { uint32_t prim_id; float t, u, v; };

PMD uint32_t perm8(uint32_t m, uint32_t oct) {   // bit k of result = bit (k ^ oct) of m
    if (oct & 1) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (oct & 2) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (oct & 4) m = ((m & 0x0Fu) << 4) | ((m >> 4) & 0x0Fu);
    return m;
}
PMD float ubyte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }

struct TravCounters { uint32_t nodes, prims, exact, overflow, fallback; };

// The reference's patch test specialised to the degenerate patch (v0, v1, v2, v2) that a triangle is: the quadratic
// coefficient a = dot(cross(p10-p00, p01-p11), d) is (+-)0 for p11 == p01, so there is one root u = -c/b.  Same
// operations in the same order as intersect_patch => same bits.
PMD bool intersect_tri_exact(f3 ro, f3 rd, float t_max, f3 p00, f3 p10, f3 p01, float* to, float* uo, float* vo) {
    float c = dot3(cross3(p00 - ro, rd), p01 - p00);
    float b = dot3(cross3(p10 - ro, rd), p01 - p10) - c;
    if (b == 0.0f) return false;
    const float u1 = -c / b;
    float eps = gamma_n(10) * (maxc3(abs3(ro)) + maxc3(abs3(rd)) + maxc3(abs3(p00)) + maxc3(abs3(p10)) +
                               maxc3(abs3(p01)) + maxc3(abs3(p01)));
    float t = t_max, u = 0.0f, v = 0.0f;
    if (u1 >= 0.0f && u1 <= 1.0f) {
        float vn, tn, p2;
        patch_root(ro, rd, p00, p10, p01, p01, u1, &vn, &tn, &p2);
        if (tn > p2 * eps && (vn >= 0.0f && vn <= p2)) { u = u1; v = vn / p2; t = tn / p2; }
    }
    if (t < t_max) { *to = t; *uo = u; *vo = v; return true; }
    return false;
}

// ---- float pre-filter for triangles (Moeller-Trumbore with forward error margins; fused multiply-adds are fine here:
// only the classification matters, and the CPU walker in oracle/wide_walk.h repeats the same operations).
//   0  the exact test cannot accept this triangle with t <= tcull     -> skipped
//   1  undecided                                                      -> exact test later
//   2  the exact test accepts it, with t <= *t_hi                     -> exact test later, *t_hi may cull now
#define PM_PRE_K 7.62939453125e-06f      /* 128 * 2^-24 */
PMD f3 fcross(f3 a, f3 b) {
    return mk3(__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x)));
}
PMD float fdot(f3 a, f3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
PMD int tri_classify(float det, float uu, float vv, float tt, float L, float S, float R, float D, float tcull, float* t_hi,
                     float* t_lo);
PMD int tri_prefilter(f3 ro, f3 rd, float R, float D, f3 v0, f3 v1, f3 v2, float tcull, float* t_hi, float* t_lo) {
    const f3 e1 = v1 - v0, e2 = v2 - v0, sv = ro - v0;
    const f3 pv = fcross(rd, e2);
    const float det = fdot(e1, pv), uu = fdot(sv, pv);
    const f3 qv = fcross(sv, e1);
    const float vv = fdot(rd, qv), tt = fdot(e2, qv);
    const float L = fmaxf(maxc3(abs3(e1)), maxc3(abs3(e2))), S = maxc3(abs3(sv));
    return tri_classify(det, uu, vv, tt, L, S, R, D, tcull, t_hi, t_lo);
}
// The flat sweep's version, from a flat-table entry (pm_layout.h PmFlatTri: e1, e2, n = e1 x e2, L, c1, kll precomputed):
//   uu = sv.(rd x e2) = e2.(sv x rd)   vv = rd.(sv x e1) = -e1.(sv x rd)   det = e1.(rd x e2) = -rd.n   tt = e2.(sv x e1) = sv.n
// 21 multiply-adds instead of 33; the magnitudes of the terms (hence the error bounds) are those of tri_prefilter, with
// S = max|ro - v0| bounded by R + max|v0| (so Sx = 2R + c1: one add).  No division: the culling distance is kept as a
// fraction cn / cd (cd > 0), "t beyond tcull" and "t_hi below tcull" are cross-multiplied, and a certain hit replaces the
// fraction by its own ((tt + ttau) * 1.00001, det - tb).  The candidate's lower bound is the fraction (ln, ld).
struct FlatCull { float cn, cd; };
struct FlatTerms { float uu, vv, det, tt, tb, ttau, dp; };
// the two halves of tri_prefilter_flat: "can this triangle be skipped" (the sweep's inner loop), and the rest for the few
// that cannot (1 or 2 as below)
// (flags: PM_FLAT_SAME_V0 / _EDGE / _N of the table entry, wave-uniform; *sh: what the previous entry left - pm_layout.h)
struct FlatShared { f3 sv, m; float e2m, sg, det, tt; };
PMD bool tri_reject_flat(f3 ro, f3 rd, float KD, float R2, f3 v0, f3 e1, f3 e2, f3 n, float L, float c1, float kll,
                         uint32_t flags, FlatShared* sh, const FlatCull& cull, FlatTerms* o) {
    if (!(flags & PM_FLAT_SAME_V0)) { sh->sv = ro - v0; sh->m = fcross(sh->sv, rd); }
    float e1m;
    if (flags & PM_FLAT_SAME_EDGE) e1m = sh->e2m; else e1m = fdot(e1, sh->m);
    sh->e2m = fdot(e2, sh->m);
    // back-facing: all four change sign.  As products with +-1 (exact): no divergent region in the sweep's inner loop, and the
    // compiler knows the results canonical (after a masked negation it re-canonicalises them in front of its min / max)
    if (!(flags & PM_FLAT_SAME_N)) {
        const float d0 = fdot(rd, n);
        sh->sg = d0 > 0.0f ? -1.0f : 1.0f;
        sh->det = -d0 * sh->sg;
        sh->tt = fdot(sh->sv, n) * sh->sg;
    }
    const float sg = sh->sg;
    o->uu = sh->e2m * sg; o->vv = e1m * -sg; o->tt = sh->tt;
    o->det = sh->det;
    const float Sx = R2 + c1;
    o->tb = (KD * L) * Sx;
    o->ttau = kll * Sx;
    o->dp = o->det + o->tb;
    return (o->uu < -o->tb) | (o->vv < -o->tb) | (o->uu + o->vv > o->dp + o->tb) | (o->tt < -o->ttau) |
           ((o->tt - o->ttau) * cull.cd > cull.cn * o->dp);
}
PMD int tri_certain_flat(const FlatTerms& f, float c1, float epsb_ray, FlatCull* cull, bool lower, float* ln, float* ld) {
    *ln = 0.0f; *ld = 1.0f;
    if ((f.uu > f.tb) & (f.vv > f.tb) & (f.uu + f.vv < (f.det - f.tb) - f.tb) & (f.det > 2.0f * f.tb) & (f.tt > f.ttau)) {
        const float hn = (f.tt + f.ttau) * 1.00001f, hd = f.det - f.tb;
        const float lo = (f.tt - f.ttau) * 0.99999f;
        // epsb >= the exact test's gamma(10) * sum of max-abs: 8e-7 * (5R + D + 4(S + L)) with S <= R + max|v0|
        const float epsb = __fmaf_rn(3.2e-6f, c1, epsb_ray);
        if ((lo > epsb * f.dp) & (hn * cull->cd < cull->cn * hd)) {
            if (lower) { cull->cn = hn; cull->cd = hd; }
            *ln = lo; *ld = f.dp;
            return 2;
        }
    }
    return 1;
}
PMD int tri_prefilter_flat(f3 ro, f3 rd, float KD, float R2, f3 v0, f3 e1, f3 e2, f3 n, float L, float c1, float kll,
                           float epsb_ray, FlatCull* cull, bool lower, float* ln, float* ld) {
    FlatTerms ft;
    FlatShared sh;
    if (tri_reject_flat(ro, rd, KD, R2, v0, e1, e2, n, L, c1, kll, 0u, &sh, *cull, &ft)) return 0;
    return tri_certain_flat(ft, c1, epsb_ray, cull, lower, ln, ld);
}
PMD int tri_classify(float det, float uu, float vv, float tt, float L, float S, float R, float D, float tcull, float* t_hi,
                     float* t_lo) {
    const float sg = det < 0.0f ? -1.0f : 1.0f;         // (see tri_reject_flat)
    det *= sg; uu *= sg; vv *= sg; tt *= sg;
    // error bounds.  The pre-filter itself works on differences (errors relative to L, S), but the exact test it must agree
    // with rounds absolute coordinates (errors relative to |ro| + |p| <= 2R + S + L), so the bounds carry Sx = S + R + L.
    const float Sx = S + R + L;
    const float tb = PM_PRE_K * D * L * Sx;            // bound for det, uu, vv
    const float ttau = PM_PRE_K * L * L * Sx;          // bound for tt
    // one branch for the common outcome (bitwise |: all five comparisons are cheaper than four nested branches)
    const bool reject = (uu < -tb) | (vv < -tb) | (uu + vv > det + 2.0f * tb) | (tt < -ttau) |
                        (tt > __fmaf_rn(tcull, det + tb, ttau));
    if (reject) return 0;
    // certain hit: inside by more than the error bounds; t bracketed by [ (tt - ttau)/(det + tb), (tt + ttau)/(det - tb) ]
    *t_lo = 0.0f;
    if ((uu > tb) & (vv > tb) & (uu + vv < det - 2.0f * tb) & (det > 2.0f * tb) & (tt > ttau)) {
        const float epsb = 8e-7f * (5.0f * R + D + 4.0f * (S + L));     // >= the exact test's gamma(10) * sum of max-abs
        const float th = ((tt + ttau) / (det - tb)) * 1.00001f;
        const float tl = ((tt - ttau) / (det + tb)) * 0.99999f;
        if (tl > epsb && th < tcull) { *t_hi = th; *t_lo = tl; return 2; }
    }
    return 1;
}

// ---- float pre-filter for bilinear patches (flat sweep): false = the ray misses the padded bounding box of the four corners
// within [0, tcull].  A bilinear patch lies inside the convex hull of its corners, and the exact test (reference arithmetic,
// ~300 instructions) accepts nothing that lies outside the surface by more than its own rounding, ~1e-6 of the coordinates
// involved; the pad is 1e-4 of them, and the slab distances are compared with a relative slack of 1e-4 as well.  inv: the
// finite reciprocals of the direction (safe_rcp_dir).  What passes is deferred to the second pass like a sphere.  The CPU
// walker repeats these operations.
PMD bool patch_prefilter(f3 ro, f3 inv, float R, f3 p00, f3 p10, f3 p01, f3 p11, float tcull) {
    const f3 lo = mk3(fminf(fminf(p00.x, p10.x), fminf(p01.x, p11.x)), fminf(fminf(p00.y, p10.y), fminf(p01.y, p11.y)),
                      fminf(fminf(p00.z, p10.z), fminf(p01.z, p11.z)));
    const f3 hi = mk3(fmaxf(fmaxf(p00.x, p10.x), fmaxf(p01.x, p11.x)), fmaxf(fmaxf(p00.y, p10.y), fmaxf(p01.y, p11.y)),
                      fmaxf(fmaxf(p00.z, p10.z), fmaxf(p01.z, p11.z)));
    const float pad = 1e-4f * fmaxf(fmaxf(maxc3(abs3(lo)), maxc3(abs3(hi))), R);
    const float ax = ((lo.x - pad) - ro.x) * inv.x, bx = ((hi.x + pad) - ro.x) * inv.x;
    const float ay = ((lo.y - pad) - ro.y) * inv.y, by = ((hi.y + pad) - ro.y) * inv.y;
    const float az = ((lo.z - pad) - ro.z) * inv.z, bz = ((hi.z + pad) - ro.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tcull));
    return tn <= tf * 1.0001f;
}

// ---- float pre-filter for spheres: false = the exact (interval) test cannot accept this sphere with t <= tcull.
// The interval test costs ~1500 instructions and a wave pays for it as soon as ONE lane runs it, so spheres that pass are
// not tested inside the walk: the (ray, sphere) pair goes to a global list and a second, dense kernel runs the exact
// tests (deferred_fix below).  Plain f32 in object space with forward error bounds; m_norm / t_norm (from the record) bound
// the rounding of the transform exactly as Transform * Point3fi does (gamma(3) * sum of |terms|, core/transform.rs).  The
// interval test itself rejects whenever its discriminant interval reaches below zero, so rejecting where the real
// discriminant is negative by more than the bound can never remove a hit.  The CPU walker repeats these operations.
#define PM_SPH_EPS 4.76837158203125e-07f      /* 8 * 2^-24 */
PMD bool sphere_prefilter(f3 ro, f3 rd, float R, float D, const float* m, float radius, float m_norm, float t_norm, float tcull) {
    const f3 o = mk3(__fmaf_rn(m[0], ro.x, __fmaf_rn(m[1], ro.y, __fmaf_rn(m[2], ro.z, m[3]))),
                     __fmaf_rn(m[4], ro.x, __fmaf_rn(m[5], ro.y, __fmaf_rn(m[6], ro.z, m[7]))),
                     __fmaf_rn(m[8], ro.x, __fmaf_rn(m[9], ro.y, __fmaf_rn(m[10], ro.z, m[11]))));
    const f3 d = mk3(__fmaf_rn(m[0], rd.x, __fmaf_rn(m[1], rd.y, m[2] * rd.z)),
                     __fmaf_rn(m[4], rd.x, __fmaf_rn(m[5], rd.y, m[6] * rd.z)),
                     __fmaf_rn(m[8], rd.x, __fmaf_rn(m[9], rd.y, m[10] * rd.z)));
    const float a = fdot(d, d), b = fdot(o, d), oo = fdot(o, o), rr = radius * radius;
    const float c = oo - rr;
    const float Om = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)), Dm = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float eo = PM_SPH_EPS * __fmaf_rn(m_norm, R, t_norm);       // per-component error of o
    const float ed = PM_SPH_EPS * (m_norm * D);                       // ... of d
    const float db = 3.0f * __fmaf_rn(Om, ed, Dm * eo);
    const float dc = 6.0f * (Om * eo), da = 6.0f * (Dm * ed);
    const float base = PM_SPH_EPS * __fmaf_rn(a, oo + rr, b * b);
    const float disc = __fmaf_rn(b, b, -(a * c));
    const float E = 4.0f * (__fmaf_rn(2.0f * fabsf(b), db, __fmaf_rn(a, dc, fabsf(c) * da)) + base);
    if (disc < -E) return false;
    const float cl = c - 2.0f * __fmaf_rn(PM_SPH_EPS, oo + rr, dc);   // origin outside the sphere by more than the bound
    if (cl > 0.0f) {
        if (b > db) return false;                                     // moving away: both roots are negative
        const float den = (db - b) + sqrtf(fmaxf(disc, 0.0f) + E) * 1.000001f;
        if (den > 0.0f && (cl / den) * 0.99999f > tcull) return false;   // the near root lies beyond tcull
    }
    return true;
}

// Appending to the deferred-sphere list.  One atomic per wave per append on a single counter serialises (measured: ~5 ns
// each, as much time as the whole sweep of the C2 scene), so a wave reserves PM_DEFER_CHUNK entries at a time and fills
// them itself: a push that does not fit fills the rest of the current reservation, reserves the next chunk and puts the
// remaining lanes there, so only a wave's LAST reservation is left partly unused; finish pads that with invalid entries
// (item == ~0) that the second pass skips.  The list therefore needs room for items + waves * PM_DEFER_CHUNK entries
// (pm_ctx.h ensure_defer).  `cap` (entries) is checked at every reservation: a reservation beyond it is not written and
// sets *overflow (reported like a traversal-stack overflow: results invalid), so the list can never be overrun.  All
// members are wave-uniform: call push / pad from wave-uniform control flow only.
#define PM_DEFER_CHUNK 256u
struct DeferWriter {
    uint4* list; uint32_t* count; uint32_t next, end, cap;
    PMD void pad(uint32_t lane) {
        for (uint32_t i = next + lane; i < end; i += 64u) list[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u);
        next = end;
    }
    PMD void push(bool defer, uint32_t lane, uint32_t item, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t* overflow) {
        const unsigned long long dm = __ballot(defer);
        const uint32_t n = (uint32_t)__popcll(dm);
        if (n == 0u) return;
        const uint32_t rank = (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
        const uint32_t room = end - next;
        if (n <= room) {
            if (defer) list[next + rank] = make_uint4(item, s0, s1, s2);
            next += n;
            return;
        }
        if (defer && rank < room) list[next + rank] = make_uint4(item, s0, s1, s2);
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(count, PM_DEFER_CHUNK);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);       // a scalar register, not 64 copies
        if (base + PM_DEFER_CHUNK > cap || base + PM_DEFER_CHUNK < base) {          // no room left: drop, report
            *overflow = 1u;
            next = 0u; end = 0u;
            return;
        }
        if (defer && rank >= room) list[base + (rank - room)] = make_uint4(item, s0, s1, s2);
        next = base + (n - room); end = base + PM_DEFER_CHUNK;
    }
};

// the same for the (rare) slow rays of the flat sweep: a list of work items, 64 reserved at a time, padding = ~0
#define PM_SLOW_CHUNK 64u
struct SlowWriter {
    uint32_t* list; uint32_t* count; uint32_t next, end, cap;
    PMD void pad(uint32_t lane) {
        for (uint32_t i = next + lane; i < end; i += 64u) list[i] = 0xFFFFFFFFu;
        next = end;
    }
    PMD void push(bool slow, uint32_t lane, uint32_t item, uint32_t* overflow) {
        const unsigned long long dm = __ballot(slow);
        const uint32_t n = (uint32_t)__popcll(dm);
        if (n == 0u) return;
        const uint32_t rank = (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
        const uint32_t room = end - next;
        if (n <= room) {
            if (slow) list[next + rank] = item;
            next += n;
            return;
        }
        if (slow && rank < room) list[next + rank] = item;
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(count, PM_SLOW_CHUNK);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
        if (base + PM_SLOW_CHUNK > cap || base + PM_SLOW_CHUNK < base) { *overflow = 1u; next = 0u; end = 0u; return; }
        if (slow && rank >= room) list[base + (rank - room)] = item;
        next = base + (n - room); end = base + PM_SLOW_CHUNK;
    }
};

#define PM_CAND_MAX 4
// Deferred leaf tests (timed closest-hit walks of triangle-only scenes).  A node step finds a leaf for ~1 lane in 7, but the
// wave pays a whole pre-filter trip (~100 instructions + the record fetch) as soon as ONE lane has one, so testing leaves
// inside the step ran the leaf code at ~10 % lane utilisation (PMC: 0.53 active lanes over the whole walk).  Instead a lane
// parks the leaves it meets in a small ring (PM_PEND_MAX entries, LDS) and the wave runs ONE pre-filter trip when at least
// PM_PEND_FLUSH walking lanes have something parked (or a ring is full); lanes whose walk has ended drain their ring in the
// refill phase, where >= `refill` lanes work together.  Culling (tcull) learns of a certain hit a few steps later, i.e. a
// ray may visit a few more nodes: results do not change (the exact round decides them), but the walk of a ray now depends
// on its wave, so the COUNTING kernels (the deterministic counters of SURVEY 8d) keep testing leaves at once.
#ifndef PM_PEND_MAX
#define PM_PEND_MAX 4          /* 0 disables; power of two */
#endif
// words of the per-lane candidate array a kernel instance needs (the ring exists only where the walk parks leaves)
#ifndef PM_PEND_ANY
#define PM_PEND_ANY 1          /* park leaves in the any-hit walk too */
#endif
#ifndef PM_PEND_MIXED
#define PM_PEND_MIXED 1        /* round 3: the MIXED walk parks its triangle and patch leaves too (spheres stay in the node step) */
#endif
#define PM_CAND_WORDS(ANY, COUNT, KIND) (2 * PM_CAND_MAX + (((ANY) && !PM_PEND_ANY) || (COUNT) || ((KIND) == PM_WALK_MIXED && !PM_PEND_MIXED) ? 0 : PM_PEND_MAX))
// what a walk kernel is compiled for (template argument KIND of trace_persistent; `false` / `true` still mean TRI / MIXED)
#define PM_WALK_TRI 0        /* the scene holds triangles only */
#define PM_WALK_MIXED 1      /* spheres, or several primitive types: no parked leaves; spheres deferred to the second pass */
#define PM_WALK_PATCH 2      /* bilinear patches only (round 3): leaves are parked like triangles' and the reference-arithmetic patch
                                test runs in the shared flush trips instead of inside the node step */
#ifndef PM_PEND_FLUSH
#define PM_PEND_FLUSH 24
#endif
#ifndef PM_PEND_FLUSH_ANY
#define PM_PEND_FLUSH_ANY 16     /* any hit: an occluder ends the walk, so learning of it early is worth smaller trips (12 / 16 / 24 / 32 measured) */
#endif
#ifndef PM_POOL_EXACT
#define PM_POOL_EXACT 1        /* 0: every finished lane runs its own exact tests (one wave trip per queue slot) */
#endif
// The flat sweep meets candidates in table order, not front to back, so now and then a ray collects more than 4; a wave
// makes as many exact-test trips as its busiest lane has candidates, so those rays are NOT given a deeper queue (measured:
// depth 6 costs 14 %) but handed to the slow list.
#define PM_FLAT_CAND_MAX 4
#ifndef PM_FLAT_ANY_CHUNK
#define PM_FLAT_ANY_CHUNK 16      /* any-hit sweep: the wave may leave between chunks of this many entries (4 / 8 / 16 measured: 10.2 / 9.65 / 9.45 ms) */
#endif
#ifndef PM_FLAT_UNROLL
#define PM_FLAT_UNROLL 4       /* triangle loop of the sweep: four table entries per scalar-load wait (1 / 2 / 4 / 8 measured: closest
                                  sweep of the C2 frame 16.87 / 16.43 / 16.25 / 16.33 ms; registers unchanged) */
#endif
// (relative to D = the largest |component| of the direction, so that a ray and the same ray with its direction scaled by any
// factor walk the same way: with round 2's absolute 1e-18 a direction of LENGTH ~1e-18 - legal, the reference finds its hits
// at t ~ 1e18 - had every component replaced and missed everything; scripts/fuzz_trace.py found it)
PMD float safe_rcp_dir(float d, float D) { return fabsf(d) < 1e-18f * D ? copysignf(1e18f / D, d) : 1.0f / d; }
#define PM_NODE_MARGIN 2.384185791015625e-07f   /* 4 * 2^-24: rounding of the pre-scaled slab evaluation */
#define PM_WALK_GUARD (1u << 22)                /* per-ray bound on node steps (a corrupt buffer cannot hang a wave) */

// The interval-arithmetic sphere test + (u, v) of sphere.rs:382-385 for the record at r, as a real (not inlined) function: the
// MIXED walk reaches it only for a third sphere of one ray and in the fallback walk.
__device__ __noinline__ bool sphere_test_cold(const uint4* __restrict__ r, f3 ro, f3 rd, float t_max, float* to, float* uo, float* vo,
                                              uint32_t* ido) {
    float m[12];
    const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
    m[0] = __uint_as_float(r0.x); m[1] = __uint_as_float(r0.y); m[2] = __uint_as_float(r0.z); m[3] = __uint_as_float(r0.w);
    m[4] = __uint_as_float(r1.x); m[5] = __uint_as_float(r1.y); m[6] = __uint_as_float(r1.z); m[7] = __uint_as_float(r1.w);
    m[8] = __uint_as_float(r2.x); m[9] = __uint_as_float(r2.y); m[10] = __uint_as_float(r2.z); m[11] = __uint_as_float(r2.w);
    const float radius = __uint_as_float(r3.x), zmin = __uint_as_float(r3.y), zmax = __uint_as_float(r3.z), phimax = __uint_as_float(r3.w);
    SphereHit sh;
    if (!intersect_sphere(ro, rd, t_max, m, radius, zmin, zmax, phimax, &sh)) return false;
    const float th0 = acosf(fminf(fmaxf(zmin / radius, -1.0f), 1.0f));
    const float th1 = acosf(fminf(fmaxf(zmax / radius, -1.0f), 1.0f));
    const float th = acosf(fminf(fmaxf(sh.p_obj.z / radius, -1.0f), 1.0f));
    *to = sh.t; *uo = sh.phi / phimax; *vo = (th - th0) / (th1 - th0); *ido = r4.x;
    return true;
}

// The reference-arithmetic test of the patch record (or of the triangle record, as the degenerate patch) at r, as a real function
// for the MIXED walk (PM_MIXED_COLD_PATCH): result (t, u, v, id bits), t = INFINITY for "no hit within t_max".
__device__ __noinline__ float4 patch_test_cold(const uint4* __restrict__ r, int is_tri, f3 ro, f3 rd, float t_max) {
    const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
    const f3 p00 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
    const f3 p10 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
    const f3 p01 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
    f3 p11 = p01; uint32_t id = r2.y;
    if (!is_tri) { const uint4 r3 = r[3]; p11 = mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w)); id = r3.x; }
    float t, u, v;
    if (!intersect_patch(ro, rd, t_max, p00, p10, p01, p11, &t, &u, &v)) return make_float4(INFINITY, 0.0f, 0.0f, __uint_as_float(id));
    return make_float4(t, u, v, __uint_as_float(id));
}
#ifndef PM_MIXED_COLD_PATCH
#define PM_MIXED_COLD_PATCH 1
#endif
#ifndef PM_PATCH_COLD
#define PM_PATCH_COLD 0         /* the same for the patch-only walk (measured, scripts/ab_lib.sh) */
#endif

// Rays through the wide BVH, persistent waves.  ANY: intersect_p (is anything hit before ray.tmax); otherwise closest hit.
//
// Every wave pulls work items from one global counter: when `refill` or more of its lanes are idle, the lanes whose walk
// has ended run their exact tests together and commit, then all idle lanes fetch the next items (one atomic per refill).
// This keeps the lanes of a wave busy although walks differ a lot in length (a wave otherwise runs as long as its
// longest walk: ~3x the mean node steps on the C3 primary rays, oracle/oracle_api.cpp wave model).
//
// Triangles go through tri_prefilter; survivors are queued (PM_CAND_MAX per ray, in LDS) and get the exact
// (reference-arithmetic) test when the walk has ended instead of diverging inside it.  Culling uses `tcull`, an upper
// bound of the final closest t that only certain hits lower; exact results never feed back into the walk, so the walk of a
// ray does not depend on its wave (the counters are deterministic).  If the exact tests do not confirm what culling
// assumed (it would take an error-analysis failure; counted in fallback), the ray is walked again with immediate exact
// tests, so results are exact regardless.
//
// MIXED: the scene holds spheres and / or patches; spheres that pass sphere_prefilter are deferred to defer_list, patches
// get the bilinear test at once.  Kernels for triangle-only scenes are compiled without any of that: it costs registers,
// i.e. occupancy (and spills inside the 4-waves-per-SIMD cap).
//   fetch(i, &ray) -> bool   load work item i (false: nothing to trace for it)
//   commit(i, found, hit)    store the result of work item i
// `lds` is this lane's column of the per-block LDS stack (entry e at lds[e * stride]); `cand` this lane's column of the
// candidate queue.
template <bool ANY, bool COUNT, int KIND, class FetchFn, class CommitFn>
PMD void trace_persistent(const PmWideNode* __restrict__ nodes, const uint4* __restrict__ prims, uint32_t n_items,
                          uint32_t* __restrict__ work_counter, uint32_t refill, uint2* lds, uint32_t* cand, int stride,
                          uint4* __restrict__ defer_list, uint32_t* __restrict__ defer_count, uint32_t defer_cap,
                          TravCounters* ctr, FetchFn fetch, CommitFn commit) {
    constexpr bool MIXED = KIND == PM_WALK_MIXED, PONLY = KIND == PM_WALK_PATCH;
    const float widen = 1.0f + 2.0f * gamma_n(3);
    const uint32_t lane = threadIdx.x & 63u;
    // perm8 for every (octant, mask) as a 2 KB table in LDS: one ds_read_u8 instead of ~15 VALU instructions, twice per node
    // step (the walk is VALU-issue bound)
    __shared__ uint8_t ptab[8 * 256];
    for (uint32_t i = threadIdx.x; i < 8u * 256u; i += blockDim.x) ptab[i] = (uint8_t)perm8(i & 0xFFu, i >> 8);
    __syncthreads();
#define PM_PERM8(m) ((uint32_t)ptab[(oct << 8) | (m)])
    bool active = false;       // this lane is walking a ray
    bool finished = false;     // its walk has ended; exact tests + commit pending
    bool more = true;          // wave-uniform: the pool may still hold items
    uint32_t item = 0, steps = 0;
    Ray ray; ray.o = mk3(0.f, 0.f, 0.f); ray.d = mk3(0.f, 0.f, 1.f); ray.tmax = 0.f;
    f3 inv = mk3(0.f, 0.f, 0.f);
    bool ngx = false, ngy = false, ngz = false;
    uint32_t oct = 0;
    float R = 0.f, D = 0.f, tmax = 0.f, tcull = 0.f;
    bool found = false, exact_mode = false;
    float t_rej = INFINITY;    // (counting kernels) nearest hit of a sphere the pre-filter rejected: the confirm path, see trace_flat
    // "a certain triangle hit was claimed": closest-hit walks do not track it - such a claim always lowers tcull below
    // ray.tmax (and when an exact sphere / patch hit lowered it instead, the confirmation below holds trivially); measured
    // +6 % on C3.  Any-hit walks test it at every step, so they keep the flag (measured).
    bool sure_any = false;
#define PM_SURE (ANY ? sure_any : (tcull < ray.tmax))
    uint32_t qn = 0;
    constexpr bool DEFERL = PM_PEND_MAX > 0 && (PM_PEND_ANY || !ANY) && !COUNT && (!MIXED || PM_PEND_MIXED);
    // Pooled exact tests (closest hit, triangle-only scenes): see the refill phase.  The finished lanes' candidates are
    // dealt out over ALL lanes of the wave, one (ray, triangle) pair each.
    constexpr bool POOL = PM_POOL_EXACT && !ANY && (KIND == PM_WALK_TRI || (MIXED && DEFERL)) && PM_LDS_STACK >= 6 && PM_CAND_MAX == 4;
    __shared__ uint8_t pool_pairs[(POOL ? 4 : 1) * 64];        // per wave: (owner lane << 2 | candidate) of the pairs in flight
    uint8_t* const my_pairs = pool_pairs + (POOL ? (threadIdx.x >> 6) * 64u : 0u);
    uint2* const lds_w = lds - lane;                           // the wave's base of the strided per-lane arrays
    uint32_t* const cand_w = cand - lane;
    uint32_t np = 0, ph = 0;                              // parked leaves: count, ring head
    uint32_t sph0 = 0xFFFFFFFFu, sph1 = 0xFFFFFFFFu;      // spheres that passed the pre-filter (deferred to sphere_fix)
    int sp = 0;
    uint2 cur = make_uint2(0u, 0u);
    HitRec hit; hit.prim_id = 0xFFFFFFFFu; hit.t = 0.f; hit.u = 0.f; hit.v = 0.f;
    uint2 spill[PM_SPILL_STACK];

    auto start_walk = [&]() {
        tmax = ray.tmax; tcull = ray.tmax; sure_any = false; found = false; qn = 0; sp = 0; steps = 0; np = 0; ph = 0; t_rej = INFINITY;
        sph0 = 0xFFFFFFFFu; sph1 = 0xFFFFFFFFu;
        hit.prim_id = 0xFFFFFFFFu; hit.t = ray.tmax; hit.u = 0.0f; hit.v = 0.0f;
        cur = make_uint2(0u, (1u << 8) | perm8(1u, oct));     // pseudo group holding only the root
        active = true;
    };

    // Exact-t TIES between triangles resolve to the LOWER primitive id, not to the triangle met first.  Round 3: with parked
    // leaves the order in which a ray's candidates are tested depends on its wave (a leaf met while the ring is full jumps the
    // queue), and two triangles hit at bit-equal t - one ray in ~10^7 of the C4 soup - made ONE pixel of the 3840 x 2160 film
    // differ from run to run.  The exact test is therefore asked for t <= tmax (tie_tmax) and tri_take decides; the result no
    // longer depends on the order of the tests, on the wave, or on the tree.  (The reference's own answer to such a tie depends
    // on the order ITS tree visits the two in: the documented tie class.  The CPU mirror follows this rule.)
    auto tie_tmax = [&]() { return ANY ? tmax : fup(tmax); };
    auto tri_take = [&](float t, float u, float v, uint32_t id) -> bool {
        if (t < tmax || (!ANY && found && t == tmax && id < hit.prim_id)) {
            tmax = t; found = true; hit.prim_id = id; hit.t = t; hit.u = u; hit.v = v;
            return true;
        }
        return false;
    };
    // pre-filter of one parked triangle (DEFERL): the triangle branch of the leaf loop below, for a unit popped from the ring
    auto drain_one = [&]() {
        const uint32_t tagged = cand[(2 * PM_CAND_MAX + (ph & (PM_PEND_MAX > 0 ? PM_PEND_MAX - 1 : 0))) * stride];
        ++ph; --np;
        // (MIXED: the ring entry carries the leaf's primitive type above the 30-bit record unit)
        const uint32_t unit = MIXED ? (tagged & 0x3FFFFFFFu) : tagged;
        const bool parked_patch = PONLY || (MIXED && (tagged >> 30) == PM_PRIM_PATCH);
        const uint4* r = prims + unit;
        const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
        const f3 p00 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
        const f3 p10 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
        const f3 p01 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
        if (parked_patch) { // a parked PATCH: the reference arithmetic now, with whoever else has one parked (no float pre-filter
                            // exists for the bilinear test; its box test was the node step's - the leaf's own quantised box)
            if ((MIXED && PM_MIXED_COLD_PATCH) || (PONLY && PM_PATCH_COLD)) {
                const float4 h = patch_test_cold(r, 0, ray.o, ray.d, tie_tmax());
                if (h.x < INFINITY && tri_take(h.x, h.y, h.z, __float_as_uint(h.w))) tcull = fminf(tcull, h.x);
                return;
            }
            const uint4 r3 = r[3];
            const f3 p11 = mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
            float t, u, v;
            if (intersect_patch(ray.o, ray.d, tie_tmax(), p00, p10, p01, p11, &t, &u, &v) && tri_take(t, u, v, r3.x)) tcull = fminf(tcull, t);
            return;
        }
        float th, tl;
        const int cls = tri_prefilter(ray.o, ray.d, R, D, p00, p10, p01, tcull, &th, &tl);
        if (cls != 0) {
            if (cls == 2) { if (ANY) sure_any = true; else tcull = th; }
            if (qn < PM_CAND_MAX) {
                cand[qn * stride] = unit; cand[(PM_CAND_MAX + qn) * stride] = __float_as_uint(tl);
                ++qn;
            } else {                                  // candidate queue full: the exact test at once
                float t, u, v;
                if (MIXED && PM_MIXED_COLD_PATCH) {
                    const float4 h = patch_test_cold(r, 1, ray.o, ray.d, tie_tmax());
                    if (h.x < INFINITY) (void)tri_take(h.x, h.y, h.z, __float_as_uint(h.w));
                } else if (MIXED) {    // (the MIXED kernels reach the triangle test through the patch code they carry anyway)
                    if (intersect_patch(ray.o, ray.d, tie_tmax(), p00, p10, p01, p01, &t, &u, &v)) (void)tri_take(t, u, v, r2.y);
                } else if (intersect_tri_exact(ray.o, ray.d, tie_tmax(), p00, p10, p01, &t, &u, &v)) (void)tri_take(t, u, v, r2.y);
            }
        }
    };

    // Work items are reserved from the global counter in chunks (one atomic per chunk, not per refill) sized so that every
    // wave of the grid gets several chunks: [wnext, wend) is this wave's current reservation (wave-uniform).
    // 16 chunks per wave measured best (scripts/ab_bvh.sh: +9 % on C3 over 4; 32 is past the optimum).  All waves of the
    // chip draw from ONE counter on purpose: they then work inside one narrow window of the items, which the XCDs' L2s and the
    // Infinity Cache share.  Giving each XCD group (blockIdx % 8) its own eighth of the items was measured 5 % (C3) to 15 %
    // (C4-shaped path) slower.
    uint32_t chunk = n_items / (gridDim.x * (blockDim.x >> 6) * 16u);
    chunk = chunk < 64u ? 64u : (chunk > 1024u ? 1024u : (chunk & ~63u));
    uint32_t wnext = 0, wend = 0;
    DeferWriter dw; dw.list = defer_list; dw.count = defer_count; dw.next = 0u; dw.end = 0u; dw.cap = defer_cap;

    for (;;) {
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(!active));
        if (n_idle == 64u || ((more || wnext < wend) && n_idle >= refill)) {
            // ---- lanes whose walk has ended: exact tests of the queued triangles, in the order they were met
            bool defer = false;
            if (DEFERL) {        // lanes whose walk has ended: what is still parked goes through the pre-filter now, together
                for (int k = 0; k < PM_PEND_MAX; ++k) {
                    // (any hit: a ray that already has its occluder needs none of what it parked)
                    const bool todo = finished && np > 0u && !(ANY && (found || sure_any));
                    if (!__any(todo)) break;
                    if (todo) drain_one();
                }
            }
            if (POOL) {
                // The exact test is ~250 instructions and the loop below makes one wave trip per queue slot with only the lanes
                // that still hold a candidate in that slot active (measured: these trips ran at ~15 % lane utilisation and took
                // a fifth of the walk's instructions).  Here the (finished lane, candidate) pairs of the wave are numbered by a
                // prefix sum and dealt out over all 64 lanes: a worker lane fetches the owner's ray with ds_bpermute, runs the
                // test against the owner's tmax and leaves (t, u, v, id) in the owner's LDS slots - the lower-bound words of
                // its queue and its (now unused) stack entries; the owner then applies its results in queue order with the
                // sequential rule `t < tmax`, so ties resolve exactly as before.  The test's only use of t_max is that final
                // comparison, so testing against the tmax of the start of the round changes nothing.
                uint32_t vmask = 0u;
                if (finished)
                    for (uint32_t qj = 0; qj < PM_CAND_MAX; ++qj)
                        if (qj < qn && !(__uint_as_float(cand[(PM_CAND_MAX + qj) * stride]) > tcull)) vmask |= 1u << qj;
                const uint32_t myn = __popc(vmask);
                const unsigned long long b0 = __ballot(myn & 1u), b1 = __ballot(myn & 2u), b2 = __ballot(myn & 4u);
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t pre = (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
                const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
                for (uint32_t base = 0; base < total; base += 64u) {
                    {   // owners publish the pairs of this window
                        uint32_t m = vmask, k = 0;
                        while (m) {
                            const uint32_t j = __ffs(m) - 1u; m &= m - 1u;
                            const uint32_t pidx = pre + k - base; ++k;
                            if (pidx < 64u) my_pairs[pidx] = (uint8_t)((lane << 2) | j);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const bool work = lane < total - base;
                    const uint32_t e = work ? (uint32_t)my_pairs[lane] : (lane << 2);
                    const uint32_t owner = e >> 2, j = e & 3u;
                    // every lane takes part in the shuffles (a disabled source lane would deliver nothing)
                    const f3 wo = mk3(__shfl(ray.o.x, (int)owner), __shfl(ray.o.y, (int)owner), __shfl(ray.o.z, (int)owner));
                    const f3 wd = mk3(__shfl(ray.d.x, (int)owner), __shfl(ray.d.y, (int)owner), __shfl(ray.d.z, (int)owner));
                    const float wtmax = __shfl(fup(tmax), (int)owner);          // t <= the owner's tmax: ties are the owner's to decide
                    if (work) {
                        const uint4* r = prims + cand_w[j * stride + owner];
                        const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                        const f3 v0 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                        const f3 v1 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                        const f3 v2 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                        if (COUNT) ctr->exact++;
                        float t = INFINITY, u = 0.0f, v = 0.0f;
                        if (!intersect_tri_exact(wo, wd, wtmax, v0, v1, v2, &t, &u, &v)) t = INFINITY;
                        cand_w[(PM_CAND_MAX + j) * stride + owner] = __float_as_uint(t);
                        lds_w[j * stride + owner] = make_uint2(__float_as_uint(u), __float_as_uint(v));
                        reinterpret_cast<uint32_t*>(lds_w)[2u * ((4u + (j >> 1)) * stride + owner) + (j & 1u)] = r2.y;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                {   // owners apply their results in queue order (ties: the lower primitive id, see tri_take).  Written out as a
                    // fixed loop over the four slots: the earlier form - `while (m)` over the set bits calling tri_take - came out
                    // of hipcc with hit.u / hit.v of the slot BEFORE a tie (ROCm 7.2; tests/test_tie_gpu.py is the regression test)
#pragma unroll
                    for (uint32_t j = 0; j < PM_CAND_MAX; ++j) {
                        if (!(vmask & (1u << j))) continue;
                        const float t = __uint_as_float(cand[(PM_CAND_MAX + j) * stride]);
                        const uint2 uv = lds[j * stride];
                        const uint32_t id = reinterpret_cast<const uint32_t*>(lds)[2u * ((4u + (j >> 1)) * stride) + (j & 1u)];
                        // (INFINITY = the test rejected it: never below a finite tmax, and not taken at tmax = inf while nothing is found)
                        if (t < tmax || (found && t == tmax && id < hit.prim_id)) {
                            tmax = t; found = true; hit.prim_id = id; hit.t = t; hit.u = __uint_as_float(uv.x); hit.v = __uint_as_float(uv.y);
                        }
                    }
                }
            }
            if (finished) {
                for (uint32_t qj = 0; qj < PM_CAND_MAX; ++qj) {
                    if (POOL || qj >= qn || (ANY && found)) continue;
                    // any hit: last candidate first - a certain hit ends the walk, so it is the last one queued, and the round
                    // is over after one test (the candidates before it are mostly the triangle the shadow ray starts on)
                    const uint32_t qi = ANY ? qn - 1u - qj : qj;
                    // a certain hit whose t bracket starts beyond the final tcull lies behind another certain hit
                    if (__uint_as_float(cand[(PM_CAND_MAX + qi) * stride]) > tcull) continue;
                    const uint4* r = prims + cand[qi * stride];
                    const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                    const f3 v0 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                    const f3 v1 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                    const f3 v2 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                    if (COUNT) ctr->exact++;
                    float t, u, v;
                    if (intersect_tri_exact(ray.o, ray.d, tie_tmax(), v0, v1, v2, &t, &u, &v)) (void)tri_take(t, u, v, r2.y);
                }
                finished = false;
                // did the exact tests confirm what culling assumed?
                const bool false_reject = COUNT && MIXED && (ANY ? (!found && t_rej < INFINITY) : t_rej < (found ? hit.t : ray.tmax));
                const bool confirmed = (!PM_SURE || (found && (ANY || hit.t <= tcull))) && !false_reject;
                if (confirmed || exact_mode) {
                    // closest hit: commit what the walk found, sphere_fix improves it; any hit: an occluded ray needs no
                    // sphere test, an unoccluded one is committed by sphere_fix
                    defer = MIXED && sph0 != 0xFFFFFFFFu && !(ANY && found);
                    if (!ANY || !defer) commit(item, found, hit);
                } else { exact_mode = true; if (COUNT) ctr->fallback++; start_walk(); }
            }
            if (MIXED) dw.push(defer, lane, item, sph0, sph1, 0xFFFFFFFFu, &ctr->overflow);      // wave-uniform control flow here
            // ---- idle lanes fetch the next items
            if (wnext >= wend && more) {
                uint32_t base = 0;
                if (lane == 0u) base = atomicAdd(work_counter, chunk);
                base = __shfl(base, 0);
                if (base < n_items) { wnext = base; wend = base + chunk < n_items ? base + chunk : n_items; }
                else more = false;
            }
            if (wnext < wend) {
                const unsigned long long want = __ballot(!active);
                if (!active) {
                    const uint32_t idx = wnext + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                    if (idx < wend && fetch(idx, &ray)) {
                        item = idx;
                        // A ray with a NaN / infinite / all-zero direction or a non-finite origin hits nothing in the reference
                        // (every comparison of its primitive tests fails) but would pass every NaN-tolerant box test, i.e.
                        // visit the whole tree: answer "miss" at once.
                        const float dsum = fabsf(ray.d.x) + fabsf(ray.d.y) + fabsf(ray.d.z);
                        const float osum = fabsf(ray.o.x) + fabsf(ray.o.y) + fabsf(ray.o.z);
                        if (!(dsum > 0.0f && dsum < INFINITY && osum < INFINITY)) {
                            HitRec miss; miss.prim_id = 0xFFFFFFFFu; miss.t = ray.tmax; miss.u = 0.0f; miss.v = 0.0f;
                            commit(idx, false, miss);
                        } else {
                        // 1/d for the slab test; a (near-)zero component gets a huge FINITE reciprocal so that the pre-scaled form
                        // q*(s/d) + (o - ro)/d stays a number (with +-inf it is inf - inf = NaN = "no constraint", and an
                        // axis-parallel ray would visit every box of its axis-aligned column)
                        R = maxc3(abs3(ray.o)); D = maxc3(abs3(ray.d));
                        inv = mk3(safe_rcp_dir(ray.d.x, D), safe_rcp_dir(ray.d.y, D), safe_rcp_dir(ray.d.z, D));
                        ngx = inv.x < 0.0f; ngy = inv.y < 0.0f; ngz = inv.z < 0.0f;
                        oct = (ngx ? 1u : 0u) | (ngy ? 2u : 0u) | (ngz ? 4u : 0u);
                        exact_mode = false;
                        start_walk();
                        }
                    }
                }
                const uint32_t nw = (uint32_t)__popcll(want);
                wnext = wnext + nw < wend ? wnext + nw : wend;
            }
            if (!__any(active) && !more && wnext >= wend) { if (MIXED) dw.pad(lane); break; }
        }
#ifdef PM_DBG_UTIL
        if (!COUNT) { const unsigned long long am = __ballot(active); if (active) { ctr->nodes++; if (lane == (uint32_t)__ffsll((long long)am) - 1u) ctr->prims++; } }
#endif
        if (active) {
            // ---- one node step (the pop of an exhausted group is merged into it)
            bool ended = (ANY && (found || PM_SURE)) || ++steps > PM_WALK_GUARD;
            uint32_t pm = cur.y & 0xFFu;
            if (!ended && pm == 0u) {
                if (sp == 0) ended = true;
                else {
                    --sp;
                    // (written so that the common case is a plain ds_read: `sp < N ? lds[..] : spill[..]` compiles to ONE flat
                    // load through a selected address, which goes the slow way round and waits on every memory counter)
                    cur = lds[(sp < PM_LDS_STACK ? sp : PM_LDS_STACK - 1) * stride];
                    if (__builtin_expect(sp >= PM_LDS_STACK, 0)) cur = spill[sp - PM_LDS_STACK];
                    pm = cur.y & 0xFFu;
                }
            }
            if (ended) { active = false; finished = true; }
            else {
                const uint32_t k = __ffs(pm) - 1u;
                const uint32_t slot = k ^ oct;
                const uint32_t imask_g = cur.y >> 8;
                const uint32_t nidx = cur.x + __popc(imask_g & ((1u << slot) - 1u));
                cur.y &= ~(1u << k);
                if (cur.y & 0xFFu) {
                    if (sp < PM_LDS_STACK) lds[sp * stride] = cur;
                    else if (sp < PM_LDS_STACK + PM_SPILL_STACK) spill[sp - PM_LDS_STACK] = cur;
                    else { ctr->overflow = 1; --sp; }
                    ++sp;
                }
                // ---- fetch the 80-byte node (5 x 16 B)
                const uint4* nq = reinterpret_cast<const uint4*>(nodes + nidx);
                const uint4 n0 = nq[0], n1 = nq[1], n2 = nq[2], n3 = nq[3], n4 = nq[4];
                if (COUNT) ctr->nodes++;
                const uint32_t imask = n0.w >> 24;
                // ---- pre-scaled slab test: t = q * (scale/d) + (origin - o)/d, margins folded into the addends
                const float sxi = __uint_as_float((n0.w & 0xFFu) << 23) * inv.x;
                const float syi = __uint_as_float(((n0.w >> 8) & 0xFFu) << 23) * inv.y;
                const float szi = __uint_as_float(((n0.w >> 16) & 0xFFu) << 23) * inv.z;
                const float ax = (__uint_as_float(n0.x) - ray.o.x) * inv.x;
                const float ay = (__uint_as_float(n0.y) - ray.o.y) * inv.y;
                const float az = (__uint_as_float(n0.z) - ray.o.z) * inv.z;
                const float mx = PM_NODE_MARGIN * __fmaf_rn(256.0f, fabsf(sxi), fabsf(ax));
                const float my = PM_NODE_MARGIN * __fmaf_rn(256.0f, fabsf(syi), fabsf(ay));
                const float mz = PM_NODE_MARGIN * __fmaf_rn(256.0f, fabsf(szi), fabsf(az));
                const float anx = ax - mx, any_ = ay - my, anz = az - mz;
                const float afx = (ax + mx) * widen, afy = (ay + my) * widen, afz = (az + mz) * widen;
                const float sxf = sxi * widen, syf = syi * widen, szf = szi * widen;
                // near / far plane bytes by direction sign: lo words n2.x n2.y | n2.z n2.w | n3.x n3.y, hi words n3.z n3.w | n4.x n4.y | n4.z n4.w
                const uint32_t qnx[2] = {ngx ? n3.z : n2.x, ngx ? n3.w : n2.y}, qfx[2] = {ngx ? n2.x : n3.z, ngx ? n2.y : n3.w};
                const uint32_t qny[2] = {ngy ? n4.x : n2.z, ngy ? n4.y : n2.w}, qfy[2] = {ngy ? n2.z : n4.x, ngy ? n2.w : n4.y};
                const uint32_t qnz[2] = {ngz ? n4.z : n3.x, ngz ? n4.w : n3.y}, qfz[2] = {ngz ? n3.x : n4.z, ngz ? n3.y : n4.w};
                uint32_t hits = 0;
                // (two children per v_pk_fma_f32 - the same IEEE fma per component - was measured in round 3: 5 % SLOWER on C3,
                // 2.5 % on the C4 closest class, profiles/r03/ab_pk_slab.log; packed f32 has no higher rate on this chip)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int w = s >> 2, kk = s & 3;
                    const float tn = fmaxf(fmaxf(__fmaf_rn(ubyte(qnx[w], kk), sxi, anx), __fmaf_rn(ubyte(qny[w], kk), syi, any_)),
                                           fmaxf(__fmaf_rn(ubyte(qnz[w], kk), szi, anz), 0.0f));
                    const float tf = fminf(fminf(__fmaf_rn(ubyte(qfx[w], kk), sxf, afx), __fmaf_rn(ubyte(qfy[w], kk), syf, afy)),
                                           fminf(__fmaf_rn(ubyte(qfz[w], kk), szf, afz), tcull));
                    if (tn <= tf) hits |= 1u << s;
                }
                // (unused child slots hold an inverted box, which the slab test rejects; should one pass on a node that is flat
                // along the only axis a ray constrains, it is a leaf of zero primitives)
                // ---- leaves of this node, front to back by octant order
                uint32_t lm = PM_PERM8(hits & ~imask);
                const uint32_t ptype = n1.y >> 30;
                const uint32_t pbase = n1.y & 0x3FFFFFFFu;
                const uint32_t metaw[2] = {n1.z, n1.w};
                while (lm && !(ANY && (found || PM_SURE))) {
                    const uint32_t kl = __ffs(lm) - 1u;
                    lm &= lm - 1u;
                    const uint32_t ls = kl ^ oct;
                    const uint32_t meta = (metaw[ls >> 2] >> (8 * (ls & 3))) & 0xFFu;
                    const uint32_t cnt = meta >> 5, off = meta & 31u;
                    for (uint32_t j = 0; j < cnt && !(ANY && (found || PM_SURE)); ++j) {
                        if (COUNT) ctr->prims++;
                        if (MIXED && ptype == PM_PRIM_SPHERE) {
                            const uint4* r = prims + pbase + (off + j) * PM_UNITS_SPHERE;
                            float m[12];
                            const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
                            m[0] = __uint_as_float(r0.x); m[1] = __uint_as_float(r0.y); m[2] = __uint_as_float(r0.z); m[3] = __uint_as_float(r0.w);
                            m[4] = __uint_as_float(r1.x); m[5] = __uint_as_float(r1.y); m[6] = __uint_as_float(r1.z); m[7] = __uint_as_float(r1.w);
                            m[8] = __uint_as_float(r2.x); m[9] = __uint_as_float(r2.y); m[10] = __uint_as_float(r2.z); m[11] = __uint_as_float(r2.w);
                            const float radius = __uint_as_float(r3.x);
                            if (!exact_mode) {
                                if (!sphere_prefilter(ray.o, ray.d, R, D, m, radius, __uint_as_float(r4.z), __uint_as_float(r4.w), tcull)) {
                                    if (COUNT) {        // confirm path (see trace_flat): the interval test all the same
                                        float st, su, sv; uint32_t sid;
                                        if (sphere_test_cold(r, ray.o, ray.d, ray.tmax, &st, &su, &sv, &sid)) t_rej = fminf(t_rej, st);
                                    }
                                    continue;
                                }
                                const uint32_t unit = pbase + (off + j) * PM_UNITS_SPHERE;
                                if (sph0 == 0xFFFFFFFFu) { sph0 = unit; continue; }
                                if (sph1 == 0xFFFFFFFFu) { sph1 = unit; continue; }
                            }                       // a third sphere for the same ray is tested at once
                            if (COUNT) ctr->exact++;
                            // (rare: a third sphere of one ray, or the fallback walk.  Out of line, so that the interval
                            // arithmetic's ~90 live values are the callee's registers and not the node step's)
                            float st, su, sv; uint32_t sid;
                            if (sphere_test_cold(r, ray.o, ray.d, tmax, &st, &su, &sv, &sid)) {
                                tmax = st; tcull = fminf(tcull, st); found = true; hit.prim_id = sid; hit.t = st; hit.u = su; hit.v = sv;
                            }
                        } else {
                            // triangle record: v0 v1 v2 id ; patch record: p00 p10 p01 p11 id
                            const bool is_tri = ptype == PM_PRIM_TRI;
                            const uint32_t unit = pbase + (off + j) * (is_tri ? PM_UNITS_TRI : PM_UNITS_PATCH);
                            if (DEFERL && !exact_mode && np < PM_PEND_MAX) {        // park it (triangle-only scene)
                                cand[(2 * PM_CAND_MAX + ((ph + np) & (PM_PEND_MAX > 0 ? PM_PEND_MAX - 1 : 0))) * stride] = MIXED ? (unit | (ptype << 30)) : unit;
                                ++np;
                                continue;
                            }
                            const uint4* r = prims + unit;
                            const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                            const f3 p00 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                            const f3 p10 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                            const f3 p01 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                            bool immediate = true;
                            if (is_tri && !exact_mode) {
                                float th, tl;
                                const int cls = tri_prefilter(ray.o, ray.d, R, D, p00, p10, p01, tcull, &th, &tl);
                                if (cls == 0) immediate = false;
                                else {
                                    if (cls == 2) { if (ANY) sure_any = true; else tcull = th; }
                                    if (qn < PM_CAND_MAX) {
                                        cand[qn * stride] = unit; cand[(PM_CAND_MAX + qn) * stride] = __float_as_uint(tl);
                                        ++qn; immediate = false;
                                    }
                                }
                            }
                            if (KIND == PM_WALK_TRI && immediate) {      // (triangles only) the queue is full, or the fallback walk
                                if (COUNT) ctr->exact++;
                                float t, u, v;
                                if (intersect_tri_exact(ray.o, ray.d, tie_tmax(), p00, p10, p01, &t, &u, &v) && tri_take(t, u, v, r2.y)) {
                                    if (exact_mode) tcull = fminf(tcull, t);
                                }
                            }
                            if (KIND != PM_WALK_TRI && immediate) {    // patches; triangles when the queue is full or in the fallback walk
                                f3 p11 = p01;
                                uint32_t id = r2.y;
                                if (!is_tri) {
                                    const uint4 r3 = r[3];
                                    p11 = mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
                                    id = r3.x;
                                }
                                if (COUNT) ctr->exact++;
                                float t, u, v;
                                if (((MIXED && PM_MIXED_COLD_PATCH) || (PONLY && PM_PATCH_COLD)) && !COUNT) {
                                    const float4 h = patch_test_cold(r, is_tri ? 1 : 0, ray.o, ray.d, tie_tmax());
                                    if (h.x < INFINITY && tri_take(h.x, h.y, h.z, __float_as_uint(h.w))) {
                                        if (exact_mode || !is_tri) tcull = fminf(tcull, h.x);
                                    }
                                } else if (is_tri) {
                                    if (intersect_patch(ray.o, ray.d, tie_tmax(), p00, p10, p01, p11, &t, &u, &v) && tri_take(t, u, v, id)) {
                                        if (exact_mode) tcull = fminf(tcull, t);
                                    }
                                } else if (intersect_patch(ray.o, ray.d, tie_tmax(), p00, p10, p01, p11, &t, &u, &v) && tri_take(t, u, v, id)) {
                                    tcull = fminf(tcull, t);        // (patches follow the triangles' tie rule: a parked patch may be tested late)
                                }
                            }
                        }
                    }
                }
                // ---- descend into the interior children that were hit
                cur = make_uint2(n1.x, (imask << 8) | PM_PERM8(hits & imask));
                // (prefetching the node the next step visits - two plain loads issued here - was measured in round 3: 2.5 - 3 %
                // slower on C3 and C4, profiles/r03/ab_prefetch_next_node.log; the walk is issue-bound, not fetch-latency-bound)
            }
        }
        if (DEFERL) {       // one pre-filter trip for the walking lanes, when enough of them have a leaf parked
            const bool have = active && np > 0u;
            const unsigned long long hm = __ballot(have);
            if (hm != 0ull && ((uint32_t)__popcll(hm) >= (ANY ? PM_PEND_FLUSH_ANY : PM_PEND_FLUSH) || __any(active && np == PM_PEND_MAX))) {
#ifdef PM_DBG_UTIL
                if (have) { ctr->exact++; if (lane == (uint32_t)__ffsll((long long)hm) - 1u) ctr->fallback++; }
#endif
                if (have) drain_one();
            }
        }
    }
}
#undef PM_SURE
#undef PM_PERM8

// ---- scenes of at most PM_FLAT_MAX_PRIMS primitives: no tree.  Every wave takes 64 consecutive work items and sweeps the
// whole flat table (pm_layout.h) in lock step: the records are wave-uniform (scalar loads), nothing diverges but the
// bookkeeping, and there are no node steps.  The sweep kernel is kept LEAN - pre-filters, the exact triangle test and (only
// in the any-hit variant for scenes that hold patches) the bilinear patch test - because its speed hangs on registers
// (occupancy): every interval-arithmetic sphere test and every rare case goes to the second pass (deferred_fix), which
// runs them with all lanes busy:
//   * spheres that pass sphere_prefilter and (closest hit) patches that pass patch_prefilter are DEFERRED: up to three per
//     ray travel in the ray's list entry;
//   * a fourth deferred primitive, a fifth triangle candidate, or exact tests that do not confirm what culling assumed make
//     the ray SLOW: the second pass redoes it from scratch with exact tests only (counted in fallback).
// Triangles go through tri_prefilter_flat into the candidate queue as in trace_persistent, then the exact round.
using FlatScene = PmFlatScene;
// Read-only scene data addressed wave-uniformly: loads through the constant address space become scalar loads (one per
// wave, into SGPRs) instead of 64 identical vector loads.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const uint32_t __attribute__((address_space(4))) * cuint_p;
struct cuint4_p {
    const u32x4_t __attribute__((address_space(4))) * p;
    PMD uint4 operator[](int i) const { const u32x4_t v = p[i]; return make_uint4(v.x, v.y, v.z, v.w); }
};
PMD cuint4_p as_uniform(const uint4* p) { cuint4_p r; r.p = (const u32x4_t __attribute__((address_space(4))) *)p; return r; }
PMD cuint_p as_uniform(const uint32_t* p) { return (cuint_p)p; }

#define PM_DEFER_PATCH 0x40000000u      /* deferred unit is a patch record (otherwise a sphere record) */
#define PM_DEFER_NONE 0xFFFFFFFFu

// COH (closest hit, timed kernels; round 3): the caller knows that a wave's 64 work items are COHERENT rays - the camera rays of an
// 8 x 8 pixel tile, bounce 0 of a pass.  When they also share one origin (a pinhole camera; checked per wave) the wave first asks,
// one lane per table entry, "can ANY ray of this tile reach triangle i?" and sweeps only the entries that say yes - 2 to 6 of the
// 36 of the C2 box.  The test is perspective projection from the common origin: with all three vertices in front of the origin
// (inside the cone of 84 degrees around the tile's first direction, so the divisions are well conditioned) a ray that meets the
// triangle has its projected direction inside the bounding box of the projected vertices; the tile's own bounding box of
// projected directions is padded by 1e-4 of its size + 2e-5 absolute (rounding of either side is ~2e-6).  Anything else - a
// vertex outside the cone, rays that do not share their origin or spread beyond 60 degrees - keeps the entry.  Culled entries are
// entries the exact test rejects for every ray of the wave, so hits are unchanged bit for bit (the queues may hold fewer useless
// candidates); the counting kernels, whose counters are the CPU mirror's, never cull.
PMD float wave_min64(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
PMD float wave_max64(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

template <bool ANY, bool COUNT, bool PATCH, bool COH = false, class FetchFn, class CommitFn>
PMD void trace_flat(FlatScene fs, const uint4* __restrict__ prims, uint32_t n_items, uint32_t* cand, int stride,
                    uint4* __restrict__ defer_list, uint32_t* __restrict__ defer_count, uint32_t* __restrict__ slow_list,
                    uint32_t* __restrict__ slow_count, uint32_t defer_cap, TravCounters* ctr, FetchFn fetch, CommitFn commit) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    DeferWriter dw; dw.list = defer_list; dw.count = defer_count; dw.next = 0u; dw.end = 0u; dw.cap = defer_cap;
    SlowWriter sw; sw.list = slow_list; sw.count = slow_count; sw.next = 0u; sw.end = 0u; sw.cap = defer_cap;
    for (uint32_t base = wave * 64u; base < n_items; base += n_waves * 64u) {
        const uint32_t item = base + lane;
        Ray ray; ray.o = mk3(0.f, 0.f, 0.f); ray.d = mk3(0.f, 0.f, 1.f); ray.tmax = 0.f;
        bool live = item < n_items && fetch(item, &ray);
        if (live) {
            const float dsum = fabsf(ray.d.x) + fabsf(ray.d.y) + fabsf(ray.d.z);
            const float osum = fabsf(ray.o.x) + fabsf(ray.o.y) + fabsf(ray.o.z);
            if (!(dsum > 0.0f && dsum < INFINITY && osum < INFINITY)) {     // degenerate ray: a miss (see trace_persistent)
                HitRec miss; miss.prim_id = 0xFFFFFFFFu; miss.t = ray.tmax; miss.u = 0.0f; miss.v = 0.0f;
                commit(item, false, miss);
                live = false;
            }
        }
        const float R = maxc3(abs3(ray.o)), D = maxc3(abs3(ray.d));
        float tmax = ray.tmax;
        float tcull = ray.tmax;                     // spheres never lower it, (any hit) patches do; triangles cull by the fraction below
        bool sure = false, found = false, slow = false;
        uint32_t qn = 0, d0 = PM_DEFER_NONE, d1 = PM_DEFER_NONE, d2 = PM_DEFER_NONE;
        HitRec hit; hit.prim_id = 0xFFFFFFFFu; hit.t = ray.tmax; hit.u = 0.0f; hit.v = 0.0f;
        auto defer_prim = [&](uint32_t tagged_unit) {
            if (d0 == PM_DEFER_NONE) d0 = tagged_unit;
            else if (d1 == PM_DEFER_NONE) d1 = tagged_unit;
            else if (d2 == PM_DEFER_NONE) d2 = tagged_unit;
            else slow = true;
        };

        // ---- patches, any hit (PATCH: the scene has some): the reference arithmetic at once - shadow rays aim AT area lights,
        // i.e. at patches, end a hair in front of them and so pass any box test; deferring them all was measured at twice the
        // sweep's time (C5 shadow sweep 22 -> 43 ms)
        if (PATCH && ANY) {
            for (uint32_t i = 0; i < fs.n_patch; ++i) {
                const uint32_t unit = as_uniform(fs.units)[fs.n_sphere + i];
                if (live && !slow && !found) {
                    if (COUNT) { ctr->prims++; ctr->exact++; }
                    const cuint4_p r = as_uniform(prims + unit);
                    const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                    float t, u, v;
                    if (intersect_patch(ray.o, ray.d, tmax, mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)),
                                        mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y)),
                                        mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x)),
                                        mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w)), &t, &u, &v)) {
                        tmax = t; tcull = fminf(tcull, t); found = true; hit.prim_id = r3.x; hit.t = t; hit.u = u; hit.v = v;
                    }
                }
            }
        }
        // ---- triangles: pre-filter, candidates queued for the exact round.  The culling distance becomes a fraction.
        FlatCull cull; cull.cn = tcull; cull.cd = 1.0f;
        const float KD = PM_PRE_K * D, R2 = 2.0f * R, epsb_ray = 8e-7f * __fmaf_rn(9.0f, R, D);
        // the table in chunks of 8 entries: an any-hit wave leaves between chunks once all its rays are decided (closest hit:
        // every live ray sweeps the whole table); inside a chunk the loop is unrolled, PM_FLAT_UNROLL entries per scalar-load wait
        constexpr uint32_t kChunk = ANY ? (uint32_t)PM_FLAT_ANY_CHUNK : 0x10000u;       // (closest hit: one chunk)
        // carried from entry to entry (a lane that tests an entry has tested all entries before it: the predicate of the
        // counting build only ever turns off)
        FlatShared shared; shared.sv = mk3(0.f, 0.f, 0.f); shared.m = mk3(0.f, 0.f, 0.f); shared.e2m = 0.f;
        shared.sg = 1.0f; shared.det = 0.f; shared.tt = 0.f;
        if constexpr (COH && !COUNT && !ANY) {
            // ---- which entries can the rays of this wave reach at all?  (see the comment above trace_flat)
            unsigned long long emask = fs.n_tri >= 64u ? ~0ull : ((1ull << fs.n_tri) - 1ull);
            const unsigned long long lv = __ballot(live);
            if (lv != 0ull) {
                const int src = __ffsll((long long)lv) - 1;
                const f3 o0 = mk3(__shfl(ray.o.x, src), __shfl(ray.o.y, src), __shfl(ray.o.z, src));
                const f3 c = mk3(__shfl(ray.d.x, src), __shfl(ray.d.y, src), __shfl(ray.d.z, src));
                const f3 ac = abs3(c);
                const f3 ax = (ac.x <= ac.y && ac.x <= ac.z) ? mk3(1.f, 0.f, 0.f) : (ac.y <= ac.z ? mk3(0.f, 1.f, 0.f) : mk3(0.f, 0.f, 1.f));
                const f3 ex = fcross(c, ax), ey = fcross(c, ex);
                const float cc = fdot(c, c), w = fdot(ray.d, c);
                const bool same = __float_as_uint(ray.o.x) == __float_as_uint(o0.x) && __float_as_uint(ray.o.y) == __float_as_uint(o0.y) &&
                                  __float_as_uint(ray.o.z) == __float_as_uint(o0.z);
                const bool ok = !live || (same && w > 0.0f && w * w > 0.25f * (cc * fdot(ray.d, ray.d)));
                if (__all(ok) && cc > 0.0f && cc < INFINITY) {
                    const float sx = live ? fdot(ray.d, ex) / w : 0.0f, sy = live ? fdot(ray.d, ey) / w : 0.0f;
                    const float x0 = wave_min64(live ? sx : INFINITY), x1 = wave_max64(live ? sx : -INFINITY);
                    const float y0 = wave_min64(live ? sy : INFINITY), y1 = wave_max64(live ? sy : -INFINITY);
                    const float px = __fmaf_rn(1e-4f, x1 - x0, 2e-5f * (1.0f + fmaxf(fabsf(x0), fabsf(x1))));
                    const float py = __fmaf_rn(1e-4f, y1 - y0, 2e-5f * (1.0f + fmaxf(fabsf(y0), fabsf(y1))));
                    bool keep = lane < fs.n_tri;
                    if (keep) {
                        const uint4* r = prims + (fs.tris[lane].unit & PM_FLAT_UNIT_MASK);
                        const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                        const f3 a0 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)) - o0;
                        const f3 a1 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y)) - o0;
                        const f3 a2 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x)) - o0;
                        const float w0 = fdot(a0, c), w1 = fdot(a1, c), w2 = fdot(a2, c);
                        // in front, and within ~84 degrees of the axis: w_k > 0.1 |a_k| |c|
                        const bool front = w0 > 0.0f && w1 > 0.0f && w2 > 0.0f && w0 * w0 > 0.01f * (cc * fdot(a0, a0)) &&
                                           w1 * w1 > 0.01f * (cc * fdot(a1, a1)) && w2 * w2 > 0.01f * (cc * fdot(a2, a2));
                        if (front) {
                            const float u0 = fdot(a0, ex) / w0, u1 = fdot(a1, ex) / w1, u2 = fdot(a2, ex) / w2;
                            const float v0 = fdot(a0, ey) / w0, v1 = fdot(a1, ey) / w1, v2 = fdot(a2, ey) / w2;
                            const float ulo = fminf(u0, fminf(u1, u2)), uhi = fmaxf(u0, fmaxf(u1, u2));
                            const float vlo = fminf(v0, fminf(v1, v2)), vhi = fmaxf(v0, fmaxf(v1, v2));
                            // (written so that a NaN keeps the entry)
                            const bool out = ulo > x1 + px || uhi < x0 - px || vlo > y1 + py || vhi < y0 - py;
                            keep = !out;
                        }
                    }
                    emask = __ballot(keep);
                }
            }
            uint32_t prev = 0xFFFFFFF0u;
            for (unsigned long long m = emask; m != 0ull; m &= m - 1ull) {
                const uint32_t i = (uint32_t)__ffsll((long long)m) - 1u;
                const cuint4_p e = as_uniform(reinterpret_cast<const uint4*>(fs.tris + i));
                const uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3];
                // the chained forms need the entry before this one to have been swept
                const uint32_t flags = (i == prev + 1u) ? q1.w : (q1.w & PM_FLAT_UNIT_MASK);
                prev = i;
                FlatTerms ft;
                if (!tri_reject_flat(ray.o, ray.d, KD, R2,
                                     mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)),
                                     mk3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                                     mk3(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z)),
                                     mk3(__uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z)),
                                     __uint_as_float(q0.w), __uint_as_float(q2.w), __uint_as_float(q3.w), flags, &shared, cull, &ft)) {
                    float ln, ld;
                    (void)tri_certain_flat(ft, __uint_as_float(q2.w), epsb_ray, &cull, true, &ln, &ld);
                    if (qn < PM_FLAT_CAND_MAX) {
                        cand[qn * stride] = q1.w & PM_FLAT_UNIT_MASK;
                        cand[(PM_FLAT_CAND_MAX + qn) * stride] = __float_as_uint(ln);
                        cand[(2 * PM_FLAT_CAND_MAX + qn) * stride] = __float_as_uint(ld);
                        ++qn;
                    } else slow = true;
                }
            }
        } else
        for (uint32_t i0 = 0; i0 < fs.n_tri; i0 += kChunk) {
        if (ANY) {
            sure = !(cull.cn == tcull && cull.cd == 1.0f);
            if (!__any(live && !slow && !(found || sure))) break;
        }
        const uint32_t i1 = i0 + kChunk < fs.n_tri ? i0 + kChunk : fs.n_tri;
#pragma unroll PM_FLAT_UNROLL
        for (uint32_t i = i0; i < i1; ++i) {
            // a certain hit always lowers the culling fraction, so "was there one" is read off it instead of being tracked
            // (tracking it makes the loop body branchier: measured 10 % of the kernel)
            if (ANY && COUNT) sure = !(cull.cn == tcull && cull.cd == 1.0f);
            const cuint4_p e = as_uniform(reinterpret_cast<const uint4*>(fs.tris + i));
            const uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3];
            // (no lane predicate around the test: an idle, slow or - any hit - already occluded lane computes along for nothing;
            // its candidates are never used, or are more occluders of a ray that has one.  The loop body is five to ten scalar
            // instructions shorter for it.  The counting build keeps the predicate: its counters are the CPU mirror's.)
            if (!COUNT || (live && !slow && !(ANY && (found || sure)))) {
                if (COUNT) ctr->prims++;
                FlatTerms ft;
                // ONE masked region for everything behind the rejection (classification and the push into the queue): the
                // common outcome - every lane rejects - then skips it with a single branch
                if (!tri_reject_flat(ray.o, ray.d, KD, R2,
                                     mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)),
                                     mk3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                                     mk3(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z)),
                                     mk3(__uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z)),
                                     __uint_as_float(q0.w), __uint_as_float(q2.w), __uint_as_float(q3.w), q1.w, &shared, cull, &ft)) {
                    float ln, ld;
                    (void)tri_certain_flat(ft, __uint_as_float(q2.w), epsb_ray, &cull, true, &ln, &ld);
                    if (qn < PM_FLAT_CAND_MAX) {
                        cand[qn * stride] = q1.w & PM_FLAT_UNIT_MASK;
                        cand[(PM_FLAT_CAND_MAX + qn) * stride] = __float_as_uint(ln);
                        cand[(2 * PM_FLAT_CAND_MAX + qn) * stride] = __float_as_uint(ld);
                        ++qn;
                    } else slow = true;         // one candidate too many: the second pass redoes this ray exactly
                }
            }
        }
        }
        sure = !(cull.cn == tcull && cull.cd == 1.0f);
        // ---- spheres and (closest hit) patches: float pre-filters; what passes is deferred to the second pass.  AFTER the
        // triangles: a certain triangle hit bounds the distance within which a sphere or a patch still matters (any hit: the ray
        // is decided and skips them altogether) - tested first, against the ray's own t_max, twice as many were deferred.
        const float tcs = (cull.cn / cull.cd) * 1.00001f;
        // CONFIRM PATH of the sphere and patch pre-filters (counting kernels, i.e. every test that compares counters): a primitive
        // a pre-filter rejects is put through its reference arithmetic all the same, and should that hit lie in front of what the
        // ray ends up with (any hit: should there be one although the ray ends up unoccluded), the ray is counted in `fallback`
        // and redone exactly by the second pass.  So a false reject can neither pass a counting run unnoticed nor falsify its
        // result; the timed kernels skip the check.
        float t_rej = INFINITY;
        for (uint32_t i = 0; i < fs.n_sphere; ++i) {
            const uint32_t unit = as_uniform(fs.units)[i];
            if (live && !slow && !(ANY && (found || sure))) {
                if (COUNT) ctr->prims++;
                const cuint4_p r = as_uniform(prims + unit);
                float m[12];
                const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
                m[0] = __uint_as_float(r0.x); m[1] = __uint_as_float(r0.y); m[2] = __uint_as_float(r0.z); m[3] = __uint_as_float(r0.w);
                m[4] = __uint_as_float(r1.x); m[5] = __uint_as_float(r1.y); m[6] = __uint_as_float(r1.z); m[7] = __uint_as_float(r1.w);
                m[8] = __uint_as_float(r2.x); m[9] = __uint_as_float(r2.y); m[10] = __uint_as_float(r2.z); m[11] = __uint_as_float(r2.w);
                if (sphere_prefilter(ray.o, ray.d, R, D, m, __uint_as_float(r3.x), __uint_as_float(r4.z), __uint_as_float(r4.w), tcs))
                    defer_prim(unit);
                else if (COUNT) {       // confirm path, as for the patches below: the interval test all the same (not counted in
                                        // `exact`, which stays the number of interval tests the timed kernels run)
                    float st, su, sv; uint32_t sid;
                    if (sphere_test_cold(prims + unit, ray.o, ray.d, ray.tmax, &st, &su, &sv, &sid)) t_rej = fminf(t_rej, st);
                }
            }
        }
        // (Why no false reject of a PATCH is expected: the reference only runs its patch test for rays that pass ITS slab test of
        // the same corner box - the leaf bounds of its BVH, bilinear_patch.rs:176-181, bounds.rs:381-404, widened by
        // 1 + 2 gamma(3) ~ 4e-7 - and patch_prefilter pads that box by 1e-4 of the coordinates and compares with a relative slack
        // of 1e-4: it accepts a superset.  Spheres: the forward error bounds of sphere_prefilter.)
        if (PATCH && !ANY) {        // (round 1 ran the reference arithmetic at once, for every ray and every patch: ~300 instructions each)
            const f3 pinv = mk3(safe_rcp_dir(ray.d.x, D), safe_rcp_dir(ray.d.y, D), safe_rcp_dir(ray.d.z, D));
            for (uint32_t i = 0; i < fs.n_patch; ++i) {
                const uint32_t unit = as_uniform(fs.units)[fs.n_sphere + i];
                if (live && !slow) {
                    if (COUNT) ctr->prims++;
                    const cuint4_p r = as_uniform(prims + unit);
                    const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                    const f3 q00 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                    const f3 q10 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                    const f3 q01 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                    const f3 q11 = mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
                    if (patch_prefilter(ray.o, pinv, R, q00, q10, q01, q11, tcs)) defer_prim(unit | PM_DEFER_PATCH);
                    else if (COUNT) {
                        ctr->exact++;
                        float t, u, v;
                        if (intersect_patch(ray.o, ray.d, ray.tmax, q00, q10, q01, q11, &t, &u, &v)) t_rej = fminf(t_rej, t);
                    }
                }
            }
        }
        // ---- exact round: the queued triangles, in the order they were met, minus the certain hits whose t bracket starts
        // beyond the final culling distance (they lie behind another certain hit); compacted first so that the wave makes
        // as many trips as its busiest lane has survivors
        {
            uint32_t m = 0;
            for (uint32_t qi = 0; qi < PM_FLAT_CAND_MAX; ++qi) {
                if (qi >= qn) continue;
                const float ln = __uint_as_float(cand[(PM_FLAT_CAND_MAX + qi) * stride]), ld = __uint_as_float(cand[(2 * PM_FLAT_CAND_MAX + qi) * stride]);
                if (!(ln * cull.cd > cull.cn * ld)) { cand[m * stride] = cand[qi * stride]; ++m; }
            }
            qn = m;
        }
        // what the closest hit must not exceed if culling was right: the fraction as a float, rounded DOWN (one value
        // kept across the exact round instead of two)
        const float tconf = fdn(cull.cn / cull.cd);
        for (uint32_t qj = 0; qj < PM_FLAT_CAND_MAX; ++qj) {
            if (!__any(live && !slow && qj < qn && !(ANY && found))) break;
            if (live && !slow && qj < qn && !(ANY && found)) {
                const uint32_t qi = ANY ? qn - 1u - qj : qj;       // any hit: last candidate (the certain hit) first
                const uint4* r = prims + cand[qi * stride];
                const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                const f3 v0 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                const f3 v1 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                const f3 v2 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                if (COUNT) ctr->exact++;
                float t, u, v;
                if (intersect_tri_exact(ray.o, ray.d, tmax, v0, v1, v2, &t, &u, &v)) {
                    tmax = t; found = true; hit.prim_id = r2.y; hit.t = t; hit.u = u; hit.v = v;
                }
            }
        }
        // ---- did the exact tests confirm what culling assumed?  If not the ray is slow too (counted; never seen)
        const bool false_reject = COUNT && live && !slow && (ANY ? (!found && t_rej < INFINITY) : t_rej < (found ? hit.t : ray.tmax));
        const bool unconfirmed = (live && !slow && sure && !(found && (ANY || hit.t <= tconf))) || false_reject;
        if (COUNT && unconfirmed) ctr->fallback++;
        // ---- commit.  Closest hit: what the triangles gave is committed now and the second pass improves it with the
        // deferred primitives.  Any hit: a ray occluded by a triangle is final; one with deferred primitives is committed by
        // the second pass.  Slow rays are committed by the second pass alone.
        const bool redo = live && (slow || unconfirmed);
        const bool pass2 = live && !redo && d0 != PM_DEFER_NONE && !(ANY && found);
        if (live && !redo && (!ANY || !pass2)) commit(item, found, hit);
        dw.push(pass2, lane, item, d0, d1, d2, &ctr->overflow);
        sw.push(redo, lane, item, &ctr->overflow);
    }
    dw.pad(lane);
    sw.pad(lane);
}

// exact tests of the second pass: a hit lowers *tmax and replaces *hit
template <bool COUNT>
PMD void fix_test_sphere(const uint4* __restrict__ prims, uint32_t unit, const Ray& ray, float* tmax, bool* found, HitRec* hit,
                         TravCounters* ctr) {
    const uint4* r = prims + unit;
    float m[12];
    const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
    m[0] = __uint_as_float(r0.x); m[1] = __uint_as_float(r0.y); m[2] = __uint_as_float(r0.z); m[3] = __uint_as_float(r0.w);
    m[4] = __uint_as_float(r1.x); m[5] = __uint_as_float(r1.y); m[6] = __uint_as_float(r1.z); m[7] = __uint_as_float(r1.w);
    m[8] = __uint_as_float(r2.x); m[9] = __uint_as_float(r2.y); m[10] = __uint_as_float(r2.z); m[11] = __uint_as_float(r2.w);
    const float radius = __uint_as_float(r3.x), zmin = __uint_as_float(r3.y), zmax = __uint_as_float(r3.z), phimax = __uint_as_float(r3.w);
    if (COUNT) ctr->exact++;
    SphereHit sh;
    if (intersect_sphere(ray.o, ray.d, *tmax, m, radius, zmin, zmax, phimax, &sh)) {
        *tmax = sh.t; *found = true; hit->prim_id = r4.x; hit->t = sh.t;
        // (u,v) of sphere.rs:382-385
        const float th0 = acosf(fminf(fmaxf(zmin / radius, -1.0f), 1.0f));
        const float th1 = acosf(fminf(fmaxf(zmax / radius, -1.0f), 1.0f));
        const float th = acosf(fminf(fmaxf(sh.p_obj.z / radius, -1.0f), 1.0f));
        hit->u = sh.phi / phimax; hit->v = (th - th0) / (th1 - th0);
    }
}
template <bool COUNT>
PMD void fix_test_patch(const uint4* __restrict__ prims, uint32_t unit, bool is_tri, const Ray& ray, float* tmax, bool* found,
                        HitRec* hit, TravCounters* ctr) {
    const uint4* r = prims + unit;
    const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
    const f3 p00 = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
    const f3 p10 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
    const f3 p01 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
    f3 p11 = p01;
    uint32_t id = r2.y;
    if (!is_tri) {
        const uint4 r3 = r[3];
        p11 = mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
        id = r3.x;
    }
    if (COUNT) ctr->exact++;
    float t, u, v;
    if (intersect_patch(ray.o, ray.d, *tmax, p00, p10, p01, p11, &t, &u, &v)) {
        *tmax = t; *found = true; hit->prim_id = id; hit->t = t; hit->u = u; hit->v = v;
    }
}

// Second pass over the list entries {work item, unit, unit, unit}: every lane runs exact tests, so the waves of this pass
// are dense.  The list has holes (the padding of the waves' reservations), so a wave first gathers valid entries into
// `buf` (LDS, 128 entries per wave) and works on 64 of them at a time.  The units are spheres / patches (PM_DEFER_PATCH)
// deferred for this ray.  Closest hit: the ray already has its best other hit committed; load_t(i) returns that t and
// commit(i, true, hit) overwrites it when a deferred primitive is nearer.  Any hit: nothing else occluded the ray and it
// has not been committed; commit(i, occluded, .).
// Then (flat sweep only) the slow list: those rays are redone from scratch, every primitive of the flat table through its
// exact test, and committed whatever the outcome.
template <bool ANY, bool COUNT, class FetchFn, class LoadTFn, class CommitFn>
PMD void deferred_fix(FlatScene fs, const uint4* __restrict__ prims, const uint4* __restrict__ defer_list,
                      const uint32_t* __restrict__ defer_count, const uint32_t* __restrict__ slow_list,
                      const uint32_t* __restrict__ slow_count, uint32_t defer_cap, uint4* buf, TravCounters* ctr, FetchFn fetch,
                      LoadTFn load_t, CommitFn commit) {
    // An overflowing pass 1 has reported itself (the call returns PM_ERR_STACK_OVERFLOW, results invalid) and nothing was
    // written for the reservations that did not fit - but the counter went on counting, so the entries between the last
    // chunk that fitted and `cap` are whatever an earlier launch left there.  Such a list is not processed at all.
    const uint32_t n = *defer_count <= defer_cap ? *defer_count : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t cnt = 0;                                   // wave-uniform: valid entries waiting in buf
    for (uint32_t base = wave * 64u; base < n || cnt > 0u; base += n_waves * 64u) {
        if (base < n) {
            uint4 ent = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
            if (base + lane < n) ent = defer_list[base + lane];
            const bool valid = ent.x != 0xFFFFFFFFu;
            const unsigned long long vm = __ballot(valid);
            if (valid) buf[cnt + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] = ent;
            cnt += (uint32_t)__popcll(vm);
            __threadfence_block();
            if (cnt < 64u && base + n_waves * 64u < n) continue;     // gather more before running the tests
        }
        const uint32_t take = cnt < 64u ? cnt : 64u;
        uint4 ent = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u), tail = ent;
        if (lane < take) ent = buf[lane];
        if (64u + lane < cnt) tail = buf[64u + lane];
        __threadfence_block();
        if (64u + lane < cnt) buf[lane] = tail;
        cnt -= take;
        __threadfence_block();
        Ray ray; ray.o = mk3(0.f, 0.f, 0.f); ray.d = mk3(0.f, 0.f, 1.f); ray.tmax = 0.f;
        const bool have = ent.x != 0xFFFFFFFFu && fetch(ent.x, &ray);
        float tmax = ray.tmax;
        if (have && !ANY) tmax = load_t(ent.x);
        bool found = false;
        HitRec hit; hit.prim_id = 0xFFFFFFFFu; hit.t = tmax; hit.u = 0.0f; hit.v = 0.0f;
        // ---- the deferred spheres / patches of this ray
        for (int k = 0; k < 3; ++k) {
            const uint32_t tagged = k == 0 ? ent.y : (k == 1 ? ent.z : ent.w);
            if (!have || tagged == PM_DEFER_NONE || (ANY && found)) continue;
            if (tagged & PM_DEFER_PATCH) fix_test_patch<COUNT>(prims, tagged & ~PM_DEFER_PATCH, false, ray, &tmax, &found, &hit, ctr);
            else fix_test_sphere<COUNT>(prims, tagged, ray, &tmax, &found, &hit, ctr);
        }
        if (have && (ANY || found)) commit(ent.x, found, hit);
    }
    // ---- slow rays (flat sweep only): everything, exactly.  Same gathering, in the first 128 words of buf.
    if (slow_list == nullptr) return;
    const uint32_t ns = *slow_count <= defer_cap ? *slow_count : 0u;
    uint32_t* ibuf = reinterpret_cast<uint32_t*>(buf);
    cnt = 0;
    for (uint32_t base = wave * 64u; base < ns || cnt > 0u; base += n_waves * 64u) {
        if (base < ns) {
            uint32_t it = 0xFFFFFFFFu;
            if (base + lane < ns) it = slow_list[base + lane];
            const bool valid = it != 0xFFFFFFFFu;
            const unsigned long long vm = __ballot(valid);
            if (valid) ibuf[cnt + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] = it;
            cnt += (uint32_t)__popcll(vm);
            __threadfence_block();
            if (cnt < 64u && base + n_waves * 64u < ns) continue;
        }
        const uint32_t take = cnt < 64u ? cnt : 64u;
        uint32_t it = 0xFFFFFFFFu, tail = 0xFFFFFFFFu;
        if (lane < take) it = ibuf[lane];
        if (64u + lane < cnt) tail = ibuf[64u + lane];
        __threadfence_block();
        if (64u + lane < cnt) ibuf[lane] = tail;
        cnt -= take;
        __threadfence_block();
        Ray ray; ray.o = mk3(0.f, 0.f, 0.f); ray.d = mk3(0.f, 0.f, 1.f); ray.tmax = 0.f;
        const bool have = it != 0xFFFFFFFFu && fetch(it, &ray);
        float tmax = ray.tmax;
        bool found = false;
        HitRec hit; hit.prim_id = 0xFFFFFFFFu; hit.t = tmax; hit.u = 0.0f; hit.v = 0.0f;
        for (uint32_t i = 0; i < fs.n_sphere; ++i) {
            const uint32_t unit = fs.units[i];
            if (have && !(ANY && found)) { if (COUNT) ctr->prims++; fix_test_sphere<COUNT>(prims, unit, ray, &tmax, &found, &hit, ctr); }
        }
        for (uint32_t i = 0; i < fs.n_patch; ++i) {
            const uint32_t unit = fs.units[fs.n_sphere + i];
            if (have && !(ANY && found)) { if (COUNT) ctr->prims++; fix_test_patch<COUNT>(prims, unit, false, ray, &tmax, &found, &hit, ctr); }
        }
        for (uint32_t i = 0; i < fs.n_tri; ++i) {
            const uint32_t unit = fs.tris[i].unit & PM_FLAT_UNIT_MASK;
            if (have && !(ANY && found)) {
                if (COUNT) { ctr->prims++; ctr->exact++; }
                const uint4* r = prims + unit;
                const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
                float t, u, v;
                if (intersect_tri_exact(ray.o, ray.d, tmax, mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)),
                                        mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y)),
                                        mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x)), &t, &u, &v)) {
                    tmax = t; found = true; hit.prim_id = r2.y; hit.t = t; hit.u = u; hit.v = v;
                }
            }
        }
        if (have) commit(it, found, hit);
    }
}

}  // namespace pmd
