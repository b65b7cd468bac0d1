// Device-side sampling, camera, interaction, BSDF, light and film arithmetic of the wavefront path tracer.
// Every function is the reference's arithmetic in the reference's operation order (-ffp-contract=off):
//   util/rng.rs, sampling/sampler.rs:104-116        PCG32 stream, murmur3 pixel hash
//   sampling/spectrum.rs:73-91,870-905              wavelength sampling
//   sampling/routines.rs                            warps, PiecewiseConstant1D/2D
//   camera/perspective.rs:130-189, core/transform.rs:446-466
//   shapes/sphere.rs:371-442, shapes/bilinear_patch.rs:86-167, core/interaction.rs:99-131,354-369
//   reflection/{base,diffuse,dielectric}.rs, materials/material.rs:122-195
//   lights/*.rs, core/ray.rs:47-66, camera/sensor.rs:114-121
// Transcendentals (sin, cos, atanh, cosh, exp) are evaluated in f64 and rounded once to f32: that is the
// correctly rounded f32 result in all but ~2^-29 of cases, which is what glibc's f32 routines (the ones
// Rust's std calls on Linux) return for all but a small fraction of inputs.
#pragma once
#include "pm_device.hpp"
#include "pm_crmath.hpp"

namespace pmd {

PMD float cr_sin(float x) { return (float)sin((double)x); }
PMD float cr_cos(float x) { return (float)cos((double)x); }
// sin and cos of one angle: one argument reduction instead of two
PMD void cr_sincos(float x, float* sn, float* cs) { double s, c; sincos((double)x, &s, &c); *sn = (float)s; *cs = (float)c; }
// ... of phi = 2 pi u (u in [0, 1)), the argument of every azimuth the sampling routines draw: pm_crmath.hpp, equal to the
// above for each of the 2^23 values of u (tests/test_crmath_cpu.py)
PMD void cr_sincos_2pi(float x, float* sn, float* cs) { double s, c; pm_sincos_2pi(x, &s, &c); *sn = (float)s; *cs = (float)c; }
// ... for the concentric disk mapping, whose angle lies in [-pi/4, 3pi/4]: no argument reduction at all.  Taylor series in
// double (|t| <= pi/4: the first neglected terms are 8e-20 and 2e-18), so the result is the correctly rounded float like
// (float)sin((double)x) except where the exact value lies within ~2^-52 (relative) of a rounding boundary.
PMD void cr_sincos_quarter(float x, float* sn, float* cs) {
    const bool upper = x > 0.78539816339744830962f;              // then t = x - pi/2 in [-pi/4, pi/4]: sin x = cos t, cos x = -sin t
    const double t = upper ? (double)x - 1.57079632679489661923 : (double)x;
    const double t2 = t * t;
    double ps = 2.8114572543455206e-15;                          // 1/17!
    ps = fma(ps, t2, -7.6471637318198164e-13);                   // -1/15!
    ps = fma(ps, t2, 1.6059043836821613e-10);                    // 1/13!
    ps = fma(ps, t2, -2.5052108385441720e-08);                   // -1/11!
    ps = fma(ps, t2, 2.7557319223985893e-06);                    // 1/9!
    ps = fma(ps, t2, -1.9841269841269841e-04);                   // -1/7!
    ps = fma(ps, t2, 8.3333333333333332e-03);                    // 1/5!
    ps = fma(ps, t2, -1.6666666666666666e-01);                   // -1/3!
    const double s = fma(ps * t2, t, t);
    double pc = 4.7794773323873853e-14;                          // 1/16!
    pc = fma(pc, t2, -1.1470745597729725e-11);                   // -1/14!
    pc = fma(pc, t2, 2.0876756987868100e-09);                    // 1/12!
    pc = fma(pc, t2, -2.7557319223985888e-07);                   // -1/10!
    pc = fma(pc, t2, 2.4801587301587302e-05);                    // 1/8!
    pc = fma(pc, t2, -1.3888888888888889e-03);                   // -1/6!
    pc = fma(pc, t2, 4.1666666666666664e-02);                    // 1/4!
    pc = fma(pc, t2, -0.5);
    const double c = fma(pc, t2, 1.0);
    *sn = (float)(upper ? c : s);
    *cs = (float)(upper ? -s : c);
}
// atanh for |x| < 1 and cosh for |x| <= 2.2 - the wavelength sampler's arguments (|x| <= 0.9706; |x| <= 2.11) - written out in
// pm_crmath.hpp: the f32 results equal (float)atanh((double)x) / (float)cosh((double)x) for every argument the sampler can
// produce (checked exhaustively on the host, tests/test_crmath_cpu.py) at a quarter of the device library's f64 operations
PMD float cr_atanh(float x) { return (float)pm_atanh_f32arg(x); }
PMD float cr_cosh(float x) { return (float)pm_cosh_small((double)x); }
// (pm_crmath.hpp: equal to (float)exp((double)x) for every f32 in [-104, 89] and beyond, to (float)log((double)x) for every
// positive f32 - both checked exhaustively on the host)
PMD float cr_exp(float x) { return (float)pm_exp_f32arg(x); }
PMD float cr_log(float x) { return (float)pm_log_f32arg(x); }
// exp(-k[c] * x) for the four wavelengths.  A grey medium has four equal coefficients, and equal arguments give equal results:
// the three repeats are then copies (same bits), and the f64 exponential - the most expensive call of a medium vertex - runs once.
PMD void cr_exp_neg4(const float* k, float x, float* out) {
    out[0] = cr_exp(-k[0] * x);
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        out[c] = out[0];
        if (k[c] != k[0]) out[c] = cr_exp(-k[c] * x);
    }
}

#define PM_PI 3.14159265358979323846f
#define PM_INV_PI 0.318309886183790671537767526745028724f
#define PM_PI_2 1.57079632679489661923132169163975144f
#define PM_PI_4 0.785398163397448309615660845819875721f
#define PM_INV_4PI 0.07957747f
#define PM_INV_2PI 0.15915494f
#define PM_ONE_MINUS_EPS (1.0f - 1.1920928955078125e-07f)

PMD float clampf(float x, float lo, float hi) { float r = x; if (r < lo) r = lo; if (r > hi) r = hi; return r; }
PMD float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
PMD float safe_sqrtf(float x) { return sqrtf(fmaxf(x, 0.0f)); }
PMHD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
PMD f3 face_fwd(f3 n, f3 v) { return (__float_as_uint(dot3(n, v)) >> 31) ? -n : n; }

// ---------------------------------------------------------------------------------------- rng
struct Rng { unsigned long long state, inc; };
PMD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
PMD uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
PMD unsigned long long hash_pixel(int x, int y, uint32_t seed) {   // murmur3_x86_128 of 8 bytes, low 64 bits
    const uint32_t c1 = 0x239b961bu, c2 = 0xab0e9789u, c3 = 0x38b34ae5u;
    uint32_t h1 = seed, h2 = seed, h3 = seed, h4 = seed;
    uint32_t k1 = (uint32_t)x, k2 = (uint32_t)y;
    k2 *= c2; k2 = rotl32(k2, 16); k2 *= c3; h2 ^= k2;
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
    h1 ^= 8u; h2 ^= 8u; h3 ^= 8u; h4 ^= 8u;
    h1 += h2; h1 += h3; h1 += h4; h2 += h1; h3 += h1; h4 += h1;
    h1 = fmix32(h1); h2 = fmix32(h2); h3 = fmix32(h3); h4 = fmix32(h4);
    h1 += h2; h1 += h3; h1 += h4; h2 += h1;
    return ((unsigned long long)h2 << 32) | (unsigned long long)h1;
}
#define PM_PCG_MULT 6364136223846793005ULL
PMD Rng rng_start(int x, int y, uint32_t seed, unsigned long long sample_index, unsigned long long dim) {
    Rng r;
    r.inc = ((sample_index * 65536ULL + dim) << 1) | 1ULL;
    r.state = hash_pixel(x, y, seed) + r.inc;
    r.state = r.state * PM_PCG_MULT + r.inc;
    return r;
}
PMD float rng_uniform(Rng* r) {
    const unsigned long long old = r->state;
    r->state = old * PM_PCG_MULT + r->inc;
    const uint32_t rot = (uint32_t)(old >> 59);
    const uint32_t xsh = (uint32_t)(((old >> 18) ^ old) >> 27);
    const uint32_t u = (xsh >> rot) | (xsh << ((32u - rot) & 31u));
    return __uint_as_float((u >> 9) | 0x3F800000u) - 1.0f;
}

// ---------------------------------------------------------------------------------------- spectra
struct DSpectrum { int kind; float a, b, c, d; int table; };   // table: offset into the float pool
struct DMaterial { int type; DSpectrum spec; float ur, vr; int remap; int texture; float su, sv, du, dv; DSpectrum spec2; };
// DIFFUSE_AREA lights do not use m: m[0..3] hold the four corner-area weights of BilinearPatch::sample's bilinear warp and
// m[4] != 0 says the patch is a rectangle (patch_area_constants below: computed once per light at upload instead of once per
// light sample - ~350 instructions, half of them f64, of every path vertex of an area-lit scene)
struct DLight { int type; float scale; float m[12]; int table; uint32_t prim_id; int two_sided; };

PMD float dense_at(const float* __restrict__ pool, int table, float lambda) {
    const float r = roundf(lambda);
    const int li = (int)r;
    return (li >= 360 && li <= 830) ? pool[table + li - 360] : 0.0f;
}
PMD float sigmoid_poly(float c0, float c1, float c2, float lambda) {
    float x = __fmaf_rn(0.0f, lambda, c0);
    x = __fmaf_rn(x, lambda, c1);
    x = __fmaf_rn(x, lambda, c2);
    if (isinf(x)) return x > 0.0f ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * sqrtf(1.0f + x * x));
}
PMD float blackbody(float lambda, float temp) {
    if (temp <= 0.0f) return 0.0f;
    const float C = 299792458.0f, H = 6.6260697e-34f, KB = 1.3806488e-23f;
    const float l = lambda * 1e-9f;
    const float l5 = ((l * l) * (l * l)) * l;
    return (2.0f * H * C * C) / (l5 * (cr_exp((H * C) / (l * KB * temp)) - 1.0f));
}
PMD float spectrum_at(const float* __restrict__ pool, const DSpectrum& s, float lambda) {
    switch (s.kind) {
        case PM_SPEC_CONSTANT: return s.a;
        case PM_SPEC_DENSE: return dense_at(pool, s.table, lambda);
        case PM_SPEC_RGB_ALBEDO: return sigmoid_poly(s.a, s.b, s.c, lambda);
        case PM_SPEC_RGB_UNBOUNDED: return s.d * sigmoid_poly(s.a, s.b, s.c, lambda);
        case PM_SPEC_RGB_ILLUMINANT: return s.d * sigmoid_poly(s.a, s.b, s.c, lambda) * dense_at(pool, s.table, lambda);
        default: return blackbody(lambda, s.a) * s.b;
    }
}

// The same at the four wavelengths of a sample: ONE switch on the kind and the table reads issued together (index clamped, value
// selected afterwards).  Lane by lane the arithmetic is that of the scalar functions; done per wavelength, every table read sat
// in its own guarded branch behind its own wait - ten serialised cache round trips per path vertex in shade_kernel.
PMD void dense_at4(const float* __restrict__ pool, int table, const float* lam, float* out) {
    int li[4];
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) li[c] = (int)roundf(lam[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = pool[table + (li[c] < 360 ? 360 : li[c] > 830 ? 830 : li[c]) - 360];
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = (li[c] >= 360 && li[c] <= 830) ? v[c] : 0.0f;
}
PMD void spectrum_at4(const float* __restrict__ pool, const DSpectrum& s, const float* lam, float* out) {
    switch (s.kind) {
        case PM_SPEC_CONSTANT:
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = s.a;
            break;
        case PM_SPEC_DENSE: dense_at4(pool, s.table, lam, out); break;
        case PM_SPEC_RGB_ALBEDO:
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = sigmoid_poly(s.a, s.b, s.c, lam[c]);
            break;
        case PM_SPEC_RGB_UNBOUNDED:
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = s.d * sigmoid_poly(s.a, s.b, s.c, lam[c]);
            break;
        case PM_SPEC_RGB_ILLUMINANT: {
            float t[4];
            dense_at4(pool, s.table, lam, t);
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = s.d * sigmoid_poly(s.a, s.b, s.c, lam[c]) * t[c];
            break;
        }
        default:
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = blackbody(lam[c], s.a) * s.b;
            break;
    }
}

PMD float sample_visible_wavelength(float u) { return 538.0f - 138.88889f * cr_atanh(0.85691062f - 1.827502f * u); }
PMD float visible_pdf(float lambda) {
    if (lambda >= 360.0f && lambda <= 830.0f) { const float ch = cr_cosh(0.0072f * (lambda - 538.0f)); return 0.003939804f / (ch * ch); }
    return 0.0f;
}

// ---------------------------------------------------------------------------------------- warps
struct f2 { float x, y; };
PMD f3 sample_uniform_sphere(f2 u) {
    const float z = 1.0f - 2.0f * u.x;
    const float r = safe_sqrtf(1.0f - z * z);
    const float phi = 2.0f * PM_PI * u.y;
    float sp, cp; cr_sincos_2pi(phi, &sp, &cp);
    return mk3(r * cp, r * sp, z);
}
PMD f3 sample_uniform_hemisphere(f2 u) {
    const float z = u.x;
    const float r = safe_sqrtf(1.0f - z * z);
    const float phi = 2.0f * PM_PI * u.y;
    float sp, cp; cr_sincos_2pi(phi, &sp, &cp);
    return mk3(r * cp, r * sp, z);
}
PMD f2 sample_disk_concentric(f2 u) {
    f2 o; o.x = 2.0f * u.x - 1.0f; o.y = 2.0f * u.y - 1.0f;
    f2 z; z.x = 0.0f; z.y = 0.0f;
    if (o.x == 0.0f && o.y == 0.0f) return z;
    float r, th;
    if (fabsf(o.x) > fabsf(o.y)) { r = o.x; th = PM_PI_4 * (o.y / o.x); }
    else { r = o.y; th = PM_PI_2 - PM_PI_4 * (o.x / o.y); }
    float st, ct; cr_sincos_quarter(th, &st, &ct);
    f2 p; p.x = ct * r; p.y = st * r;
    return p;
}
PMD f2 sample_disk_polar(f2 u) {
    const float r = sqrtf(u.x);
    const float th = 2.0f * PM_PI * u.y;
    float st, ct; cr_sincos_2pi(th, &st, &ct);
    f2 p; p.x = r * ct; p.y = r * st;
    return p;
}
PMD f3 sample_cosine_hemisphere(f2 u) {
    const f2 d = sample_disk_concentric(u);
    return mk3(d.x, d.y, safe_sqrtf(1.0f - d.x * d.x - d.y * d.y));
}

// PiecewiseConstant1D::sample over tables in the float pool: func[n], cdf[n+1]
PMD void pc1d_sample(const float* __restrict__ func, const float* __restrict__ cdf, int n, float integral, float mn, float mx,
                     float u, float* value, float* pdf, int* offset) {
    int left = 0, right = n;        // cdf.len() - 1
    while (left < right) {
        const int mid = left + (right - left) / 2;
        if (cdf[mid] <= u) left = mid + 1; else right = mid;
    }
    int o = left == 0 ? 0 : (left >= n ? n - 1 : left - 1);
    float du = u - cdf[o];
    const float w = cdf[o + 1] - cdf[o];
    if (w > 0.0f) du /= w;
    *pdf = integral > 0.0f ? func[o] / integral : 0.0f;
    *value = lerpf(mn, mx, ((float)o + du) / (float)n);
    *offset = o;
}
PMD float sample_linear(float u, float a, float b) {
    if (u == 0.0f || a == 0.0f) return 0.0f;
    const float x = u * (a + b) / (a + sqrtf(lerpf(a * a, b * b, u)));
    return fminf(x, PM_ONE_MINUS_EPS);
}
PMD float sample_tent(float u, float r) {
    const float sum_w = 0.0f + 0.5f + 0.5f;
    float up = u * sum_w;
    if (up == sum_w) up = fdn(up);
    int off = 0; float sum = 0.0f;
    while (sum + 0.5f <= up) { sum += 0.5f; ++off; }
    const float ur = fminf((up - sum) / 0.5f, PM_ONE_MINUS_EPS);
    if (off == 0) return -r + r * sample_linear(ur, 0.0f, 1.0f);
    return r * sample_linear(ur, 1.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------- render constants
struct DFilter { int kind; float rx, ry; int nx, ny; int vals, cond_cdf, cond_int, marg_func, marg_cdf; float marg_int; };
// 64 B; flags bit 0 flip normal, bit 1 vertex normals (12 floats at pool[nrm_off]), bit 2 vertex uv (8 floats at pool[uv_off]:
// uv00 uv10 uv01 uv11), bits 8.. = 1 + index of the DiffuseAreaLight on this primitive (0: none); mat = material id.  Round 3:
// material and light ride in the record, and in a scene without spheres record i belongs to primitive id i - so a path vertex
// goes hit -> record -> material where it went hit -> (prim_ref, prim_material, prim_light) -> (record, material)
struct DShadePatch { float p[12]; uint32_t flags; uint32_t nrm_off; uint32_t uv_off; uint32_t mat; };
struct DShadeSphere { float rfo[12]; float ofr[12]; float radius, z_min, z_max, phi_max, th_min, th_max; uint32_t flags; uint32_t mat; };  // 128 B

struct RenderConst {
    float cfr[16];            // camera_from_raster
    float rfc[12];            // render_from_camera rows 0..2 (affine)
    float lens_radius, focal_distance;
    DFilter filter;
    int sensor_table;         // 3 x 471 in the pool
    float imaging_ratio, max_component_value;
    int integrator, max_depth, sample_lights, sample_bsdf;
    uint32_t seed;
    int n_lights, n_infinite;
    int terminal_emission;    // the scene has an infinite or an area light: a path's last vertex can add emission (see dead_end)
    float scene_radius;
    // homogeneous medium of PM_INTEGRATOR_VOL_PATH (media/medium.rs:41-97): 471-entry tables in the pool, HG g, bounds
    int medium_on, med_sigma_a, med_sigma_s;
    float med_g, med_bounds[6];
    // pass geometry
    int win_x0, win_y0, win_w, win_h, tiles_x, npix_pad;
    int film_x0, film_y0, film_w;
    int s_base, s_count;
};

PMD void filter_sample(const float* __restrict__ pool, const DFilter& f, f2 u, f2* p, float* weight) {
    if (f.kind == PM_FILTER_BOX) { p->x = lerpf(-f.rx, f.rx, u.x); p->y = lerpf(-f.ry, f.ry, u.y); *weight = 1.0f; return; }
    if (f.kind == PM_FILTER_TRIANGLE) { p->x = sample_tent(u.x, f.rx); p->y = sample_tent(u.y, f.ry); *weight = 1.0f; return; }
    float d1, pdf1, d0, pdf0; int v, uo;
    pc1d_sample(pool + f.marg_func, pool + f.marg_cdf, f.ny, f.marg_int, -f.ry, f.ry, u.y, &d1, &pdf1, &v);
    // conditional rows: func = |vals| (the Gaussian table is non-negative), cdf rows of nx+1
    pc1d_sample(pool + f.vals + v * f.nx, pool + f.cond_cdf + v * (f.nx + 1), f.nx, pool[f.cond_int + v], -f.rx, f.rx, u.x,
                &d0, &pdf0, &uo);
    p->x = d0; p->y = d1;
    *weight = pool[f.vals + v * f.nx + uo] / (pdf0 * pdf1);
}

// ---------------------------------------------------------------------------------------- camera
PMD void generate_camera_ray(const RenderConst& rc, f2 pf, f2 pl, f3* o_out, f3* d_out) {
    const float* m = rc.cfr;
    float x = pf.x * m[0] + pf.y * m[1] + 0.0f * m[2] + m[3];
    float y = pf.x * m[4] + pf.y * m[5] + 0.0f * m[6] + m[7];
    float z = pf.x * m[8] + pf.y * m[9] + 0.0f * m[10] + m[11];
    const float w = pf.x * m[12] + pf.y * m[13] + 0.0f * m[14] + m[15];
    if (w == 0.0f) { x /= w; y /= w; z /= w; }   // B11: never normalises
    f3 o = mk3(0.0f, 0.0f, 0.0f);
    f3 d = norm3(mk3(x, y, z));
    if (rc.lens_radius > 0.0f) {
        const f2 dl = sample_disk_concentric(pl);
        const float lx = dl.x * rc.lens_radius, ly = dl.y * rc.lens_radius;
        const float tf = rc.focal_distance / d.z;
        const f3 pfoc = o + d * tf;
        o = mk3(lx, ly, 0.0f);
        d = norm3(pfoc - o);
    }
    Iv ox, oy, oz;
    xf_point_exact(rc.rfc, o, &ox, &oy, &oz);
    const float* r = rc.rfc;
    const f3 dr = mk3(d.x * r[0] + d.y * r[1] + d.z * r[2], d.x * r[4] + d.y * r[5] + d.z * r[6], d.x * r[8] + d.y * r[9] + d.z * r[10]);
    const float lsq = dot3(dr, dr);
    if (lsq > 0.0f) {
        const f3 err = mk3((ox.hi - ox.lo) / 2.0f, (oy.hi - oy.lo) / 2.0f, (oz.hi - oz.lo) / 2.0f);
        const float dt = dot3(abs3(dr), err) / lsq;
        ox = ox + iv_exact(dr.x) * dt; oy = oy + iv_exact(dr.y) * dt; oz = oz + iv_exact(dr.z) * dt;
    }
    *o_out = mk3(iv_mid(ox), iv_mid(oy), iv_mid(oz));
    *d_out = dr;
}

// ---------------------------------------------------------------------------------------- interaction
struct SurfInt { Iv px, py, pz; f3 n, ns, dpdus; };

PMD f3 offset_ray_origin(const SurfInt& si, f3 w) {   // core/ray.rs:47-66
    const f3 err = mk3((si.px.hi - si.px.lo) / 2.0f, (si.py.hi - si.py.lo) / 2.0f, (si.pz.hi - si.pz.lo) / 2.0f);
    const float d = dot3(abs3(si.n), err);
    f3 off = si.n * d;
    if (dot3(si.n, w) < 0.0f) off = -off;
    f3 po = mk3(iv_mid(si.px), iv_mid(si.py), iv_mid(si.pz)) + off;
    if (off.x > 0.0f) po.x = fup(po.x); else if (off.x < 0.0f) po.x = fdn(po.x);
    if (off.y > 0.0f) po.y = fup(po.y); else if (off.y < 0.0f) po.y = fdn(po.y);
    if (off.z > 0.0f) po.z = fup(po.z); else if (off.z < 0.0f) po.z = fdn(po.z);
    return po;
}
PMD f3 si_p(const SurfInt& si) { return mk3(iv_mid(si.px), iv_mid(si.py), iv_mid(si.pz)); }

// Transform::rotate_from_to (core/transform.rs:180-201) applied to a vector
PMD f3 rotate_from_to_apply(f3 from, f3 to, f3 x) {
    f3 refl;
    if (fabsf(from.x) < 0.72f && fabsf(to.x) < 0.72f) refl = mk3(1.f, 0.f, 0.f);
    else if (fabsf(from.y) < 0.72f && fabsf(to.y) < 0.72f) refl = mk3(0.f, 1.f, 0.f);
    else refl = mk3(0.f, 0.f, 1.f);
    const f3 u = refl - from, v = refl - to;
    const float ua[3] = {u.x, u.y, u.z}, va[3] = {v.x, v.y, v.z};
    const float uu = dot3(u, u), vv = dot3(v, v), uv = dot3(u, v);
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float m[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            m[j] = (i == j ? 1.0f : 0.0f) - 2.0f / uu * ua[i] * ua[j] - 2.0f / vv * va[i] * va[j] + 4.0f * uv / (uu * vv) * va[i] * ua[j];
        r[i] = m[0] * x.x + m[1] * x.y + m[2] * x.z;
    }
    return mk3(r[0], r[1], r[2]);
}
PMD SurfInt patch_interaction(const DShadePatch& sp, float u, float v, const float* __restrict__ pool) {
    const f3 p00 = mk3(sp.p[0], sp.p[1], sp.p[2]), p10 = mk3(sp.p[3], sp.p[4], sp.p[5]);
    const f3 p01 = mk3(sp.p[6], sp.p[7], sp.p[8]), p11 = mk3(sp.p[9], sp.p[10], sp.p[11]);
    const f3 pt = lerp3(lerp3(p00, p01, v), lerp3(p10, p11, v), u);
    f3 dpdu = lerp3(p10, p11, v) - lerp3(p00, p01, v);
    const f3 dpdv = lerp3(p01, p11, u) - lerp3(p00, p10, u);
    // extension: on the collapsed edge of a degenerate patch (a triangle hit exactly at v = 1) dp/du is the zero vector and
    // the reference's normal and shading frame become NaN; take the direction dp/du has everywhere else on the patch
    if (dpdu.x == 0.0f && dpdu.y == 0.0f && dpdu.z == 0.0f) dpdu = p10 - p00;
    const f3 pabs = abs3(p00) + abs3(p10) + abs3(p01) + abs3(p11);
    const f3 perr = pabs * gamma_n(6);
    f3 n = norm3(cross3(dpdu, dpdv));
    if (sp.flags & 1u) n = n * -1.0f;
    SurfInt si;
    si.px = iv_err(pt.x, perr.x); si.py = iv_err(pt.y, perr.y); si.pz = iv_err(pt.z, perr.z);
    si.n = n; si.ns = n; si.dpdus = dpdu;
    if (sp.flags & 2u) {      // bilinear_patch.rs:150-164 + set_shading_geometry(.., orientation_is_authoritative = true)
        const float* q = pool + sp.nrm_off;
        const f3 n00 = mk3(q[0], q[1], q[2]), n10 = mk3(q[3], q[4], q[5]), n01 = mk3(q[6], q[7], q[8]), n11 = mk3(q[9], q[10], q[11]);
        f3 ns = lerp3(lerp3(n00, n01, v), lerp3(n10, n11, v), u);
        if (dot3(ns, ns) > 0.0f) {
            ns = norm3(ns);
            si.dpdus = rotate_from_to_apply(norm3(si.n), ns, dpdu);
            if (__float_as_uint(dot3(si.n, ns)) >> 31) si.n = -si.n;     // Normal3f::face_forward: sign bit of the dot product
            si.ns = ns;
        }
    }
    return si;
}
PMD f3 xf_normal12(const float* mi, f3 n) {   // Transform * Normal3f with the INVERSE matrix rows in mi
    return mk3(n.x * mi[0] + n.y * mi[4] + n.z * mi[8], n.x * mi[1] + n.y * mi[5] + n.z * mi[9], n.x * mi[2] + n.y * mi[6] + n.z * mi[10]);
}
PMD bool swaps_handedness12(const float* m) {
    const float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    return det < 0.0f;
}
// Transform * Point3fi for p = ph +- perr (approximate-p branch, core/transform.rs:371-404), affine 3 x 4 matrix m
PMD void xf_point_fi12(const float* m, f3 ph, f3 perr, Iv* px, Iv* py, Iv* pz) {
    const Iv ix = iv_err(ph.x, perr.x), iy = iv_err(ph.y, perr.y), iz = iv_err(ph.z, perr.z);
    const float x = iv_mid(ix), y = iv_mid(iy), z = iv_mid(iz);
    const float ex = (ix.hi - ix.lo) / 2.0f, ey = (iy.hi - iy.lo) / 2.0f, ez = (iz.hi - iz.lo) / 2.0f;
    const float g3 = gamma_n(3);
    const float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    const float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    const float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    float e0, e1, e2;
    if (ex == 0.0f && ey == 0.0f && ez == 0.0f && (ix.hi - ix.lo) == 0.0f && (iy.hi - iy.lo) == 0.0f && (iz.hi - iz.lo) == 0.0f) {
        e0 = g3 * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
        e1 = g3 * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
        e2 = g3 * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    } else {
        e0 = (g3 + 1.0f) * (fabsf(m[0]) * ex + fabsf(m[1]) * ey + fabsf(m[2]) * ez) + g3 * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
        e1 = (g3 + 1.0f) * (fabsf(m[4]) * ex + fabsf(m[5]) * ey + fabsf(m[6]) * ez) + g3 * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
        e2 = (g3 + 1.0f) * (fabsf(m[8]) * ex + fabsf(m[9]) * ey + fabsf(m[10]) * ez) + g3 * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    }
    *px = iv_err(xp, e0); *py = iv_err(yp, e1); *pz = iv_err(zp, e2);
}
// recomputes the reference's QuadricIntersection.p_obj from the ray and t_hit (sphere.rs:320-333), then
// interaction_from_intersection (sphere.rs:371-442) and SurfaceInteraction::transform
PMD SurfInt sphere_interaction(const DShadeSphere& s, f3 ro, f3 rd, float t_hit) {
    Iv ox, oy, oz, dx, dy, dz;
    xf_point_exact(s.ofr, ro, &ox, &oy, &oz);
    xf_vec_exact(s.ofr, rd, &dx, &dy, &dz);
    f3 ph; float phi;
    sphere_point_phi(mk3(iv_mid(ox), iv_mid(oy), iv_mid(oz)), mk3(iv_mid(dx), iv_mid(dy), iv_mid(dz)), t_hit, s.radius, &ph, &phi);
    const float z_radius = sqrtf(ph.x * ph.x + ph.y * ph.y);
    const float cos_phi = ph.x / z_radius, sin_phi = ph.y / z_radius;
    const float cos_theta = ph.z / s.radius;
    const f3 dpdu = mk3(-s.phi_max * ph.y, s.phi_max * ph.x, 0.0f);
    const float sin_theta = safe_sqrtf(1.0f - cos_theta * cos_theta);
    const f3 dpdv = mk3(ph.z * cos_phi, ph.z * sin_phi, -s.radius * sin_theta) * (s.th_max - s.th_min);
    const f3 perr = abs3(ph) * gamma_n(5);
    const bool flip = ((s.flags & 1u) != 0u) != swaps_handedness12(s.rfo);
    f3 n = norm3(cross3(dpdu, dpdv));
    if (flip) n = n * -1.0f;
    const float* m = s.rfo;
    SurfInt si;
    xf_point_fi12(m, ph, perr, &si.px, &si.py, &si.pz);
    si.n = xf_normal12(s.ofr, norm3(n));
    si.dpdus = mk3(dpdu.x * m[0] + dpdu.y * m[1] + dpdu.z * m[2], dpdu.x * m[4] + dpdu.y * m[5] + dpdu.z * m[6],
                   dpdu.x * m[8] + dpdu.y * m[9] + dpdu.z * m[10]);
    si.ns = face_fwd(xf_normal12(s.ofr, n), si.n);
    return si;
}

// ---------------------------------------------------------------------------------------- area lights, medium (extensions)
// sampling/routines.rs:219-246 (B14 reproduced: sample_bilinear hands (y, w0, w2) to the (a, b, t) lerp)
PMD float bilinear_pdf(f2 p, const float* w) {
    float w_sum = 0.0f;
    for (int i = 0; i < 4; ++i) w_sum += w[i];
    if (p.x < 0.0f || p.x > 1.0f || p.y < 0.0f || p.y > 1.0f) return 0.0f;
    if (w_sum == 0.0f) return 1.0f;
    return 4.0f * ((1.0f - p.x) * (1.0f - p.y) * w[0] + p.x * (1.0f - p.y) * w[1] + p.y * (1.0f - p.x) * w[2] + p.x * p.y * w[3]) / w_sum;
}
PMD f2 sample_bilinear(f2 u, const float* w) {
    f2 r;
    r.y = sample_linear(u.y, w[0] + w[1], w[2] + w[3]);
    r.x = sample_linear(u.x, lerpf(r.y, w[0], w[2]), lerpf(r.y, w[1], w[3]));
    return r;
}
// media/phase_function.rs:57-80
PMD float henyey_greenstein(float cos_theta, float g) {
    const float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return PM_INV_4PI * (1.0f - g * g) / (denom * safe_sqrtf(denom));
}
PMD f3 sample_henyey_greenstein(f3 wo, float g, f2 u, float* pdf) {
    float cos_theta;
    if (fabsf(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u.x;
    else {
        const float q = (1.0f - g * g) / (1.0f + g - 2.0f * g * u.x);
        cos_theta = -1.0f / (2.0f * g) * (1.0f + g * g - q * q);
    }
    const float sin_theta = safe_sqrtf(1.0f - cos_theta * cos_theta);
    const float phi = 2.0f * PM_PI * u.y;
    f3 x, y;                                           // Frame::from_z(wo) via Vec3f::coordinate_system (core/vec.rs:743-757)
    if (fabsf(wo.x) > fabsf(wo.y)) x = mk3(-wo.z, 0.0f, wo.x) / sqrtf(wo.x * wo.x + wo.z * wo.z);
    else x = mk3(0.0f, wo.z, -wo.y) / sqrtf(wo.y * wo.y + wo.z * wo.z);
    y = cross3(wo, x);
    const float st = clampf(sin_theta, -1.0f, 1.0f), ct = clampf(cos_theta, -1.0f, 1.0f);
    float sp, cp; cr_sincos_2pi(phi, &sp, &cp);
    *pdf = henyey_greenstein(cos_theta, g);
    return (st * cp) * x + (st * sp) * y + ct * wo;
}
// shapes/bilinear_patch.rs:739-773
PMHD bool f3_eq(f3 a, f3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
PMHD float dist2(f3 a, f3 b) { const f3 d = a - b; return dot3(d, d); }
PMHD bool patch_is_rectangle(f3 p00, f3 p10, f3 p01, f3 p11) {
    if (f3_eq(p00, p01) || f3_eq(p01, p11) || f3_eq(p11, p10) || f3_eq(p10, p00)) return false;
    const f3 n = norm3(cross3(p10 - p00, p01 - p00));
    if (fabsf(dot3(norm3(p11 - p00), n)) > 1e-5f) return false;
    const f3 c = (p00 + p10 + p01 + p11) / 4.0f;
    const float d0 = dist2(p00, c), d1 = dist2(p10, c), d2 = dist2(p01, c), d3 = dist2(p11, c);
    if (fabsf(d0 - d0) / d0 > 1e-4f || fabsf(d1 - d0) / d0 > 1e-4f || fabsf(d2 - d0) / d0 > 1e-4f || fabsf(d3 - d0) / d0 > 1e-4f) return false;
    return true;
}
// what BilinearPatch::sample needs of the patch alone (bilinear_patch.rs:268-283): is it a rectangle, else the corner areas
PMHD void patch_area_constants(f3 p00, f3 p10, f3 p01, f3 p11, float* w4, bool* rect) {
    *rect = patch_is_rectangle(p00, p10, p01, p11);
    w4[0] = len3(cross3(p10 - p00, p01 - p00)); w4[1] = len3(cross3(p10 - p00, p11 - p10));
    w4[2] = len3(cross3(p01 - p00, p11 - p01)); w4[3] = len3(cross3(p11 - p10, p11 - p01));
}
struct ShapeSample { f3 p; f3 n; float pdf; };
// BilinearPatch::sample + the area-sampling branch of sample_with_context (bilinear_patch.rs:260-352); the spherical-
// rectangle branch is todo!() upstream (sampling/routines.rs:307-327), so every patch is sampled by area
PMD bool patch_sample_ctx(const DShadePatch& sp, const float* __restrict__ pool, const float* w, bool rect, f3 ref_p, f2 u,
                          ShapeSample* out) {
    const f3 p00 = mk3(sp.p[0], sp.p[1], sp.p[2]), p10 = mk3(sp.p[3], sp.p[4], sp.p[5]);
    const f3 p01 = mk3(sp.p[6], sp.p[7], sp.p[8]), p11 = mk3(sp.p[9], sp.p[10], sp.p[11]);
    f2 uv; float pdf;
    if (!rect) { uv = sample_bilinear(u, w); pdf = bilinear_pdf(uv, w); }     // (w, rect: patch_area_constants, once per light)
    else { uv = u; pdf = 1.0f; }
    const f3 pu0 = lerp3(p00, p01, uv.y), pu1 = lerp3(p10, p11, uv.y);
    const f3 pt = lerp3(pu0, pu1, uv.x);
    const f3 dpdu = pu1 - pu0;
    const f3 dpdv = lerp3(p01, p11, uv.x) - lerp3(p00, p10, uv.x);
    if (dot3(dpdu, dpdu) == 0.0f || dot3(dpdv, dpdv) == 0.0f) return false;
    f3 n = norm3(cross3(dpdu, dpdv));
    if (sp.flags & 2u) {
        const float* q = pool + sp.nrm_off;
        const f3 n00 = mk3(q[0], q[1], q[2]), n10 = mk3(q[3], q[4], q[5]), n01 = mk3(q[6], q[7], q[8]), n11 = mk3(q[9], q[10], q[11]);
        const f3 ns = lerp3(lerp3(n00, n01, uv.y), lerp3(n10, n11, uv.y), uv.x);
        n = face_fwd(n, ns);
    } else if (sp.flags & 1u) n = -n;
    pdf = pdf / len3(cross3(dpdu, dpdv));
    // Point3fi::new_fi(p, p_err).midpoints(): the interval is [down(p - e), up(p + e)]
    const f3 pabs = abs3(p00) + abs3(p01) + abs3(p10) + abs3(p11);
    const f3 perr = pabs * gamma_n(6);
    const f3 pm = mk3(iv_mid(iv_err(pt.x, perr.x)), iv_mid(iv_err(pt.y, perr.y)), iv_mid(iv_err(pt.z, perr.z)));
    f3 wi = pm - ref_p;
    if (dot3(wi, wi) == 0.0f) return false;
    wi = norm3(wi);
    pdf /= fabsf(dot3(n, -wi)) / dist2(ref_p, pm);
    if (isinf(pdf)) return false;
    out->p = pm; out->n = n; out->pdf = pdf;
    return true;
}
// Sphere::sample (sphere.rs:104-133): a uniform point of the full sphere, pdf = 1 / area() (:100-102)
PMD void sphere_sample(const DShadeSphere& s, f2 u, ShapeSample* out) {
    f3 p_obj = mk3(0.0f, 0.0f, 0.0f) + sample_uniform_sphere(u) * s.radius;
    p_obj = p_obj * (s.radius / len3(p_obj - mk3(0.0f, 0.0f, 0.0f)));
    const f3 p_err = abs3(p_obj) * gamma_n(5);
    f3 n = norm3(xf_normal12(s.ofr, p_obj));
    if (s.flags & 1u) n = n * -1.0f;
    Iv px, py, pz;
    xf_point_fi12(s.rfo, p_obj, p_err, &px, &py, &pz);
    out->p = mk3(iv_mid(px), iv_mid(py), iv_mid(pz));
    out->n = n;
    out->pdf = 1.0f / (s.phi_max * s.radius * (s.z_max - s.z_min));
}
// Sphere::sample_with_context (sphere.rs:135-213).  ctx_p: the receiving point; p_origin: the same point offset towards the
// sphere's centre along the surface normal (ShapeSampleContext::offset_ray_origin_towards), or ctx_p itself in a medium
PMD f3 sphere_center(const DShadeSphere& s) {
    const float* m = s.rfo;
    return mk3(0.0f * m[0] + 0.0f * m[1] + 0.0f * m[2] + m[3], 0.0f * m[4] + 0.0f * m[5] + 0.0f * m[6] + m[7],
               0.0f * m[8] + 0.0f * m[9] + 0.0f * m[10] + m[11]);
}
PMD bool sphere_sample_ctx(const DShadeSphere& s, f3 ctx_p, f3 p_origin, f2 u, ShapeSample* out) {
    const f3 p_center = sphere_center(s);
    if (dist2(p_origin, p_center) <= s.radius * s.radius) {        // inside: by area, converted to solid angle
        sphere_sample(s, u, out);
        f3 wi = out->p - ctx_p;
        if (dot3(wi, wi) == 0.0f) return false;
        wi = norm3(wi);
        out->pdf /= fabsf(dot3(out->n, -wi)) / dist2(ctx_p, out->p);
        return !isinf(out->pdf);
    }
    const float sin_theta_max = s.radius / len3(ctx_p - p_center);
    const float sin2_theta_max = sin_theta_max * sin_theta_max;
    const float cos_theta_max = safe_sqrtf(1.0f - sin2_theta_max);
    float one_minus_cos_theta_max = 1.0f - cos_theta_max;
    float cos_theta = (cos_theta_max - 1.0f) * u.x + 1.0f;
    float sin2_theta = 1.0f - cos_theta * cos_theta;
    if (sin2_theta_max < 0.00068523f) {
        sin2_theta = sin2_theta_max * u.x;
        cos_theta = sqrtf(1.0f - sin2_theta);
        one_minus_cos_theta_max = sin2_theta_max / 2.0f;
    }
    const float cos_alpha = sin2_theta / sin_theta_max + cos_theta * safe_sqrtf(1.0f - sin2_theta / sin_theta_max * sin2_theta_max);
    const float sin_alpha = safe_sqrtf(1.0f - cos_alpha * cos_alpha);
    const float phi = u.y * 2.0f * PM_PI;
    const float st = clampf(sin_alpha, -1.0f, 1.0f), ct = clampf(cos_alpha, -1.0f, 1.0f);
    float sp, cp; cr_sincos_2pi(phi, &sp, &cp);
    const f3 w = mk3(st * cp, st * sp, ct);
    const f3 fz = norm3(p_center - ctx_p);
    f3 fx, fy;                                         // Frame::from_z via Vec3f::coordinate_system (core/vec.rs:743-757)
    if (fabsf(fz.x) > fabsf(fz.y)) fx = mk3(-fz.z, 0.0f, fz.x) / sqrtf(fz.x * fz.x + fz.z * fz.z);
    else fx = mk3(0.0f, fz.z, -fz.y) / sqrtf(fz.y * fz.y + fz.z * fz.z);
    fy = cross3(fz, fx);
    const f3 mw = -w;
    f3 n = mw.x * fx + mw.y * fy + mw.z * fz;
    const f3 pt = p_center + n * s.radius;
    if (s.flags & 1u) n = n * -1.0f;
    const f3 p_err = abs3(pt) * gamma_n(5);
    const Iv ix = iv_err(pt.x, p_err.x), iy = iv_err(pt.y, p_err.y), iz = iv_err(pt.z, p_err.z);
    out->p = mk3(iv_mid(ix), iv_mid(iy), iv_mid(iz));
    out->n = n;
    out->pdf = 1.0f / (2.0f * PM_PI * one_minus_cos_theta_max);
    return true;
}
// the part [t0, t1] of the ray's [0, t_max] inside the medium's bounds
PMD bool medium_segment(const float* b, f3 ro, f3 rd, float t_max, float* t0o, float* t1o) {
    float t0 = 0.0f, t1 = t_max;
    const float o[3] = {ro.x, ro.y, ro.z}, d[3] = {rd.x, rd.y, rd.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / d[a];
        float tn = (b[a] - o[a]) * inv, tf = (b[3 + a] - o[a]) * inv;
        if (tn > tf) { const float x = tn; tn = tf; tf = x; }
        if (tn > t0) t0 = tn;
        if (tf < t1) t1 = tf;
    }
    if (!(t0 < t1)) return false;
    *t0o = t0; *t1o = t1;
    return true;
}

// ---------------------------------------------------------------------------------------- BxDFs
#define PM_BXDF_REFLECTION 1
#define PM_BXDF_TRANSMISSION 2
#define PM_BXDF_DIFFUSE 4
#define PM_BXDF_GLOSSY 8
#define PM_BXDF_SPECULAR 16

struct Spec4 { float v[4]; };
PMD Spec4 spec_c(float c) { Spec4 s; s.v[0] = c; s.v[1] = c; s.v[2] = c; s.v[3] = c; return s; }
PMD bool spec_zero(const Spec4& s) { return s.v[0] == 0.0f && s.v[1] == 0.0f && s.v[2] == 0.0f && s.v[3] == 0.0f; }

PMD float cos2t(f3 w) { return w.z * w.z; }
PMD float sin2t(f3 w) { return fmaxf(1.0f - cos2t(w), 0.0f); }
PMD float sint(f3 w) { return sqrtf(sin2t(w)); }
PMD float tan2t(f3 w) { return sin2t(w) / cos2t(w); }
PMD float cosphi(f3 w) { const float st = sint(w); return st == 0.0f ? 1.0f : clampf(w.x / st, -1.0f, 1.0f); }
PMD float sinphi(f3 w) { const float st = sint(w); return st == 0.0f ? 0.0f : clampf(w.y / st, -1.0f, 1.0f); }

struct TR { float ax, ay; };
PMD bool tr_smooth(TR t) { return fmaxf(t.ax, t.ay) < 0.001f; }
PMD float tr_lambda(TR t, f3 w) {
    const float t2 = tan2t(w);
    if (!isfinite(t2)) return 0.0f;
    const float cx = cosphi(w) * t.ax, sy = sinphi(w) * t.ay;
    const float a2 = cx * cx + sy * sy;
    return (sqrtf(1.0f + a2 * t2) - 1.0f) / 2.0f;
}
PMD float tr_masking(TR t, f3 w) { return 1.0f / (1.0f + tr_lambda(t, w)); }
PMD float tr_g(TR t, f3 wo, f3 wi) { return 1.0f / (1.0f + tr_lambda(t, wo) + tr_lambda(t, wi)); }
PMD float tr_d(TR t, f3 m) {
    const float t2 = tan2t(m);
    if (!isfinite(t2)) return 0.0f;
    const float c2 = cos2t(m); const float c4 = c2 * c2;
    const float a = cosphi(m) / t.ax, b = sinphi(m) / t.ay;
    const float e = t2 * (a * a + b * b);
    const float ope = 1.0f + e;
    return 1.0f / (PM_PI * t.ax * t.ay * c4 * (ope * ope));
}
PMD float tr_pdf(TR t, f3 w, f3 m) { return tr_masking(t, w) / fabsf(w.z) * tr_d(t, m) * fabsf(dot3(w, m)); }
PMD f3 tr_sample_wm(TR t, f3 w, f2 u) {
    f3 wh = norm3(mk3(t.ax * w.x, t.ay * w.y, w.z));
    if (wh.z < 0.0f) wh = -wh;
    const f3 t1 = wh.z < 0.99999f ? norm3(cross3(mk3(0.0f, 0.0f, 1.0f), wh)) : mk3(1.0f, 0.0f, 0.0f);
    const f3 t2 = cross3(wh, t1);
    f2 p = sample_disk_polar(u);
    const float h = sqrtf(1.0f - p.x * p.x);
    p.y = lerpf(h, p.y, (1.0f + wh.z) / 2.0f);
    const float pz = sqrtf(fmaxf(1.0f - (p.x * p.x + p.y * p.y), 0.0f));
    const f3 nh = p.x * t1 + p.y * t2 + pz * wh;
    return norm3(mk3(t.ax * nh.x, t.ay * nh.y, fmaxf(nh.z, 1e-6f)));
}
PMD float fresnel_dielectric(float cti, float eta) {
    cti = clampf(cti, -1.0f, 1.0f);
    if (cti < 0.0f) { eta = 1.0f / eta; cti = -cti; }
    const float s2i = 1.0f - cti * cti;
    const float s2t = s2i / (eta * eta);
    if (s2t >= 1.0f) return 1.0f;
    const float ctt = safe_sqrtf(1.0f - s2t);
    const float rpa = (eta * cti - ctt) / (eta * cti + ctt);
    const float rpe = (cti - eta * ctt) / (cti + eta * ctt);
    return (rpa * rpa + rpe * rpe) / 2.0f;
}
PMD bool refract(f3 wi, f3 n, float eta, f3* wt, float* etap) {
    float cti = dot3(n, wi);
    if (cti < 0.0f) { eta = 1.0f / eta; cti = -cti; n = -n; }
    const float s2i = fmaxf(1.0f - cti * cti, 0.0f);
    const float s2t = s2i / (eta * eta);
    if (s2t >= 1.0f) return false;
    const float ctt = safe_sqrtf(1.0f - s2t);
    *wt = (-wi) / eta + (cti / eta - ctt) * n;
    *etap = eta;
    return true;
}

struct Bsdf { int type; Spec4 refl; float eta; TR tr; f3 fx, fy, fz; };
PMD int bsdf_flags(const Bsdf& b) {
    if (b.type == PM_MAT_DIFFUSE) return PM_BXDF_DIFFUSE | PM_BXDF_REFLECTION;
    const int f = b.eta == 1.0f ? PM_BXDF_TRANSMISSION : (PM_BXDF_REFLECTION | PM_BXDF_TRANSMISSION);
    return f | (tr_smooth(b.tr) ? PM_BXDF_SPECULAR : PM_BXDF_GLOSSY);
}
PMD f3 to_local(const Bsdf& b, f3 v) { return mk3(dot3(v, b.fx), dot3(v, b.fy), dot3(v, b.fz)); }
PMD f3 from_local(const Bsdf& b, f3 v) { return v.x * b.fx + v.y * b.fy + v.z * b.fz; }

PMD Spec4 bxdf_eval(const Bsdf& b, f3 wo, f3 wi) {
    if (b.type == PM_MAT_DIFFUSE) {
        if (wo.z * wi.z > 0.0f) { Spec4 r; for (int i = 0; i < 4; ++i) r.v[i] = b.refl.v[i] * PM_INV_PI; return r; }
        return spec_c(0.0f);
    }
    if (b.eta == 1.0f || tr_smooth(b.tr)) return spec_c(0.0f);
    const float cto = wo.z, cti = wi.z;
    const bool refl = cto * cti > 0.0f;
    const float etap = refl ? 1.0f : (cto > 0.0f ? b.eta : 1.0f / b.eta);
    f3 wm = wi * etap + wo;
    if (cti == 0.0f || cto == 0.0f || dot3(wm, wm) == 0.0f) return spec_c(0.0f);
    wm = face_fwd(norm3(wm), mk3(0.0f, 0.0f, 1.0f));
    const float idm = dot3(wi, wm), odm = dot3(wo, wm);
    if (idm * cti < 0.0f || odm * cto < 0.0f) return spec_c(0.0f);
    const float F = fresnel_dielectric(odm, b.eta);
    if (refl) return spec_c(tr_d(b.tr, wm) * tr_g(b.tr, wo, wi) * F / fabsf(4.0f * cti * cto));
    const float s = idm + odm / etap;
    const float denom = s * s * cti * cto;
    float ft = tr_d(b.tr, wm) * (1.0f - F) * tr_g(b.tr, wo, wi) * fabsf(idm * odm / denom);
    ft /= etap * etap;
    return spec_c(ft);
}
PMD Spec4 bsdf_eval(const Bsdf& b, f3 wo_r, f3 wi_r) {
    const f3 wi = to_local(b, wi_r), wo = to_local(b, wo_r);
    if (wo.z != 0.0f) return bxdf_eval(b, wo, wi);
    return spec_c(0.0f);
}
struct BsdfSample { Spec4 value; f3 wi; float pdf; int flags; };
PMD bool bxdf_sample(const Bsdf& b, f3 wo, float uc, f2 u, BsdfSample* bs) {
    if (b.type == PM_MAT_DIFFUSE) {
        f3 wi = sample_cosine_hemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        bs->pdf = fabsf(wi.z) * PM_INV_PI;
        for (int i = 0; i < 4; ++i) bs->value.v[i] = b.refl.v[i] * PM_INV_PI;
        bs->wi = wi; bs->flags = PM_BXDF_DIFFUSE | PM_BXDF_REFLECTION;
        return true;
    }
    if (b.eta == 1.0f || tr_smooth(b.tr)) {
        const float R = fresnel_dielectric(wo.z, b.eta), T = 1.0f - R;
        if (R == 0.0f && T == 0.0f) return false;
        if (uc < R / (R + T)) {
            const f3 wi = mk3(-wo.x, -wo.y, wo.z);
            bs->value = spec_c(R / fabsf(wi.z)); bs->wi = wi; bs->pdf = R / (R + T);
            bs->flags = PM_BXDF_SPECULAR | PM_BXDF_REFLECTION;
            return true;
        }
        f3 wi; float etap;
        if (!refract(wo, mk3(0.0f, 0.0f, 1.0f), b.eta, &wi, &etap)) return false;
        float ft = T / fabsf(wi.z);
        ft /= etap * etap;
        bs->value = spec_c(ft); bs->wi = wi; bs->pdf = T / (R + T);
        bs->flags = PM_BXDF_SPECULAR | PM_BXDF_TRANSMISSION;
        return true;
    }
    const f3 wm = tr_sample_wm(b.tr, wo, u);
    const float R = fresnel_dielectric(dot3(wo, wm), b.eta), T = 1.0f - R;
    if (R == 0.0f && T == 0.0f) return false;
    if (uc < R / (R + T)) {
        const f3 wi = (-wo) + (2.0f * dot3(wo, wm)) * wm;
        if (!(wo.z * wi.z > 0.0f)) return false;
        bs->pdf = tr_pdf(b.tr, wo, wm) / (4.0f * fabsf(dot3(wo, wi))) * (R / (R + T));
        bs->value = spec_c(tr_d(b.tr, wm) * tr_g(b.tr, wo, wi) * R / (4.0f * wi.z * wo.z));
        bs->wi = wi; bs->flags = PM_BXDF_GLOSSY | PM_BXDF_REFLECTION;
        return true;
    }
    f3 wi; float etap;
    if (!refract(wo, wm, b.eta, &wi, &etap)) return false;
    if (wo.z * wi.z > 0.0f || wi.z == 0.0f) return false;
    const float idm = dot3(wi, wm), odm = dot3(wo, wm);
    const float s = idm + odm / etap;
    const float denom = s * s;
    const float dwm_dwi = fabsf(idm) / denom;
    bs->pdf = tr_pdf(b.tr, wo, wm) * dwm_dwi * (T / (R + T));
    float ft = T * tr_d(b.tr, wm) * tr_g(b.tr, wo, wi) * fabsf(idm * odm / (wi.z * wo.z * denom));
    ft /= etap * etap;
    bs->value = spec_c(ft); bs->wi = wi; bs->flags = PM_BXDF_GLOSSY | PM_BXDF_TRANSMISSION;
    return true;
}
PMD bool bsdf_sample(const Bsdf& b, f3 wo_r, float uc, f2 u, BsdfSample* bs) {
    const f3 wo = to_local(b, wo_r);
    if (wo.z == 0.0f) return false;
    if (!bxdf_sample(b, wo, uc, u, bs)) return false;
    if (spec_zero(bs->value) || bs->pdf == 0.0f || bs->wi.z == 0.0f) return false;
    bs->wi = from_local(b, bs->wi);
    return true;
}

}  // namespace pmd
