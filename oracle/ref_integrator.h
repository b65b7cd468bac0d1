// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header).
//
// CPU restatement of the reference's per-sample pipeline and integrators, from the flattened scene
// description (the same information a host hands to the C ABI):
//   integrators/base.rs:82-219      image_tile_render / eval_pixel_sample
//   integrators/simple_path.rs:105-269, integrators/random_walk.rs:106-173
//   sampling/sampler.rs:37-52       get_camera_sample;  imaging/filter.rs  (Box/Triangle/Gaussian+FilterSampler)
//   sampling/routines.rs            warps, PiecewiseConstant1D/2D
//   camera/perspective.rs:130-189, core/transform.rs:446-466 (mul_ray)
//   core/interaction.rs, shapes/{sphere,bilinear_patch}.rs interaction_from_intersection
//   reflection/{base,diffuse,dielectric}.rs, materials/material.rs:122-195
//   lights/{point,directional,infinite}.rs, sampling/light_sampler.rs:33-44
//   camera/film.rs:160-192,261-279, camera/sensor.rs:114-121
// Ray differentials are not restated: they only feed SurfaceInteraction::compute_differentials, whose
// output (texture footprints) no constant texture reads (materials/textures.rs:63-90); no random number
// is drawn for them, so dropping them changes no pixel.
#pragma once
#include <thread>
#include <vector>

#include "ref_render.h"
#include "ref_scene.h"

namespace ora {

static const int NSP = 4;   // N_SPECTRUM_SAMPLES, spectrum.rs:25
struct Spec4 { float v[NSP]; };
static inline Spec4 spec_const(float c) { Spec4 s; for (int i = 0; i < NSP; ++i) s.v[i] = c; return s; }
static inline bool spec_all_zero(const Spec4& s) { for (int i = 0; i < NSP; ++i) if (s.v[i] != 0.0f) return false; return true; }

struct Wavelengths { float lambda[NSP]; float pdf[NSP]; };
// spectrum.rs:870-883
static inline Wavelengths sample_visible(float u) {
    Wavelengths w;
    for (int i = 0; i < NSP; ++i) {
        float up = u + (float)i / (float)NSP;
        if (up > 1.0f) up -= 1.0f;
        w.lambda[i] = sample_visible_wavelengths(up);
    }
    for (int i = 0; i < NSP; ++i) w.pdf[i] = visible_wavelengths_pdf(w.lambda[i]);
    return w;
}
// spectrum.rs:893-905
static inline void terminate_secondary(Wavelengths* w) {
    bool term = true;
    for (int i = 1; i < NSP; ++i) if (w->pdf[i] != 0.0f) term = false;
    if (!term) { w->pdf[0] /= (float)NSP; for (int i = 1; i < NSP; ++i) w->pdf[i] = 0.0f; }
}

// ---- spectra (flattened) ----------------------------------------------------------------------
enum { SPEC_CONSTANT = 0, SPEC_DENSE = 1, SPEC_RGB_ALBEDO = 2, SPEC_RGB_UNBOUNDED = 3, SPEC_RGB_ILLUMINANT = 4, SPEC_BLACKBODY = 5 };
struct OSpectrum { int32_t kind; float a, b, c, d; const float* table; };   // see include/pbrt_mi355.h PmSpectrum

static inline float dense_at(const float* table, float lambda) {   // spectrum.rs:225-239, 360..830
    float r = std::round(lambda);                                    // Rust round(): half away from zero
    long li = (long)r;
    if (li >= 360 && li <= 830) return table[li - 360];
    return 0.0f;
}
static inline float sigmoid_poly(float c0, float c1, float c2, float lambda) {   // color_space.rs:243-280
    const float c[3] = {c2, c1, c0};
    float x = evaluate_polynomial(lambda, c, 3);
    if (std::isinf(x)) return x > 0.0f ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * std::sqrt(1.0f + x * x));
}
static inline float blackbody(float lambda, float temp) {   // spectrum.rs:378-400
    if (temp <= 0.0f) return 0.0f;
    const float C = 299792458.0f, H = 6.6260697e-34f, KB = 1.3806488e-23f;
    float l = lambda * 1e-9f;
    float l5 = l * l * l * l * l;    // powi(5): llvm.powi expands by repeated squaring: ((l*l)*(l*l))*l
    l5 = ((l * l) * (l * l)) * l;
    return (2.0f * H * C * C) / (l5 * (t_exp((H * C) / (l * KB * temp)) - 1.0f));
}
static inline float spectrum_at(const OSpectrum& s, float lambda) {
    switch (s.kind) {
        case SPEC_CONSTANT: return s.a;
        case SPEC_DENSE: return dense_at(s.table, lambda);
        case SPEC_RGB_ALBEDO: return sigmoid_poly(s.a, s.b, s.c, lambda);
        case SPEC_RGB_UNBOUNDED: return s.d * sigmoid_poly(s.a, s.b, s.c, lambda);
        case SPEC_RGB_ILLUMINANT: return s.d * sigmoid_poly(s.a, s.b, s.c, lambda) * dense_at(s.table, lambda);
        case SPEC_BLACKBODY: return blackbody(lambda, s.a) * s.b;
    }
    return 0.0f;
}
static inline Spec4 spectrum_sample(const OSpectrum& s, const Wavelengths& w) {
    Spec4 r; for (int i = 0; i < NSP; ++i) r.v[i] = spectrum_at(s, w.lambda[i]); return r;
}

// ---- flattened scene description ----------------------------------------------------------------
enum { MAT_DIFFUSE = 0, MAT_DIELECTRIC = 1 };
enum { TEX_NONE = 0, TEX_CHECKERBOARD = 1 };
// texture / su sv du dv / spec2: the checkerboard extension (see include/pbrt_mi355.h PmMaterial)
struct OMaterial { int32_t type; OSpectrum spec; float u_rough, v_rough; int32_t remap_roughness;
                   int32_t texture; float su, sv, du, dv; OSpectrum spec2; };
enum { LIGHT_POINT = 0, LIGHT_DIRECTIONAL = 1, LIGHT_UNIFORM_INFINITE = 2, LIGHT_DIFFUSE_AREA = 3 };
struct OLight { int32_t type; float scale; float render_from_light[16]; const float* spectrum; uint32_t prim_id; int32_t two_sided; };
// media/medium.rs:41-69 HomogeneousMedium: sigma_a / sigma_s densely sampled and already scaled by sigma_scale, HG asymmetry g;
// bounds (extension, replaces MediumInterface): the medium fills this axis-aligned box of render space (+-inf = everywhere)
struct OMedium { const float* sigma_a; const float* sigma_s; float g; float bounds[6]; };
struct OCamera { float camera_from_raster[16]; float render_from_camera[16]; float lens_radius, focal_distance; };
enum { FILTER_BOX = 0, FILTER_TRIANGLE = 1, FILTER_GAUSSIAN = 2 };
struct OFilm {
    int32_t full_res[2]; int32_t bounds[4];     // pixel_bounds x0,y0,x1,y1
    float max_component_value, imaging_ratio;
    float out_rgb_from_sensor[9];
    const float* sensor_bar;                    // r_bar,g_bar,b_bar: 3 x 471
    int32_t filter_kind; float filter_radius[2];
    int32_t filter_nx, filter_ny; const float* filter_vals;   // tabulated filter (Gaussian), row-major
};
enum { INTEGRATOR_SIMPLE_PATH = 0, INTEGRATOR_RANDOM_WALK = 1, INTEGRATOR_VOL_PATH = 2 };
struct ORender {
    int32_t integrator, max_depth, sample_lights, sample_bsdf;
    int32_t spp; uint32_t seed;
    int32_t window[4];      // x0,y0,x1,y1 (inside pixel_bounds)
    int32_t s0, s1;         // sample-index window
};

// ---- sampling warps (sampling/routines.rs) --------------------------------------------------------
struct P2 { float x, y; };
static inline V3 sample_uniform_sphere(P2 u) {          // :260-266
    float z = 1.0f - 2.0f * u.x;
    float r = safe_sqrt(1.0f - z * z);
    float phi = 2.0f * PI_F * u.y;
    return V3(r * t_cos(phi), r * t_sin(phi), z);
}
static inline V3 sample_uniform_hemisphere(P2 u) {      // :270-276
    float z = u.x;
    float r = safe_sqrt(1.0f - z * z);
    float phi = 2.0f * PI_F * u.y;
    return V3(r * t_cos(phi), r * t_sin(phi), z);
}
static inline P2 sample_uniform_disk_concentric(P2 u) {  // :278-298
    P2 o{2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f};
    if (o.x == 0.0f && o.y == 0.0f) return P2{0.0f, 0.0f};
    float r, theta;
    if (std::fabs(o.x) > std::fabs(o.y)) { r = o.x; theta = FRAC_PI_4 * (o.y / o.x); }
    else { r = o.y; theta = FRAC_PI_2 - FRAC_PI_4 * (o.x / o.y); }
    return P2{t_cos(theta) * r, t_sin(theta) * r};
}
static inline P2 sample_uniform_disk_polar(P2 u) {       // :300-305
    float r = std::sqrt(u.x);
    float theta = 2.0f * PI_F * u.y;
    return P2{r * t_cos(theta), r * t_sin(theta)};
}
static inline V3 sample_cosine_hemisphere(P2 u) {        // :329-334
    P2 d = sample_uniform_disk_concentric(u);
    float z = safe_sqrt(1.0f - d.x * d.x - d.y * d.y);
    return V3(d.x, d.y, z);
}
static inline int find_interval(int size, const std::vector<float>& cdf, float u) {   // util/routines.rs:1-32, pred cdf[i] <= u
    if (size < 2) return 0;
    int left = 0, right = size - 1;
    while (left < right) {
        int mid = left + (right - left) / 2;
        if (cdf[mid] <= u) left = mid + 1; else right = mid;
    }
    if (left == 0) return 0;
    if (left >= size - 1) return size - 2;
    return left - 1;
}
struct PC1D {                                            // :341-427
    std::vector<float> func, cdf; float mn, mx, integral;
    void init(const float* f, int n, float mn_, float mx_) {
        mn = mn_; mx = mx_;
        func.resize(n); for (int i = 0; i < n; ++i) func[i] = std::fabs(f[i]);
        cdf.resize(n + 1); cdf[0] = 0.0f;
        for (int i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + func[i - 1] * (mx - mn) / (float)n;
        integral = cdf[n];
        if (integral == 0.0f) { for (int i = 1; i <= n; ++i) cdf[i] = (float)i / (float)n; }
        else { for (int i = 0; i <= n; ++i) cdf[i] /= integral; }
    }
    void sample(float u, float* value, float* pdf, int* offset) const {
        int o = find_interval((int)cdf.size(), cdf, u);
        float du = u - cdf[o];
        if (cdf[o + 1] - cdf[o] > 0.0f) du /= cdf[o + 1] - cdf[o];
        *pdf = integral > 0.0f ? func[o] / integral : 0.0f;
        *value = lerpf(mn, mx, ((float)o + du) / (float)func.size());
        *offset = o;
    }
};
// sample_discrete / sample_linear / sample_tent (:41-132) for the triangle filter
static inline float sample_linear(float u, float a, float b) {
    if (u == 0.0f || a == 0.0f) return 0.0f;
    float x = u * (a + b) / (a + std::sqrt(lerpf(a * a, b * b, u)));
    return rmin(x, ONE_MINUS_EPSILON);
}
static inline float sample_tent(float u, float r) {
    // sample_discrete(&[0.5, 0.5], u)
    float sum_w = 0.0f + 0.5f + 0.5f;
    float up = u * sum_w;
    if (up == sum_w) up = next_float_down(up);
    int off = 0; float sum = 0.0f; const float w[2] = {0.5f, 0.5f};
    while (sum + w[off] <= up) { sum += w[off]; ++off; }
    float ur = rmin((up - sum) / w[off], ONE_MINUS_EPSILON);
    if (off == 0) return -r + r * sample_linear(ur, 0.0f, 1.0f);
    return r * sample_linear(ur, 1.0f, 0.0f);
}

// sampling/routines.rs:219-246.  B14 REPRODUCED: sample_bilinear passes (y, w0, w2) to a lerp whose signature is (a, b, t).
static inline float bilinear_pdf(P2 p, const float w[4]) {
    float w_sum = 0.0f; for (int i = 0; i < 4; ++i) w_sum += w[i];
    if (p.x < 0.0f || p.x > 1.0f || p.y < 0.0f || p.y > 1.0f) return 0.0f;
    if (w_sum == 0.0f) return 1.0f;
    return 4.0f * ((1.0f - p.x) * (1.0f - p.y) * w[0] + p.x * (1.0f - p.y) * w[1] + p.y * (1.0f - p.x) * w[2] + p.x * p.y * w[3]) / w_sum;
}
static inline P2 sample_bilinear(P2 u, const float w[4]) {
    float y = sample_linear(u.y, w[0] + w[1], w[2] + w[3]);
    float x = sample_linear(u.x, lerpf(y, w[0], w[2]), lerpf(y, w[1], w[3]));
    return P2{x, y};
}
static inline float sample_exponential(float u, float a) { return -t_log(1.0f - u) / a; }   // routines.rs:148-150

// media/phase_function.rs:57-80
static inline float henyey_greenstein(float cos_theta, float g) {
    float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return INV_4_PI * (1.0f - g * g) / (denom * safe_sqrt(denom));
}
static inline V3 sample_henyey_greenstein(V3 wo, float g, P2 u, float* pdf) {
    float cos_theta;
    if (std::fabs(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u.x;
    else {
        float q = (1.0f - g * g) / (1.0f + g - 2.0f * g * u.x);
        cos_theta = -1.0f / (2.0f * g) * (1.0f + g * g - q * q);       // powi(2) == x * x
    }
    float sin_theta = safe_sqrt(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * PI_F * u.y;
    V3 x, y;                                                          // Frame::from_z(wo): (z, x, y) = wo.coordinate_system()
    coordinate_system(wo, &x, &y);
    float st = clampf(sin_theta, -1.0f, 1.0f), ct = clampf(cos_theta, -1.0f, 1.0f);     // spherical_direction, vec_math.rs:40-46
    V3 l(st * t_cos(phi), st * t_sin(phi), ct);
    *pdf = henyey_greenstein(cos_theta, g);
    return l.x * x + l.y * y + l.z * wo;                              // Frame::from_local
}

// ---- shape sampling of bilinear patches (shapes/bilinear_patch.rs:256-353, 739-773) ----------------------------------
static inline bool v3_eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline bool patch_is_rectangle(const Patch& p) {
    if (v3_eq(p.p00, p.p01) || v3_eq(p.p01, p.p11) || v3_eq(p.p11, p.p10) || v3_eq(p.p10, p.p00)) return false;
    V3 n = normalized(cross(p.p10 - p.p00, p.p01 - p.p00));
    if (absdot(normalized(p.p11 - p.p00), n) > 1e-5f) return false;
    V3 c = (p.p00 + p.p10 + p.p01 + p.p11) / 4.0f;
    float d2[4] = {distance_squared(p.p00, c), distance_squared(p.p10, c), distance_squared(p.p01, c), distance_squared(p.p11, c)};
    for (int i = 0; i < 4; ++i) if (std::fabs(d2[i] - d2[0]) / d2[0] > 1e-4f) return false;
    return true;
}
struct ShapeSample { V3fi pi; V3 n; float pdf; };
// BilinearPatch::sample (:260-318); no image distribution, no mesh uvs
static inline bool patch_sample(const Patch& p, P2 u, ShapeSample* out) {
    P2 uv; float pdf;
    if (!patch_is_rectangle(p)) {
        const float w[4] = {length(cross(p.p10 - p.p00, p.p01 - p.p00)), length(cross(p.p10 - p.p00, p.p11 - p.p10)),
                            length(cross(p.p01 - p.p00, p.p11 - p.p01)), length(cross(p.p11 - p.p10, p.p11 - p.p01))};
        uv = sample_bilinear(u, w); pdf = bilinear_pdf(uv, w);
    } else { uv = u; pdf = 1.0f; }
    V3 pu0 = vlerp(p.p00, p.p01, uv.y), pu1 = vlerp(p.p10, p.p11, uv.y);
    V3 pt = vlerp(pu0, pu1, uv.x);
    V3 dpdu = pu1 - pu0;
    V3 dpdv = vlerp(p.p01, p.p11, uv.x) - vlerp(p.p00, p.p10, uv.x);
    if (length_squared(dpdu) == 0.0f || length_squared(dpdv) == 0.0f) return false;
    V3 n = normalized(cross(dpdu, dpdv));
    if (p.has_normals) {
        V3 ns = vlerp(vlerp(p.n00, p.n01, uv.y), vlerp(p.n10, p.n11, uv.y), uv.x);
        n = face_forward(n, ns);
    } else if (p.flip_normal) n = -n;
    V3 pabs = vabs(p.p00) + vabs(p.p01) + vabs(p.p10) + vabs(p.p11);
    out->pi = V3fi::with_err(pt, gamma_n(6) * pabs);
    out->n = n;
    out->pdf = pdf / length(cross(dpdu, dpdv));
    return true;
}
// BilinearPatch::sample_with_context (:320-352), the area-sampling branch.  The spherical-rectangle branch of rectangles ends
// in sample_spherical_rectangle, which is todo!() upstream (sampling/routines.rs:307-327): every patch takes this branch.
static inline bool patch_sample_ctx(const Patch& p, V3 ref_p, P2 u, ShapeSample* out) {
    if (!patch_sample(p, u, out)) return false;
    V3 wi = out->pi.midpoints() - ref_p;
    if (length_squared(wi) == 0.0f) return false;
    wi = normalized(wi);
    out->pdf /= absdot(out->n, -wi) / distance_squared(ref_p, out->pi.midpoints());
    if (std::isinf(out->pdf)) return false;
    return true;
}

// ---- shape sampling of spheres (shapes/sphere.rs:100-213) -------------------------------------------------------------
// Sphere::sample (:104-133): a uniform point of the FULL sphere (z / phi clipping is ignored there), pdf = 1 / area() where
// area() = phi_max * radius * (z_max - z_min) (:100-102).  uv is computed upstream but read by nobody here.
static inline bool sphere_sample(const Sphere& s, P2 u, ShapeSample* out) {
    V3 p_obj = V3(0.0f, 0.0f, 0.0f) + s.radius * sample_uniform_sphere(u);
    p_obj = p_obj * (s.radius / distance(p_obj, V3(0.0f, 0.0f, 0.0f)));
    V3 p_err = gamma_n(5) * vabs(p_obj);
    V3 n = normalized(xf_normal(s.object_from_render, p_obj));        // render_from_object * Normal3f: transposed inverse
    if (s.reverse_orientation) n = n * -1.0f;
    out->pi = xf_point_fi(s.render_from_object, V3fi::with_err(p_obj, p_err));
    out->n = n;
    out->pdf = 1.0f / (s.phi_max * s.radius * (s.z_max - s.z_min));
    return true;
}
// Sphere::sample_with_context (:135-213).  The context is a surface point (pi with its error, geometric normal n: the origin
// used for the inside / outside decision is offset along n, shapes/base.rs:93-99) or - has_n false - a point in a medium,
// where upstream would unwrap a None normal; such a context takes the point itself (extension, as pbrt-v4 does).
static inline V3 sphere_center(const Sphere& s) { return xf_point(s.render_from_object, V3(0.0f, 0.0f, 0.0f)); }
// ctx_p: ctx.pi.midpoints(); p_origin: ctx.offset_ray_origin_towards(p_center), or ctx_p itself for a point in a medium
static inline bool sphere_sample_ctx(const Sphere& s, V3 ctx_p, V3 p_origin, P2 u, ShapeSample* out) {
    V3 p_center = sphere_center(s);
    if (distance_squared(p_origin, p_center) <= s.radius * s.radius) {         // inside: by area, converted to solid angle
        if (!sphere_sample(s, u, out)) return false;
        V3 wi = out->pi.midpoints() - ctx_p;
        if (length_squared(wi) == 0.0f) return false;
        wi = normalized(wi);
        out->pdf /= absdot(out->n, -wi) / distance_squared(ctx_p, out->pi.midpoints());
        if (std::isinf(out->pdf)) return false;
        return true;
    }
    // outside: uniformly within the cone the sphere subtends
    float sin_theta_max = s.radius / distance(ctx_p, p_center);
    float sin2_theta_max = sin_theta_max * sin_theta_max;
    float cos_theta_max = safe_sqrt(1.0f - sin2_theta_max);
    float one_minus_cos_theta_max = 1.0f - cos_theta_max;
    float cos_theta = (cos_theta_max - 1.0f) * u.x + 1.0f;
    float sin2_theta = 1.0f - cos_theta * cos_theta;
    if (sin2_theta_max < 0.00068523f) {               // < sin^2(1.5 deg): Taylor expansion
        sin2_theta = sin2_theta_max * u.x;
        cos_theta = std::sqrt(1.0f - sin2_theta);
        one_minus_cos_theta_max = sin2_theta_max / 2.0f;
    }
    // (as written upstream: sin2_theta / sin_theta_max * sin2_theta_max under the root, :180-181)
    float cos_alpha = sin2_theta / sin_theta_max + cos_theta * safe_sqrt(1.0f - sin2_theta / sin_theta_max * sin2_theta_max);
    float sin_alpha = safe_sqrt(1.0f - cos_alpha * cos_alpha);
    float phi = u.y * 2.0f * PI_F;
    float st = clampf(sin_alpha, -1.0f, 1.0f), ct = clampf(cos_alpha, -1.0f, 1.0f);      // spherical_direction, vec_math.rs:40-46
    V3 w(st * t_cos(phi), st * t_sin(phi), ct);
    V3 fz = normalized(p_center - ctx_p), fx, fy;                      // Frame::from_z (frame.rs:62-65)
    coordinate_system(fz, &fx, &fy);
    V3 mw = -w;
    V3 n = mw.x * fx + mw.y * fy + mw.z * fz;                          // from_local (frame.rs:68-71)
    V3 pt = p_center + s.radius * n;
    if (s.reverse_orientation) n = n * -1.0f;
    V3 p_err = gamma_n(5) * vabs(pt);
    out->pi = V3fi::with_err(pt, p_err);
    out->n = n;
    out->pdf = 1.0f / (2.0f * PI_F * one_minus_cos_theta_max);
    return true;
}

struct FilterSamplerO {                                  // imaging/filter.rs:177-225 + routines.rs:436-503
    int kind; float rx, ry; int nx = 0, ny = 0; std::vector<float> vals; std::vector<PC1D> cond; PC1D marg;
    void init(const OFilm& f) {
        kind = f.filter_kind; rx = f.filter_radius[0]; ry = f.filter_radius[1];
        if (kind != FILTER_GAUSSIAN) return;
        nx = f.filter_nx; ny = f.filter_ny; vals.assign(f.filter_vals, f.filter_vals + (size_t)nx * ny);
        cond.resize(ny);
        std::vector<float> mf(ny);
        for (int v = 0; v < ny; ++v) { cond[v].init(&vals[(size_t)v * nx], nx, -rx, rx); mf[v] = cond[v].integral; }
        marg.init(mf.data(), ny, -ry, ry);
    }
    void sample(P2 u, P2* p, float* weight) const {
        if (kind == FILTER_BOX) { p->x = lerpf(-rx, rx, u.x); p->y = lerpf(-ry, ry, u.y); *weight = 1.0f; return; }
        if (kind == FILTER_TRIANGLE) { p->x = sample_tent(u.x, rx); p->y = sample_tent(u.y, ry); *weight = 1.0f; return; }
        float d1, pdf1, d0, pdf0; int v, uo;
        marg.sample(u.y, &d1, &pdf1, &v);
        cond[v].sample(u.x, &d0, &pdf0, &uo);
        p->x = d0; p->y = d1;
        *weight = vals[(size_t)v * nx + uo] / (pdf0 * pdf1);
    }
};

// ---- surface interaction (only the fields the integrators read) -------------------------------------
struct SurfInt { V3fi pi; V3 n, ns, dpdus; uint32_t prim_id; float u = 0.0f, v = 0.0f; };

static inline SurfInt sphere_interaction(const Sphere& s, const Hit& h) {   // sphere.rs:371-442 + interaction.rs:99-131,354-369
    V3 ph = h.p_obj;
    float z_radius = std::sqrt(ph.x * ph.x + ph.y * ph.y);
    float cos_phi = ph.x / z_radius, sin_phi = ph.y / z_radius;
    float cos_theta = ph.z / s.radius;
    V3 dpdu(-s.phi_max * ph.y, s.phi_max * ph.x, 0.0f);
    float sin_theta = safe_sqrt(1.0f - cos_theta * cos_theta);
    V3 dpdv = (s.theta_z_max - s.theta_z_min) * V3(ph.z * cos_phi, ph.z * sin_phi, -s.radius * sin_theta);
    V3 perr = gamma_n(5) * vabs(ph);
    bool flip = s.reverse_orientation ^ swaps_handedness(s.render_from_object);
    V3 n = normalized(cross(dpdu, dpdv));
    if (flip) n = n * -1.0f;
    V3 ns = n;
    SurfInt si;
    // SurfaceInteraction::transform
    si.pi = xf_point_fi(s.render_from_object, V3fi::with_err(ph, perr));
    si.n = xf_normal(s.object_from_render, normalized(n));
    si.dpdus = xf_vec(s.render_from_object, dpdu);
    si.ns = face_forward(xf_normal(s.object_from_render, ns), si.n);
    si.prim_id = h.prim_id;
    return si;
}
// Transform::rotate_from_to (core/transform.rs:180-201), upper 3x3 only (it is applied to vectors)
static inline void rotate_from_to(V3 from, V3 to, float m[3][3]) {
    V3 refl;
    if (std::fabs(from.x) < 0.72f && std::fabs(to.x) < 0.72f) refl = V3(1, 0, 0);
    else if (std::fabs(from.y) < 0.72f && std::fabs(to.y) < 0.72f) refl = V3(0, 1, 0);
    else refl = V3(0, 0, 1);
    V3 u = refl - from, v = refl - to;
    const float ua[3] = {u.x, u.y, u.z}, va[3] = {v.x, v.y, v.z};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            m[i][j] = (i == j ? 1.0f : 0.0f) - 2.0f / dot(u, u) * ua[i] * ua[j] - 2.0f / dot(v, v) * va[i] * va[j] +
                      4.0f * dot(u, v) / (dot(u, u) * dot(v, v)) * va[i] * ua[j];
}
static inline SurfInt patch_interaction(const Patch& p, const Hit& h) {     // bilinear_patch.rs:86-167
    float u = h.u, v = h.v;
    V3 pt = vlerp(vlerp(p.p00, p.p01, v), vlerp(p.p10, p.p11, v), u);
    V3 dpdu = vlerp(p.p10, p.p11, v) - vlerp(p.p00, p.p01, v);
    V3 dpdv = vlerp(p.p01, p.p11, u) - vlerp(p.p00, p.p10, u);
    // extension shared with the product (pm_shade.hpp): a triangle is the degenerate patch (v0,v1,v2,v2); exactly on its
    // collapsed edge dp/du is the zero vector and the reference's normal would be NaN
    if (dpdu.x == 0.0f && dpdu.y == 0.0f && dpdu.z == 0.0f) dpdu = p.p10 - p.p00;
    V3 pabs = vabs(p.p00) + vabs(p.p10) + vabs(p.p01) + vabs(p.p11);
    V3 perr = gamma_n(6) * pabs;
    V3 n = normalized(cross(dpdu, dpdv));
    if (p.flip_normal) n = n * -1.0f;
    SurfInt si;
    si.pi = V3fi::with_err(pt, perr);
    si.n = n; si.ns = n; si.dpdus = dpdu; si.prim_id = h.prim_id;
    if (p.has_normals) {      // bilinear_patch.rs:150-164 + SurfaceInteraction::set_shading_geometry(.., true) (interaction.rs:134-158)
        V3 ns = vlerp(vlerp(p.n00, p.n01, v), vlerp(p.n10, p.n11, v), u);
        if (length_squared(ns) > 0.0f) {
            ns = normalized(ns);
            float rot[3][3];
            rotate_from_to(normalized(si.n), ns, rot);
            si.n = face_forward(si.n, ns);
            si.ns = ns;
            si.dpdus = V3(rot[0][0] * dpdu.x + rot[0][1] * dpdu.y + rot[0][2] * dpdu.z,
                          rot[1][0] * dpdu.x + rot[1][1] * dpdu.y + rot[1][2] * dpdu.z,
                          rot[2][0] * dpdu.x + rot[2][1] * dpdu.y + rot[2][2] * dpdu.z);
        }
    }
    return si;
}

// ---- BxDFs (local shading space) ---------------------------------------------------------------------
enum { BXDF_REFLECTION = 1, BXDF_TRANSMISSION = 2, BXDF_DIFFUSE = 4, BXDF_GLOSSY = 8, BXDF_SPECULAR = 16 };
struct BSDFSample { Spec4 value; V3 wi; float pdf; int flags; };

static inline float cos2_theta(V3 w) { return w.z * w.z; }
static inline float sin2_theta(V3 w) { return rmax(1.0f - cos2_theta(w), 0.0f); }
static inline float sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
static inline float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
static inline float cos_phi(V3 w) { float st = sin_theta(w); return st == 0.0f ? 1.0f : clampf(w.x / st, -1.0f, 1.0f); }
static inline float sin_phi(V3 w) { float st = sin_theta(w); return st == 0.0f ? 0.0f : clampf(w.y / st, -1.0f, 1.0f); }

struct TrowbridgeReitz {          // dielectric.rs:416-512
    float ax, ay;
    bool effectively_smooth() const { return rmax(ax, ay) < 0.001f; }
    float lambda(V3 w) const {
        float t2 = tan2_theta(w);
        if (!std::isfinite(t2)) return 0.0f;
        float cx = cos_phi(w) * ax, sy = sin_phi(w) * ay;
        float alpha2 = cx * cx + sy * sy;
        return (std::sqrt(1.0f + alpha2 * t2) - 1.0f) / 2.0f;
    }
    float masking(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
    float masking_shadowing(V3 wo, V3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    float density(V3 m) const {
        float t2 = tan2_theta(m);
        if (!std::isfinite(t2)) return 0.0f;
        float c2 = cos2_theta(m); float c4 = c2 * c2;
        float a = cos_phi(m) / ax, b = sin_phi(m) / ay;
        float e = t2 * (a * a + b * b);
        float ope = 1.0f + e;
        return 1.0f / (PI_F * ax * ay * c4 * (ope * ope));
    }
    float pdf(V3 w, V3 m) const { return masking(w) / std::fabs(w.z) * density(m) * absdot(w, m); }
    V3 sample_microfacet_n(V3 w, P2 u) const {
        V3 wh = normalized(V3(ax * w.x, ay * w.y, w.z));
        if (wh.z < 0.0f) wh = -wh;
        V3 t1 = wh.z < 0.99999f ? normalized(cross(V3(0, 0, 1), wh)) : V3(1, 0, 0);
        V3 t2 = cross(wh, t1);
        P2 p = sample_uniform_disk_polar(u);
        float h = std::sqrt(1.0f - p.x * p.x);
        p.y = lerpf(h, p.y, (1.0f + wh.z) / 2.0f);
        float pz = std::sqrt(rmax(1.0f - (p.x * p.x + p.y * p.y), 0.0f));
        V3 nh = p.x * t1 + p.y * t2 + pz * wh;
        return normalized(V3(ax * nh.x, ay * nh.y, rmax(nh.z, 1e-6f)));
    }
};
static inline float fresnel_dielectric(float cti, float eta) {   // dielectric.rs:514-539
    cti = clampf(cti, -1.0f, 1.0f);
    if (cti < 0.0f) { eta = 1.0f / eta; cti = -cti; }
    float s2i = 1.0f - cti * cti;
    float s2t = s2i / (eta * eta);
    if (s2t >= 1.0f) return 1.0f;
    float ctt = safe_sqrt(1.0f - s2t);
    float rpa = (eta * cti - ctt) / (eta * cti + ctt);
    float rpe = (cti - eta * ctt) / (cti + eta * ctt);
    return (rpa * rpa + rpe * rpe) / 2.0f;
}
static inline bool refract(V3 wi, V3 n, float eta, V3* wt, float* etap) {   // dielectric.rs:541-567
    float cti = dot(n, wi);
    if (cti < 0.0f) { eta = 1.0f / eta; cti = -cti; n = -n; }
    float s2i = rmax(1.0f - cti * cti, 0.0f);
    float s2t = s2i / (eta * eta);
    if (s2t >= 1.0f) return false;
    float ctt = safe_sqrt(1.0f - s2t);
    *wt = -wi / eta + (cti / eta - ctt) * n;
    *etap = eta;
    return true;
}
static inline V3 reflect(V3 wo, V3 n) { return -wo + 2.0f * dot(wo, n) * n; }   // dielectric.rs:569-571

struct BxDF {
    int type; Spec4 refl; float eta; TrowbridgeReitz tr;
    int flags() const {
        if (type == MAT_DIFFUSE) return BXDF_DIFFUSE | BXDF_REFLECTION;
        int f = eta == 1.0f ? BXDF_TRANSMISSION : (BXDF_REFLECTION | BXDF_TRANSMISSION);
        return f | (tr.effectively_smooth() ? BXDF_SPECULAR : BXDF_GLOSSY);
    }
    Spec4 eval(V3 wo, V3 wi) const {   // TransportMode::Radiance
        if (type == MAT_DIFFUSE) {
            if (wo.z * wi.z > 0.0f) { Spec4 r; for (int i = 0; i < NSP; ++i) r.v[i] = refl.v[i] * FRAC_1_PI; return r; }
            return spec_const(0.0f);
        }
        if (eta == 1.0f || tr.effectively_smooth()) return spec_const(0.0f);
        float cto = wo.z, cti = wi.z;
        bool refl_ = cto * cti > 0.0f;
        float etap = refl_ ? 1.0f : (cto > 0.0f ? eta : 1.0f / eta);
        V3 wm = wi * etap + wo;
        if (cti == 0.0f || cto == 0.0f || length_squared(wm) == 0.0f) return spec_const(0.0f);
        wm = face_forward(normalized(wm), V3(0, 0, 1));
        float idm = dot(wi, wm), odm = dot(wo, wm);
        if (idm * cti < 0.0f || odm * cto < 0.0f) return spec_const(0.0f);
        float F = fresnel_dielectric(odm, eta);
        if (refl_) return spec_const(tr.density(wm) * tr.masking_shadowing(wo, wi) * F / std::fabs(4.0f * cti * cto));
        float s = idm + odm / etap;
        float denom = s * s * cti * cto;
        float ft = tr.density(wm) * (1.0f - F) * tr.masking_shadowing(wo, wi) * std::fabs(idm * odm / denom);
        ft /= etap * etap;
        return spec_const(ft);
    }
    bool sample(V3 wo, float uc, P2 u, BSDFSample* bs) const {   // sample_flags = all
        if (type == MAT_DIFFUSE) {
            V3 wi = sample_cosine_hemisphere(u);
            if (wo.z < 0.0f) wi.z *= -1.0f;
            bs->pdf = std::fabs(wi.z) * FRAC_1_PI;
            for (int i = 0; i < NSP; ++i) bs->value.v[i] = refl.v[i] * FRAC_1_PI;
            bs->wi = wi; bs->flags = BXDF_DIFFUSE | BXDF_REFLECTION;
            return true;
        }
        if (eta == 1.0f || tr.effectively_smooth()) {
            float R = fresnel_dielectric(wo.z, eta), T = 1.0f - R;
            float pr = R, pt = T;
            if (pr == 0.0f && pt == 0.0f) return false;
            if (uc < pr / (pr + pt)) {
                V3 wi(-wo.x, -wo.y, wo.z);
                bs->value = spec_const(R / std::fabs(wi.z)); bs->wi = wi; bs->pdf = pr / (pr + pt);
                bs->flags = BXDF_SPECULAR | BXDF_REFLECTION;
                return true;
            }
            V3 wi; float etap;
            if (!refract(wo, V3(0, 0, 1), eta, &wi, &etap)) return false;
            float ft = T / std::fabs(wi.z);
            ft /= etap * etap;
            bs->value = spec_const(ft); bs->wi = wi; bs->pdf = pt / (pr + pt);
            bs->flags = BXDF_SPECULAR | BXDF_TRANSMISSION;
            return true;
        }
        V3 wm = tr.sample_microfacet_n(wo, u);
        float R = fresnel_dielectric(dot(wo, wm), eta), T = 1.0f - R;
        float pr = R, pt = T;
        if (pr == 0.0f && pt == 0.0f) return false;
        if (uc < pr / (pr + pt)) {
            V3 wi = reflect(wo, wm);
            if (!(wo.z * wi.z > 0.0f)) return false;
            bs->pdf = tr.pdf(wo, wm) / (4.0f * absdot(wo, wi)) * (pr / (pr + pt));
            bs->value = spec_const(tr.density(wm) * tr.masking_shadowing(wo, wi) * R / (4.0f * wi.z * wo.z));
            bs->wi = wi; bs->flags = BXDF_GLOSSY | BXDF_REFLECTION;
            return true;
        }
        V3 wi; float etap;
        if (!refract(wo, wm, eta, &wi, &etap)) return false;
        if (wo.z * wi.z > 0.0f || wi.z == 0.0f) return false;
        float idm = dot(wi, wm), odm = dot(wo, wm);
        float s = idm + odm / etap;
        float denom = s * s;
        float dwm_dwi = std::fabs(idm) / denom;
        bs->pdf = tr.pdf(wo, wm) * dwm_dwi * (pt / (pr + pt));
        float ft = T * tr.density(wm) * tr.masking_shadowing(wo, wi) * std::fabs(idm * odm / (wi.z * wo.z * denom));
        ft /= etap * etap;
        bs->value = spec_const(ft); bs->wi = wi; bs->flags = BXDF_GLOSSY | BXDF_TRANSMISSION;
        return true;
    }
};

struct BSDF {   // reflection/base.rs:88-184, core/frame.rs:57-80
    BxDF bxdf; V3 fx, fy, fz;
    void init_frame(V3 ns, V3 dpdus) { fx = normalized(dpdus); fz = ns; fy = cross(fz, fx); }
    V3 to_local(V3 v) const { return V3(dot(v, fx), dot(v, fy), dot(v, fz)); }
    V3 from_local(V3 v) const { return v.x * fx + v.y * fy + v.z * fz; }
    Spec4 eval(V3 wo_r, V3 wi_r) const {
        V3 wi = to_local(wi_r), wo = to_local(wo_r);
        if (wo.z != 0.0f) return bxdf.eval(wo, wi);
        return spec_const(0.0f);
    }
    bool sample(V3 wo_r, float uc, P2 u, BSDFSample* bs) const {
        V3 wo = to_local(wo_r);
        if (wo.z == 0.0f) return false;   // (sample_flags = all always intersects the bxdf flags)
        if (!bxdf.sample(wo, uc, u, bs)) return false;
        if (spec_all_zero(bs->value) || bs->pdf == 0.0f || bs->wi.z == 0.0f) return false;
        bs->wi = from_local(bs->wi);
        return true;
    }
};

// ---- the renderer ---------------------------------------------------------------------------------------
// dead_final_rays: closest-hit rays of SimplePath / VolPath traced towards the LAST vertex of a path whose result cannot reach
// the film - that vertex only adds emission (simple_path.rs:156-158,124-132), which needs (specular_bounce || !sample_lights)
// and an infinite or area light in the scene.  The reference traces them; the device elides them (same film), so its ray
// counts are compared with closest_rays - dead_final_rays.
struct RenderCounters { uint64_t closest_rays = 0, shadow_rays = 0, samples = 0, bounces = 0, dead_final_rays = 0; };

struct Renderer {
    const Scene* scene;
    std::vector<uint32_t> prim_material;      // by prim id
    std::vector<OMaterial> materials;
    std::vector<OLight> lights;
    std::vector<int> infinite_lights;
    std::vector<int> prim_area_light;         // by prim id: index of the DiffuseAreaLight on that primitive, or -1
    std::vector<int> prim_patch;              // by prim id: index into scene->patches (-1 for spheres)
    std::vector<int> prim_sphere;             // by prim id: index into scene->spheres (-1 for patches / triangles)
    bool have_medium = false; OMedium medium{};
    OCamera cam; OFilm film; FilterSamplerO fsampler;
    V3 scene_center; float scene_radius;
    std::vector<double> pixels;               // 4 doubles per pixel: rgb_sum[3], weight_sum (film.rs:302-307)

    void prepare() {
        fsampler.init(film);
        b3_bounding_sphere(scene->bounds(), &scene_center, &scene_radius);   // integrators/base.rs:54-70
        infinite_lights.clear();
        for (size_t i = 0; i < lights.size(); ++i) if (lights[i].type == LIGHT_UNIFORM_INFINITE) infinite_lights.push_back((int)i);
        prim_area_light.assign(prim_material.size(), -1); prim_patch.assign(prim_material.size(), -1);
        prim_sphere.assign(prim_material.size(), -1);
        for (const Prim& p : scene->prims) if (p.type != PRIM_SPHERE && p.id < prim_patch.size()) prim_patch[p.id] = (int)p.index;
        for (const Prim& p : scene->prims) if (p.type == PRIM_SPHERE && p.id < prim_sphere.size()) prim_sphere[p.id] = (int)p.index;
        for (size_t i = 0; i < lights.size(); ++i)          // GeometricPrimitive.area_light (primitives/single.rs:13-19)
            if (lights[i].type == LIGHT_DIFFUSE_AREA && lights[i].prim_id < prim_area_light.size()) prim_area_light[lights[i].prim_id] = (int)i;
        int w = film.bounds[2] - film.bounds[0], h = film.bounds[3] - film.bounds[1];
        pixels.assign((size_t)w * h * 4, 0.0);
    }
    M4 m4(const float* p) const { M4 m; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m.m[r][c] = p[r * 4 + c]; return m; }

    Spec4 light_L(const OLight& l, const Wavelengths& w) const {   // scale * spectrum.sample(lambda)
        Spec4 r; for (int i = 0; i < NSP; ++i) r.v[i] = dense_at(l.spectrum, w.lambda[i]) * l.scale; return r;
    }
    // DiffuseAreaLight::L (extension: the reference has no area light, lights/base.rs:18-22; pbrt-v4 semantics): one-sided
    // lights emit on the side their surface normal points to
    Spec4 area_L(const OLight& l, V3 n, V3 w, const Wavelengths& wl) const {
        if (!l.two_sided && dot(n, w) < 0.0f) return spec_const(0.0f);
        return light_L(l, wl);
    }
    // homogeneous medium (media/medium.rs:78-96): sigma_a, sigma_s at the wavelengths; sigma_maj = sigma_a + sigma_s
    void medium_sigmas(const Wavelengths& w, Spec4* sa, Spec4* ss, Spec4* st) const {
        for (int i = 0; i < NSP; ++i) {
            sa->v[i] = dense_at(medium.sigma_a, w.lambda[i]); ss->v[i] = dense_at(medium.sigma_s, w.lambda[i]);
            st->v[i] = sa->v[i] + ss->v[i];
        }
    }
    // the part [t0, t1] of the ray's [0, t_max] that lies inside the medium's bounds (slab clip); false if empty
    bool medium_segment(V3 ro, V3 rd, float t_max, float* t0o, float* t1o) const {
        float t0 = 0.0f, t1 = t_max;
        for (int a = 0; a < 3; ++a) {
            float inv = 1.0f / rd[a];
            float tn = (medium.bounds[a] - ro[a]) * inv, tf = (medium.bounds[3 + a] - ro[a]) * inv;
            if (tn > tf) { float x = tn; tn = tf; tf = x; }
            if (tn > t0) t0 = tn;
            if (tf < t1) t1 = tf;
        }
        if (!(t0 < t1)) return false;
        *t0o = t0; *t1o = t1;
        return true;
    }
    // transmittance of the shadow segment o + t * dir, t in [0, t_max]: exp(-sigma_t * length inside the medium)
    Spec4 shadow_tr(V3 o, V3 dir, float t_max, const Wavelengths& w) const {
        Spec4 tr = spec_const(1.0f);
        float t0, t1;
        if (!have_medium || !medium_segment(o, dir, t_max, &t0, &t1)) return tr;
        Spec4 sa, ss, st; medium_sigmas(w, &sa, &ss, &st);
        float seg = (t1 - t0) * length(dir);
        for (int i = 0; i < NSP; ++i) tr.v[i] = t_exp(-st.v[i] * seg);
        return tr;
    }
    // Light::sample_li for every light type.  ref_p: the receiving point; returns false when the sample carries nothing.
    // ctx: the receiving SURFACE point (sphere emitters offset their inside / outside test along its normal); null in a medium
    bool sample_light(const OLight& lt, V3 ref_p, P2 u2, const Wavelengths& w, Spec4* Ls, V3* wi, float* pdf, V3* p_light,
                      const SurfInt* ctx = nullptr) const {
        if (lt.type == LIGHT_UNIFORM_INFINITE) {
            *wi = sample_uniform_sphere(u2); *pdf = INV_4_PI; *Ls = light_L(lt, w);
            *p_light = ref_p + *wi * (2.0f * scene_radius);
        } else if (lt.type == LIGHT_DIRECTIONAL) {
            *wi = normalized(xf_vec(m4(lt.render_from_light), V3(0, 0, 1))); *pdf = 1.0f; *Ls = light_L(lt, w);
            *p_light = ref_p + *wi * (2.0f * scene_radius);
        } else if (lt.type == LIGHT_POINT) {
            V3 p = xf_point(m4(lt.render_from_light), V3(0, 0, 0));
            *wi = normalized(p - ref_p); *pdf = 1.0f;
            *Ls = light_L(lt, w);
            float d2 = distance_squared(p, ref_p);
            for (int i = 0; i < NSP; ++i) Ls->v[i] /= d2;
            *p_light = p;
        } else {      // DiffuseAreaLight::sample_li over BilinearPatch::sample_with_context
            int pi = lt.prim_id < prim_patch.size() ? prim_patch[lt.prim_id] : -1;
            int si = lt.prim_id < prim_sphere.size() ? prim_sphere[lt.prim_id] : -1;
            if (pi < 0 && si < 0) return false;
            ShapeSample ss;
            if (si >= 0) {          // ... or over Sphere::sample_with_context
                const Sphere& sph = scene->spheres[si];
                V3 ctx_p = ctx ? ctx->pi.midpoints() : ref_p;
                V3 p_origin = ctx ? offset_ray_origin(ctx->pi, ctx->n, sphere_center(sph) - ctx_p) : ctx_p;
                if (!sphere_sample_ctx(sph, ctx_p, p_origin, u2, &ss)) return false;
            } else if (!patch_sample_ctx(scene->patches[pi], ref_p, u2, &ss)) return false;
            V3 pl = ss.pi.midpoints();
            if (ss.pdf == 0.0f || length_squared(pl - ref_p) == 0.0f) return false;
            *wi = normalized(pl - ref_p);
            *Ls = area_L(lt, ss.n, -*wi, w);
            *pdf = ss.pdf; *p_light = pl;
        }
        return true;
    }

    SurfInt make_interaction(const Hit& h) const {
        const Prim& p = scene->prims[h.prim_slot];
        SurfInt si = p.type == PRIM_SPHERE ? sphere_interaction(scene->spheres[p.index], h) : patch_interaction(scene->patches[p.index], h);
        si.u = h.u; si.v = h.v;
        return si;
    }
    BSDF make_bsdf(const SurfInt& si, Wavelengths* w) const {   // material.rs:122-195
        const OMaterial& m = materials[prim_material[si.prim_id]];
        BSDF b;
        b.bxdf.type = m.type;
        if (m.type == MAT_DIFFUSE) {
            bool second = false;
            if (m.texture == TEX_CHECKERBOARD) {       // extension: point-sampled 2-D checkerboard over UvMapping::map (mappings.rs:66-84)
                float s = si.u, t = si.v;
                int pi = si.prim_id < prim_patch.size() ? prim_patch[si.prim_id] : -1;
                if (pi >= 0 && scene->patches[pi].has_uv) {
                    const Patch& pt = scene->patches[pi];
                    s = lerpf(lerpf(pt.uv00[0], pt.uv01[0], si.v), lerpf(pt.uv10[0], pt.uv11[0], si.v), si.u);
                    t = lerpf(lerpf(pt.uv00[1], pt.uv01[1], si.v), lerpf(pt.uv10[1], pt.uv11[1], si.v), si.u);
                }
                float ss = m.su * s + m.du, tt = m.sv * t + m.dv;
                second = (((int)std::floor(ss) + (int)std::floor(tt)) & 1) != 0;
            }
            Spec4 r = spectrum_sample(second ? m.spec2 : m.spec, *w);
            for (int i = 0; i < NSP; ++i) r.v[i] = clampf(r.v[i], 0.0f, 1.0f);
            b.bxdf.refl = r; b.bxdf.eta = 1.0f; b.bxdf.tr = TrowbridgeReitz{0, 0};
        } else {
            b.bxdf.eta = spectrum_at(m.spec, w->lambda[0]);
            if (m.spec.kind != SPEC_CONSTANT) terminate_secondary(w);
            float ur = m.u_rough, vr = m.v_rough;
            if (m.remap_roughness) { ur = std::sqrt(ur); vr = std::sqrt(vr); }
            b.bxdf.tr = TrowbridgeReitz{ur, vr};
            b.bxdf.refl = spec_const(0.0f);
        }
        b.init_frame(si.ns, si.dpdus);
        return b;
    }

    // perspective.rs:130-189 + transform.rs:446-466
    void generate_ray(P2 p_film, P2 p_lens, V3* o_out, V3* d_out) const {
        M4 cfr = m4(cam.camera_from_raster), rfc = m4(cam.render_from_camera);
        V3 pc = xf_point(cfr, V3(p_film.x, p_film.y, 0.0f));
        V3 o(0, 0, 0);
        V3 d = normalized(pc);
        if (cam.lens_radius > 0.0f) {
            P2 dl = sample_uniform_disk_concentric(p_lens);
            P2 pl{dl.x * cam.lens_radius, dl.y * cam.lens_radius};
            float tf = cam.focal_distance / d.z;
            V3 pf = o + d * tf;
            o = V3(pl.x, pl.y, 0.0f);
            d = normalized(pf - o);
        }
        V3fi oi = xf_point_fi(rfc, V3fi::exact(o));
        V3 dr = xf_vec(rfc, d);
        float lsq = length_squared(dr);
        if (lsq > 0.0f) {
            float dt = dot(vabs(dr), oi.error()) / lsq;
            oi = V3fi(oi.x + Interval::exact(dr.x) * dt, oi.y + Interval::exact(dr.y) * dt, oi.z + Interval::exact(dr.z) * dt);
        }
        *o_out = oi.midpoints(); *d_out = dr;
    }

    bool unoccluded(const SurfInt& si, V3 p_light, RenderCounters* rc) const {   // integrators/base.rs:39-44
        V3 dir = p_light - si.pi.midpoints();
        V3 o = offset_ray_origin(si.pi, si.n, dir);
        rc->shadow_rays++;
        return !scene->intersect_p(o, dir, 1.0f - SHADOW_EPSILON);
    }

    // SimplePathIntegrator::incident_radiance (simple_path.rs:105-269).  vol: the medium-aware variant (INTEGRATOR_VOL_PATH, an
    // extension - the reference has no volumetric integrator): before the surface is considered, the part of the ray inside the
    // homogeneous medium is sampled for a real collision (distance ~ sigma_t[lambda_0], no null collisions since sigma_maj ==
    // sigma_t): absorbed -> the path ends; scattered -> a path vertex with next-event estimation through the phase function and a
    // Henyey-Greenstein sampled continuation.  Shadow rays are attenuated by the closed-form transmittance.
    Spec4 li_simple_path(V3 ro, V3 rd, Wavelengths* w, Pcg32* rng, const ORender& rp, RenderCounters* rc, bool vol = false) const {
        Spec4 L = spec_const(0.0f), beta = spec_const(1.0f);
        bool specular_bounce = true;
        int depth = 0;
        const bool use_medium = vol && have_medium;
        bool terminal_emission = !infinite_lights.empty();
        for (const auto& l : lights) if (l.type == LIGHT_DIFFUSE_AREA) terminal_emission = true;
        while (!spec_all_zero(beta)) {
            Hit h;
            rc->closest_rays++;
            if (depth == rp.max_depth && !((specular_bounce || !rp.sample_lights) && terminal_emission)) rc->dead_final_rays++;
            bool hit = scene->intersect(ro, rd, std::numeric_limits<float>::infinity(), &h);
            if (use_medium) {
                float t0, t1;
                if (medium_segment(ro, rd, hit ? h.t : std::numeric_limits<float>::infinity(), &t0, &t1)) {
                    float dlen = length(rd), seg = (t1 - t0) * dlen;
                    float u = rng->uniform_float(), um = rng->uniform_float();
                    Spec4 sa, ss, st; medium_sigmas(*w, &sa, &ss, &st);
                    float s = st.v[0] > 0.0f ? sample_exponential(u, st.v[0]) : std::numeric_limits<float>::infinity();
                    if (s < seg) {
                        if (um < sa.v[0] / st.v[0]) break;                       // absorbed (the medium does not emit)
                        float T0 = t_exp(-st.v[0] * s);
                        for (int i = 0; i < NSP; ++i) beta.v[i] *= t_exp(-st.v[i] * s) * ss.v[i] / (T0 * ss.v[0]);
                        if (depth == rp.max_depth) break;
                        depth++;
                        rc->bounces++;
                        V3 p = ro + rd * (t0 + s / dlen);
                        V3 wo = -rd / dlen;
                        if (rp.sample_lights) {
                            float ul = rng->uniform_float();
                            if (!lights.empty()) {
                                size_t nl = lights.size();
                                float fi = ul * (float)nl;
                                size_t idx = fi > 0.0f ? (size_t)fi : 0; if (idx > nl - 1) idx = nl - 1;
                                float prob = 1.0f / (float)nl;
                                P2 u2; u2.x = rng->uniform_float(); u2.y = rng->uniform_float();
                                Spec4 Ls; V3 wi, p_light; float pdf;
                                if (sample_light(lights[idx], p, u2, *w, &Ls, &wi, &pdf, &p_light) && !spec_all_zero(Ls) && pdf > 0.0f) {
                                    float f = henyey_greenstein(dot(wo, wi), medium.g);
                                    if (f != 0.0f) {
                                        V3 dir = p_light - p;
                                        rc->shadow_rays++;
                                        if (!scene->intersect_p(p, dir, 1.0f - SHADOW_EPSILON)) {
                                            Spec4 tr = shadow_tr(p, dir, 1.0f - SHADOW_EPSILON, *w);
                                            float den = prob * pdf;
                                            for (int i = 0; i < NSP; ++i) L.v[i] += beta.v[i] * f * Ls.v[i] * tr.v[i] / den;
                                        }
                                    }
                                }
                            }
                        }
                        P2 u2; u2.x = rng->uniform_float(); u2.y = rng->uniform_float();
                        float ppdf;
                        V3 wi = sample_henyey_greenstein(wo, medium.g, u2, &ppdf);      // value == pdf: beta is unchanged
                        if (ppdf == 0.0f) break;
                        specular_bounce = false;
                        ro = p; rd = wi;
                        continue;
                    }
                    float Tm0 = t_exp(-st.v[0] * seg);                           // no collision: T_maj / pdf of passing through
                    for (int i = 0; i < NSP; ++i) beta.v[i] *= t_exp(-st.v[i] * seg) / Tm0;
                }
            }
            if (!hit) {
                if (!rp.sample_lights || specular_bounce)
                    for (int li : infinite_lights) { Spec4 le = light_L(lights[li], *w); for (int i = 0; i < NSP; ++i) L.v[i] += beta.v[i] * le.v[i]; }
                break;
            }
            SurfInt si = make_interaction(h);
            // emitted_radiance (simple_path.rs:156-158, interaction.rs:179-184): only primitives that carry an area light emit
            if ((!rp.sample_lights || specular_bounce) && prim_area_light[si.prim_id] >= 0) {
                Spec4 le = area_L(lights[prim_area_light[si.prim_id]], si.n, -rd, *w);
                for (int i = 0; i < NSP; ++i) L.v[i] += beta.v[i] * le.v[i];
            }
            if (depth == rp.max_depth) break;
            depth++;
            rc->bounces++;
            BSDF bsdf = make_bsdf(si, w);
            V3 wo = -rd;
            if (rp.sample_lights) {
                float ul = rng->uniform_float();
                if (!lights.empty()) {
                    size_t nl = lights.size();
                    float fi = ul * (float)nl;
                    size_t idx = fi > 0.0f ? (size_t)fi : 0; if (idx > nl - 1) idx = nl - 1;
                    float prob = 1.0f / (float)nl;
                    P2 u2; u2.x = rng->uniform_float(); u2.y = rng->uniform_float();
                    Spec4 Ls; V3 wi, p_light; float pdf;
                    V3 pm = si.pi.midpoints();
                    if (sample_light(lights[idx], pm, u2, *w, &Ls, &wi, &pdf, &p_light, &si) && !spec_all_zero(Ls) && pdf > 0.0f) {
                        Spec4 f = bsdf.eval(wo, wi);
                        float ad = absdot(wi, si.ns);
                        for (int i = 0; i < NSP; ++i) f.v[i] *= ad;
                        if (!spec_all_zero(f) && unoccluded(si, p_light, rc)) {
                            float den = prob * pdf;
                            if (use_medium) {
                                V3 dir = p_light - pm;
                                Spec4 tr = shadow_tr(offset_ray_origin(si.pi, si.n, dir), dir, 1.0f - SHADOW_EPSILON, *w);
                                for (int i = 0; i < NSP; ++i) L.v[i] += beta.v[i] * f.v[i] * Ls.v[i] * tr.v[i] / den;
                            } else
                                for (int i = 0; i < NSP; ++i) L.v[i] += beta.v[i] * f.v[i] * Ls.v[i] / den;
                        }
                    }
                }
            }
            if (rp.sample_bsdf) {
                float uc = rng->uniform_float();
                P2 u2; u2.x = rng->uniform_float(); u2.y = rng->uniform_float();
                BSDFSample bs;
                if (!bsdf.sample(wo, uc, u2, &bs)) break;
                float ad = absdot(bs.wi, si.ns);
                for (int i = 0; i < NSP; ++i) beta.v[i] *= bs.value.v[i] * ad / bs.pdf;
                specular_bounce = (bs.flags & BXDF_SPECULAR) != 0;
                ro = offset_ray_origin(si.pi, si.n, bs.wi); rd = bs.wi;
            } else {
                int fl = bsdf.bxdf.flags();
                V3 wi; float pdf;
                P2 u2; u2.x = rng->uniform_float(); u2.y = rng->uniform_float();
                if ((fl & BXDF_REFLECTION) && (fl & BXDF_TRANSMISSION)) { wi = sample_uniform_sphere(u2); pdf = INV_4_PI; }
                else {
                    wi = sample_uniform_hemisphere(u2); pdf = INV_2_PI;
                    float pr = dot(wo, si.n) * dot(wi, si.n);
                    bool flip_r = (fl & BXDF_REFLECTION) && pr < 0.0f;
                    bool flip_t = (fl & BXDF_TRANSMISSION) && pr > 0.0f;
                    if (flip_r || flip_t) wi = -wi;
                }
                Spec4 f = bsdf.eval(wo, wi);
                float ad = absdot(wi, si.ns);
                for (int i = 0; i < NSP; ++i) beta.v[i] *= f.v[i] * ad / pdf;
                specular_bounce = false;
                ro = offset_ray_origin(si.pi, si.n, wi); rd = wi;
            }
        }
        return L;
    }

    Spec4 li_random_walk(V3 ro, V3 rd, Wavelengths* w, Pcg32* rng, const ORender& rp, int depth, RenderCounters* rc) const {
        Hit h;
        rc->closest_rays++;
        if (!scene->intersect(ro, rd, std::numeric_limits<float>::infinity(), &h)) {
            Spec4 le = spec_const(0.0f);   // Sum::sum folds from with_single_value(0.0)
            for (int li : infinite_lights) { Spec4 l = light_L(lights[li], *w); for (int i = 0; i < NSP; ++i) le.v[i] += l.v[i]; }
            return le;
        }
        Spec4 le = spec_const(0.0f);
        if (depth == rp.max_depth) return le;
        rc->bounces++;
        SurfInt si = make_interaction(h);
        BSDF bsdf = make_bsdf(si, w);
        V3 wo = -rd;
        P2 u; u.x = rng->uniform_float(); u.y = rng->uniform_float();
        V3 wp = sample_uniform_sphere(u);
        Spec4 f = bsdf.eval(wo, wp);
        if (spec_all_zero(f)) return le;
        float ad = absdot(wp, si.ns);
        for (int i = 0; i < NSP; ++i) f.v[i] *= ad;
        V3 o = offset_ray_origin(si.pi, si.n, wp);
        Spec4 li = li_random_walk(o, wp, w, rng, rp, depth + 1, rc);
        Spec4 r;
        const float pdf = 1.0f / (4.0f * PI_F);
        for (int i = 0; i < NSP; ++i) r.v[i] = le.v[i] + f.v[i] * li.v[i] / pdf;
        return r;
    }

    // film.rs:160-192 + sensor.rs:114-121
    void add_sample(int px, int py, const Spec4& L, const Wavelengths& w, float weight) {
        Spec4 Lp; for (int i = 0; i < NSP; ++i) Lp.v[i] = w.pdf[i] != 0.0f ? L.v[i] / w.pdf[i] : 0.0f;
        float rgb[3];
        for (int c = 0; c < 3; ++c) {
            float sum = 0.0f;
            for (int i = 0; i < NSP; ++i) sum += dense_at(film.sensor_bar + 471 * c, w.lambda[i]) * Lp.v[i];
            rgb[c] = sum / (float)NSP * film.imaging_ratio;
        }
        float m = rgb[0]; if (!(rgb[1] < m)) m = rgb[1]; if (!(rgb[2] < m)) m = rgb[2];
        if (m > film.max_component_value) { float s = film.max_component_value / m; for (int c = 0; c < 3; ++c) rgb[c] *= s; }
        int wdt = film.bounds[2] - film.bounds[0];
        double* p = &pixels[((size_t)(py - film.bounds[1]) * wdt + (px - film.bounds[0])) * 4];
        double wd = (double)weight;
        for (int c = 0; c < 3; ++c) p[c] += wd * (double)rgb[c];
        p[3] += wd;
    }

    // integrators/base.rs:158-219; returns the unweighted radiance too (for per-sample checks)
    void eval_pixel_sample(int px, int py, int sample_index, const ORender& rp, RenderCounters* rc, Spec4* L_out,
                           Wavelengths* w_out, float* fw_out) {
        Pcg32 rng = start_pixel_sample(px, py, rp.seed, (uint64_t)sample_index, 0);
        float lu = rng.uniform_float();
        Wavelengths w = sample_visible(lu);
        P2 uf; uf.x = rng.uniform_float(); uf.y = rng.uniform_float();
        P2 fp; float fweight;
        fsampler.sample(uf, &fp, &fweight);
        P2 p_film{(float)px + fp.x + 0.5f, (float)py + fp.y + 0.5f};
        (void)rng.uniform_float();   // time (B13: evaluates to 0 for a 0..1 shutter; no geometry reads it)
        P2 p_lens; p_lens.x = rng.uniform_float(); p_lens.y = rng.uniform_float();
        V3 ro, rd;
        generate_ray(p_film, p_lens, &ro, &rd);
        Spec4 L = rp.integrator == INTEGRATOR_RANDOM_WALK ? li_random_walk(ro, rd, &w, &rng, rp, 0, rc)
                                                          : li_simple_path(ro, rd, &w, &rng, rp, rc, rp.integrator == INTEGRATOR_VOL_PATH);
        for (int i = 0; i < NSP; ++i) L.v[i] = L.v[i] * 1.0f;   // * camera_ray.weight
        rc->samples++;
        if (L_out) { *L_out = L; *w_out = w; *fw_out = fweight; }
    }

    // image_tile_render: every (pixel, sample) is independent and a pixel's samples are added in sample-index
    // order, so the result does not depend on the tile schedule; rows are handed to threads.
    RenderCounters render(const ORender& rp, int nthreads) {
        int x0 = rp.window[0], y0 = rp.window[1], x1 = rp.window[2], y1 = rp.window[3];
        std::vector<RenderCounters> rcs(std::max(1, nthreads));
        std::atomic<int> next_row(y0);
        auto work = [&](int tid) {
            for (;;) {
                int y = next_row.fetch_add(1);
                if (y >= y1) break;
                for (int x = x0; x < x1; ++x)
                    for (int s = rp.s0; s < rp.s1; ++s) {
                        Spec4 L; Wavelengths w; float fw;
                        eval_pixel_sample(x, y, s, rp, &rcs[tid], &L, &w, &fw);
                        add_sample(x, y, L, w, fw);
                    }
            }
        };
        if (nthreads <= 1) work(0);
        else { std::vector<std::thread> th; for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t); for (auto& t : th) t.join(); }
        RenderCounters tot;
        for (auto& r : rcs) { tot.closest_rays += r.closest_rays; tot.shadow_rays += r.shadow_rays; tot.samples += r.samples; tot.bounces += r.bounces; tot.dead_final_rays += r.dead_final_rays; }
        return tot;
    }

    // film.rs:261-279 (splats are never emitted by these integrators; their term is 0)
    void get_image(float* rgb_out) const {
        int wdt = film.bounds[2] - film.bounds[0], hgt = film.bounds[3] - film.bounds[1];
        for (size_t i = 0; i < (size_t)wdt * hgt; ++i) {
            const double* p = &pixels[i * 4];
            float rgb[3] = {(float)p[0], (float)p[1], (float)p[2]};
            if (p[3] != 0.0) { float ws = (float)p[3]; for (int c = 0; c < 3; ++c) rgb[c] /= ws; }
            for (int r = 0; r < 3; ++r) {
                float acc = 0.0f;
                for (int c = 0; c < 3; ++c) acc += film.out_rgb_from_sensor[r * 3 + c] * rgb[c];
                rgb_out[i * 3 + r] = acc;
            }
        }
    }
};

}  // namespace ora
