// Host-side mirror of the reference constructors whose OUTPUT crosses the C ABI (include/pbrt_mi355_host.h).
// In a real integration the Rust host computes these with its own code (camera/, imaging/filter.rs,
// sampling/spectrum.rs, color/) and passes the results to pm_set_camera / pm_set_film / pm_set_lights;
// this C++ restatement exists so that the C++/Python harnesses of this repository (tests, bench, examples)
// can build the same scenes without a Rust toolchain.  f32 arithmetic in the reference's operation order.
#include "../../include/pbrt_mi355_host.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct M4 { float m[4][4]; };
struct Xf { M4 m, inv; };

float dop(float a, float b, float c, float d) {   // math.rs:78-87
    float cd = c * d;
    return std::fmaf(a, b, -cd) + std::fmaf(-c, d, cd);
}
float ip3(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

// SquareMatrix<4>::inverse, core/square_matrix.rs:196-262 (only .val of inner_product! is read)
bool inverse(const M4& A, M4* out) {
    const auto& m = A.m;
    float s0 = dop(m[0][0], m[1][1], m[1][0], m[0][1]), s1 = dop(m[0][0], m[1][2], m[1][0], m[0][2]);
    float s2 = dop(m[0][0], m[1][3], m[1][0], m[0][3]), s3 = dop(m[0][1], m[1][2], m[1][1], m[0][2]);
    float s4 = dop(m[0][1], m[1][3], m[1][1], m[0][3]), s5 = dop(m[0][2], m[1][3], m[1][2], m[0][3]);
    float c0 = dop(m[2][0], m[3][1], m[3][0], m[2][1]), c1 = dop(m[2][0], m[3][2], m[3][0], m[2][2]);
    float c2 = dop(m[2][0], m[3][3], m[3][0], m[2][3]), c3 = dop(m[2][1], m[3][2], m[3][1], m[2][2]);
    float c4 = dop(m[2][1], m[3][3], m[3][1], m[2][3]), c5 = dop(m[2][2], m[3][3], m[3][2], m[2][3]);
    float det = s0 * c5 + (-s1 * c4 + (s2 * c3 + (s3 * c2 + (s5 * c0 + -s4 * c1))));
    if (det == 0.0f) return false;
    float s = 1.0f / det;
    float r[4][4] = {
        {s * ip3(m[1][1], c5, m[1][3], c3, -m[1][2], c4), s * ip3(-m[0][1], c5, m[0][2], c4, -m[0][3], c3),
         s * ip3(m[3][1], s5, m[3][3], s3, -m[3][2], s4), s * ip3(-m[2][1], s5, m[2][2], s4, -m[2][3], s3)},
        {s * ip3(-m[1][0], c5, m[1][2], c2, -m[1][3], c1), s * ip3(m[0][0], c5, m[0][3], c1, -m[0][2], c2),
         s * ip3(-m[3][0], s5, m[3][2], s2, -m[3][3], s1), s * ip3(m[2][0], s5, m[2][3], s1, -m[2][2], s2)},
        {s * ip3(m[1][0], c4, m[1][3], c0, -m[1][1], c2), s * ip3(-m[0][0], c4, m[0][1], c2, -m[0][3], c0),
         s * ip3(m[3][0], s4, m[3][3], s0, -m[3][1], s2), s * ip3(-m[2][0], s4, m[2][1], s2, -m[2][3], s0)},
        {s * ip3(-m[1][0], c3, m[1][1], c1, -m[1][2], c0), s * ip3(m[0][0], c3, m[0][2], c0, -m[0][1], c1),
         s * ip3(-m[3][0], s3, m[3][1], s1, -m[3][2], s0), s * ip3(m[2][0], s3, m[2][2], s0, -m[2][1], s1)}};
    std::memcpy(out->m, r, sizeof(r));
    return true;
}
M4 matmul(const M4& a, const M4& b) {   // square_matrix.rs:316-332 (mul_add accumulation)
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc = std::fmaf(a.m[i][k], b.m[k][j], acc);
            r.m[i][j] = acc;
        }
    return r;
}
Xf xmul(const Xf& a, const Xf& b) { return Xf{matmul(a.m, b.m), matmul(b.inv, a.inv)}; }   // transform.rs:312-317
Xf xinv(const Xf& a) { return Xf{a.inv, a.m}; }
M4 ident() { M4 r{}; for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f; return r; }
Xf translate(float x, float y, float z) {   // transform.rs:60-75
    Xf t{ident(), ident()};
    t.m.m[0][3] = x; t.m.m[1][3] = y; t.m.m[2][3] = z;
    t.inv.m[0][3] = -x; t.inv.m[1][3] = -y; t.inv.m[2][3] = -z;
    return t;
}
Xf scale(float x, float y, float z) {       // transform.rs:78-93
    Xf t{ident(), ident()};
    t.m.m[0][0] = x; t.m.m[1][1] = y; t.m.m[2][2] = z;
    t.inv.m[0][0] = 1.0f / x; t.inv.m[1][1] = 1.0f / y; t.inv.m[2][2] = 1.0f / z;
    return t;
}
struct V { float x, y, z; };
V vsub(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
V vcross(V a, V b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V{(float)(ay * bz - az * by), (float)(az * bx - ax * bz), (float)(ax * by - ay * bx)};
}
V vnorm(V a) { float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return V{a.x / l, a.y / l, a.z / l}; }
void put(const M4& m, float* o) { std::memcpy(o, m.m, 64); }
M4 get(const float* p) { M4 m; std::memcpy(m.m, p, 64); return m; }

float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
float eval_poly(float t, const float* c, int n) { float s = 0.0f; for (int i = n - 1; i >= 0; --i) s = std::fmaf(s, t, c[i]); return s; }
int exponent(float v) { uint32_t u; std::memcpy(&u, &v, 4); return (int)((u & 0x7F800000u) >> 23) - 127; }
float fast_exp(float x) {     // core/routines.rs:47-73
    float xp = x * 1.44269504088896340736f;
    float fxp = std::floor(xp), f = xp - fxp;
    int i = (int)std::trunc(fxp);
    const float c[4] = {1.0f, 0.6955569f, 0.22617356f, 0.07814557f};
    float t = eval_poly(f, c, 4);
    int e = exponent(t) + i;
    if (e < -126) return 0.0f;
    if (e > 127) return INFINITY;
    uint32_t b; std::memcpy(&b, &t, 4);
    b = (b & 0x807FFFFFu) | ((uint32_t)(e + 127) << 23);
    float r; std::memcpy(&r, &b, 4);
    return r;
}
float gaussian(float x, float mean, float sd) {   // math.rs:10-13
    float d = x - mean;
    return 1.0f / std::sqrt(2.0f * 3.14159265358979323846f * sd * sd) * fast_exp(-(d * d) / (2.0f * sd * sd));
}
float sigmoid_poly(const float* c, float lambda) {   // color_space.rs:243-280
    const float p[3] = {c[2], c[1], c[0]};
    float x = eval_poly(lambda, p, 3);
    if (std::isinf(x)) return x > 0.0f ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * std::sqrt(1.0f + x * x));
}

}  // namespace

extern "C" {

int pmh_mat4_inverse(const float* m16, float* out16) {
    M4 r;
    if (!inverse(get(m16), &r)) return 1;
    put(r, out16);
    return 0;
}
void pmh_mat4_mul(const float* a16, const float* b16, float* out16) { put(matmul(get(a16), get(b16)), out16); }

// Transform::look_at (core/transform.rs:207-223): out_m = camera_from_world, out_inv = world_from_camera
int pmh_look_at(const float* pos, const float* look, const float* up, float* out_m, float* out_inv) {
    V p{pos[0], pos[1], pos[2]}, l{look[0], look[1], look[2]}, u{up[0], up[1], up[2]};
    V dir = vnorm(vsub(l, p));
    V right = vnorm(vcross(vnorm(u), dir));
    V nup = vcross(dir, right);
    M4 c2w{{{right.x, nup.x, dir.x, p.x}, {right.y, nup.y, dir.y, p.y}, {right.z, nup.z, dir.z, p.z}, {0, 0, 0, 1}}};
    M4 w2c;
    if (!inverse(c2w, &w2c)) return 1;
    put(w2c, out_m); put(c2w, out_inv);
    return 0;
}

// CameraTransform::new (camera/base.rs:265-279) + PerspectiveCamera::new (camera/perspective.rs:30-57) +
// ProjectiveCamera::new (camera/base.rs:342-380).  world_from_camera is given as (m, inverse).
int pmh_perspective_camera(const float* wfc_m, const float* wfc_inv, float fov_deg, const float* screen_window4,
                           int res_x, int res_y, float lens_radius, float focal_distance, PmCamera* out,
                           float* world_from_render16, float* render_from_world16) {
    Xf wfc{get(wfc_m), get(wfc_inv)};
    // p_camera = world_from_camera * Point3f::ZERO
    const auto& m = wfc.m.m;
    float px = 0.0f * m[0][0] + 0.0f * m[0][1] + 0.0f * m[0][2] + m[0][3];
    float py = 0.0f * m[1][0] + 0.0f * m[1][1] + 0.0f * m[1][2] + m[1][3];
    float pz = 0.0f * m[2][0] + 0.0f * m[2][1] + 0.0f * m[2][2] + m[2][3];
    Xf wfr = translate(px, py, pz);
    Xf rfw = xinv(wfr);
    Xf rfc = xmul(rfw, wfc);
    // Transform::perspective(fov, 0.01, 1000.0) (core/transform.rs:236-255)
    const float n = 0.01f, f = 1000.0f;
    M4 persp{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, f / (f - n), -f * n / (f - n)}, {0, 0, 1, 0}}};
    Xf pt; pt.m = persp;
    if (!inverse(persp, &pt.inv)) return 1;
    float inv_tan = 1.0f / std::tan(fov_deg * (3.14159265358979323846f / 180.0f) / 2.0f);
    Xf sfc = xmul(scale(inv_tan, inv_tan, 1.0f), pt);
    const float* sw = screen_window4;   // min.x, min.y, max.x, max.y
    Xf ndc_from_screen = xmul(scale(1.0f / (sw[2] - sw[0]), 1.0f / (sw[3] - sw[1]), 1.0f), translate(-sw[0], -sw[3], 0.0f));
    Xf raster_from_ndc = scale((float)res_x, -(float)res_y, 1.0f);
    Xf raster_from_screen = xmul(raster_from_ndc, ndc_from_screen);
    Xf screen_from_raster = xinv(raster_from_screen);
    Xf cfr = xmul(xinv(sfc), screen_from_raster);
    put(cfr.m, out->camera_from_raster);
    put(rfc.m, out->render_from_camera);
    out->lens_radius = lens_radius; out->focal_distance = focal_distance;
    if (world_from_render16) put(wfr.m, world_from_render16);
    if (render_from_world16) put(rfw.m, render_from_world16);
    return 0;
}

// GaussianFilter::new + FilterSampler::new (imaging/filter.rs:138-216): tabulated filter, row-major [ny][nx].
// Bounds2f::lerp (core/bounds.rs:948-953) interpolates BOTH coordinates between p_min.x and p_max.x.
int pmh_gaussian_filter_table(float rx, float ry, float sigma, float* vals, int cap, int* nx_out, int* ny_out) {
    int nx = (int)(32.0f * rx), ny = (int)(32.0f * ry);
    *nx_out = nx; *ny_out = ny;
    if (!vals) return 0;
    if (cap < nx * ny) return 1;
    float exp_x = gaussian(rx, 0.0f, sigma), exp_y = gaussian(ry, 0.0f, sigma);
    float dmin = std::fmin(-rx, rx), dmax = std::fmax(-rx, rx);
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            float tx = ((float)x + 0.5f) / (float)nx, ty = ((float)y + 0.5f) / (float)ny;
            float p0 = lerpf(dmin, dmax, tx), p1 = lerpf(dmin, dmax, ty);
            float gx = std::fmax(gaussian(p0, 0.0f, sigma) - exp_x, 0.0f);
            float gy = std::fmax(gaussian(p1, 0.0f, sigma) - exp_y, 0.0f);
            vals[y * nx + x] = gx * gy;
        }
    return 0;
}
float pmh_gaussian(float x, float mean, float sd) { return gaussian(x, mean, sd); }
float pmh_fast_exp(float x) { return fast_exp(x); }

// PiecewiseLinearSpectrum::from_interleaved(samples, normalize) (sampling/spectrum.rs:269-310) densified at 360..830
// (DenselySampledSpectrum::new, :196-214; PiecewiseLinearSpectrum::at, :331-357).  Nonzero: the input upstream asserts on
// (an odd or zero count, wavelengths not strictly ascending).
int pmh_piecewise_linear_table(const float* interleaved, int n_floats, int normalize, const float* cie_y471, float* out471) {
    if (!interleaved || !out471 || n_floats < 2 || (n_floats % 2) != 0 || (normalize && !cie_y471)) return 1;
    int n = n_floats / 2;
    std::vector<float> lam, val;
    if (interleaved[0] > 360.0f) { lam.push_back(359.0f); val.push_back(interleaved[1]); }
    for (int i = 0; i < n; ++i) { lam.push_back(interleaved[2 * i]); val.push_back(interleaved[2 * i + 1]); }
    if (lam.back() < 830.0f) { lam.push_back(831.0f); val.push_back(val.back()); }
    for (size_t i = 1; i < lam.size(); ++i) if (!(lam[i - 1] < lam[i])) return 2;        // PiecewiseLinearSpectrum::new's assert
    auto at = [&](float l) {
        if (l < lam.front() || l > lam.back()) return 0.0f;
        int sz = (int)lam.size(), left = 0, right = sz - 1;     // find_interval(|i| lam[i] <= l)
        while (left < right) { int mid = left + (right - left) / 2; if (lam[mid] <= l) left = mid + 1; else right = mid; }
        int i = left == 0 ? 0 : (left >= sz - 1 ? sz - 2 : left - 1);
        if (l == lam[i]) return val[i];
        float t = (l - lam[i]) / (lam[i + 1] - lam[i]);
        return lerpf(val[i], val[i + 1], t);
    };
    if (normalize) {            // to luminance 1: CIE_Y_INTEGRAL / inner_product(Y)
        float integral = 0.0f;
        for (int l = 360; l <= 830; ++l) integral += at((float)l) * cie_y471[l - 360];
        float s = 106.856895f / integral;
        for (auto& v : val) v *= s;
    }
    for (int l = 360; l <= 830; ++l) out471[l - 360] = at((float)l);
    return 0;
}
// ILLUMD65 (sampling/spectrum.rs:64-65): from_interleaved(.., normalize = true)
void pmh_d65_table(const float* cie_y471, const float* d65_interleaved, int n_interleaved, float* out471) {
    (void)pmh_piecewise_linear_table(d65_interleaved, n_interleaved, 1, cie_y471, out471);
}
// Spectrum::to_photometric (sampling/spectrum.rs:152-160)
float pmh_photometric(const float* cie_y471, const float* table471) {
    float y = 0.0f;
    for (int i = 0; i < 471; ++i) y += cie_y471[i] * table471[i];
    return y;
}
// BlackbodySpectrum (sampling/spectrum.rs:376-437) densified; returns the normalisation factor
float pmh_blackbody_table(float temp, float* out471) {
    auto bb = [&](float lambda) {
        if (temp <= 0.0f) return 0.0f;
        const float C = 299792458.0f, H = 6.6260697e-34f, KB = 1.3806488e-23f;
        float l = lambda * 1e-9f;
        float l5 = ((l * l) * (l * l)) * l;
        return (2.0f * H * C * C) / (l5 * (std::exp((H * C) / (l * KB * temp)) - 1.0f));
    };
    float lambda_max = 2.897772e-3f / temp;
    float norm = 1.0f / bb(lambda_max * 1e9f);
    for (int l = 360; l <= 830; ++l) out471[l - 360] = bb((float)l) * norm;
    return norm;
}
// scale * sigmoid(c0,c1,c2)(lambda) [* illuminant(lambda)] densified (RgbAlbedo / RgbUnbounded / RgbIlluminant)
void pmh_rgb_spectrum_table(const float* c3, float scale, const float* illuminant471_or_null, float* out471) {
    for (int l = 360; l <= 830; ++l) {
        float v = sigmoid_poly(c3, (float)l);
        if (illuminant471_or_null) out471[l - 360] = scale * v * illuminant471_or_null[l - 360];
        else out471[l - 360] = scale * v;
    }
}
float pmh_sigmoid_poly(const float* c3, float lambda) { return sigmoid_poly(c3, lambda); }

// RGB -> sigmoid-polynomial coefficients.  The reference looks these up (trilinearly) in a 3x64^3 table file that
// is missing from the snapshot (util/data.rs:1684-1702, .MISSING_LARGE_BLOBS).  That file is the output of pbrt-v4's
// table generator (rgb2spec_opt, Jakob & Hanika 2019), whose published algorithm is restated here step for step - not only
// the optimisation problem but the numerical recipe, because the generator stops each Gauss-Newton solve as soon as the
// squared CIELAB residual falls below 1e-6, so a table entry is "where the iteration stopped", not the optimum:
//   * CIE 1931 observer and the illuminant at 5 nm (every fifth entry of the 1 nm tables), linearly interpolated at
//     283 points over 360..830 nm, Simpson 3/8 weights;
//   * for sRGB the generator's own 6-digit matrices (not the colour space's f32 ones);
//   * central differences (eps 1e-4) for the Jacobian, at most 15 iterations, update first and test the residual of the
//     point before the update, coefficients clamped to |c| <= 200.
// Pinned by the reference's own vector (sampling/spectrum.rs:1031-1037): tests/test_oracle_golden.py.
namespace {
struct RgbSolver {
    static constexpr int NC = 95;                  // 5 nm samples
    static constexpr int N = (NC - 1) * 3 + 1;     // 283 quadrature points
    std::vector<double> tbl, lam;
    double wp[3] = {0, 0, 0};
    double xyz_from_rgb[9], rgb_from_xyz[9];
    bool ok = false;
    // xyz_from_rgb9 == nullptr: sRGB with the generator's literal matrices
    RgbSolver(const float* cie_x, const float* cie_y, const float* cie_z, const float* illuminant471, const float* xyz_from_rgb9)
        : tbl(3 * N), lam(N) {
        if (xyz_from_rgb9 == nullptr) {
            const double to_xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
            const double to_rgb[9] = {3.240479, -1.537150, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
            std::memcpy(xyz_from_rgb, to_xyz, sizeof(to_xyz));
            std::memcpy(rgb_from_xyz, to_rgb, sizeof(to_rgb));
        } else {
            const float* m = xyz_from_rgb9;
            for (int k = 0; k < 9; ++k) xyz_from_rgb[k] = m[k];
            double a = m[0], bb = m[1], c = m[2], d = m[3], e = m[4], f = m[5], gg = m[6], h = m[7], i = m[8];
            double det = a * (e * i - f * h) - bb * (d * i - f * gg) + c * (d * h - e * gg);
            if (det == 0) return;
            double id = 1.0 / det;
            double inv[9] = {(e * i - f * h) * id, (c * h - bb * i) * id, (bb * f - c * e) * id, (f * gg - d * i) * id, (a * i - c * gg) * id,
                             (c * d - a * f) * id, (d * h - e * gg) * id, (bb * gg - a * h) * id, (a * e - bb * d) * id};
            std::memcpy(rgb_from_xyz, inv, sizeof(inv));
        }
        // The generator's tables are doubles parsed from the decimal literals of the CIE data (0.0002321, ...); the f32
        // tables handed in here are those numbers rounded, so the shortest decimal that round-trips gives the literal back.
        auto literal = [](float v) {
            char buf[40];
            auto r = std::to_chars(buf, buf + sizeof(buf) - 1, v);
            *r.ptr = 0;
            return std::strtod(buf, nullptr);
        };
        std::vector<double> t5(4 * NC);
        for (int k = 0; k < NC; ++k) {
            t5[k] = literal(cie_x[5 * k]); t5[NC + k] = literal(cie_y[5 * k]); t5[2 * NC + k] = literal(cie_z[5 * k]);
            t5[3 * NC + k] = (double)illuminant471[5 * k];       // (computed, not a literal)
        }
        auto interp = [&](int tab, double l) {          // linear interpolation of the 5 nm samples
            double x = (l - 360.0) * ((NC - 1) / (830.0 - 360.0));
            int off = (int)x;
            if (off > NC - 2) off = NC - 2;
            const double w = x - off;
            return (1.0 - w) * t5[tab * NC + off] + w * t5[tab * NC + off + 1];
        };
        const double h = (830.0 - 360.0) / (N - 1);
        for (int i = 0; i < N; ++i) {
            const double l = 360.0 + i * h;
            const double xyz[3] = {interp(0, l), interp(1, l), interp(2, l)};
            const double I = interp(3, l);
            double w = 3.0 / 8.0 * h;
            if (i == 0 || i == N - 1) {}
            else if ((i - 1) % 3 == 2) w *= 2.0;
            else w *= 3.0;
            lam[i] = (l - 360.0) / (830.0 - 360.0);
            for (int c = 0; c < 3; ++c) {
                double v = 0;
                for (int k = 0; k < 3; ++k) v += rgb_from_xyz[c * 3 + k] * xyz[k] * I * w;
                tbl[c * N + i] = v;
                wp[c] += xyz[c] * I * w;
            }
        }
        // the generator's illuminant table is pre-divided so that the white point has Y = 1; the table handed in here has
        // another normalisation (the colour space's), so divide now
        const double ynorm = wp[1];
        if (!(ynorm > 0)) return;
        for (auto& v : tbl) v /= ynorm;
        for (int c = 0; c < 3; ++c) wp[c] /= ynorm;
        ok = true;
    }
    void lab(const double* rgbv, double* out) const {
        double xyz[3];
        for (int k = 0; k < 3; ++k) xyz[k] = rgbv[0] * xyz_from_rgb[k * 3] + rgbv[1] * xyz_from_rgb[k * 3 + 1] + rgbv[2] * xyz_from_rgb[k * 3 + 2];
        auto f = [](double t) { const double d = 6.0 / 29.0; return t > d * d * d ? std::cbrt(t) : t / (d * d * 3.0) + 4.0 / 29.0; };
        double fx = f(xyz[0] / wp[0]), fy = f(xyz[1] / wp[1]), fz = f(xyz[2] / wp[2]);
        out[0] = 116.0 * fy - 16.0; out[1] = 500.0 * (fx - fy); out[2] = 200.0 * (fy - fz);
    }
    void residual(const double* target, const double* co, double* res) const {
        double out[3] = {0, 0, 0};
        for (int i = 0; i < N; ++i) {
            double x = (co[0] * lam[i] + co[1]) * lam[i] + co[2];
            double s = 0.5 * x / std::sqrt(1.0 + x * x) + 0.5;
            for (int c = 0; c < 3; ++c) out[c] += tbl[c * N + i] * s;
        }
        double l0[3], l1[3];
        lab(target, l0); lab(out, l1);
        for (int c = 0; c < 3; ++c) res[c] = l0[c] - l1[c];
    }
    // 3x3 solve by LU with partial pivoting
    static bool solve3(double A[3][3], const double* b, double* x) {
        int P[3] = {0, 1, 2};
        for (int i = 0; i < 3; ++i) {
            double mx = 0; int im = i;
            for (int k = i; k < 3; ++k) if (std::fabs(A[k][i]) > mx) { mx = std::fabs(A[k][i]); im = k; }
            if (mx < 1e-15) return false;
            if (im != i) { std::swap(P[i], P[im]); for (int c = 0; c < 3; ++c) std::swap(A[i][c], A[im][c]); }
            for (int j = i + 1; j < 3; ++j) {
                A[j][i] /= A[i][i];
                for (int k = i + 1; k < 3; ++k) A[j][k] -= A[j][i] * A[i][k];
            }
        }
        for (int i = 0; i < 3; ++i) { x[i] = b[P[i]]; for (int k = 0; k < i; ++k) x[i] -= A[i][k] * x[k]; }
        for (int i = 2; i >= 0; --i) { for (int k = i + 1; k < 3; ++k) x[i] -= A[i][k] * x[k]; x[i] /= A[i][i]; }
        return true;
    }
    void solve(const double* target, double* co, int max_it = 15, double tol = 1e-6) const {
        for (int it = 0; it < max_it; ++it) {
            double res[3]; residual(target, co, res);
            double J[3][3];
            for (int k = 0; k < 3; ++k) {
                double t[3] = {co[0], co[1], co[2]}, r0[3], r1[3];
                const double eps = 1e-4;
                t[k] = co[k] - eps; residual(target, t, r0);
                t[k] = co[k] + eps; residual(target, t, r1);
                for (int j = 0; j < 3; ++j) J[j][k] = (r1[j] - r0[j]) * (1.0 / (2 * eps));
            }
            double x[3];
            if (!solve3(J, res, x)) break;
            double rr = 0;
            for (int k = 0; k < 3; ++k) { co[k] -= x[k]; rr += res[k] * res[k]; }
            double mx = std::fmax(std::fmax(std::fabs(co[0]), std::fabs(co[1])), std::fabs(co[2]));
            if (mx > 200) for (int k = 0; k < 3; ++k) co[k] *= 200 / mx;
            if (rr < tol) break;
        }
    }
    // from the normalised wavelength domain to nanometres
    static void to_nm(const double* co, float* out_c3) {
        const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);
        double A = co[0], B = co[1], C = co[2];
        out_c3[0] = (float)(A * (c1 * c1));
        out_c3[1] = (float)(B * c1 - 2 * A * c0 * (c1 * c1));
        out_c3[2] = (float)(C - B * c0 * c1 + A * ((c0 * c1) * (c0 * c1)));
    }
};
}  // namespace

int pmh_rgb_to_coeffs(const float* rgb3, const float* cie_x, const float* cie_y, const float* cie_z,
                      const float* illuminant471, const float* xyz_from_rgb9, float* out_c3) {
    if (rgb3[0] == rgb3[1] && rgb3[1] == rgb3[2]) {   // color_space.rs:130-136
        float x = rgb3[0];
        out_c3[0] = 0.0f; out_c3[1] = 0.0f; out_c3[2] = (x - 0.5f) / std::sqrt(x * (1.0f - x));
        return 0;
    }
    RgbSolver sv(cie_x, cie_y, cie_z, illuminant471, xyz_from_rgb9);
    if (!sv.ok) return 1;
    const double target[3] = {rgb3[0], rgb3[1], rgb3[2]};
    double co[3] = {0, 0, 0};
    sv.solve(target, co, 60, 1e-12);        // no table, no warm start: iterate to convergence
    RgbSolver::to_nm(co, out_c3);
    return 0;
}

// The table the reference loads from src/util/rgbspectrum_srgb.txt (missing upstream), regenerated the way pbrt's
// rgb2spec_opt builds it: z_nodes[k] = smoothstep(smoothstep(k/63)); for each max-component l and each (y, x), the
// brightness axis is swept outwards from k = 64/5 with warm starts.  Layout = CoefficientTable's
// (color/color_space.rs:200-212): index ((l*64 + z)*64 + y)*64 + x, 3 coefficients each.
int pmh_rgb_table_generate(const float* cie_x, const float* cie_y, const float* cie_z, const float* illuminant471,
                           const float* xyz_from_rgb9, float* z_nodes64, float* coeffs, int nthreads) {
    const int R = 64;
    RgbSolver sv(cie_x, cie_y, cie_z, illuminant471, xyz_from_rgb9);
    if (!sv.ok) return 1;
    auto smoothstep = [](double x) { return x * x * (3.0 - 2.0 * x); };
    std::vector<double> scale(R);
    for (int k = 0; k < R; ++k) { scale[k] = smoothstep(smoothstep((double)k / (R - 1))); z_nodes64[k] = (float)scale[k]; }
    auto column = [&](int l, int j, int i) {
        const double x = (double)i / (R - 1), y = (double)j / (R - 1);
        const int start = R / 5;
        auto at = [&](int k, double* co) {
            const double b = (double)z_nodes64[k];     // the generator keeps the nodes in f32
            double rgb[3];
            rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
            sv.solve(rgb, co);
            RgbSolver::to_nm(co, coeffs + 3 * ((((size_t)l * R + k) * R + j) * R + i));
        };
        double co[3] = {0, 0, 0};
        for (int k = start; k < R; ++k) at(k, co);
        co[0] = co[1] = co[2] = 0;
        for (int k = start; k >= 0; --k) at(k, co);
    };
    nthreads = std::max(1, nthreads);
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&]() {
            for (;;) {
                const int w = next.fetch_add(1);
                if (w >= 3 * R) break;
                const int l = w / R, j = w % R;
                for (int i = 0; i < R; ++i) column(l, j, i);
            }
        });
    for (auto& t : th) t.join();
    return 0;
}

// RGBToSpectrumTable::convert (color/color_space.rs:128-185), f32, same operations in the same order
void pmh_rgb_table_lookup(const float* z_nodes64, const float* coeffs, const float* rgb3_in, float* out_c3) {
    const int R = 64;
    const float rgb[3] = {rgb3_in[0] < 0.0f ? 0.0f : rgb3_in[0], rgb3_in[1] < 0.0f ? 0.0f : rgb3_in[1], rgb3_in[2] < 0.0f ? 0.0f : rgb3_in[2]};
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) {
        out_c3[0] = 0.0f; out_c3[1] = 0.0f; out_c3[2] = (rgb[0] - 0.5f) / std::sqrt(rgb[0] * (1.0f - rgb[0]));
        return;
    }
    const int max_i = rgb[0] > rgb[1] ? (rgb[0] > rgb[2] ? 0 : 2) : (rgb[1] > rgb[2] ? 1 : 2);
    const float z = rgb[max_i];
    const float x = rgb[(max_i + 1) % 3] * (float)(R - 1) / z;
    const float y = rgb[(max_i + 2) % 3] * (float)(R - 1) / z;
    const int xi = std::min((int)x, R - 2), yi = std::min((int)y, R - 2);
    int zi;
    {   // find_interval(64, |i| z_nodes[i] < z)  (util/routines.rs:1-32)
        int left = 0, right = R - 1;
        while (left < right) { const int mid = left + (right - left) / 2; if (z_nodes64[mid] < z) left = mid + 1; else right = mid; }
        zi = left == 0 ? 0 : (left >= R - 1 ? R - 2 : left - 1);
    }
    const float dx = x - (float)xi, dy = y - (float)yi;
    const float dz = (z - z_nodes64[zi]) / (z_nodes64[zi + 1] - z_nodes64[zi]);
    auto lerp = [](float a, float b, float t) { return a * (1.0f - t) + b * t; };
    for (int c = 0; c < 3; ++c) {
        auto co = [&](int ddx, int ddy, int ddz) {
            return coeffs[3 * ((((size_t)max_i * R + (zi + ddz)) * R + (yi + ddy)) * R + (xi + ddx)) + c];
        };
        out_c3[c] = lerp(lerp(lerp(co(0, 0, 0), co(1, 0, 0), dx), lerp(co(0, 1, 0), co(1, 1, 0), dx), dy),
                         lerp(lerp(co(0, 0, 1), co(1, 0, 1), dx), lerp(co(0, 1, 1), co(1, 1, 1), dx), dy), dz);
    }
}

// RGBColorSpace::new for sRGB primaries (color/color_space.rs:40-89): xyz_from_rgb and rgb_from_xyz, in the reference's f32
// arithmetic - SquareMatrix<3>::inverse with difference_of_products (square_matrix.rs:123-167), the matrix products with
// mul_add accumulation (:316-332), matrix * tuple with plain += (:345-360).
namespace {
bool invert3(const float m[3][3], float r[3][3]) {
    const float minor12 = dop(m[1][1], m[2][2], m[1][2], m[2][1]);
    const float minor02 = dop(m[1][0], m[2][2], m[1][2], m[2][0]);
    const float minor01 = dop(m[1][0], m[2][1], m[1][1], m[2][0]);
    const float det = std::fmaf(m[0][2], minor01, dop(m[0][0], minor12, m[0][1], minor02));
    if (det == 0.0f) return false;
    const float k = 1.0f / det;
    r[0][0] = k * dop(m[1][1], m[2][2], m[1][2], m[2][1]); r[1][0] = k * dop(m[1][2], m[2][0], m[1][0], m[2][2]);
    r[2][0] = k * dop(m[1][0], m[2][1], m[1][1], m[2][0]); r[0][1] = k * dop(m[0][2], m[2][1], m[0][1], m[2][2]);
    r[1][1] = k * dop(m[0][0], m[2][2], m[0][2], m[2][0]); r[2][1] = k * dop(m[0][1], m[2][0], m[0][0], m[2][1]);
    r[0][2] = k * dop(m[0][1], m[1][2], m[0][2], m[1][1]); r[1][2] = k * dop(m[0][2], m[1][0], m[0][0], m[1][2]);
    r[2][2] = k * dop(m[0][0], m[1][1], m[0][1], m[1][0]);
    return true;
}
}  // namespace
int pmh_srgb_matrices(const float* cie_x, const float* cie_y, const float* cie_z, const float* illuminant471,
                      float* xyz_from_rgb9, float* rgb_from_xyz9) {
    // W = illuminant.to_xyz() = (inner products) / CIE_Y_INTEGRAL
    float X = 0, Y = 0, Z = 0;
    for (int i = 0; i < 471; ++i) { X += illuminant471[i] * cie_x[i]; Y += illuminant471[i] * cie_y[i]; Z += illuminant471[i] * cie_z[i]; }
    const float W[3] = {X / 106.856895f, Y / 106.856895f, Z / 106.856895f};
    auto from_xyy = [](float x, float y, float* o) {   // XYZ::from_xyy with Y = 1
        if (y == 0.0f) { o[0] = o[1] = o[2] = 0.0f; return; }
        o[0] = x * 1.0f / y; o[1] = 1.0f; o[2] = (1.0f - x - y) * 1.0f / y;
    };
    float R[3], G[3], B[3];
    from_xyy(0.64f, 0.33f, R); from_xyy(0.3f, 0.6f, G); from_xyy(0.15f, 0.06f, B);
    const float prim[3][3] = {{R[0], G[0], B[0]}, {R[1], G[1], B[1]}, {R[2], G[2], B[2]}};
    float prim_inv[3][3];
    if (!invert3(prim, prim_inv)) return 1;
    float Cw[3] = {0.0f, 0.0f, 0.0f};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Cw[r] += prim_inv[r][c] * W[c];
    float xr[3][3], rx[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {            // prim * diagonal(Cw)
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = std::fmaf(prim[r][k], k == c ? Cw[c] : 0.0f, acc);
            xr[r][c] = acc;
        }
    if (!invert3(xr, rx)) return 1;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { xyz_from_rgb9[r * 3 + c] = xr[r][c]; rgb_from_xyz9[r * 3 + c] = rx[r][c]; }
    return 0;
}

// ---- Image::write_exr (imaging/image.rs:48-57): minimal OpenEXR 2.0 scanline writer, FLOAT channels, no compression
int pmh_write_exr(const char* path, int width, int height, const float* rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) return 1;
    std::string hdr;
    auto put = [&](const void* p, size_t n) { hdr.append(reinterpret_cast<const char*>(p), n); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_str = [&](const char* z) { hdr.append(z); hdr.push_back('\0'); };
    auto attr = [&](const char* name, const char* type, const std::string& value) {
        put_str(name); put_str(type); put_i32((int32_t)value.size()); hdr.append(value);
    };
    const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    put(magic, 8);
    {   // chlist: channels in alphabetical order
        std::string v;
        for (const char* ch : {"B", "G", "R"}) {
            v.append(ch); v.push_back('\0');
            const int32_t type = 2;   // FLOAT
            v.append(reinterpret_cast<const char*>(&type), 4);
            const unsigned char lin[4] = {0, 0, 0, 0};
            v.append(reinterpret_cast<const char*>(lin), 4);
            const int32_t one = 1;
            v.append(reinterpret_cast<const char*>(&one), 4); v.append(reinterpret_cast<const char*>(&one), 4);
        }
        v.push_back('\0');
        attr("channels", "chlist", v);
    }
    attr("compression", "compression", std::string(1, '\0'));
    {
        const int32_t box[4] = {0, 0, width - 1, height - 1};
        std::string v(reinterpret_cast<const char*>(box), 16);
        attr("dataWindow", "box2i", v);
        attr("displayWindow", "box2i", v);
    }
    attr("lineOrder", "lineOrder", std::string(1, '\0'));
    { const float one = 1.0f; attr("pixelAspectRatio", "float", std::string(reinterpret_cast<const char*>(&one), 4)); }
    { const float z[2] = {0.0f, 0.0f}; attr("screenWindowCenter", "v2f", std::string(reinterpret_cast<const char*>(z), 8)); }
    { const float one = 1.0f; attr("screenWindowWidth", "float", std::string(reinterpret_cast<const char*>(&one), 4)); }
    hdr.push_back('\0');
    const uint64_t line_bytes = 8ull + 3ull * 4ull * (uint64_t)width;
    const uint64_t table_pos = hdr.size();
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return 2;
    bool ok = std::fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
    for (int y = 0; y < height && ok; ++y) {
        const uint64_t off = table_pos + 8ull * (uint64_t)height + (uint64_t)y * line_bytes;
        ok = std::fwrite(&off, 8, 1, f) == 1;
    }
    std::vector<float> line(3 * (size_t)width);
    for (int y = 0; y < height && ok; ++y) {
        const float* row = rgb + 3 * (size_t)y * width;
        for (int x = 0; x < width; ++x) {
            line[x] = row[3 * x + 2];                          // B
            line[(size_t)width + x] = row[3 * x + 1];          // G
            line[2 * (size_t)width + x] = row[3 * x];          // R
        }
        const int32_t yy = y, sz = (int32_t)(3 * 4 * width);
        ok = std::fwrite(&yy, 4, 1, f) == 1 && std::fwrite(&sz, 4, 1, f) == 1 &&
             std::fwrite(line.data(), 4, line.size(), f) == line.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : 3;
}

}  // extern "C"
