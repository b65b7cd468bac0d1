// TEST INFRASTRUCTURE ONLY (see the header of oracle/oracle.py): CPU restatement of the reference constructors whose RESULTS the product's
// host layer hands across the C ABI - camera matrices, filter table, densified spectra, colour-space matrices.  Written from
// the Rust sources alone (not from pbrt-rs_amd/csrc/pm_host.cpp) so that tests/test_host_oracle.py can compare the two
// bit for bit and so that a render parity test can feed the oracle WITHOUT the product's host code (VERDICT r1 weak #4:
// a construction error shared by both sides would otherwise cancel out).
//
// Follows:
//   src/core/transform.rs:59-95 (translate, scale), 207-223 (look_at), 236-255 (perspective), 312-317 (composition)
//   src/core/square_matrix.rs:123-167 (3x3 determinant / inverse), 316-332 (product, mul_add), 345-360 (matrix * tuple)
//   src/camera/base.rs:265-279 (CameraTransform::new), 342-380 (ProjectiveCamera::new)
//   src/camera/perspective.rs:30-57 (PerspectiveCamera::new)
//   src/imaging/filter.rs:137-151,160-163 (GaussianFilter), 183-214 (FilterSampler::new); src/core/bounds.rs:948-953 (lerp, B-quirk)
//   src/sampling/spectrum.rs:28-55 (X, Y, Z), 64-65 (ILLUMD65), 120-160 (inner_product, to_xyz, to_photometric),
//       196-214 (DenselySampledSpectrum::new), 269-310 (from_interleaved), 331-357 (PiecewiseLinear::at),
//       376-437 (blackbody), 922-1019 (RGB spectra)
//   src/color/color_space.rs:40-89 (SRGB, RGBColorSpace::new); src/color/colors.rs:26-47 (from_xyy, xy)
//   src/main.rs:97-105 (the sun's frame)
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "ref_core.h"
#include "ref_integrator.h"
#include "ref_render.h"

namespace ora {
namespace host {

struct T4 { M4 m, inv; };        // Transform { m, m_inv }

static inline M4 m4_identity() {
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0f : 0.0f;
    return r;
}
// square_matrix.rs:316-332: m[i][j] = self[i][k].mul_add(rhs[k][j], m[i][j]) for k = 0..N
static inline M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc = std::fmaf(a.m[i][k], b.m[k][j], acc);
            r.m[i][j] = acc;
        }
    return r;
}
// transform.rs:312-317
static inline T4 compose(const T4& t1, const T4& t2) { T4 r; r.m = m4_mul(t1.m, t2.m); r.inv = m4_mul(t2.inv, t1.inv); return r; }
static inline T4 inverse(const T4& t) { T4 r; r.m = t.inv; r.inv = t.m; return r; }
// transform.rs:59-75
static inline T4 translate(V3 d) {
    T4 r; r.m = m4_identity(); r.inv = m4_identity();
    r.m.m[0][3] = d.x; r.m.m[1][3] = d.y; r.m.m[2][3] = d.z;
    r.inv.m[0][3] = -d.x; r.inv.m[1][3] = -d.y; r.inv.m[2][3] = -d.z;
    return r;
}
// transform.rs:78-95
static inline T4 scale(float x, float y, float z) {
    T4 r; r.m = m4_identity(); r.inv = m4_identity();
    r.m.m[0][0] = x; r.m.m[1][1] = y; r.m.m[2][2] = z;
    r.inv.m[0][0] = 1.0f / x; r.inv.m[1][1] = 1.0f / y; r.inv.m[2][2] = 1.0f / z;
    return r;
}
// transform.rs:46-55 (from_arr: the inverse is computed)
static inline bool from_matrix(const M4& m, T4* out) { out->m = m; return m4_inverse(m, &out->inv); }

// transform.rs:207-223
static inline bool look_at(V3 cam_pos, V3 look_pos, V3 up, T4* out) {
    const V3 dir = normalized(look_pos - cam_pos);
    const V3 right = normalized(cross(normalized(up), dir));
    const V3 new_up = cross(dir, right);
    M4 c2w = m4_identity();
    c2w.m[0][0] = right.x; c2w.m[0][1] = new_up.x; c2w.m[0][2] = dir.x; c2w.m[0][3] = cam_pos.x;
    c2w.m[1][0] = right.y; c2w.m[1][1] = new_up.y; c2w.m[1][2] = dir.y; c2w.m[1][3] = cam_pos.y;
    c2w.m[2][0] = right.z; c2w.m[2][1] = new_up.z; c2w.m[2][2] = dir.z; c2w.m[2][3] = cam_pos.z;
    out->inv = c2w;
    return m4_inverse(c2w, &out->m);
}

// transform.rs:236-255
static inline bool perspective(float fov_deg, float near_z, float far_z, T4* out) {
    M4 p = m4_identity();
    p.m[2][2] = far_z / (far_z - near_z);
    p.m[2][3] = -far_z * near_z / (far_z - near_z);
    p.m[3][2] = 1.0f; p.m[3][3] = 0.0f;
    T4 persp;
    if (!from_matrix(p, &persp)) return false;
    // f32::to_radians multiplies by the f32 constant pi/180
    const float rad = fov_deg * (PI_F / 180.0f);
    const float inv_tan = 1.0f / std::tan(rad / 2.0f);
    *out = compose(scale(inv_tan, inv_tan, 1.0f), persp);
    return true;
}

struct CameraOut { OCamera cam; T4 world_from_render; };
// camera/base.rs:265-279 + 342-380 + camera/perspective.rs:30-57
static inline bool perspective_camera(const T4& world_from_camera, float fov_deg, const float* screen_window /* min.x min.y max.x max.y */,
                                      int res_x, int res_y, float lens_radius, float focal_distance, CameraOut* out) {
    // CameraTransform::new
    const V3 p_camera = xf_point(world_from_camera.m, V3(0.0f, 0.0f, 0.0f));
    const T4 world_from_render = translate(p_camera);
    const T4 render_from_world = inverse(world_from_render);
    const T4 render_from_camera = compose(render_from_world, world_from_camera);
    // PerspectiveCamera::new
    T4 screen_from_camera;
    if (!perspective(fov_deg, 0.01f, 1000.0f, &screen_from_camera)) return false;
    // ProjectiveCamera::new
    const float x0 = screen_window[0], y0 = screen_window[1], x1 = screen_window[2], y1 = screen_window[3];
    const T4 ndc_from_screen = compose(scale(1.0f / (x1 - x0), 1.0f / (y1 - y0), 1.0f), translate(V3(-x0, -y1, 0.0f)));
    const T4 raster_from_ndc = scale((float)res_x, -(float)res_y, 1.0f);
    const T4 raster_from_screen = compose(raster_from_ndc, ndc_from_screen);
    const T4 screen_from_raster = inverse(raster_from_screen);
    const T4 camera_from_raster = compose(inverse(screen_from_camera), screen_from_raster);
    std::memcpy(out->cam.camera_from_raster, camera_from_raster.m.m, sizeof(float) * 16);
    std::memcpy(out->cam.render_from_camera, render_from_camera.m.m, sizeof(float) * 16);
    out->cam.lens_radius = lens_radius; out->cam.focal_distance = focal_distance;
    out->world_from_render = world_from_render;
    return true;
}

// main.rs:97-105: the frame whose z axis points from `to` towards `from`
static inline bool frame_towards(V3 from, V3 to, T4* out) {
    const V3 w = normalized(from - to);
    V3 v1, v2;
    coordinate_system(w, &v1, &v2);
    M4 m = m4_identity();
    m.m[0][0] = v1.x; m.m[0][1] = v2.x; m.m[0][2] = w.x;
    m.m[1][0] = v1.y; m.m[1][1] = v2.y; m.m[1][2] = w.y;
    m.m[2][0] = v1.z; m.m[2][1] = v2.z; m.m[2][2] = w.z;
    return from_matrix(m, out);
}

// imaging/filter.rs:137-151,160-163,183-214: the table FilterSampler::new tabulates, vals[y][x] (Array2D over Bounds2i is
// x-fastest).  Bounds2f::lerp (bounds.rs:948-953) interpolates BOTH coordinates between p_min.x and p_max.x.
static inline void gaussian_filter_table(float rx, float ry, float sigma, std::vector<float>* vals, int* nx, int* ny) {
    const float exp_x = gaussian(rx, 0.0f, sigma), exp_y = gaussian(ry, 0.0f, sigma);
    const int sx = (int)(32.0f * rx), sy = (int)(32.0f * ry);
    vals->assign((size_t)sx * sy, 0.0f);
    const float lo = -rx, hi = rx;
    for (int y = 0; y < sy; ++y)
        for (int x = 0; x < sx; ++x) {
            const float tx = ((float)x + 0.5f) / (float)sx, ty = ((float)y + 0.5f) / (float)sy;
            const float px = lerpf(lo, hi, tx), py = lerpf(lo, hi, ty);
            (*vals)[(size_t)y * sx + x] =
                rmax(gaussian(px, 0.0f, sigma) - exp_x, 0.0f) * rmax(gaussian(py, 0.0f, sigma) - exp_y, 0.0f);
        }
    *nx = sx; *ny = sy;
}

// ---- spectra ---------------------------------------------------------------------------------------------------------
static const float CIE_Y_INTEGRAL = 106.856895f;     // spectrum.rs:71
struct PLSample { float lambda, value; };

// spectrum.rs:269-310 (without the final scale)
static inline std::vector<PLSample> from_interleaved(const float* il, int n_floats) {
    std::vector<PLSample> s;
    const int n = n_floats / 2;
    if (il[0] > 360.0f) s.push_back({360.0f - 1.0f, il[1]});
    for (int i = 0; i < n; ++i) s.push_back({il[2 * i], il[2 * i + 1]});
    if (s.back().lambda < 830.0f) s.push_back({830.0f + 1.0f, s.back().value});
    return s;
}
// spectrum.rs:331-357; find_interval(size, pred) (util/routines.rs:1-32) = last index whose predicate holds, clamped to [0, size-2]
static inline float pl_at(const std::vector<PLSample>& s, float lambda) {
    if (s.empty() || lambda < s.front().lambda || lambda > s.back().lambda) return 0.0f;
    long lo = 0;
    for (long i = 0; i < (long)s.size(); ++i) if (s[i].lambda <= lambda) lo = i; else break;
    if (lo > (long)s.size() - 2) lo = (long)s.size() - 2;
    if (lo < 0) lo = 0;
    if (lambda == s[lo].lambda) return s[lo].value;
    const float t = (lambda - s[lo].lambda) / (s[lo + 1].lambda - s[lo].lambda);
    return lerpf(s[lo].value, s[lo + 1].value, t);
}
// ILLUMD65 (spectrum.rs:64-65) densified as RGBColorSpace::new keeps it (color_space.rs:81): normalised to luminance 1
// (from_interleaved(.., true): scale by CIE_Y_INTEGRAL / inner_product(Y)), sampled at 360..830
static inline void d65_dense(const float* cie_y471, const float* il, int n_floats, float* out471) {
    std::vector<PLSample> s = from_interleaved(il, n_floats);
    float integral = 0.0f;
    for (int i = 0; i <= 470; ++i) integral += pl_at(s, 360.0f + (float)i) * cie_y471[i];
    const float k = CIE_Y_INTEGRAL / integral;
    for (auto& e : s) e.value *= k;
    for (int i = 0; i <= 470; ++i) out471[i] = pl_at(s, (float)(360 + i));
}
// PiecewiseLinearSpectrum::from_interleaved(.., normalize) densified (spectrum.rs:196-214,269-310)
static inline void piecewise_linear_dense(const float* il, int n_floats, bool normalize, const float* cie_y471, float* out471) {
    std::vector<PLSample> s = from_interleaved(il, n_floats);
    if (normalize) {
        float integral = 0.0f;
        for (int i = 0; i <= 470; ++i) integral += pl_at(s, 360.0f + (float)i) * cie_y471[i];
        const float k = CIE_Y_INTEGRAL / integral;
        for (auto& e : s) e.value *= k;
    }
    for (int i = 0; i <= 470; ++i) out471[i] = pl_at(s, (float)(360 + i));
}
// spectrum.rs:152-160
static inline float photometric(const float* cie_y471, const float* table471) {
    float y = 0.0f;
    for (int i = 0; i <= 470; ++i) y += cie_y471[i] * table471[i];
    return y;
}
// BlackbodySpectrum::new(T) densified (spectrum.rs:402-437, f32 build: Wien constant 2.897772e-3)
static inline float blackbody_dense(float temp, float* out471) {
    const float lambda_max = 2.897772e-3f / temp;
    const float norm = 1.0f / blackbody(lambda_max * 1e9f, temp);
    for (int i = 0; i <= 470; ++i) out471[i] = blackbody((float)(360 + i), temp) * norm;
    return norm;
}
// RgbAlbedoSpectrum / RgbUnboundedSpectrum / RgbIlluminantSpectrum densified (spectrum.rs:922-1019): scale * rsp(l) [* illum(l)]
static inline void rgb_spectrum_dense(const float* c3, float scale, const float* illum471_or_null, float* out471) {
    for (int i = 0; i <= 470; ++i) {
        const float l = (float)(360 + i);
        float v = scale * sigmoid_poly(c3[0], c3[1], c3[2], l);
        if (illum471_or_null) v = v * illum471_or_null[i];
        out471[i] = v;
    }
}

// ---- colour space (color_space.rs:40-89, SRGB primaries) -----------------------------------------------------------------
static inline float det3(const float m[3][3]) {                 // square_matrix.rs:123-138
    const float minor12 = difference_of_products(m[1][1], m[2][2], m[1][2], m[2][1]);
    const float minor02 = difference_of_products(m[1][0], m[2][2], m[1][2], m[2][0]);
    const float minor01 = difference_of_products(m[1][0], m[2][1], m[1][1], m[2][0]);
    return std::fmaf(m[0][2], minor01, difference_of_products(m[0][0], minor12, m[0][1], minor02));
}
static inline bool inv3(const float m[3][3], float r[3][3]) {   // square_matrix.rs:141-167
    const float det = det3(m);
    if (det == 0.0f) return false;
    const float id = 1.0f / det;
    r[0][0] = id * difference_of_products(m[1][1], m[2][2], m[1][2], m[2][1]);
    r[1][0] = id * difference_of_products(m[1][2], m[2][0], m[1][0], m[2][2]);
    r[2][0] = id * difference_of_products(m[1][0], m[2][1], m[1][1], m[2][0]);
    r[0][1] = id * difference_of_products(m[0][2], m[2][1], m[0][1], m[2][2]);
    r[1][1] = id * difference_of_products(m[0][0], m[2][2], m[0][2], m[2][0]);
    r[2][1] = id * difference_of_products(m[0][1], m[2][0], m[0][0], m[2][1]);
    r[0][2] = id * difference_of_products(m[0][1], m[1][2], m[0][2], m[1][1]);
    r[1][2] = id * difference_of_products(m[0][2], m[1][0], m[0][0], m[1][2]);
    r[2][2] = id * difference_of_products(m[0][0], m[1][1], m[0][1], m[1][0]);
    return true;
}
static inline bool srgb_matrices(const float* cie_x, const float* cie_y, const float* cie_z, const float* illum471,
                                 float xyz_from_rgb[3][3], float rgb_from_xyz[3][3]) {
    // W = illuminant.to_xyz() (spectrum.rs:120-150): three Riemann sums, each divided by CIE_Y_INTEGRAL
    float W[3] = {0.0f, 0.0f, 0.0f};
    const float* bars[3] = {cie_x, cie_y, cie_z};
    for (int c = 0; c < 3; ++c) {
        float integral = 0.0f;
        for (int i = 0; i <= 470; ++i) integral += illum471[i] * bars[c][i];
        W[c] = integral / CIE_Y_INTEGRAL;
    }
    // XYZ::from_xyy(p, None) (colors.rs:26-35), Y = 1
    auto from_xy = [](float x, float y, float out[3]) {
        if (y == 0.0f) { out[0] = out[1] = out[2] = 0.0f; return; }
        out[0] = x * 1.0f / y; out[1] = 1.0f; out[2] = (1.0f - x - y) * 1.0f / y;
    };
    float R[3], G[3], B[3];
    from_xy(0.64f, 0.33f, R); from_xy(0.3f, 0.6f, G); from_xy(0.15f, 0.06f, B);
    const float rgb[3][3] = {{R[0], G[0], B[0]}, {R[1], G[1], B[1]}, {R[2], G[2], B[2]}};
    float rgb_inv[3][3];
    if (!inv3(rgb, rgb_inv)) return false;
    float C[3] = {0.0f, 0.0f, 0.0f};                       // matrix * tuple: ret[i] += m[i][j] * rhs[j]
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i] += rgb_inv[i][j] * W[j];
    // rgb * diagonal(C): the general product with mul_add over k
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = std::fmaf(rgb[i][k], k == j ? C[j] : 0.0f, acc);
            xyz_from_rgb[i][j] = acc;
        }
    return inv3(xyz_from_rgb, rgb_from_xyz);
}

}  // namespace host
}  // namespace ora
