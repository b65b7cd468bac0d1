// pbrt_mi355.hpp — C++17 host-side mirror of the reference's interface for the ray-intersection / path-tracing path,
// header-only, over the C ABI of libpbrt_mi355.so (pbrt_mi355.h, pbrt_mi355_host.h).
//
// The reference is Rust; its constructors are mirrored here by name and argument meaning so that code written against
// pbrt-rs reads the same (reference file:line per type):
//   Transform                 core/transform.rs:22-255        (translate, look_at, from_matrix, inverse, operator*)
//   Sphere                    shapes/sphere.rs:34-41 (builder arguments)
//   BilinearPatchMesh         shapes/bilinear_patch.rs:631-685
//   TriangleMesh              extension: a triangle is the degenerate patch (v0, v1, v2, v2)
//   Spectra                   sampling/spectrum.rs (CIE tables, sRGB colour space, Rgb*/Blackbody spectrum constructors)
//   GpuAggregate              primitives/aggregates.rs:25-47,532-655  BVHAggregate::new / bounds / intersect / intersect_p
//   WavefrontIntegrator       integrators/simple_path.rs:40-66, integrators/base.rs:32-45  ::new / render()
// Errors: every failing C call throws pbrt_mi355::Error with the library's message (the reference panics).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "pbrt_mi355_host.h"

namespace pbrt_mi355 {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------- Transform
struct Transform {
    std::array<float, 16> m{}, m_inv{};
    static Transform identity() { Transform t; for (int i = 0; i < 4; ++i) t.m[5 * i] = t.m_inv[5 * i] = 1.0f; return t; }
    static Transform translate(float x, float y, float z) {
        Transform t = identity();
        t.m[3] = x; t.m[7] = y; t.m[11] = z; t.m_inv[3] = -x; t.m_inv[7] = -y; t.m_inv[11] = -z;
        return t;
    }
    static Transform from_matrix(const std::array<float, 16>& a) {
        Transform t; t.m = a;
        if (pmh_mat4_inverse(a.data(), t.m_inv.data()) != 0) throw Error("Transform::from_matrix: singular matrix");
        return t;
    }
    static Transform look_at(const float (&pos)[3], const float (&look)[3], const float (&up)[3]) {
        Transform t;
        if (pmh_look_at(pos, look, up, t.m.data(), t.m_inv.data()) != 0) throw Error("Transform::look_at: degenerate");
        return t;
    }
    Transform inverse() const { Transform t; t.m = m_inv; t.m_inv = m; return t; }
    Transform operator*(const Transform& o) const {
        Transform t;
        pmh_mat4_mul(m.data(), o.m.data(), t.m.data());
        pmh_mat4_mul(o.m_inv.data(), m_inv.data(), t.m_inv.data());
        return t;
    }
    // Transform * Point3f (core/transform.rs:320-335; never divides by w)
    std::array<float, 3> point(float x, float y, float z) const {
        return {x * m[0] + y * m[1] + z * m[2] + m[3], x * m[4] + y * m[5] + z * m[6] + m[7], x * m[8] + y * m[9] + z * m[10] + m[11]};
    }
};

// ---------------------------------------------------------------------------------------------- shapes
struct Sphere {
    float radius, z_min, z_max, phi_max;
    Transform render_from_object;
    bool reverse_orientation = false;
};
struct BilinearPatchMesh {
    std::vector<float> positions;      // render space, 3 per vertex
    std::vector<uint32_t> indices;     // 4 per patch: p00 p10 p01 p11
    std::vector<float> normals;        // optional: 3 per vertex, render space
    bool reverse_orientation = false, transform_swaps_handedness = false;
};
struct TriangleMesh {
    std::vector<float> positions;      // render space
    std::vector<uint32_t> indices;     // 3 per triangle
};
using Primitive = std::variant<Sphere, BilinearPatchMesh, TriangleMesh>;

// ---------------------------------------------------------------------------------------------- spectra
class Spectra {
public:
    std::vector<float> cie_x, cie_y, cie_z, d65;
    std::array<float, 9> xyz_from_rgb{}, rgb_from_xyz{};

    // `tables_bin`: pbrt-rs_amd/data/cie_tables.bin (CIE 1931 observer + D65, made by scripts/extract_cie_tables.py)
    explicit Spectra(const std::string& tables_bin) {
        std::FILE* f = std::fopen(tables_bin.c_str(), "rb");
        if (!f) throw Error("Spectra: cannot open " + tables_bin);
        char magic[8];
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PMCIE1\0\0", 8) == 0;
        cie_x.resize(471); cie_y.resize(471); cie_z.resize(471);
        ok = ok && std::fread(cie_x.data(), 4, 471, f) == 471 && std::fread(cie_y.data(), 4, 471, f) == 471 &&
             std::fread(cie_z.data(), 4, 471, f) == 471;
        uint32_t n = 0;
        ok = ok && std::fread(&n, 4, 1, f) == 1 && n < 4096;
        std::vector<float> d65i(n);
        ok = ok && std::fread(d65i.data(), 4, n, f) == n;
        std::fclose(f);
        if (!ok) throw Error("Spectra: malformed " + tables_bin);
        d65.resize(471);
        pmh_d65_table(cie_y.data(), d65i.data(), (int)n, d65.data());
        if (pmh_srgb_matrices(cie_x.data(), cie_y.data(), cie_z.data(), d65.data(), xyz_from_rgb.data(), rgb_from_xyz.data()) != 0)
            throw Error("Spectra: sRGB matrices");
        load_or_generate_rgb_table(tables_bin);
    }
    float photometric(const std::vector<float>& table) const { return pmh_photometric(cie_y.data(), table.data()); }
    // RGBColorSpace::to_rgb_coeffs for sRGB (color/color_space.rs:110-184): trilinear lookup in the 3 x 64^3 table.  The
    // reference reads it from src/util/rgbspectrum_srgb.txt, missing upstream; it is regenerated once with the published
    // generator's recipe (pmh_rgb_table_generate) and cached as `rgbspectrum_srgb.bin` next to the CIE tables.
    std::array<float, 3> rgb_coeffs(float r, float g, float b) const {
        const float rgb[3] = {std::fmax(r, 0.0f), std::fmax(g, 0.0f), std::fmax(b, 0.0f)};
        std::array<float, 3> c{};
        pmh_rgb_table_lookup(z_nodes_.data(), coeffs_.data(), rgb, c.data());
        return c;
    }
    // RgbAlbedoSpectrum::new(&SRGB, rgb)
    PmSpectrum rgb_albedo(float r, float g, float b) const {
        const auto c = rgb_coeffs(r, g, b);
        return PmSpectrum{PM_SPEC_RGB_ALBEDO, c[0], c[1], c[2], 0.0f, nullptr};
    }
    // RgbIlluminantSpectrum::new(&SRGB, rgb), densified as lights do (lights/base.rs:177-188)
    std::vector<float> rgb_illuminant_table(float r, float g, float b) const {
        const float m = std::fmax(r, std::fmax(g, b));
        const float scale = 2.0f * m;
        const auto c = scale != 0.0f ? rgb_coeffs(r / scale, g / scale, b / scale) : rgb_coeffs(0, 0, 0);
        std::vector<float> out(471);
        pmh_rgb_spectrum_table(c.data(), scale, d65.data(), out.data());
        return out;
    }
    // PiecewiseLinearSpectrum::from_interleaved({lambda0, v0, lambda1, v1, ...}, normalize) densified (sampling/spectrum.rs:269-310)
    std::vector<float> piecewise_linear_table(const std::vector<float>& interleaved, bool normalize = false) const {
        std::vector<float> out(471);
        if (pmh_piecewise_linear_table(interleaved.data(), (int)interleaved.size(), normalize ? 1 : 0, cie_y.data(), out.data()) != 0)
            throw Error("piecewise-linear spectrum: an even, non-zero number of values with strictly ascending wavelengths is required");
        return out;
    }
    std::vector<float> blackbody_table(float kelvin) const { std::vector<float> out(471); pmh_blackbody_table(kelvin, out.data()); return out; }
    // PixelSensor::with_xyz_matching: r_bar, g_bar, b_bar = X, Y, Z
    std::vector<float> sensor_xyz_tables() const {
        std::vector<float> t(cie_x); t.insert(t.end(), cie_y.begin(), cie_y.end()); t.insert(t.end(), cie_z.begin(), cie_z.end());
        return t;
    }

private:
    std::vector<float> z_nodes_, coeffs_;
    void load_or_generate_rgb_table(const std::string& tables_bin) {
        const size_t n = 3ull * 64 * 64 * 64 * 3;
        z_nodes_.assign(64, 0.0f); coeffs_.assign(n, 0.0f);
        const size_t slash = tables_bin.find_last_of('/');
        const std::string path = (slash == std::string::npos ? std::string() : tables_bin.substr(0, slash + 1)) + "rgbspectrum_srgb.bin";
        if (std::FILE* f = std::fopen(path.c_str(), "rb")) {
            char magic[8];
            const bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PMRGBT2\0", 8) == 0 &&
                            std::fread(z_nodes_.data(), 4, 64, f) == 64 && std::fread(coeffs_.data(), 4, n, f) == n;
            std::fclose(f);
            if (ok) return;
        }
        if (pmh_rgb_table_generate(cie_x.data(), cie_y.data(), cie_z.data(), d65.data(), nullptr, z_nodes_.data(),
                                   coeffs_.data(), 8) != 0)
            throw Error("Spectra: rgb table generation failed");
        if (std::FILE* f = std::fopen(path.c_str(), "wb")) {      // best effort cache
            std::fwrite("PMRGBT2\0", 1, 8, f); std::fwrite(z_nodes_.data(), 4, 64, f); std::fwrite(coeffs_.data(), 4, n, f);
            std::fclose(f);
        }
    }
};

// ---------------------------------------------------------------------------------------------- scene description
struct Material { int type = PM_MAT_DIFFUSE; PmSpectrum spectrum{}; float u_roughness = 0, v_roughness = 0; bool remap_roughness = false; };
inline Material diffuse(const PmSpectrum& reflectance) { Material m; m.type = PM_MAT_DIFFUSE; m.spectrum = reflectance; return m; }
inline Material dielectric(float eta, float u_rough = 0, float v_rough = 0, bool remap = false) {
    Material m; m.type = PM_MAT_DIELECTRIC; m.spectrum = PmSpectrum{PM_SPEC_CONSTANT, eta, 0, 0, 0, nullptr};
    m.u_roughness = u_rough; m.v_roughness = v_rough; m.remap_roughness = remap; return m;
}
struct Light { int type; float scale; Transform render_from_light = Transform::identity(); std::vector<float> spectrum; };

struct Film {
    int full_resolution[2] = {0, 0};
    int pixel_bounds[4] = {0, 0, 0, 0};
    float max_component_value = std::numeric_limits<float>::infinity(), imaging_ratio = 1.0f;
    std::array<float, 9> output_rgb_from_sensor_rgb{};
    std::vector<float> sensor;           // 3 x 471
    int filter_kind = PM_FILTER_GAUSSIAN; float filter_radius[2] = {1.5f, 1.5f};
    int filter_nx = 0, filter_ny = 0; std::vector<float> filter_vals;
};
// RGBFilm::new(.. sensor: PixelSensor::with_xyz_matching(&SRGB, None, ratio), filter: GaussianFilter::new(radius, sigma))
inline Film make_film(const Spectra& sp, int w, int h, float rx = 1.5f, float ry = 1.5f, float sigma = 0.5f) {
    Film f;
    f.full_resolution[0] = w; f.full_resolution[1] = h; f.pixel_bounds[2] = w; f.pixel_bounds[3] = h;
    f.output_rgb_from_sensor_rgb = sp.rgb_from_xyz;
    f.sensor = sp.sensor_xyz_tables();
    f.filter_radius[0] = rx; f.filter_radius[1] = ry;
    pmh_gaussian_filter_table(rx, ry, sigma, nullptr, 0, &f.filter_nx, &f.filter_ny);
    f.filter_vals.resize((size_t)f.filter_nx * f.filter_ny);
    if (pmh_gaussian_filter_table(rx, ry, sigma, f.filter_vals.data(), (int)f.filter_vals.size(), &f.filter_nx, &f.filter_ny) != 0)
        throw Error("make_film: filter table");
    return f;
}

struct SceneDescription {
    std::vector<Primitive> prims;
    std::vector<uint32_t> prim_materials;   // one id per primitive (meshes: one per face)
    std::vector<Material> materials;
    std::vector<Light> lights;
    PmCamera camera{};
    Transform render_from_world = Transform::identity();
    Film film;
};
// PerspectiveCamera::builder()...build() (camera/perspective.rs:30-86): fills sd.camera and sd.render_from_world
inline void set_perspective_camera(SceneDescription& sd, const Transform& world_from_camera, float fov_deg, int w, int h,
                                   float lens_radius = 0.0f, float focal_distance = 3.0e38f) {
    const float sw[4] = {-1.0f, -1.0f, 1.0f, 1.0f};
    Transform rfw;
    if (pmh_perspective_camera(world_from_camera.m.data(), world_from_camera.m_inv.data(), fov_deg, sw, w, h, lens_radius,
                               focal_distance, &sd.camera, rfw.m_inv.data(), rfw.m.data()) != 0)
        throw Error("set_perspective_camera failed");
    sd.render_from_world = rfw;
}

// ---------------------------------------------------------------------------------------------- GpuAggregate
class GpuAggregate {
public:
    PmAccelInfo info{};
    GpuAggregate(const std::vector<Primitive>& prims, int max_prims_in_node = 4, int split_method = PM_SPLIT_SAH, int device_id = 0) {
        if (pm_create(device_id, &ctx_) != PM_OK) throw Error("pm_create failed (is a gfx950 GPU visible? there is no CPU fallback)");
        try {
            for (const auto& p : prims) {
                uint32_t first = 0;
                if (const auto* s = std::get_if<Sphere>(&p)) {
                    PmSphere ps{};
                    ps.radius = s->radius; ps.z_min = s->z_min; ps.z_max = s->z_max; ps.phi_max_deg = s->phi_max;
                    std::memcpy(ps.render_from_object, s->render_from_object.m.data(), 64);
                    std::memcpy(ps.object_from_render, s->render_from_object.m_inv.data(), 64);
                    ps.reverse_orientation = s->reverse_orientation ? 1u : 0u;
                    check(pm_add_spheres(ctx_, &ps, 1, &first, (uint32_t)sizeof(PmSphere)));
                } else if (const auto* m = std::get_if<BilinearPatchMesh>(&p)) {
                    check(pm_add_patch_mesh(ctx_, m->positions.data(), (uint32_t)(m->positions.size() / 3), m->indices.data(),
                                            (uint32_t)(m->indices.size() / 4), m->normals.empty() ? nullptr : m->normals.data(),
                                            m->reverse_orientation, m->transform_swaps_handedness, &first));
                } else {
                    const auto& t = std::get<TriangleMesh>(p);
                    check(pm_add_triangle_mesh(ctx_, t.positions.data(), (uint32_t)(t.positions.size() / 3), t.indices.data(),
                                               (uint32_t)(t.indices.size() / 3), &first));
                }
            }
            check(pm_build_accel(ctx_, split_method, max_prims_in_node, &info, (uint32_t)sizeof(PmAccelInfo)));
        } catch (...) { pm_destroy(ctx_); ctx_ = nullptr; throw; }
    }
    ~GpuAggregate() { if (ctx_) pm_destroy(ctx_); }
    GpuAggregate(const GpuAggregate&) = delete;
    GpuAggregate& operator=(const GpuAggregate&) = delete;

    // Primitive trait, batched
    void bounds(float lo[3], float hi[3]) const { for (int k = 0; k < 3; ++k) { lo[k] = info.bounds[k]; hi[k] = info.bounds[3 + k]; } }
    void intersect(const PmRay* rays, uint64_t n, PmHit* hits) { check(pm_trace_closest(ctx_, rays, n, hits)); }     // host or device pointers
    void intersect_p(const PmRay* rays, uint64_t n, uint8_t* occluded) { check(pm_trace_any(ctx_, rays, n, occluded)); }
    float last_kernel_ms() { float ms = 0; check(pm_last_kernel_ms(ctx_, &ms)); return ms; }
    pm_ctx* ctx() { return ctx_; }
    void check(int rc) const {
        if (rc != PM_OK) { const char* m = pm_last_error(ctx_); throw Error("pbrt_mi355 status " + std::to_string(rc) + ": " + (m ? m : "")); }
    }
private:
    pm_ctx* ctx_ = nullptr;
};

// ---------------------------------------------------------------------------------------------- WavefrontIntegrator
class WavefrontIntegrator {
public:
    PmRenderParams params{};
    // SimplePathIntegrator::new(max_depth, sample_lights, sample_bsdf, camera, sampler(spp, seed), aggregate, lights)
    WavefrontIntegrator(int max_depth, bool sample_lights, bool sample_bsdf, const SceneDescription& sd, int spp, uint32_t seed = 0,
                        int device_id = 0, int integrator = PM_INTEGRATOR_SIMPLE_PATH)
        : sd_(sd), agg_(sd.prims, 4, PM_SPLIT_SAH, device_id) {
        std::vector<PmMaterial> mats(sd_.materials.size());
        for (size_t i = 0; i < mats.size(); ++i) {
            mats[i].type = sd_.materials[i].type; mats[i].spectrum = sd_.materials[i].spectrum;
            mats[i].u_roughness = sd_.materials[i].u_roughness; mats[i].v_roughness = sd_.materials[i].v_roughness;
            mats[i].remap_roughness = sd_.materials[i].remap_roughness ? 1 : 0;
        }
        agg_.check(pm_set_materials(agg_.ctx(), mats.data(), (uint32_t)mats.size(), (uint32_t)sizeof(PmMaterial)));
        agg_.check(pm_set_prim_materials(agg_.ctx(), sd_.prim_materials.data(), (uint32_t)sd_.prim_materials.size()));
        std::vector<PmLight> lights(sd_.lights.size());
        for (size_t i = 0; i < lights.size(); ++i) {
            lights[i].type = sd_.lights[i].type; lights[i].scale = sd_.lights[i].scale;
            std::memcpy(lights[i].render_from_light, sd_.lights[i].render_from_light.m.data(), 64);
            lights[i].spectrum = sd_.lights[i].spectrum.data();
        }
        agg_.check(pm_set_lights(agg_.ctx(), lights.data(), (uint32_t)lights.size(), (uint32_t)sizeof(PmLight)));
        agg_.check(pm_set_camera(agg_.ctx(), &sd_.camera, (uint32_t)sizeof(PmCamera)));
        PmFilm f{};
        std::memcpy(f.full_resolution, sd_.film.full_resolution, 8); std::memcpy(f.pixel_bounds, sd_.film.pixel_bounds, 16);
        f.max_component_value = sd_.film.max_component_value; f.imaging_ratio = sd_.film.imaging_ratio;
        std::memcpy(f.output_rgb_from_sensor_rgb, sd_.film.output_rgb_from_sensor_rgb.data(), 36);
        f.sensor_rgb_bar = sd_.film.sensor.data();
        f.filter_kind = sd_.film.filter_kind; f.filter_radius[0] = sd_.film.filter_radius[0]; f.filter_radius[1] = sd_.film.filter_radius[1];
        f.filter_nx = sd_.film.filter_nx; f.filter_ny = sd_.film.filter_ny; f.filter_vals = sd_.film.filter_vals.data();
        agg_.check(pm_set_film(agg_.ctx(), &f, (uint32_t)sizeof(PmFilm)));
        params.integrator = integrator; params.max_depth = max_depth; params.sample_lights = sample_lights; params.sample_bsdf = sample_bsdf;
        params.spp = spp; params.seed = seed;
        std::memcpy(params.window, sd_.film.pixel_bounds, 16);
        params.s0 = 0; params.s1 = spp;
    }
    int width() const { return sd_.film.pixel_bounds[2] - sd_.film.pixel_bounds[0]; }
    int height() const { return sd_.film.pixel_bounds[3] - sd_.film.pixel_bounds[1]; }
    // Integrate::render(): the whole image, or params.window / params.s0..s1 of it
    PmRenderStats render() { agg_.check(pm_render(agg_.ctx(), &params, (uint32_t)sizeof(PmRenderParams))); PmRenderStats st{}; agg_.check(pm_get_render_stats(agg_.ctx(), &st, (uint32_t)sizeof(PmRenderStats))); return st; }
    // the same in the reference's waves of sample indices (integrators/base.rs:85-127: [0,1) [1,2) [2,4) ... doubling up to 64);
    // after each wave on_wave(wave_start, wave_end) may read a progressive image.  Final film bit-identical to render().
    template <class F>
    PmRenderStats render_waves(F on_wave) {
        PmRenderStats st{};
        const int spp = params.spp;
        int wave_start = 0, wave_end = 1, next_wave_size = 1;
        while (wave_start < spp) {
            PmRenderParams p = params;
            p.s0 = wave_start; p.s1 = wave_end;
            agg_.check(pm_render(agg_.ctx(), &p, (uint32_t)sizeof(PmRenderParams)));
            on_wave(wave_start, wave_end);
            wave_start = wave_end;
            wave_end = std::min(spp, wave_end + next_wave_size);
            next_wave_size = std::min(2 * next_wave_size, 64);
        }
        agg_.check(pm_get_render_stats(agg_.ctx(), &st, (uint32_t)sizeof(PmRenderStats)));
        return st;
    }
    void clear() { agg_.check(pm_film_clear(agg_.ctx())); }
    std::vector<double> film() { std::vector<double> v(4 * (size_t)width() * height()); agg_.check(pm_film_read(agg_.ctx(), v.data())); return v; }
    std::vector<float> image() { std::vector<float> v(3 * (size_t)width() * height()); agg_.check(pm_film_image(agg_.ctx(), v.data())); return v; }
    // RGBFilm::write_image -> Image::write_exr
    void write_exr(const std::string& path) {
        const auto img = image();
        if (pmh_write_exr(path.c_str(), width(), height(), img.data()) != 0) throw Error("cannot write " + path);
    }
    GpuAggregate& aggregate() { return agg_; }
private:
    SceneDescription sd_;      // owns every table the C structs point to
    GpuAggregate agg_;
};

// ---------------------------------------------------------------------------------------------- the main.rs scene
// src/main.rs:43-234 verbatim: 3 spheres + floor quad, uniform sky + blackbody sun, SimplePathIntegrator's inputs.
inline SceneDescription main_rs_scene(const Spectra& sp, int w = 400, int h = 400) {
    SceneDescription sd;
    const float pos[3] = {3.0f, 4.0f, 1.5f}, look[3] = {0.5f, 0.5f, 0.0f}, up[3] = {0.0f, 0.0f, 1.0f};
    const Transform camera_from_world = Transform::look_at(pos, look, up);
    sd.film = make_film(sp, w, h, 1.5f, 1.5f, 0.5f);
    set_perspective_camera(sd, camera_from_world.inverse(), 45.0f, w, h, 0.0f, 10e30f);
    const Transform& rfw = sd.render_from_world;
    // lights (main.rs:89-112)
    Light sky; sky.type = PM_LIGHT_UNIFORM_INFINITE; sky.spectrum = sp.rgb_illuminant_table(0.4f, 0.45f, 0.5f);
    sky.scale = 1.0f / sp.photometric(sp.d65);
    sd.lights.push_back(sky);
    float wv[3] = {-30.0f - 0.0f, 40.0f - 0.0f, 100.0f - 1.0f};
    {   const float len = std::sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]); for (float& c : wv) c /= len; }
    float v1[3], v2[3];      // Vec3f::coordinate_system (core/vec.rs:738-757)
    if (std::fabs(wv[0]) > std::fabs(wv[1])) {
        const float l = std::sqrt(wv[0] * wv[0] + wv[2] * wv[2]); v1[0] = -wv[2] / l; v1[1] = 0.0f; v1[2] = wv[0] / l;
    } else {
        const float l = std::sqrt(wv[1] * wv[1] + wv[2] * wv[2]); v1[0] = 0.0f; v1[1] = wv[2] / l; v1[2] = -wv[1] / l;
    }
    v2[0] = (float)((double)wv[1] * v1[2] - (double)wv[2] * v1[1]);
    v2[1] = (float)((double)wv[2] * v1[0] - (double)wv[0] * v1[2]);
    v2[2] = (float)((double)wv[0] * v1[1] - (double)wv[1] * v1[0]);
    const std::array<float, 16> sun_m = {v1[0], v2[0], wv[0], 0, v1[1], v2[1], wv[1], 0, v1[2], v2[2], wv[2], 0, 0, 0, 0, 1};
    Light sun; sun.type = PM_LIGHT_DIRECTIONAL; sun.spectrum = sp.blackbody_table(3000.0f);
    sun.scale = 1.5f / sp.photometric(sun.spectrum);
    sun.render_from_light = rfw * Transform::from_matrix(sun_m);
    sd.lights.push_back(sun);
    // materials
    sd.materials = {dielectric(1.5f), diffuse(sp.rgb_albedo(0.8f, 0.8f, 0.8f)), diffuse(sp.rgb_albedo(0.8f, 0.45f, 0.15f)),
                    diffuse(sp.rgb_albedo(0.8f, 0.25f, 0.3f))};
    auto sphere = [&](float radius, float x, float y, float z) {
        Sphere s; s.radius = radius; s.z_min = -radius; s.z_max = radius; s.phi_max = 360.0f;
        s.render_from_object = rfw * Transform::translate(x, y, z);
        return s;
    };
    BilinearPatchMesh floor;
    const Transform fx = rfw * Transform::translate(0.0f, 0.0f, -1.0f);
    const float corners[4][3] = {{-20, -20, 0}, {20, -20, 0}, {-20, 20, 0}, {20, 20, 0}};
    for (const auto& c : corners) { const auto p = fx.point(c[0], c[1], c[2]); floor.positions.insert(floor.positions.end(), p.begin(), p.end()); }
    floor.indices = {0, 1, 2, 3};
    sd.prims = {sphere(1.0f, 0, 0, 0), floor, sphere(1.5f, -3.5f, -2.0f, 0.5f), sphere(0.75f, -0.3f, -3.0f, -0.25f)};
    sd.prim_materials = {0, 1, 2, 3};     // main.rs:217-219 order: sphere, floor, sphere2, sphere3
    return sd;
}

}  // namespace pbrt_mi355
