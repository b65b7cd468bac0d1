// Host-side sanitizer run of the library (scripts/host_sanitize.sh builds the library's HOST code and this program with
// -fsanitize=address,undefined; the GPU code is not instrumented and no kernel is launched: every context here is a host-only one,
// device_id = -1).  Exercises what runs on the host: scene ingestion, the three host tree builders + the wide-node collapse and
// quantisation, the flat table, buffer export, argument validation incl. the ABI handshake, and the pmh_* constructors with the
// OpenEXR writer.  Exit code 0 and no sanitizer report = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "pbrt_mi355.hpp"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static int export_buffers(pm_ctx* c) {
    uint64_t need_n = 0, need_p = 0;
    if (pm_get_node_buffer(c, nullptr, 0, &need_n) != PM_OK || pm_get_prim_buffer(c, nullptr, 0, &need_p) != PM_OK) return 1;
    std::vector<unsigned char> nb(need_n), pb(need_p);
    if (pm_get_node_buffer(c, nb.data(), nb.size(), &need_n) != PM_OK) return 1;
    if (pm_get_prim_buffer(c, pb.data(), pb.size(), &need_p) != PM_OK) return 1;
    // a buffer that is one byte short must be refused, not overrun
    if (need_n > 1) { std::vector<unsigned char> tight(need_n - 1); if (pm_get_node_buffer(c, tight.data(), tight.size(), &need_n) == PM_OK) return 1; }
    return 0;
}

int main(int argc, char** argv) {
    namespace pm = pbrt_mi355;
    const char* tables = argc > 1 ? argv[1] : "pbrt-rs_amd/data/cie_tables.bin";
    CHECK(pm_abi_version() == PM_ABI_VERSION);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    // ---- triangle soup through every host split method and several leaf sizes
    const uint32_t n_tri = 20000;
    std::vector<float> pos; std::vector<uint32_t> idx;
    for (uint32_t t = 0; t < n_tri; ++t) {
        const float cx = U(rng), cy = U(rng), cz = U(rng);
        for (int k = 0; k < 3; ++k) { pos.push_back(cx + 0.02f * U(rng)); pos.push_back(cy + 0.02f * U(rng)); pos.push_back(cz + 0.02f * U(rng)); idx.push_back(3 * t + k); }
    }
    const int splits[] = {PM_SPLIT_SAH, PM_SPLIT_MIDDLE, PM_SPLIT_EQUAL_COUNTS, PM_SPLIT_HLBVH};
    for (int split : splits)
        for (int leaf : {1, 4, 255}) {
            pm_ctx* c = nullptr;
            CHECK(pm_create(-1, &c) == PM_OK);
            uint32_t first = 0;
            CHECK(pm_add_triangle_mesh(c, pos.data(), (uint32_t)(pos.size() / 3), idx.data(), n_tri, &first) == PM_OK);
            PmAccelInfo info{};
            CHECK(pm_build_accel(c, split, leaf, &info, (uint32_t)sizeof(info)) == PM_OK);
            CHECK(info.n_nodes > 0 && info.flat_prims == 0);
            CHECK(export_buffers(c) == 0);
            // a caller compiled against another revision of the header
            CHECK(pm_build_accel(c, split, leaf, &info, (uint32_t)sizeof(info) - 4) == PM_ERR_INVALID);
            // host-only contexts refuse to launch
            PmRay ray{}; PmHit hit{};
            CHECK(pm_trace_closest(c, &ray, 1, &hit) != PM_OK);
            pm_destroy(c);
        }
    // ---- the main.rs scene (spheres + a patch mesh) and its renderer state, all on the host
    {
        pm::Spectra spectra(tables);
        pm::SceneDescription sd = pm::main_rs_scene(spectra, 64, 48);
        pm::GpuAggregate agg(sd.prims, 4, PM_SPLIT_SAH, -1);
        CHECK(agg.info.n_prims > 0);
        CHECK(export_buffers(agg.ctx()) == 0);
        // geometry edits: clear and rebuild with a patch mesh that carries normals and uv
        CHECK(pm_clear_geometry(agg.ctx()) == PM_OK);
        const float q[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0.2f};
        const float nrm[12] = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
        const float uv[8] = {0, 0, 1, 0, 0, 1, 1, 1};
        const uint32_t i4[4] = {0, 1, 2, 3};
        uint32_t first = 0;
        CHECK(pm_add_patch_mesh_uv(agg.ctx(), q, 4, i4, 1, nrm, uv, 0, 0, &first) == PM_OK);
        const uint32_t bad4[4] = {0, 1, 2, 9};       // index out of range: refused, nothing read past the array
        CHECK(pm_add_patch_mesh(agg.ctx(), q, 4, bad4, 1, nullptr, 0, 0, &first) != PM_OK);
        PmAccelInfo info{};
        CHECK(pm_build_accel(agg.ctx(), PM_SPLIT_SAH, 4, &info, (uint32_t)sizeof(info)) == PM_OK);
        CHECK(info.flat_prims == 1);
        CHECK(export_buffers(agg.ctx()) == 0);
        // renderer state setters validate on the host
        std::vector<PmMaterial> mats(1);
        std::memset(mats.data(), 0, sizeof(PmMaterial));
        mats[0].type = 77;
        CHECK(pm_set_materials(agg.ctx(), mats.data(), 1, (uint32_t)sizeof(PmMaterial)) != PM_OK);
        CHECK(pm_set_materials(agg.ctx(), mats.data(), 1, 8u) == PM_ERR_INVALID);
        PmRenderParams rp{};
        CHECK(pm_render(agg.ctx(), &rp, (uint32_t)sizeof(rp)) != PM_OK);      // host-only context
    }
    // ---- host constructors + the OpenEXR writer
    {
        std::vector<float> img(3 * 17 * 9);
        for (size_t i = 0; i < img.size(); ++i) img[i] = 0.001f * (float)i;
        CHECK(pmh_write_exr("/tmp/pm_host_sanitize.exr", 17, 9, img.data()) == 0);
        std::vector<float> vals(48 * 48);
        int nx = 0, ny = 0;
        CHECK(pmh_gaussian_filter_table(1.5f, 1.5f, 0.5f, vals.data(), (int)vals.size(), &nx, &ny) == 0 && nx == 48 && ny == 48);
        CHECK(pmh_gaussian_filter_table(1.5f, 1.5f, 0.5f, vals.data(), 100, &nx, &ny) != 0);      // capacity too small: refused
        float inv[16], m[16] = {2, 0, 0, 1, 0, 3, 0, 2, 0, 0, 4, 3, 0, 0, 0, 1};
        CHECK(pmh_mat4_inverse(m, inv) == 0);
        float sing[16] = {0};
        CHECK(pmh_mat4_inverse(sing, inv) != 0);
    }
    std::printf("host sanitize check: ok\n");
    return 0;
}
