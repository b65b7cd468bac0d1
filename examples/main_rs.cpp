// The reference binary's hardcoded scene (src/main.rs:43-234) rendered through the C++ host mirror of the C ABI:
//   main_rs <width> <height> <spp> <out.exr> [film.f64] [tables.bin]
// Writes the image as OpenEXR (Image::write_exr) and, optionally, the raw f64 film (rgb_sum[3], weight_sum per pixel).
#include <chrono>
#include <string>
#include <cstdio>
#include <cstdlib>

#include "pbrt_mi355.hpp"

int main(int argc, char** argv) {
    namespace pm = pbrt_mi355;
    const int w = argc > 1 ? std::atoi(argv[1]) : 400, h = argc > 2 ? std::atoi(argv[2]) : 400;
    const int spp = argc > 3 ? std::atoi(argv[3]) : 256;
    const char* out = argc > 4 ? argv[4] : "main_rs.exr";
    const char* film_path = argc > 5 ? argv[5] : nullptr;
    const char* tables = argc > 6 ? argv[6] : "pbrt-rs_amd/data/cie_tables.bin";
    try {
        pm::Spectra spectra(tables);
        pm::SceneDescription scene = pm::main_rs_scene(spectra, w, h);
        // main.rs:221-233: SimplePathIntegrator::new(5, true, true, camera, sampler, bvh, lights).render()
        pm::WavefrontIntegrator integrator(5, true, true, scene, spp);
        const auto t0 = std::chrono::steady_clock::now();
        // argv[7] = "waves": the reference's wave schedule (integrators/base.rs:85-127) with a progress line per wave; same film
        const bool waves = argc > 7 && std::string(argv[7]) == "waves";
        const PmRenderStats st = waves ? integrator.render_waves([&](int w0, int w1) {
                                             std::fprintf(stderr, "wave [%d, %d) of %d spp done\n", w0, w1, spp);
                                         })
                                       : integrator.render();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        integrator.write_exr(out);
        if (film_path) {
            const auto film = integrator.film();
            std::FILE* f = std::fopen(film_path, "wb");
            if (!f || std::fwrite(film.data(), 8, film.size(), f) != film.size()) { std::fprintf(stderr, "cannot write %s\n", film_path); return 2; }
            std::fclose(f);
        }
        std::printf("rendered %dx%d @ %d spp: %llu samples, %llu closest + %llu shadow rays, %.1f ms device, %.3f s wall -> %s\n", w, h, spp,
                    (unsigned long long)st.samples, (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays, st.kernel_ms, dt, out);
    } catch (const pm::Error& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
