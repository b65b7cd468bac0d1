// Host check of pbrt-rs_amd/csrc/pm_crmath.hpp: every argument the wavelength sampler can produce (u = j * 2^-23, j < 2^23,
// IndependentSampler's floats) against glibc's f64 functions rounded to f32.  Prints: calls, atanh mismatches, cosh mismatches,
// worst f64 relative errors.
#define PM_CR_FN static inline
#include "../../pbrt-rs_amd/csrc/pm_crmath.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>

int main() {
    unsigned long mism_a = 0, mism_c = 0, calls = 0;
    double worst_a = 0.0, worst_c = 0.0;
    for (uint32_t j = 0; j < (1u << 23); ++j) {
        const float u = (float)j * 1.1920928955078125e-07f;
        const float x = 0.85691062f - 1.827502f * u;
        const double ra = atanh((double)x), ma = pm_atanh_f32arg(x);
        const float fa = (float)ra, fm = (float)ma;
        if (std::memcmp(&fa, &fm, 4) != 0) ++mism_a;
        if (ra != 0.0) worst_a = std::fmax(worst_a, std::fabs((ma - ra) / ra));
        const float lambda = 538.0f - 138.88889f * fa;
        if (lambda >= 360.0f && lambda <= 830.0f) {
            const float y = 0.0072f * (lambda - 538.0f);
            const double rc = cosh((double)y), mc = pm_cosh_small((double)y);
            const float fc = (float)rc, fmc = (float)mc;
            if (std::memcmp(&fc, &fmc, 4) != 0) ++mism_c;
            worst_c = std::fmax(worst_c, std::fabs((mc - rc) / rc));
        }
        ++calls;
    }
    std::printf("%lu %lu %lu %.3e %.3e\n", calls, mism_a, mism_c, worst_a, worst_c);
    return 0;
}
