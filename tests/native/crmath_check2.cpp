// Host check of pm_crmath.hpp's sincos / exp / log against glibc's f64 functions rounded to f32.
//   crmath_check2 sincos            every phi = 2 pi u, u = j 2^-23 (the sampling routines' argument)
//   crmath_check2 exp  <stride>     every <stride>-th f32 in [-104, 89] (beyond: 0 / inf in f32)
//   crmath_check2 log  <stride>     every <stride>-th positive finite f32
// Prints: calls, mismatches (f32 results that differ), worst f64 relative error.  stride 1 = exhaustive (minutes, threaded).
#define PM_CR_FN static inline
#include "../../pbrt-rs_amd/csrc/pm_crmath.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static inline bool same(float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a != a && b != b); }
struct Acc { unsigned long calls = 0, mism = 0; double worst = 0.0; };

template <class F> static Acc run_bits(uint32_t lo, uint32_t hi, uint32_t stride, F f) {
    const int nt = 8;
    std::vector<Acc> acc(nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (uint64_t b = (uint64_t)lo + (uint64_t)t * stride; b <= hi; b += (uint64_t)nt * stride) {
                const uint32_t bb = (uint32_t)b;
                float x;
                std::memcpy(&x, &bb, 4);
                f(x, &acc[t]);
            }
        });
    for (auto& t : th) t.join();
    Acc r;
    for (const Acc& a : acc) { r.calls += a.calls; r.mism += a.mism; r.worst = std::fmax(r.worst, a.worst); }
    return r;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "sincos";
    const uint32_t stride = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1u;
    Acc r;
    if (!std::strcmp(what, "sincos")) {
        for (uint32_t j = 0; j < (1u << 23); ++j) {
            const float u = (float)j * 1.1920928955078125e-07f;
            const float phi = 2.0f * 3.14159265358979323846f * u;
            double s, c;
            pm_sincos_2pi(phi, &s, &c);
            const double rs = std::sin((double)phi), rc = std::cos((double)phi);
            if (!same((float)s, (float)rs)) ++r.mism;
            if (!same((float)c, (float)rc)) ++r.mism;
            if (rs != 0.0) r.worst = std::fmax(r.worst, std::fabs((s - rs) / rs));
            if (rc != 0.0) r.worst = std::fmax(r.worst, std::fabs((c - rc) / rc));
            ++r.calls;
        }
    } else if (!std::strcmp(what, "exp")) {
        auto f = [](float x, Acc* a) {
            const double m = pm_exp_f32arg(x), g = std::exp((double)x);
            if (!same((float)m, (float)g)) ++a->mism;
            if (g > 1e-300 && g < 1e300) a->worst = std::fmax(a->worst, std::fabs((m - g) / g));
            ++a->calls;
        };
        const float lo_neg = -0.0f, hi_neg = -104.0f, hi_pos = 89.0f;
        uint32_t a, b;
        std::memcpy(&a, &lo_neg, 4); std::memcpy(&b, &hi_neg, 4);
        r = run_bits(a, b, stride, f);                       // -0 .. -104 (bit patterns ascend with magnitude)
        std::memcpy(&b, &hi_pos, 4);
        const Acc p = run_bits(0u, b, stride, f);            // +0 .. 89
        r.calls += p.calls; r.mism += p.mism; r.worst = std::fmax(r.worst, p.worst);
        // out of range and special values
        const float sp[] = {-105.0f, -200.0f, -745.0f, -746.5f, -1e30f, -HUGE_VALF, 90.0f, 709.0f, 711.0f, 1e30f, HUGE_VALF, NAN};
        for (float x : sp) { if (!same((float)pm_exp_f32arg(x), (float)std::exp((double)x))) ++r.mism; ++r.calls; }
    } else {
        auto f = [](float x, Acc* a) {
            const double m = pm_log_f32arg(x), g = std::log((double)x);
            if (!same((float)m, (float)g)) ++a->mism;
            if (g != 0.0) a->worst = std::fmax(a->worst, std::fabs((m - g) / g));
            ++a->calls;
        };
        r = run_bits(1u, 0x7F7FFFFFu, stride, f);
        const float sp[] = {0.0f, -0.0f, -1.0f, HUGE_VALF, -HUGE_VALF, NAN};
        for (float x : sp) { if (!same((float)pm_log_f32arg(x), (float)std::log((double)x))) ++r.mism; ++r.calls; }
    }
    std::printf("%lu %lu %.3e\n", r.calls, r.mism, r.worst);
    return 0;
}
