// The f64 transcendentals of the path, written out for the argument ranges the path uses instead of calling the device library.
//
// The reference computes `538 - 138.888889 * atanh(0.85691062 - 1.82750197 u)` and `0.0039398042 / cosh^2(0.0072 (lambda - 538))`
// in f32 with the platform's libm (sampling/spectrum.rs:1004-1018), azimuths as sin / cos of 2 pi u (sampling/routines.rs),
// transmittance and free-flight distances with exp / log (media/medium.rs).  Here - as in the oracle's correctly-rounded mode -
// the value is the f64 function rounded to f32 (pm_shade.hpp).  The device library's f64 routines are written for < 1 ulp over
// the whole f64 range and cost ~125 f64 operations per call: 1062 of raygen_kernel's 2300 vector instructions were f64 (at half
// the f32 issue rate), 673 of them additions of its double-double arithmetic.  The forms below are exact enough for the
// rounding to f32 (error <= 1.4 ulp of f64: the rounded f32 could differ from the correctly rounded one only where the exact
// value lies within ~3e-16 (relative) of an f32 rounding boundary) at 15 to 40 f64 operations.  tests/test_crmath_cpu.py
// compiles this header for the host and compares with glibc: every argument the samplers can produce (2^23 values of u) for
// atanh, cosh and sincos, and - tests/native/crmath_check2.cpp, run in full once - every f32 of [-104, 89] for exp and every
// positive f32 for log.  No f32 result differs.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef PM_CR_FN
#define PM_CR_FN __device__ __forceinline__
#endif

// atanh(x) = log((1 + x) / (1 - x)) / 2 for |x| < 1 (f32 argument).  With a = 1 + x, b = 1 - x (exact in f64) and k ~ log2(a / b):
//   (a / b) / 2^k = m in about [1/sqrt 2, sqrt 2],  log m = 2 atanh(s),  s = (m - 1) / (m + 1) = (a - 2^k b) / (a + 2^k b)
// Numerator and denominator are exact (operands of <= 25 bits, exponents within 2^7), so s carries ONE rounding (the division),
// and |s| <= 0.1716: 2 atanh(s) = 2 s + s R(s^2) with the degree-7 minimax polynomial R of the well-known fdlibm e_log.c
// (Lg1 .. Lg7, error < 2^-58.45).  For small |x|: k = 0, s = x exactly, and the result keeps full relative accuracy.
PM_CR_FN double pm_atanh_f32arg(float x) {
    const double a = 1.0 + (double)x, b = 1.0 - (double)x;
    // k from the bits of the f32 quotient (integer operations only: host and device pick the same k)
    const float qf = (1.0f + x) / (1.0f - x);
    uint32_t qb;
    __builtin_memcpy(&qb, &qf, 4);
    const int k = (int)((qb >> 23) & 0xFFu) - 127 + ((qb & 0x7FFFFFu) > 0x3504F3u ? 1 : 0);
    const double b2 = ldexp(b, k);
    const double s = (a - b2) / (a + b2);
    const double z = s * s;
    double r = 1.479819860511658591e-01;                                 // Lg7
    r = fma(r, z, 1.531383769920937332e-01);                            // Lg6
    r = fma(r, z, 1.818357216161805012e-01);                            // Lg5
    r = fma(r, z, 2.222219843214978396e-01);                            // Lg4
    r = fma(r, z, 2.857142874366239149e-01);                            // Lg3
    r = fma(r, z, 3.999999999940941908e-01);                            // Lg2
    r = fma(r, z, 6.666666666666735130e-01);                            // Lg1
    r *= z;
    // (k ln2 + 2 s + s R) / 2, ln2 split as in e_log.c so that k * ln2_hi is exact
    const double dk = (double)k;
    const double lo = fma(0.5 * s, r, dk * (0.5 * 1.90821492927058770002e-10));
    return fma(dk, 0.5 * 6.93147180369123816490e-01, s + lo);
}

// cosh(y) for |y| <= 2.2 (the sampler's 0.0072 (lambda - 538) lies in [-1.29, 2.11]): the even Taylor series to y^26 / 26!
// (the first neglected term is 2.2^28 / 28! = 1.3e-20).  All terms are positive, so Horner's scheme loses nothing to
// cancellation: ~14 roundings, each scaled down by the terms behind it.
PM_CR_FN double pm_cosh_small(double y) {
    const double z = y * y;
    double c = 2.4795962632247976e-27;               // 1/26!
    c = fma(c, z, 1.6117375710961184e-24);           // 1/24!
    c = fma(c, z, 8.8967913924505741e-22);           // 1/22!
    c = fma(c, z, 4.1103176233121648e-19);           // 1/20!
    c = fma(c, z, 1.5619206968586225e-16);           // 1/18!
    c = fma(c, z, 4.7794773323873853e-14);           // 1/16!
    c = fma(c, z, 1.1470745597729725e-11);           // 1/14!
    c = fma(c, z, 2.0876756987868100e-09);           // 1/12!
    c = fma(c, z, 2.7557319223985888e-07);           // 1/10!
    c = fma(c, z, 2.4801587301587302e-05);           // 1/8!
    c = fma(c, z, 1.3888888888888889e-03);           // 1/6!
    c = fma(c, z, 4.1666666666666664e-02);           // 1/4!
    c = fma(c, z, 0.5);
    return fma(c, z, 1.0);
}

// sin and cos of phi = 2 pi u, u in [0, 1) - the only argument the sampling routines pass (uniform sphere / hemisphere, polar
// disk, Henyey-Greenstein: phi in [0, 6.2831855]).  Quadrant j = round(phi / (pi/2)) in 0 .. 4, t = phi - j pi/2 with pi/2 in two
// parts (the first has 33 bits, so j * pio2_hi and the first difference are exact; fdlibm's split), then the Taylor series of
// pm_shade.hpp cr_sincos_quarter (|t| <= pi/4: the first neglected terms are 8e-20 and 2e-18) and the quadrant's rotation.
PM_CR_FN void pm_sincos_2pi(float phi, double* sn, double* cs) {
    const int j = (int)rintf(phi * 0.636619772367581343f);
    const double dj = (double)j;
    const double t = fma(-dj, 6.07710050650619224932e-11, fma(-dj, 1.57079632673412561417e+00, (double)phi));
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
    const bool swap = (j & 1) != 0;
    const double a = swap ? c : s, b = swap ? s : c;             // j odd: sin = +-cos t, cos = -+sin t
    *sn = (j & 2) ? -a : a;
    *cs = ((j + 1) & 2) ? -b : b;
}

// exp(x), f32 argument (the medium's transmittance exp(-sigma d); the blackbody's exponent): k = round(x / ln 2),
// r = x - k ln2 in two parts (k * ln2_hi exact), |r| <= 0.347, Taylor series to r^13 / 13! (next term 4e-18), scaled by 2^k.
// Arguments beyond the f64 range give 0 / inf as the library does; NaN stays NaN.
PM_CR_FN double pm_exp_f32arg(float x) {
    if (!(x > -746.0f)) return x != x ? (double)x : 0.0;
    if (x > 710.0f) return HUGE_VAL;
    const double dx = (double)x;
    const double dk = rint(dx * 1.44269504088896338700e+00);
    const double r = fma(-dk, 1.90821492927058770002e-10, fma(-dk, 6.93147180369123816490e-01, dx));
    double p = 1.6059043836821613e-10;                           // 1/13!
    p = fma(p, r, 2.0876756987868100e-09);                       // 1/12!
    p = fma(p, r, 2.5052108385441720e-08);                       // 1/11!
    p = fma(p, r, 2.7557319223985888e-07);                       // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);                       // 1/9!
    p = fma(p, r, 2.4801587301587302e-05);                       // 1/8!
    p = fma(p, r, 1.9841269841269841e-04);                       // 1/7!
    p = fma(p, r, 1.3888888888888889e-03);                       // 1/6!
    p = fma(p, r, 8.3333333333333332e-03);                       // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);                       // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);                       // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p * r, r, r);                                        // r + r^2 (1/2 + ...)
    return ldexp(1.0 + p, (int)dk);
}

// log(x), f32 argument > 0 (the medium's free-flight distance -log(1 - u)): x = 2^k m, m in [1/sqrt 2, sqrt 2) from the bits,
// s = (m - 1) / (m + 1) with an exact numerator and denominator, log m = 2 s + s R(s^2) as in pm_atanh_f32arg.
// 0 -> -inf, negative -> NaN, inf / NaN -> themselves, as the library does.
PM_CR_FN double pm_log_f32arg(float x) {
    if (!(x > 0.0f) || !(x < HUGE_VALF)) return x == 0.0f ? -HUGE_VAL : (x < 0.0f ? (double)NAN : (double)x);
    double m = (double)x;                                        // (f32 denormals become normal f64 numbers)
    uint64_t mb;
    __builtin_memcpy(&mb, &m, 8);
    int k = (int)((mb >> 52) & 0x7FFu) - 1023;
    mb = (mb & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;   // m in [1, 2)
    __builtin_memcpy(&m, &mb, 8);
    if (m > 1.4142135623730951) { m *= 0.5; ++k; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double r = 1.479819860511658591e-01;                         // Lg7
    r = fma(r, z, 1.531383769920937332e-01);                     // Lg6
    r = fma(r, z, 1.818357216161805012e-01);                     // Lg5
    r = fma(r, z, 2.222219843214978396e-01);                     // Lg4
    r = fma(r, z, 2.857142874366239149e-01);                     // Lg3
    r = fma(r, z, 3.999999999940941908e-01);                     // Lg2
    r = fma(r, z, 6.666666666666735130e-01);                     // Lg1
    r *= z;
    const double dk = (double)k;
    const double lo = fma(s, r, dk * 1.90821492927058770002e-10);
    return fma(dk, 6.93147180369123816490e-01, 2.0 * s + lo);
}
