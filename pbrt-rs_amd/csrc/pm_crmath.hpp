// Two f64 evaluations of raygen_kernel's wavelength sampling, written out instead of calling the device library.
//
// The reference computes `538 - 138.888889 * atanh(0.85691062 - 1.82750197 u)` and `0.0039398042 / cosh^2(0.0072 (lambda - 538))`
// in f32 with the platform's libm (sampling/spectrum.rs:1004-1018).  Here - as in the oracle's correctly-rounded mode - the
// value is the f64 function rounded to f32 (pm_shade.hpp).  The device library's f64 atanh / cosh are written for < 1 ulp over
// the whole f64 range and cost ~125 f64 operations per call: 1062 of raygen_kernel's 2300 vector instructions were f64 (at half
// the f32 issue rate), 673 of them additions of its double-double arithmetic.  The two forms below are exact enough for the
// rounding to f32 (error <= ~2 ulp of f64: the rounded f32 differs from the correctly rounded one only where the exact value
// lies within ~4e-16 (relative) of an f32 rounding boundary, i.e. for about 1 argument in 10^8) at ~35 and ~15 f64 operations.
// tests/test_crmath_cpu.py compiles this header for the host and compares every argument the sampler can produce (2^23 values
// of u) with glibc.
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
