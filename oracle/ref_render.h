// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header).
//
// CPU restatement of the reference's sampling / camera / film / integrator path:
//   src/util/rng.rs, src/sampling/sampler.rs (IndependentSampler),
//   src/core/routines.rs (fast_exp, erf_inv), src/math.rs (gaussian),
//   src/core/square_matrix.rs (4x4 inverse), src/sampling/spectrum.rs.
//
// Third-party arithmetic that is NOT under /root/reference (Cargo.lock pins;
// sources not on disk; restated from the published algorithms; the reference
// holds no test for them => "parity unpinned"):
//   rand_pcg 0.3.1  Pcg32 = Lcg64Xsh32 (PCG-XSH-RR 64/32, O'Neill 2014)
//   rand 0.8.5      Uniform<f32>::new_inclusive(0, 1-2^-23) -> (u32 >> 9) * 2^-23
//   murmur3 0.5.2   murmur3_x86_128 (Appleby's MurmurHash3_x86_128), low 64 bits
#pragma once
#include "ref_core.h"

namespace ora {

// ------------------------------------------------------------------ rng (src/util/rng.rs)
struct Pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t MULT = 6364136223846793005ULL;
    // rand_pcg Lcg64Xsh32::new(state, stream): increment = (stream << 1) | 1; state += inc; step()
    static Pcg32 create(uint64_t st, uint64_t stream) {
        Pcg32 r;
        r.inc = (stream << 1) | 1ULL;
        r.state = st + r.inc;
        r.state = r.state * MULT + r.inc;
        return r;
    }
    uint32_t next_u32() {
        uint64_t old = state;
        state = state * MULT + inc;
        uint32_t rot = (uint32_t)(old >> 59);
        uint32_t xsh = (uint32_t)(((old >> 18) ^ old) >> 27);
        return (xsh >> rot) | (xsh << ((32 - rot) & 31));
    }
    // rng.rs:19,34-36: Uniform::new_inclusive(0.0, 1-2^-23).sample(): scale == 1.0, low == 0.0
    float uniform_float() {
        uint32_t u = next_u32();
        float v12 = u2f((u >> 9) | 0x3F800000u);
        return (v12 - 1.0f) * 1.0f + 0.0f;
    }
};

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
// rng.rs:47-50 over bytes_of(Point2i{x,y}) = 8 little-endian bytes; u128 as u64 = h2<<32 | h1
static inline uint64_t hash_pixel(int32_t x, int32_t y, uint32_t seed) {
    const uint32_t c1 = 0x239b961bu, c2 = 0xab0e9789u, c3 = 0x38b34ae5u;
    uint32_t h1 = seed, h2 = seed, h3 = seed, h4 = seed;
    uint32_t k1 = (uint32_t)x, k2 = (uint32_t)y;   // tail bytes 0..3 -> k1, 4..7 -> k2
    k2 *= c2; k2 = rotl32(k2, 16); k2 *= c3; h2 ^= k2;
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
    const uint32_t len = 8;
    h1 ^= len; h2 ^= len; h3 ^= len; h4 ^= len;
    h1 += h2; h1 += h3; h1 += h4; h2 += h1; h3 += h1; h4 += h1;
    h1 = fmix32(h1); h2 = fmix32(h2); h3 = fmix32(h3); h4 = fmix32(h4);
    h1 += h2; h1 += h3; h1 += h4; h2 += h1; h3 += h1; h4 += h1;
    (void)h3; (void)h4;
    return ((uint64_t)h2 << 32) | (uint64_t)h1;
}
// sampler.rs:104-116
static inline Pcg32 start_pixel_sample(int32_t x, int32_t y, uint32_t seed, uint64_t sample_index, uint64_t dim) {
    uint64_t inc = sample_index * 65536ULL + dim;
    return Pcg32::create(hash_pixel(x, y, seed), inc);
}

// ------------------------------------------------------------------ scalar helpers
static inline int exponent_f32(float v) {   // float.rs:65-72
    return (int)((f2u(v) & 0x7F800000u) >> 23) - 127;
}
static inline float fast_exp(float x) {     // routines.rs:47-73
    float xp = x * 1.44269504088896340736f;
    float fxp = std::floor(xp);
    float f = xp - fxp;
    int i = (int)std::trunc(fxp);
    const float c[4] = {1.0f, 0.6955569f, 0.22617356f, 0.07814557f};
    float two_to_f = evaluate_polynomial(f, c, 4);
    int e = exponent_f32(two_to_f) + i;
    if (e < -126) return 0.0f;
    if (e > 127) return std::numeric_limits<float>::infinity();
    uint32_t bits = f2u(two_to_f);
    bits &= 0x807FFFFFu;
    bits |= (uint32_t)(e + 127) << 23;
    return u2f(bits);
}
static inline float gaussian(float x, float mean, float sd) {   // math.rs:10-13
    float d = x - mean;
    float ef = fast_exp(-(d * d) / (2.0f * sd * sd));
    return 1.0f / std::sqrt(2.0f * PI_F * sd * sd) * ef;
}
static inline float erf_inv(float a) {      // routines.rs:89-120
    float t = std::log(rmax(std::fmaf(a, -a, 1.0f), -std::numeric_limits<float>::max()));
    float p;
    if (std::fabs(t) > 6.125f) {
        p = 3.03697567e-10f;
        p = std::fmaf(p, t, 2.93243101e-8f);
        p = std::fmaf(p, t, 1.22150334e-6f);
        p = std::fmaf(p, t, 2.84108955e-5f);
        p = std::fmaf(p, t, 3.93552968e-4f);
        p = std::fmaf(p, t, 3.02698812e-3f);
        p = std::fmaf(p, t, 4.83185798e-3f);
        p = std::fmaf(p, t, -2.64646143e-1f);
        p = std::fmaf(p, t, 8.40016484e-1f);
    } else {
        p = 5.43877832e-9f;
        p = std::fmaf(p, t, 1.43286059e-7f);
        p = std::fmaf(p, t, 1.22775396e-6f);
        p = std::fmaf(p, t, 1.12962631e-7f);
        p = std::fmaf(p, t, -5.61531961e-5f);
        p = std::fmaf(p, t, -1.47697705e-4f);
        p = std::fmaf(p, t, 2.31468701e-3f);
        p = std::fmaf(p, t, 1.15392562e-2f);
        p = std::fmaf(p, t, -2.32015476e-1f);
        p = std::fmaf(p, t, 8.86226892e-1f);
    }
    return a * p;
}

// square_matrix.rs:196-262.  inner_product!(..).val is the right-nested sum of rounded products
// (math.rs:89-104: only .val is read, the compensation terms are discarded).
static inline float ip3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return a0 * b0 + (a1 * b1 + a2 * b2);
}
static inline bool m4_inverse(const M4& M, M4* out) {
    const auto& m = M.m;
    float s0 = difference_of_products(m[0][0], m[1][1], m[1][0], m[0][1]);
    float s1 = difference_of_products(m[0][0], m[1][2], m[1][0], m[0][2]);
    float s2 = difference_of_products(m[0][0], m[1][3], m[1][0], m[0][3]);
    float s3 = difference_of_products(m[0][1], m[1][2], m[1][1], m[0][2]);
    float s4 = difference_of_products(m[0][1], m[1][3], m[1][1], m[0][3]);
    float s5 = difference_of_products(m[0][2], m[1][3], m[1][2], m[0][3]);
    float c0 = difference_of_products(m[2][0], m[3][1], m[3][0], m[2][1]);
    float c1 = difference_of_products(m[2][0], m[3][2], m[3][0], m[2][2]);
    float c2 = difference_of_products(m[2][0], m[3][3], m[3][0], m[2][3]);
    float c3 = difference_of_products(m[2][1], m[3][2], m[3][1], m[2][2]);
    float c4 = difference_of_products(m[2][1], m[3][3], m[3][1], m[2][3]);
    float c5 = difference_of_products(m[2][2], m[3][3], m[3][2], m[2][3]);
    float det = s0 * c5 + (-s1 * c4 + (s2 * c3 + (s3 * c2 + (s5 * c0 + -s4 * c1))));
    if (det == 0.0f) return false;
    float s = 1.0f / det;
    float inv[4][4] = {
        {s * ip3(m[1][1], c5, m[1][3], c3, -m[1][2], c4), s * ip3(-m[0][1], c5, m[0][2], c4, -m[0][3], c3),
         s * ip3(m[3][1], s5, m[3][3], s3, -m[3][2], s4), s * ip3(-m[2][1], s5, m[2][2], s4, -m[2][3], s3)},
        {s * ip3(-m[1][0], c5, m[1][2], c2, -m[1][3], c1), s * ip3(m[0][0], c5, m[0][3], c1, -m[0][2], c2),
         s * ip3(-m[3][0], s5, m[3][2], s2, -m[3][3], s1), s * ip3(m[2][0], s5, m[2][3], s1, -m[2][2], s2)},
        {s * ip3(m[1][0], c4, m[1][3], c0, -m[1][1], c2), s * ip3(-m[0][0], c4, m[0][1], c2, -m[0][3], c0),
         s * ip3(m[3][0], s4, m[3][3], s0, -m[3][1], s2), s * ip3(-m[2][0], s4, m[2][1], s2, -m[2][3], s0)},
        {s * ip3(-m[1][0], c3, m[1][1], c1, -m[1][2], c0), s * ip3(m[0][0], c3, m[0][2], c0, -m[0][1], c1),
         s * ip3(-m[3][0], s3, m[3][1], s1, -m[3][2], s0), s * ip3(m[2][0], s3, m[2][2], s0, -m[2][1], s1)}};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out->m[i][j] = inv[i][j];
    return true;
}

// ------------------------------------------------------------------ wavelengths (spectrum.rs:73-91)
static inline float visible_wavelengths_pdf(float lambda) {
    if (lambda >= 360.0f && lambda <= 830.0f) {
        float ch = t_cosh(0.0072f * (lambda - 538.0f));
        return 0.003939804f / (ch * ch);
    }
    return 0.0f;
}
static inline float sample_visible_wavelengths(float u) {
    return 538.0f - 138.88889f * t_atanh(0.85691062f - 1.827502f * u);
}

}  // namespace ora
