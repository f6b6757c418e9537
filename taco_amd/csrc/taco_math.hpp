// taco_math.hpp -- device-side scalar math for the fused step kernel (gfx950).
//
// Everything here is built from IEEE-754 fp32 +,-,*,fma, correctly rounded / and sqrt, and integer ops only, so a
// result depends on nothing but the inputs (no ocml transcendental, no fast-math approximation instruction).  That is
// what lets the kernel's trajectories, rewards and done flags be compared BIT-FOR-BIT with an independent CPU
// implementation of the same published algorithms (Cephes-style single-precision kernels, Philox4x32-10).
// The translation unit is compiled with -ffp-contract=off: an fma happens only where this file writes fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace taco {

#define TD __device__ __forceinline__
// LDS pointers that stay LDS pointers when passed around (through a generic pointer the accesses become flat loads)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kQuarterPi = 0.78539816339744830962f;

TD float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TD float absf(float x) { return __builtin_fabsf(x); }
TD uint32_t bits(float f) { return __builtin_bit_cast(uint32_t, f); }
TD float from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
TD float with_sign_of(float mag, float sgn) { return from_bits((bits(mag) & 0x7fffffffu) | (bits(sgn) & 0x80000000u)); }
TD float nanf32() { return from_bits(0x7fc00000u); }

// torch.clamp semantics: min(max(x, lo), hi); a NaN x stays NaN
TD float clampf(float x, float lo, float hi) {
    float t = (x < lo) ? lo : x;
    return (t > hi) ? hi : t;
}

// the same for CONSTANT bounds lo < hi: one v_med3_f32 plus the NaN pass-through (v_med3 alone returns lo for a NaN x)
TD float clamp_const(float x, float lo, float hi) {
    const float r = __builtin_amdgcn_fmed3f(x, lo, hi);
    return (x != x) ? x : r;
}

// FIN = the caller has established (wave-uniformly) that x cannot be a NaN: the pass-through select is dead
template <bool FIN> TD float clamp_const_t(float x, float lo, float hi) {
    if constexpr (FIN) return __builtin_amdgcn_fmed3f(x, lo, hi);
    else return clamp_const(x, lo, hi);
}

// ---- x / c for a divisor known in advance, without a division: q = x * RN(1/c) plus one fma correction.  Bit-identical to
// the IEEE quotient for every divisor it is used with (0.001, 0.75, 3, 3.3, 6, 100, 1000, 4500, 9000, pi: checked
// exhaustively over all signed mantissas) for |x| up to ~1e34; 3 instructions instead of the ~10 of v_div_scale/.../v_div_fixup.
TD float div_const(float x, float c, float rc) {
    float q = x * rc;
    float r = fma(-q, c, x);
    return fma(r, rc, q);
}
#define TACO_DIVC(x, c) ::taco::div_const((x), (c), 1.0f / (c))

// ---- sin / cos: k = rint(x * 2/pi); 3-term Cody-Waite reduction; minimax kernels on |r| <= pi/4
TD float sin_poly(float r) {
    float z = r * r;
    float p = fma(fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    return fma(p * z, r, r);
}
TD float cos_poly(float r) {
    float z = r * r;
    float p = fma(fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    return fma(p * z, z, fma(-0.5f, z, 1.0f));
}
TD void sincos(float x, float &s, float &c) {
    if (!(absf(x) < 1048576.0f)) { s = c = nanf32(); return; }
    float k = __builtin_rintf(x * 0.63661977236758134308f);
    float r = fma(-k, 1.5703125f, x);
    r = fma(-k, 4.837512969970703125e-4f, r);
    r = fma(-k, 7.54978995489188216e-8f, r);
    float sr = sin_poly(r), cr = cos_poly(r);
    int n = (int)k & 3;
    float ss = (n & 1) ? cr : sr;
    float cc = (n & 1) ? sr : cr;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

// ---- atan2: octant reduction with one division, odd minimax polynomial on |t| <= tan(pi/8)
TD float atan2(float y, float x) {
    if (x != x || y != y) return nanf32();
    float ax = absf(x), ay = absf(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float r;
    if (mx == 0.0f) {
        r = 0.0f;
    } else {
        bool big = mn > 0.41421356237309503f * mx;
        float num = big ? mn - mx : mn;
        float den = big ? mn + mx : mx;
        float t = num / den;
        float z = t * t;
        float p = fma(fma(fma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
        r = fma(p * z, t, t);
        if (big) r = kQuarterPi + r;
    }
    if (ay > ax) r = kHalfPi - r;
    if (bits(x) & 0x80000000u) r = kPi - r;
    return with_sign_of(r, y);
}

// ---- asin on [-1, 1]
TD float asin(float x) {
    float a = absf(x);
    if (!(a <= 1.0f)) return nanf32();
    bool big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    float s = big ? __builtin_sqrtf(z) : a;
    float p = fma(fma(fma(fma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    float r = fma(s * z, p, s);
    if (big) r = kHalfPi - (r + r);
    return with_sign_of(r, x);
}

// ---- natural log of a normal positive finite float (Box-Muller only)
TD float log(float x) {
    uint32_t u = bits(x);
    int e = (int)((u >> 23) & 0xffu) - 126;
    float m = from_bits((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fma(p, m, -1.1514610310e-1f);
    p = fma(p, m, 1.1676998740e-1f);
    p = fma(p, m, -1.2420140846e-1f);
    p = fma(p, m, 1.4249322787e-1f);
    p = fma(p, m, -1.6668057665e-1f);
    p = fma(p, m, 2.0000714765e-1f);
    p = fma(p, m, -2.4999993993e-1f);
    p = fma(p, m, 3.3333331174e-1f);
    float fe = (float)e;
    float y = (m * z) * p;
    y = fma(-2.12194440e-4f, fe, y);
    y = fma(-0.5f, z, y);
    return fma(0.693359375f, fe, m + y);
}

// ---- exp / tanh / sigmoid for the policy forward (Cephes single-precision kernels, explicit fma, one division in sigmoid/tanh)
TD float expf_own(float x) {
    // the argument is clamped to the range in which p * 2^k neither overflows nor goes subnormal: exp(x >= 88.72) = 3.4e38,
    // exp(x <= -87.34) = 1.2e-38 (sigmoid / tanh saturate long before); straight-line code, a NaN comes back as NaN.
    // p * 2^k is one v_ldexp_f32 (= C's ldexpf: exact here, the clamp keeps the result normal).
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -87.33654475055310898657f), 88.72283905206835f);
    const float k = __builtin_rintf(xc * 1.44269504088896341f);
    float r = fma(-k, 0.693359375f, xc);
    r = fma(-k, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fma(p, r, 1.3981999507e-3f);
    p = fma(p, r, 8.3334519073e-3f);
    p = fma(p, r, 4.1665795894e-2f);
    p = fma(p, r, 1.6666665459e-1f);
    p = fma(p, r, 5.0000001201e-1f);
    p = fma(p, z, r) + 1.0f;
    const float e = __builtin_amdgcn_ldexpf(p, (int)k);
    return (x != x) ? x : e;
}
TD float sigmoid_own(float x) { return 1.0f / (1.0f + expf_own(-x)); }
// both forms are evaluated and one is selected (no divergent control flow in the LSTM epilogue)
TD float tanh_own(float x) {
    const float z = absf(x);
    const float zc = __builtin_fminf(z, 44.0f);
    const float big = with_sign_of(1.0f - 2.0f / (expf_own(zc + zc) + 1.0f), x);
    const float z2 = x * x;
    float p = -5.70498872745e-3f;
    p = fma(p, z2, 2.06390887954e-2f);
    p = fma(p, z2, -5.37397155531e-2f);
    p = fma(p, z2, 1.33314422036e-1f);
    p = fma(p, z2, -3.33332819422e-1f);
    const float small = fma(p * z2, x, x);
    const float t = (z >= 0.625f) ? big : small;
    return (x != x) ? x : t;
}

// One LSTM cell update (torch.nn.LSTM gate order i f g o) with the activations FUSED so that three divisions serve five activations:
//   sigmoid(a) = 1 / (1 + e^-a),  tanh(x) = sign(x) (1 - t) / (1 + t) with t = e^(-2|x|)   (absolute error <= 1e-7 everywhere)
//   i * g = sign(ag) (1 - tg) / ((1 + ei)(1 + tg)),   f * c = c / (1 + ef),   h = o * tanh(c') = sign(c') (1 - tc) / ((1 + eo)(1 + tc))
TD void lstm_cell(float ai, float af, float ag, float ao, float c_old, float &c_new, float &h) {
    const float ei = expf_own(-ai), ef = expf_own(-af), eo = expf_own(-ao);
    const float zg = absf(ag);
    const float tg = expf_own(-(zg + zg));
    const float ig = with_sign_of(1.0f - tg, ag) / ((1.0f + ei) * (1.0f + tg));
    const float c = c_old / (1.0f + ef) + ig;
    const float zc = absf(c);
    const float tc = expf_own(-(zc + zc));
    c_new = c;
    h = with_sign_of(1.0f - tc, c) / ((1.0f + eo) * (1.0f + tc));
}

// The same cell on the hardware's transcendental pipe (v_exp_f32 = 2^x and v_rcp_f32, 1 ulp each) for the batched critic kernels, whose
// SIMD time is MFMA time + VALU time with no slack (profiles/r03_q_rollout_pmc.json): 8 quarter-rate + ~19 full-rate instructions per
// cell instead of 142.  Same three-division form; |error| <= 3e-7 on c' and h against lstm_cell (tests/test_policy_gpu.py), so the
// values stay inside the 1e-5 bar to the reference's vectors but are no longer bit-identical to the CPU restatement of the policy.
// e^-a for a < -88 is +inf: 1/(1+inf) = 0 and the numerators are finite, so saturated gates give exact zeros, never NaN.
TD void lstm_cell_fast(float ai, float af, float ag, float ao, float c_old, float &c_new, float &h) {
    constexpr float L2E = 1.44269504088896341f;
    const float ei = __builtin_amdgcn_exp2f(-L2E * ai), ef = __builtin_amdgcn_exp2f(-L2E * af), eo = __builtin_amdgcn_exp2f(-L2E * ao);
    const float tg = __builtin_amdgcn_exp2f((-2.0f * L2E) * absf(ag));
    const float ig = with_sign_of(1.0f - tg, ag) * __builtin_amdgcn_rcpf((1.0f + ei) * (1.0f + tg));
    const float c = fma(c_old, __builtin_amdgcn_rcpf(1.0f + ef), ig);
    const float tc = __builtin_amdgcn_exp2f((-2.0f * L2E) * absf(c));
    c_new = c;
    h = with_sign_of(1.0f - tc, c) * __builtin_amdgcn_rcpf((1.0f + eo) * (1.0f + tc));
}
template <bool EXACT>
TD void lstm_cell_batched(float ai, float af, float ag, float ao, float c_old, float &c_new, float &h) {
    if constexpr (EXACT) lstm_cell(ai, af, ag, ao, c_old, c_new, h);
    else lstm_cell_fast(ai, af, ag, ao, c_old, c_new, h);
}

// ---- Philox4x32-10 (Salmon et al., SC'11).  counter = (global env id, step index, stream, block), key = seed
struct U4 { uint32_t x, y, z, w; };
TD U4 philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // ONE 64-bit product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the same bits, half the
        // quarter-rate multiplies -- a block 134 -> 104 ns per wavefront at 4 wavefronts per SIMD (tools/ubench/intmul_cndmask.hip,
        // profiles/r05_b_ubench_intmul_cndmask.txt)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        c0 = h1 ^ c1 ^ k0;
        c1 = l1;
        c2 = h0 ^ c3 ^ k1;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
// 24-bit uniform in [0, 1)
TD float uniform(uint32_t b) { return (float)(b >> 8) * 5.9604644775390625e-8f; }

// norms / cross products with the accumulation shape the reference's CPU kernels use (see DESIGN.md "arithmetic")
TD float norm2(float a, float b) { return __builtin_sqrtf(fma(b, b, a * a)); }
TD float norm3(float a, float b, float c) { return __builtin_sqrtf(fma(c, c, fma(b, b, a * a))); }
TD float cross_term(float a1, float b2, float a2, float b1) { return fma(a1, b2, -(a2 * b1)); }

}  // namespace taco
