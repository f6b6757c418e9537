// Round 5 price list for two things the round-4 review found in the throughput form's ISA:
//   (1) Philox rounds as v_mul_hi_u32 + v_mul_lo_u32 pairs vs ONE v_mad_u64_u32 per product -- raw instruction rates and a whole Philox4x32-10
//       block written both ways in C (same bits by construction), at 2 and 4 wavefronts per SIMD;
//   (2) v_cndmask_b32 ALONE with a static mask (bank2's 9.5 ns line) vs inside a realistic mix (one select per four fma, mask from a v_cmp just
//       ahead, VCC and SGPR-pair masks) vs the bit-exact alternatives (v_bfi_b32 on a precomputed mask, v_med3_f32).
//   hipcc --offload-arch=gfx950 -O2 -o intmul_cndmask intmul_cndmask.hip && ./intmul_cndmask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "vcc", "s8", "s9"

// every pattern is 8 instructions; REP4 -> 32 per loop trip
#define PATTERNS(X) \
    X(0, "v_fma_f32 rotating (reference)", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_fma_f32 v23, v7, v8, v13\n v_fma_f32 v24, v4, v9, v14\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n v_fma_f32 v27, v7, v8, v13\n") \
    X(1, "v_mul_lo_u32", "v_mul_lo_u32 v20, v4, v9\n v_mul_lo_u32 v21, v5, v10\n v_mul_lo_u32 v22, v6, v11\n v_mul_lo_u32 v23, v7, v8\n v_mul_lo_u32 v24, v4, v9\n v_mul_lo_u32 v25, v5, v10\n v_mul_lo_u32 v26, v6, v11\n v_mul_lo_u32 v27, v7, v8\n") \
    X(2, "v_mul_hi_u32", "v_mul_hi_u32 v20, v4, v9\n v_mul_hi_u32 v21, v5, v10\n v_mul_hi_u32 v22, v6, v11\n v_mul_hi_u32 v23, v7, v8\n v_mul_hi_u32 v24, v4, v9\n v_mul_hi_u32 v25, v5, v10\n v_mul_hi_u32 v26, v6, v11\n v_mul_hi_u32 v27, v7, v8\n") \
    X(3, "v_mad_u64_u32 (one per hi+lo product; counts as ONE instr)", "v_mad_u64_u32 v[20:21], s[8:9], v4, v9, 0\n v_mad_u64_u32 v[22:23], s[8:9], v5, v10, 0\n v_mad_u64_u32 v[24:25], s[8:9], v6, v11, 0\n v_mad_u64_u32 v[26:27], s[8:9], v7, v8, 0\n v_mad_u64_u32 v[28:29], s[8:9], v4, v9, 0\n v_mad_u64_u32 v[30:31], s[8:9], v5, v10, 0\n v_mad_u64_u32 v[32:33], s[8:9], v6, v11, 0\n v_mad_u64_u32 v[34:35], s[8:9], v7, v8, 0\n") \
    X(4, "v_mul_hi_u32 with an SGPR multiplier (what Philox compiles to: VOP3 takes no literal)", "v_mul_hi_u32 v20, s4, v9\n v_mul_hi_u32 v21, s4, v10\n v_mul_hi_u32 v22, s4, v11\n v_mul_hi_u32 v23, s4, v8\n v_mul_hi_u32 v24, s4, v9\n v_mul_hi_u32 v25, s4, v10\n v_mul_hi_u32 v26, s4, v11\n v_mul_hi_u32 v27, s4, v8\n") \
    X(5, "v_cndmask vcc, mask static, ALONE (bank2's line)", "v_cndmask_b32 v20, v4, v9, vcc\n v_cndmask_b32 v21, v5, v10, vcc\n v_cndmask_b32 v22, v6, v11, vcc\n v_cndmask_b32 v23, v7, v8, vcc\n v_cndmask_b32 v24, v4, v9, vcc\n v_cndmask_b32 v25, v5, v10, vcc\n v_cndmask_b32 v26, v6, v11, vcc\n v_cndmask_b32 v27, v7, v8, vcc\n") \
    X(6, "v_cndmask s[8:9] (e64), mask static, ALONE", "v_cndmask_b32 v20, v4, v9, s[8:9]\n v_cndmask_b32 v21, v5, v10, s[8:9]\n v_cndmask_b32 v22, v6, v11, s[8:9]\n v_cndmask_b32 v23, v7, v8, s[8:9]\n v_cndmask_b32 v24, v4, v9, s[8:9]\n v_cndmask_b32 v25, v5, v10, s[8:9]\n v_cndmask_b32 v26, v6, v11, s[8:9]\n v_cndmask_b32 v27, v7, v8, s[8:9]\n") \
    X(7, "mix: 6 fma + v_cmp vcc + dependent v_cndmask vcc", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_cmp_gt_f32 vcc, v4, v9\n v_fma_f32 v23, v7, v8, v13\n v_cndmask_b32 v24, v4, v9, vcc\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n") \
    X(8, "mix: 6 fma + v_cmp s[8:9] + dependent v_cndmask s[8:9]", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_cmp_gt_f32 s[8:9], v4, v9\n v_fma_f32 v23, v7, v8, v13\n v_cndmask_b32 v24, v4, v9, s[8:9]\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n") \
    X(9, "mix: 7 fma + v_cndmask vcc (mask static)", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_fma_f32 v27, v6, v11, v12\n v_fma_f32 v23, v7, v8, v13\n v_cndmask_b32 v24, v4, v9, vcc\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n") \
    X(10, "mix: 7 fma + v_bfi_b32 (select by a VGPR mask)", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_fma_f32 v27, v6, v11, v12\n v_fma_f32 v23, v7, v8, v13\n v_bfi_b32 v24, v16, v4, v9\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n") \
    X(11, "mix: 7 fma + v_med3_f32 (VGPR bounds)", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_fma_f32 v27, v6, v11, v12\n v_fma_f32 v23, v7, v8, v13\n v_med3_f32 v24, v4, v9, v14\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n") \
    X(12, "v_cndmask vcc, static mask, DISTINCT sources every time (dst rotating)", "v_cndmask_b32 v20, v4, v9, vcc\n v_cndmask_b32 v21, v5, v10, vcc\n v_cndmask_b32 v22, v6, v11, vcc\n v_cndmask_b32 v23, v7, v8, vcc\n v_cndmask_b32 v24, v12, v13, vcc\n v_cndmask_b32 v25, v14, v15, vcc\n v_cndmask_b32 v26, v16, v17, vcc\n v_cndmask_b32 v27, v18, v19, vcc\n") \
    X(13, "v_bfi_b32 ALONE", "v_bfi_b32 v20, v16, v4, v9\n v_bfi_b32 v21, v16, v5, v10\n v_bfi_b32 v22, v16, v6, v11\n v_bfi_b32 v23, v16, v7, v8\n v_bfi_b32 v24, v16, v4, v9\n v_bfi_b32 v25, v16, v5, v10\n v_bfi_b32 v26, v16, v6, v11\n v_bfi_b32 v27, v16, v7, v8\n")

#define NAME(i, n, s) n,
static const char *names[] = {PATTERNS(NAME)};
constexpr int NPAT = sizeof(names) / sizeof(names[0]);

template <int P>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    asm volatile("v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.5\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 2.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n"
                 "v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0x0000ffff\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 2.0\n v_mov_b32 v19, 3.0\n"
                 "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
                 "v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n"
                 "s_mov_b64 vcc, 0x0f0f0f0f\n s_mov_b64 s[8:9], 0x33333333\n s_mov_b32 s4, 0xD2511F53\n" ::: CLOB, "s4");
    for (int i = 0; i < iters; ++i) {
#define BODY(idx, n, s) if (P == idx) asm volatile(REP4(s) ::: CLOB, "s4");
        PATTERNS(BODY)
    }
    float acc;
    asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v27\n v_add_f32 %0, %0, v24" : "=v"(acc)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int P> void run(float *d) {
    const int iters = 4000;
    for (int k_per_cu : {2, 4}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 32;
        printf("%-78s waves/SIMD=%d : %6.3f ns/instr/SIMD\n", names[P], k_per_cu, ms * 1e6 / (n * k_per_cu));
    }
}
template <int P> struct All { static void go(float *d) { All<P - 1>::go(d); run<P>(d); } };
template <> struct All<-1> { static void go(float *) {} };

// ---- a whole Philox4x32-10 block, both ways (same bits): per-block time at full occupancy
template <bool WIDE>
__device__ __forceinline__ void philox(uint32_t k0, uint32_t k1, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0, l0, h1, l1;
        if constexpr (WIDE) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            h0 = (uint32_t)(p0 >> 32); l0 = (uint32_t)p0; h1 = (uint32_t)(p1 >> 32); l1 = (uint32_t)p1;
        } else {
            h0 = __umulhi(0xD2511F53u, c0); l0 = 0xD2511F53u * c0; h1 = __umulhi(0xCD9E8D57u, c2); l1 = 0xCD9E8D57u * c2;
        }
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
template <bool WIDE>
__global__ __launch_bounds__(256) void pk(uint32_t *out, int blocks) {
    uint32_t acc = 0;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int b = 0; b < blocks; ++b) {
        uint32_t c0 = gid, c1 = (uint32_t)b, c2 = 4, c3 = acc & 1;   // (the dependence on acc keeps the blocks in order)
        philox<WIDE>(0x1234567u, 0x89abcdeu, c0, c1, c2, c3);
        acc ^= c0 ^ c1 ^ c2 ^ c3;
    }
    out[gid] = acc;
}
template <bool WIDE> double time_philox(uint32_t *d, int k_per_cu, uint32_t *first) {
    const int blocks = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(pk<WIDE>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(pk<WIDE>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, blocks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(first, d, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    return ms * 1e6 / ((double)blocks * k_per_cu);   // ns per Philox block per wavefront slot of a SIMD
}

int main() {
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    All<NPAT - 1>::go(d);
    uint32_t a[64], b[64];
    for (int k_per_cu : {1, 2, 4}) {
        const double t0 = time_philox<false>((uint32_t *)d, k_per_cu, a), t1 = time_philox<true>((uint32_t *)d, k_per_cu, b);
        bool same = true;
        for (int i = 0; i < 64; ++i) same = same && a[i] == b[i];
        printf("Philox4x32-10 block, waves/SIMD=%d : mul_hi + mul_lo %7.1f ns   64-bit products %7.1f ns   (%s)\n", k_per_cu, t0, t1, same ? "same bits" : "BITS DIFFER");
    }
    return 0;
}
