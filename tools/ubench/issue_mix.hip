// Issue cost per wave-instruction on one SIMD, by instruction class and by waves per SIMD.
// grid = 256 CUs x k blocks of 256 threads (one wave per SIMD per block), every wave runs `iters` x 32 copies of one pattern.
// Reports ns per instruction per SIMD = time / (iters * 32 * n_instr_in_pattern * k) and shader-clock cycles (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

enum { PKMUL, PKFMA, FMA, MUL2, ADD2, CHAIN, CNDMASK, CMPCND, MED3, RCP, SQRT, MULHI, MULLO, FMAAK, XOR, DIVSEQ, LDSRD, CVT, NPAT };
static const char *names[NPAT] = {"v_pk_mul_f32 (2 ops each)", "v_pk_fma_f32 (2 ops each)", "v_fma_f32 (8 indep)", "v_mul_f32 e32 indep", "v_add_f32 e32 indep", "v_fma dependent chain", "v_cndmask vcc (static)",
                                  "v_cmp+s_nop1+v_cndmask", "v_med3_f32", "v_rcp_f32", "v_sqrt_f32", "v_mul_hi_u32", "v_mul_lo_u32", "v_fmaak_f32 literal",
                                  "v_xor_b32", "full IEEE div sequence(10)", "ds_read_b128", "v_cvt_f32_u32"};
static const int ninstr[NPAT] = {8, 8, 8, 8, 8, 8, 8, 8 * 2, 8, 8, 8, 8, 8, 8, 8, 10, 4, 8};

template <int P>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc) {
    __shared__ float4 sh[256];
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pm = {m, m}, pc = {c, c};
    sh[threadIdx.x] = float4{a0, a1, a2, a3};
    __syncthreads();
    uint32_t addr = threadIdx.x * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (P == PKMUL) asm volatile(REP4("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        if (P == PKFMA) asm volatile(REP4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
        if (P == FMA) asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == MUL2) asm volatile(REP4("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == ADD2) asm volatile(REP4("v_add_f32 %0, %9, %0\n v_add_f32 %1, %9, %1\n v_add_f32 %2, %9, %2\n v_add_f32 %3, %9, %3\n"
                             "v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_add_f32 %6, %9, %6\n v_add_f32 %7, %9, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == CHAIN) asm volatile(REP32("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
        if (P == CNDMASK) asm volatile(REP4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
        if (P == CMPCND) asm volatile(REP4("v_cmp_gt_f32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_f32 vcc, %1, %8\n s_nop 1\n v_cndmask_b32 %1, %1, %9, vcc\n"
                             "v_cmp_gt_f32 vcc, %2, %8\n s_nop 1\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_f32 vcc, %3, %8\n s_nop 1\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_gt_f32 vcc, %4, %8\n s_nop 1\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_gt_f32 vcc, %5, %8\n s_nop 1\n v_cndmask_b32 %5, %5, %9, vcc\n"
                             "v_cmp_gt_f32 vcc, %6, %8\n s_nop 1\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_gt_f32 vcc, %7, %8\n s_nop 1\n v_cndmask_b32 %7, %7, %9, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
        if (P == MED3) asm volatile(REP4("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                             "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == RCP) asm volatile(REP4("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (P == SQRT) asm volatile(REP4("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (P == MULHI) asm volatile(REP4("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                             "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == MULLO) asm volatile(REP4("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == FMAAK) asm volatile(REP4("v_fmaak_f32 %0, %0, %8, 0x3f800123\n v_fmaak_f32 %1, %1, %8, 0x3f800123\n v_fmaak_f32 %2, %2, %8, 0x3f800123\n v_fmaak_f32 %3, %3, %8, 0x3f800123\n"
                             "v_fmaak_f32 %4, %4, %8, 0x3f800123\n v_fmaak_f32 %5, %5, %8, 0x3f800123\n v_fmaak_f32 %6, %6, %8, 0x3f800123\n v_fmaak_f32 %7, %7, %8, 0x3f800123\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == XOR) asm volatile(REP4("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                             "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        if (P == CVT) asm volatile(REP4("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (P == DIVSEQ) {
#pragma unroll
            for (int r = 0; r < 32; ++r) a0 = a1 / (a0 + 2.0f) + 1.0f;  // compiler's IEEE sequence (+2 adds): counted as 10
        }
        if (P == LDSRD) {
            float4 v0, v1, v2, v3;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr & 2047u));
                a0 += v0.x + v1.y + v2.z + v3.w;
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int P> void run(float *d, unsigned long long *dc, int k_per_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, 10, dc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, iters, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 32 / 8 * ninstr[P] * (P == DIVSEQ ? 8.0 / 1 * 1 : 1) ;
    const double per = (P == DIVSEQ) ? (double)iters * 32 * 10 : (P == LDSRD ? (double)iters * 8 * 4 : n);
    printf("%-28s waves/SIMD=%d : %8.3f ms  %6.2f ns/instr/SIMD  wave-0 clock ticks/instr=%.2f\n", names[P], k_per_cu, ms, ms * 1e6 / (per * k_per_cu), (double)c / per);
}
template <int P> void all(float *d, unsigned long long *dc) { for (int kk : {1, 2, 4}) run<P>(d, dc, kk); }
int main() {
    float *d; unsigned long long *dc; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); (void)hipMalloc(&dc, 8);
    all<PKMUL>(d, dc); all<PKFMA>(d, dc); all<FMA>(d, dc); all<MUL2>(d, dc); all<ADD2>(d, dc); all<CHAIN>(d, dc); all<CNDMASK>(d, dc); all<CMPCND>(d, dc); all<MED3>(d, dc);
    all<RCP>(d, dc); all<SQRT>(d, dc); all<MULHI>(d, dc); all<MULLO>(d, dc); all<FMAAK>(d, dc); all<XOR>(d, dc); all<DIVSEQ>(d, dc); all<LDSRD>(d, dc); all<CVT>(d, dc);
    return 0;
}
