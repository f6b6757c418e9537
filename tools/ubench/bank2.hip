// Second VALU issue-cost table (follow-up of bank.hip): which operand KINDS and register patterns make a wave64 VALU instruction cost two
// issue passes on gfx950.  Same harness: grid = 256 CUs x k blocks of 256 threads, iters x 32 instructions per wave, hard-coded registers.
//   hipcc --offload-arch=gfx950 -O2 -o bank2 bank2.hip && ./bank2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "vcc", "s8", "s9"

#define PATTERNS(X) \
    X(0, "v_mul v, s4, v9 (VOP2, SGPR src0)", "v_mul_f32 v20, s4, v9\n v_mul_f32 v21, s4, v9\n v_mul_f32 v22, s4, v9\n v_mul_f32 v23, s4, v9\n v_mul_f32 v24, s4, v9\n v_mul_f32 v25, s4, v9\n v_mul_f32 v26, s4, v9\n v_mul_f32 v27, s4, v9\n") \
    X(1, "v_add v, s4, v9", "v_add_f32 v20, s4, v9\n v_add_f32 v21, s4, v9\n v_add_f32 v22, s4, v9\n v_add_f32 v23, s4, v9\n v_add_f32 v24, s4, v9\n v_add_f32 v25, s4, v9\n v_add_f32 v26, s4, v9\n v_add_f32 v27, s4, v9\n") \
    X(2, "v_fma v4, v9, s4 (SGPR src2)", "v_fma_f32 v20, v4, v9, s4\n v_fma_f32 v21, v4, v9, s4\n v_fma_f32 v22, v4, v9, s4\n v_fma_f32 v23, v4, v9, s4\n v_fma_f32 v24, v4, v9, s4\n v_fma_f32 v25, v4, v9, s4\n v_fma_f32 v26, v4, v9, s4\n v_fma_f32 v27, v4, v9, s4\n") \
    X(3, "v_fmac v20+i, s4, v9", "v_fmac_f32 v20, s4, v9\n v_fmac_f32 v21, s4, v9\n v_fmac_f32 v22, s4, v9\n v_fmac_f32 v23, s4, v9\n v_fmac_f32 v24, s4, v9\n v_fmac_f32 v25, s4, v9\n v_fmac_f32 v26, s4, v9\n v_fmac_f32 v27, s4, v9\n") \
    X(4, "v_fma rotating, banks all different", "v_fma_f32 v20, v4, v9, v14\n v_fma_f32 v21, v5, v10, v15\n v_fma_f32 v22, v6, v11, v12\n v_fma_f32 v23, v7, v8, v13\n v_fma_f32 v24, v4, v9, v14\n v_fma_f32 v25, v5, v10, v15\n v_fma_f32 v26, v6, v11, v12\n v_fma_f32 v27, v7, v8, v13\n") \
    X(5, "v_fma rotating, two sources same bank", "v_fma_f32 v20, v4, v8, v13\n v_fma_f32 v21, v5, v9, v14\n v_fma_f32 v22, v6, v10, v15\n v_fma_f32 v23, v7, v11, v12\n v_fma_f32 v24, v4, v8, v13\n v_fma_f32 v25, v5, v9, v14\n v_fma_f32 v26, v6, v10, v15\n v_fma_f32 v27, v7, v11, v12\n") \
    X(6, "v_fma rotating, three sources same bank", "v_fma_f32 v20, v4, v8, v12\n v_fma_f32 v21, v5, v9, v13\n v_fma_f32 v22, v6, v10, v14\n v_fma_f32 v23, v7, v11, v15\n v_fma_f32 v24, v4, v8, v12\n v_fma_f32 v25, v5, v9, v13\n v_fma_f32 v26, v6, v10, v14\n v_fma_f32 v27, v7, v11, v15\n") \
    X(7, "v_fmac rotating srcs different banks, dst any", "v_fmac_f32 v20, v4, v9\n v_fmac_f32 v21, v5, v10\n v_fmac_f32 v22, v6, v11\n v_fmac_f32 v23, v7, v8\n v_fmac_f32 v24, v4, v9\n v_fmac_f32 v25, v5, v10\n v_fmac_f32 v26, v6, v11\n v_fmac_f32 v27, v7, v8\n") \
    X(8, "v_fmac srcs same bank, dst OTHER bank", "v_fmac_f32 v21, v4, v8\n v_fmac_f32 v22, v4, v8\n v_fmac_f32 v23, v4, v8\n v_fmac_f32 v25, v4, v8\n v_fmac_f32 v26, v4, v8\n v_fmac_f32 v27, v4, v8\n v_fmac_f32 v29, v4, v8\n v_fmac_f32 v30, v4, v8\n") \
    X(9, "v_fmac srcs same bank, dst SAME bank", "v_fmac_f32 v20, v4, v8\n v_fmac_f32 v24, v4, v8\n v_fmac_f32 v28, v4, v8\n v_fmac_f32 v32, v4, v8\n v_fmac_f32 v20, v4, v8\n v_fmac_f32 v24, v4, v8\n v_fmac_f32 v28, v4, v8\n v_fmac_f32 v32, v4, v8\n") \
    X(10, "v_fmac srcs differ, dst bank = src0 bank", "v_fmac_f32 v20, v4, v9\n v_fmac_f32 v24, v4, v9\n v_fmac_f32 v28, v4, v9\n v_fmac_f32 v32, v4, v9\n v_fmac_f32 v20, v4, v9\n v_fmac_f32 v24, v4, v9\n v_fmac_f32 v28, v4, v9\n v_fmac_f32 v32, v4, v9\n") \
    X(11, "v_cmp_gt vcc only", "v_cmp_gt_f32 vcc, v4, v9\n v_cmp_gt_f32 vcc, v5, v10\n v_cmp_gt_f32 vcc, v6, v11\n v_cmp_gt_f32 vcc, v7, v8\n v_cmp_gt_f32 vcc, v4, v9\n v_cmp_gt_f32 vcc, v5, v10\n v_cmp_gt_f32 vcc, v6, v11\n v_cmp_gt_f32 vcc, v7, v8\n") \
    X(12, "v_cmp_gt s[8:9] (e64)", "v_cmp_gt_f32 s[8:9], v4, v9\n v_cmp_gt_f32 s[8:9], v5, v10\n v_cmp_gt_f32 s[8:9], v6, v11\n v_cmp_gt_f32 s[8:9], v7, v8\n v_cmp_gt_f32 s[8:9], v4, v9\n v_cmp_gt_f32 s[8:9], v5, v10\n v_cmp_gt_f32 s[8:9], v6, v11\n v_cmp_gt_f32 s[8:9], v7, v8\n") \
    X(13, "v_cndmask vcc (mask static)", "v_cndmask_b32 v20, v4, v9, vcc\n v_cndmask_b32 v21, v5, v10, vcc\n v_cndmask_b32 v22, v6, v11, vcc\n v_cndmask_b32 v23, v7, v8, vcc\n v_cndmask_b32 v24, v4, v9, vcc\n v_cndmask_b32 v25, v5, v10, vcc\n v_cndmask_b32 v26, v6, v11, vcc\n v_cndmask_b32 v27, v7, v8, vcc\n") \
    X(14, "v_mul e64 with neg modifier", "v_mul_f32 v20, -v4, v9\n v_mul_f32 v21, -v5, v10\n v_mul_f32 v22, -v6, v11\n v_mul_f32 v23, -v7, v8\n v_mul_f32 v24, -v4, v9\n v_mul_f32 v25, -v5, v10\n v_mul_f32 v26, -v6, v11\n v_mul_f32 v27, -v7, v8\n") \
    X(15, "v_fma with neg modifier, banks differ", "v_fma_f32 v20, -v4, v9, v14\n v_fma_f32 v21, -v5, v10, v15\n v_fma_f32 v22, -v6, v11, v12\n v_fma_f32 v23, -v7, v8, v13\n v_fma_f32 v24, -v4, v9, v14\n v_fma_f32 v25, -v5, v10, v15\n v_fma_f32 v26, -v6, v11, v12\n v_fma_f32 v27, -v7, v8, v13\n") \
    X(16, "v_mov v, s4", "v_mov_b32 v20, s4\n v_mov_b32 v21, s4\n v_mov_b32 v22, s4\n v_mov_b32 v23, s4\n v_mov_b32 v24, s4\n v_mov_b32 v25, s4\n v_mov_b32 v26, s4\n v_mov_b32 v27, s4\n") \
    X(17, "v_max + v_min literal (clamp by two VOP2)", "v_max_f32 v20, 0x42c80000, v4\n v_min_f32 v20, 0x447a0000, v20\n v_max_f32 v21, 0x42c80000, v5\n v_min_f32 v21, 0x447a0000, v21\n v_max_f32 v22, 0x42c80000, v6\n v_min_f32 v22, 0x447a0000, v22\n v_max_f32 v23, 0x42c80000, v7\n v_min_f32 v23, 0x447a0000, v23\n") \
    X(18, "v_med3 v, v, v (bounds in VGPRs, banks differ)", "v_med3_f32 v20, v4, v9, v14\n v_med3_f32 v21, v5, v9, v14\n v_med3_f32 v22, v6, v9, v14\n v_med3_f32 v23, v7, v9, v14\n v_med3_f32 v24, v4, v9, v14\n v_med3_f32 v25, v5, v9, v14\n v_med3_f32 v26, v6, v9, v14\n v_med3_f32 v27, v7, v9, v14\n") \
    X(19, "v_sub v, v4, v9", "v_sub_f32 v20, v4, v9\n v_sub_f32 v21, v5, v10\n v_sub_f32 v22, v6, v11\n v_sub_f32 v23, v7, v8\n v_sub_f32 v24, v4, v9\n v_sub_f32 v25, v5, v10\n v_sub_f32 v26, v6, v11\n v_sub_f32 v27, v7, v8\n") \
    X(20, "v_fmamk v, v4, lit, v9", "v_fmamk_f32 v20, v4, 0x3f800123, v9\n v_fmamk_f32 v21, v5, 0x3f800123, v10\n v_fmamk_f32 v22, v6, 0x3f800123, v11\n v_fmamk_f32 v23, v7, 0x3f800123, v8\n v_fmamk_f32 v24, v4, 0x3f800123, v9\n v_fmamk_f32 v25, v5, 0x3f800123, v10\n v_fmamk_f32 v26, v6, 0x3f800123, v11\n v_fmamk_f32 v27, v7, 0x3f800123, v8\n") \
    X(21, "v_fma v4, v9, 1.0 (inline constant src2)", "v_fma_f32 v20, v4, v9, 1.0\n v_fma_f32 v21, v5, v10, 1.0\n v_fma_f32 v22, v6, v11, 1.0\n v_fma_f32 v23, v7, v8, 1.0\n v_fma_f32 v24, v4, v9, 1.0\n v_fma_f32 v25, v5, v10, 1.0\n v_fma_f32 v26, v6, v11, 1.0\n v_fma_f32 v27, v7, v8, 1.0\n") \
    X(22, "v_mul_f32 e64 no modifiers (VOP3 encoding of a 2-source op)", "v_mul_f32_e64 v20, v4, v9\n v_mul_f32_e64 v21, v5, v10\n v_mul_f32_e64 v22, v6, v11\n v_mul_f32_e64 v23, v7, v8\n v_mul_f32_e64 v24, v4, v9\n v_mul_f32_e64 v25, v5, v10\n v_mul_f32_e64 v26, v6, v11\n v_mul_f32_e64 v27, v7, v8\n") \
    X(23, "v_mul v, v4, v9 (reference, rotating)", "v_mul_f32 v20, v4, v9\n v_mul_f32 v21, v5, v10\n v_mul_f32 v22, v6, v11\n v_mul_f32 v23, v7, v8\n v_mul_f32 v24, v4, v9\n v_mul_f32 v25, v5, v10\n v_mul_f32 v26, v6, v11\n v_mul_f32 v27, v7, v8\n")

#define NAME(i, n, s) n,
static const char *names[] = {PATTERNS(NAME)};
constexpr int NPAT = sizeof(names) / sizeof(names[0]);

template <int P>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc) {
    asm volatile("v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n"
                 "v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
                 "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
                 "v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n"
                 "s_mov_b32 s4, 1.0\n s_mov_b32 s5, 2.0\n s_mov_b64 vcc, -1\n" ::: CLOB, "s4", "s5");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define BODY(idx, n, s) if (P == idx) asm volatile(REP4(s) ::: CLOB);
        PATTERNS(BODY)
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float acc;
    asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v27" : "=v"(acc)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int P> void run(float *d, unsigned long long *dc) {
    const int iters = 4000;
    for (int k_per_cu : {2, 4}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, 10, dc);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, iters, dc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 32;
        printf("%-60s waves/SIMD=%d : %6.3f ns/instr/SIMD\n", names[P], k_per_cu, ms * 1e6 / (n * k_per_cu));
    }
}
template <int P> struct All { static void go(float *d, unsigned long long *dc) { All<P - 1>::go(d, dc); run<P>(d, dc); } };
template <> struct All<-1> { static void go(float *, unsigned long long *) {} };
int main() {
    float *d; unsigned long long *dc; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); (void)hipMalloc(&dc, 16);
    All<NPAT - 1>::go(d, dc);
    return 0;
}
