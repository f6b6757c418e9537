// VALU issue cost vs the REGISTER NUMBERS of the operands (is there a VGPR bank structure a register allocator can trip over?), and the
// shader clock the chip holds under a saturating VALU load (s_memtime ticks per s_memrealtime 100 MHz tick).
// grid = 256 CUs x k blocks of 256 threads; every wave runs iters x 32 instructions of one pattern on hard-coded registers.
//   hipcc --offload-arch=gfx950 -O2 -o bank bank.hip && ./bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35"

struct Pat { const char *name; };
enum { FMA_SAME, FMA_DIFF, FMA_2SAME, MUL_SAME, MUL_DIFF, FMAC_SAME, FMAC_DIFF, FMA_SGPR, FMAAK, MUL_LIT, ADD_INL, MED3_DIFF, MED3_CONST, MAX3, CNDMASK_SGPR, NPAT };
static const char *names[NPAT] = {"v_fma  srcs v4 v8 v12 (all = 0 mod 4)", "v_fma  srcs v4 v9 v14 (0 1 2 mod 4)", "v_fma  srcs v4 v8 v13", "v_mul  srcs v4 v8", "v_mul  srcs v4 v9",
                                  "v_fmac dst v20+i srcs v4 v8", "v_fmac dst v20+i srcs v4 v9", "v_fma  s4 v9 v14", "v_fmaak v4 v9 lit", "v_mul  lit v4", "v_add  1.0 v4",
                                  "v_med3 v4 v9 v14", "v_med3 v4 s4 v9", "v_max3 v4 v9 v14", "v_cndmask v4 v9 s[6:7]"};

template <int P>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc) {
    asm volatile("v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n"
                 "v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
                 "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
                 "s_mov_b32 s4, 1.0\n s_mov_b32 s5, 2.0\n s_mov_b64 s[6:7], -1\n" ::: CLOB, "s4", "s5", "s6", "s7");
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define EIGHT(op, a, b, c) op " v20, " a ", " b c "\n" op " v21, " a ", " b c "\n" op " v22, " a ", " b c "\n" op " v23, " a ", " b c "\n" \
                           op " v24, " a ", " b c "\n" op " v25, " a ", " b c "\n" op " v26, " a ", " b c "\n" op " v27, " a ", " b c "\n"
        if (P == FMA_SAME) asm volatile(REP4(EIGHT("v_fma_f32", "v4", "v8", ", v12")) ::: CLOB);
        if (P == FMA_DIFF) asm volatile(REP4(EIGHT("v_fma_f32", "v4", "v9", ", v14")) ::: CLOB);
        if (P == FMA_2SAME) asm volatile(REP4(EIGHT("v_fma_f32", "v4", "v8", ", v13")) ::: CLOB);
        if (P == MUL_SAME) asm volatile(REP4(EIGHT("v_mul_f32", "v4", "v8", "")) ::: CLOB);
        if (P == MUL_DIFF) asm volatile(REP4(EIGHT("v_mul_f32", "v4", "v9", "")) ::: CLOB);
        if (P == FMAC_SAME) asm volatile(REP4(EIGHT("v_fmac_f32", "v4", "v8", "")) ::: CLOB);
        if (P == FMAC_DIFF) asm volatile(REP4(EIGHT("v_fmac_f32", "v4", "v9", "")) ::: CLOB);
        if (P == FMA_SGPR) asm volatile(REP4(EIGHT("v_fma_f32", "s4", "v9", ", v14")) ::: CLOB);
        if (P == FMAAK) asm volatile(REP4(EIGHT("v_fmaak_f32", "v4", "v9", ", 0x3f800123")) ::: CLOB);
        if (P == MUL_LIT) asm volatile(REP4(EIGHT("v_mul_f32", "0x3f800123", "v4", "")) ::: CLOB);
        if (P == ADD_INL) asm volatile(REP4(EIGHT("v_add_f32", "1.0", "v4", "")) ::: CLOB);
        if (P == MED3_DIFF) asm volatile(REP4(EIGHT("v_med3_f32", "v4", "v9", ", v14")) ::: CLOB);
        if (P == MED3_CONST) asm volatile(REP4(EIGHT("v_med3_f32", "v4", "s4", ", v9")) ::: CLOB);
        if (P == MAX3) asm volatile(REP4(EIGHT("v_max3_f32", "v4", "v9", ", v14")) ::: CLOB);
        if (P == CNDMASK_SGPR) asm volatile(REP4(EIGHT("v_cndmask_b32", "v4", "v9", ", s[6:7]")) ::: CLOB);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc;
    asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v27" : "=v"(acc)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int P> void run(float *d, unsigned long long *dc, int k_per_cu) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, 10, dc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(256 * k_per_cu), dim3(256), 0, 0, d, iters, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 32;
    printf("%-40s waves/SIMD=%d : %8.3f ms  %6.3f ns/instr/SIMD  ticks/instr/wave=%.2f  memtime/memrealtime=%.2f (x100 MHz)\n", names[P], k_per_cu, ms,
           ms * 1e6 / (n * k_per_cu), (double)c[0] / n, (double)c[0] / (double)c[1]);
}
template <int P> void all(float *d, unsigned long long *dc) { for (int kk : {1, 2, 4}) run<P>(d, dc, kk); }
int main() {
    float *d; unsigned long long *dc; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); (void)hipMalloc(&dc, 16);
    all<FMA_SAME>(d, dc); all<FMA_DIFF>(d, dc); all<FMA_2SAME>(d, dc); all<MUL_SAME>(d, dc); all<MUL_DIFF>(d, dc); all<FMAC_SAME>(d, dc); all<FMAC_DIFF>(d, dc);
    all<FMA_SGPR>(d, dc); all<FMAAK>(d, dc); all<MUL_LIT>(d, dc); all<ADD_INL>(d, dc); all<MED3_DIFF>(d, dc); all<MED3_CONST>(d, dc); all<MAX3>(d, dc); all<CNDMASK_SGPR>(d, dc);
    return 0;
}
