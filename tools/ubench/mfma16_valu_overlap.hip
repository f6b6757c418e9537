// How much VALU / transcendental work hides under v_mfma_f32_16x16x32_f16 (16 cycles of the SIMD's matrix pipe) INSIDE one wavefront's
// instruction stream on gfx950?  (the design question of the split LSTM's cells: its chains and its cells run back to back today)
// One workgroup per CU; 64 threads (one wavefront on a SIMD) or 512 (two per SIMD, both the same stream).  Stream: 1024 x [one MFMA on one of
// four independent accumulators, then NV independent v_fma_f32 (KIND 0) / v_exp_f32 (KIND 1) / alternating v_exp, v_rcp, v_fma, v_fma (KIND 2)].
// Reported: shader-clock cycles of wave 0 of workgroup 0 per iteration.
//   hipcc --offload-arch=gfx950 -O2 -o mfma16_valu_overlap mfma16_valu_overlap.hip && ./mfma16_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND, bool MFMA>
__global__ void k(float *out, unsigned long long *cyc, float seed) {
    f4 a[4];
    for (int j = 0; j < 4; ++j) a[j] = f4{seed, 0, 0, 0};
    h8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)seed; y[j] = (_Float16)(seed * 0.5f); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (MFMA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a[q]) : "v"(x), "v"(y));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = (q * NV + j) & 7;
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(seed));
                else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else {
                    if ((j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                    else if ((j & 3) == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(seed));
                }
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j) s += a[j][j];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int KIND, bool MFMA>
static void run(const char *what, int threads, float *o, unsigned long long *c) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NV, KIND, MFMA>), dim3(256), dim3(threads), 0, 0, o, c, 1.0f);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-46s %3d threads: %7.1f cycles per MFMA slot\n", what, threads, (double)h / 1024.0);
}

int main() {
    float *o; unsigned long long *c;
    (void)hipMalloc(&o, 4); (void)hipMalloc(&c, 8);
    for (int threads : {64, 512}) {
        run<0, 0, true>("MFMA alone", threads, o, c);
        run<1, 0, true>("MFMA + 1 v_fma", threads, o, c);
        run<2, 0, true>("MFMA + 2 v_fma", threads, o, c);
        run<3, 0, true>("MFMA + 3 v_fma", threads, o, c);
        run<4, 0, true>("MFMA + 4 v_fma", threads, o, c);
        run<6, 0, true>("MFMA + 6 v_fma", threads, o, c);
        run<1, 1, true>("MFMA + 1 v_exp", threads, o, c);
        run<2, 1, true>("MFMA + 2 v_exp", threads, o, c);
        run<4, 2, true>("MFMA + (exp, rcp, fma, fma)", threads, o, c);
        run<2, 0, false>("2 v_fma alone", threads, o, c);
        run<4, 0, false>("4 v_fma alone", threads, o, c);
        run<2, 1, false>("2 v_exp alone", threads, o, c);
        run<4, 2, false>("(exp, rcp, fma, fma) alone", threads, o, c);
    }
    return 0;
}
