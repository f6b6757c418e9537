// Can the VALU work of one wavefront issue under the 16-bit MFMAs of the OTHER wavefront of its SIMD on gfx950 -- and does it help to keep the
// MFMA wavefront from presenting its next MFMA while the matrix pipe is still busy?  (round 6; the design question of taco_critic_lstm_pair_split_kernel:
// tools/ubench/mfma_valu found f32 / bf16 MFMAs of wave w and v_fma of wave w + 4 finishing in the SUM of their times, although ONE wavefront that
// interleaves them overlaps them.  Hypothesis: an MFMA that waits for the pipe sits in the SIMD's VALU issue stage and blocks the other wavefront.)
// One workgroup of 512 threads on every CU: wavefronts w and w + 4 share SIMD w.  Waves 0-3 run role X, waves 4-7 role Y:
//   M<k>: 1024 v_mfma_f32_16x16x32_f16 on 4 independent accumulators, `s_nop k-1` behind each (k = 0: none)
//   V: 4096 independent v_fma_f32      E: 4096 v_exp_f32        -: return at once
// Reported: shader-clock cycles of wave 0 and wave 4 (workgroup 0).
//   hipcc --offload-arch=gfx950 -O2 -o mfma16_cross_wave mfma16_cross_wave.hip && ./mfma16_cross_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NOP>
__device__ __forceinline__ void gap() {
    if constexpr (NOP == 1) asm volatile("s_nop 0");
    else if constexpr (NOP == 2) asm volatile("s_nop 1");
    else if constexpr (NOP == 4) asm volatile("s_nop 3");
    else if constexpr (NOP == 6) asm volatile("s_nop 5");
    else if constexpr (NOP == 8) asm volatile("s_nop 7");
    else if constexpr (NOP == 10) asm volatile("s_nop 9");
    else if constexpr (NOP == 12) asm volatile("s_nop 11");
    else if constexpr (NOP == 14) asm volatile("s_nop 13");
    else if constexpr (NOP == 16) asm volatile("s_nop 15");
    else if constexpr (NOP == 100) asm volatile("s_sleep 0");   // (the scheduler's own "give the slot away" hint)
    else if constexpr (NOP == 200) asm volatile("s_setprio 0");
}

// ROLE: 0 none; 1000 + k: M<k>; 2: V; 6: E
template <int ROLE>
__device__ __forceinline__ float work(float seed) {
    if constexpr (ROLE >= 1000) {
        constexpr int NOP = ROLE - 1000;
        f4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        h8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (_Float16)seed; y[j] = (_Float16)(seed * 0.5f); }
        for (int i = 0; i < 256; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a0, 0, 0, 0); gap<NOP>();
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a1, 0, 0, 0); gap<NOP>();
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a2, 0, 0, 0); gap<NOP>();
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a3, 0, 0, 0); gap<NOP>();
        }
        return a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (ROLE == 2) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = seed + j;
        for (int i = 0; i < 512; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(seed));
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j];
        return s;
    } else if constexpr (ROLE == 6) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = seed + j;
        for (int i = 0; i < 512; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j];
        return s;
    }
    return 0.0f;
}

template <int X, int Y, int PRIO>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    if constexpr (PRIO == 1) { if (wave >= 4) asm volatile("s_setprio 3"); }   // the VALU wavefront at raised priority
    if constexpr (PRIO == 2) { if (wave < 4) asm volatile("s_setprio 3"); }    // the MFMA wavefront at raised priority
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    float r = wave < 4 ? work<X>(seed) : work<Y>(seed);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

template <int X, int Y, int PRIO = 0>
void run(const char *name, float *d, unsigned long long *dc) {
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<X, Y, PRIO>), dim3(256), dim3(512), 0, 0, d, dc, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s wave 0: %8llu cycles   wave 4: %8llu cycles\n", name, h[0], h[1]);
}

int main() {
    float *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    hipMalloc(&dc, 2 * sizeof(unsigned long long));
    run<1000, 0>("f16 MFMA 16x16x32 x1024 | -", d, dc);
    run<0, 2>("- | v_fma x4096", d, dc);
    run<0, 6>("- | v_exp x4096", d, dc);
    run<1000, 2>("f16 MFMA x1024 | v_fma x4096", d, dc);
    run<1000, 6>("f16 MFMA x1024 | v_exp x4096", d, dc);
    run<1000, 1000>("f16 MFMA x1024 | f16 MFMA x1024", d, dc);
    run<1001, 0>("(MFMA, s_nop 0) x1024 | -", d, dc);
    run<1002, 0>("(MFMA, s_nop 1) x1024 | -", d, dc);
    run<1004, 0>("(MFMA, s_nop 3) x1024 | -", d, dc);
    run<1008, 0>("(MFMA, s_nop 7) x1024 | -", d, dc);
    run<1012, 0>("(MFMA, s_nop 11) x1024 | -", d, dc);
    run<1016, 0>("(MFMA, s_nop 15) x1024 | -", d, dc);
    run<1001, 2>("(MFMA, s_nop 0) x1024 | v_fma x4096", d, dc);
    run<1002, 2>("(MFMA, s_nop 1) x1024 | v_fma x4096", d, dc);
    run<1004, 2>("(MFMA, s_nop 3) x1024 | v_fma x4096", d, dc);
    run<1006, 2>("(MFMA, s_nop 5) x1024 | v_fma x4096", d, dc);
    run<1008, 2>("(MFMA, s_nop 7) x1024 | v_fma x4096", d, dc);
    run<1010, 2>("(MFMA, s_nop 9) x1024 | v_fma x4096", d, dc);
    run<1012, 2>("(MFMA, s_nop 11) x1024 | v_fma x4096", d, dc);
    run<1014, 2>("(MFMA, s_nop 13) x1024 | v_fma x4096", d, dc);
    run<1016, 2>("(MFMA, s_nop 15) x1024 | v_fma x4096", d, dc);
    run<1008, 6>("(MFMA, s_nop 7) x1024 | v_exp x4096", d, dc);
    run<1012, 6>("(MFMA, s_nop 11) x1024 | v_exp x4096", d, dc);
    run<1100, 2>("(MFMA, s_sleep 0) x1024 | v_fma x4096", d, dc);
    run<1100, 0>("(MFMA, s_sleep 0) x1024 | -", d, dc);
    run<1000, 2, 1>("f16 MFMA x1024 | v_fma x4096 at s_setprio 3", d, dc);
    run<1000, 2, 2>("f16 MFMA x1024 at s_setprio 3 | v_fma x4096", d, dc);
    run<1008, 2, 1>("(MFMA, s_nop 7) x1024 | v_fma x4096 at s_setprio 3", d, dc);
    return 0;
}
