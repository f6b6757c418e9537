// Do f32 MFMAs and plain VALU work of the two wavefronts of one SIMD overlap on gfx950?  (the design question of taco_critic_lstm_kernel)
// One workgroup of 512 threads on every CU: wavefronts w and w + 4 share SIMD w.  Role X of waves 0-3 and role Y of waves 4-7:
//   M = 1024 v_mfma_f32_16x16x4_f32 on 4 independent accumulators, V = 4096 independent v_fma_f32, B = 1024 v_mfma_f32_32x32x16_bf16,
//   - = return at once.  Reported: shader-clock cycles of wave 0 and of wave 4 (workgroup 0).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ float work(float seed) {
    if constexpr (ROLE == 1) {
        f4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = seed, y = seed * 0.5f;
        for (int i = 0; i < 256; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
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
    } else if constexpr (ROLE == 5) {  // 4096 independent v_pk_fma_f32 (two f32 fma per lane each)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v[8];
        for (int j = 0; j < 8; ++j) v[j] = f2{seed + j, seed - j};
        const f2 sd = {seed, seed};
        for (int i = 0; i < 512; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(sd));
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
        return s;
    } else if constexpr (ROLE == 6) {  // 4096 v_exp_f32 (transcendental pipe)
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = seed + j;
        for (int i = 0; i < 512; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j];
        return s;
    } else if constexpr (ROLE == 3) {
        f16v a0 = {}, a1 = {};
        bf8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (__bf16)seed; y[j] = (__bf16)(seed * 0.5f); }
        for (int i = 0; i < 512; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        }
        return a0[0] + a1[5];
    }
    else if constexpr (ROLE >= 30) {  // in-wave interleave with the bf16 MFMA: one v_mfma_f32_32x32x16_bf16, then ROLE - 30 v_fma, 1024 times
        f16v a0 = {}, a1 = {};
        bf8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (__bf16)seed; y[j] = (__bf16)(seed * 0.5f); }
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = seed + j;
        constexpr int NV = ROLE - 30;
#define FILLB(a) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a, 0, 0, 0); _Pragma("unroll") for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(seed));
        for (int i = 0; i < 512; ++i) { FILLB(a0) FILLB(a1) }
        float s = a0[0] + a1[5];
        for (int j = 0; j < 8; ++j) s += v[j];
        return s;
    }
    else if constexpr (ROLE >= 10) {  // in-wave interleave: one f32 MFMA, then ROLE - 10 independent v_fma, 1024 times
        f4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = seed, y = seed * 0.5f;
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = seed + j;
        constexpr int NV = ROLE - 10;
#define FILL(a) a = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a, 0, 0, 0); _Pragma("unroll") for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(seed));
        for (int i = 0; i < 256; ++i) { FILL(a0) FILL(a1) FILL(a2) FILL(a3) }
        float s = a0[0] + a1[1] + a2[2] + a3[3];
        for (int j = 0; j < 8; ++j) s += v[j];
        return s;
    }
    return 0.0f;
}

template <int X, int Y>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    float r = wave < 4 ? work<X>(seed) : work<Y>(seed);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

template <int X, int Y>
void run(const char *name, float *d, unsigned long long *dc) {
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<X, Y>), dim3(256), dim3(512), 0, 0, d, dc, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s wave 0: %8llu cycles   wave 4: %8llu cycles\n", name, h[0], h[1]);
}

int main() {
    float *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    hipMalloc(&dc, 2 * sizeof(unsigned long long));
    run<1, 0>("f32 MFMA x1024 | -", d, dc);
    run<0, 2>("- | v_fma x4096", d, dc);
    run<1, 2>("f32 MFMA x1024 | v_fma x4096", d, dc);
    run<1, 1>("f32 MFMA x1024 | f32 MFMA x1024", d, dc);
    run<2, 2>("v_fma x4096 | v_fma x4096", d, dc);
    run<3, 0>("bf16 MFMA 32x32x16 x1024 | -", d, dc);
    run<3, 2>("bf16 MFMA 32x32x16 x1024 | v_fma x4096", d, dc);
    run<14, 0>("(MFMA + 4 v_fma) x1024 | -", d, dc);
    run<15, 0>("(MFMA + 5 v_fma) x1024 | -", d, dc);
    run<16, 0>("(MFMA + 6 v_fma) x1024 | -", d, dc);
    run<18, 0>("(MFMA + 8 v_fma) x1024 | -", d, dc);
    run<2, 1>("v_fma x4096 | f32 MFMA x1024", d, dc);
    run<5, 0>("v_pk_fma_f32 x4096 | -", d, dc);
    run<5, 5>("v_pk_fma_f32 x4096 | same", d, dc);
    run<6, 0>("v_exp_f32 x4096 | -", d, dc);
    run<1, 6>("f32 MFMA x1024 | v_exp_f32 x4096", d, dc);
    run<34, 0>("(bf16 MFMA + 4 v_fma) x1024 | -", d, dc);
    run<36, 0>("(bf16 MFMA + 6 v_fma) x1024 | -", d, dc);
    run<14, 14>("(MFMA + 4 v_fma) x1024 | same", d, dc);
    run<16, 16>("(MFMA + 6 v_fma) x1024 | same", d, dc);
    return 0;
}
