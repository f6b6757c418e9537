// One wavefront per CU on 64 CUs (the 4 096-env launch geometry): issue rate of a lone wavefront for
// independent / dependent FMA streams, and of two or four wavefronts sharing one SIMD (256-thread blocks put one wave on each SIMD;
// 64-thread blocks x k on the same CU land on different SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEP>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
        if (DEP)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(m), "v"(c));
        else
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int DEP> void run(const char* name, float* d, int blocks, int threads) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s blocks=%4d threads=%3d : %.3f ms -> %.2f ns per wave-instruction\n", name, blocks, threads, ms, ms * 1e6 / (8.0 * iters));
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 256 * sizeof(float));
    for (int b : {1, 64, 256, 1024}) { run<0>("independent", d, b, 64); run<1>("dependent", d, b, 64); }
    run<0>("independent", d, 64, 256); run<0>("independent", d, 64, 512); run<0>("independent", d, 64, 1024);
    return 0;
}
