// Does a wavefront with only 16 or 32 of its 64 lanes active issue VALU instructions faster?  (If it did, the 4 096-env launch could
// spread its 16 384 quad lanes over 1 024 partial wavefronts, one per SIMD.)  Dependent and independent FMA streams, lone wavefront per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEP>
__global__ void k(float* out, int iters, int active) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float m = 1.0000001f, c = 1e-9f;
    if ((int)(threadIdx.x & 63) < active) {
        for (int i = 0; i < iters; ++i) {
            if (DEP)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(m), "v"(c));
            else
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int DEP> void run(const char* name, float* d, int blocks, int threads, int active) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(threads), 0, 0, d, 100, active);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(threads), 0, 0, d, iters, active);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s blocks=%4d threads=%3d active=%2d : %.3f ms -> %.2f ns per wave-instruction\n", name, blocks, threads, active, ms, ms * 1e6 / (8.0 * iters));
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 256 * sizeof(float));
    for (int act : {64, 32, 16, 8, 1}) { run<0>("independent", d, 256, 64, act); run<1>("dependent", d, 256, 64, act); }
    for (int thr : {32, 16}) { run<0>("independent", d, 256, thr, 64); run<1>("dependent", d, 256, thr, 64); }
    return 0;
}
