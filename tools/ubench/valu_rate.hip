// Micro-benchmark: issue rate of plain vs packed fp32 FMA on gfx950 at 1/2/4/8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    if (PK == 0) {
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    } else if (PK == 1) {
        float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
        }
        a0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    } else {  // dependent chain
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int PK> void run(const char* name, float* d) {
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256-thread blocks = 4 waves = 1 per SIMD; wps blocks per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(256), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = 8.0 * iters;  // instructions per wave
        printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr  (per-SIMD: %.2f ns/instr)\n", name, wps, ms, ms * 1e6 / inst, ms * 1e6 / inst / wps);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32 indep x8", d);
    run<1>("v_pk_fma_f32 indep x4", d);
    run<2>("v_fma_f32 dependent", d);
    return 0;
}
