#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float qp(float x, int ctrl_unused);
template <int CTRL> __device__ __forceinline__ float dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// quad_perm helpers: ctrl = p0 | p1<<2 | p2<<4 | p3<<6
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, unsigned long long *cyc) {
    float x = threadIdx.x * 1e-3f + 1.0f, y = x + 0.5f, z = x * 0.25f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) { x = x * 0.999f + dpp<QP(1, 2, 0, 3)>(x); }                       // dependent: mul, add(dpp of the fresh value)
            if (MODE == 1) { x = x * 0.999f + dpp<QP(1, 2, 0, 3)>(y); y = y * 0.999f + dpp<QP(2, 0, 1, 3)>(x); }  // two chains feeding each other
            if (MODE == 2) { float a = dpp<QP(1, 2, 0, 3)>(x), b = dpp<QP(2, 0, 1, 3)>(x), c = dpp<QP(1, 2, 0, 3)>(y), d = dpp<QP(2, 0, 1, 3)>(y);
                             float cr = __builtin_fmaf(a, d, -(b * c)); x = x + 1e-3f * cr; y = y - 1e-3f * cr; }  // cross-product shape
            if (MODE == 3) { x = x * 0.999f + y; y = y * 0.999f + x; }  // no dpp reference
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x + y + z;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(float *d, unsigned long long *dc, const char *name, int ops) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(64), dim3(64), 0, 0, d, 10, dc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(64), dim3(64), 0, 0, d, iters, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms  %.2f ns per unrolled body (%d source ops)\n", name, ms, ms * 1e6 / (iters * 16.0), ops);
}
int main() {
    float *d; unsigned long long *dc; (void)hipMalloc(&d, 64 * 64 * 4); (void)hipMalloc(&dc, 8);
    run<0>(d, dc, "dependent mul + add_dpp(fresh)", 2);
    run<1>(d, dc, "two chains cross-feeding via dpp", 4);
    run<2>(d, dc, "cross-product shape (4 dpp + 6 alu)", 10);
    run<3>(d, dc, "no dpp, 2 mul + 2 add", 4);
    return 0;
}
