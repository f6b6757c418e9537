// Floor of a dependent kernel launch vs launch SHAPE: K back-to-back launches of a kernel that only reads the shader clock, for several
// (workgroups, threads per workgroup, LDS per workgroup) combinations, timed by events on one stream.  Question: would the 4 096-env launch
// (256 workgroups x 256 threads, 9 KB of LDS) get over its ~3 us launch boundary faster as fewer, larger workgroups?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_WORDS>
__global__ void tiny_k(unsigned long long *out) {
    __shared__ float pad[LDS_WORDS > 0 ? LDS_WORDS : 1];
    if (LDS_WORDS > 0) pad[threadIdx.x % LDS_WORDS] = 1.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = __builtin_readcyclecounter() + (LDS_WORDS > 0 ? (unsigned long long)pad[0] : 0ull);
}
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int K = 3000;
    struct { int g, b, lds; } shapes[] = {{1, 64, 0}, {256, 64, 0}, {256, 256, 0}, {256, 256, 1}, {128, 512, 0}, {128, 512, 1}, {64, 1024, 0}, {128, 256, 0}, {64, 256, 0},
                                          {512, 128, 0}, {1024, 64, 0}, {1024, 256, 0}, {4096, 64, 0}};
    for (auto s : shapes) {
        for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(tiny_k<0>, dim3(1), dim3(64), 0, 0, d);
        (void)hipEventRecord(e0);
        for (int k = 0; k < K; ++k) {
            if (s.lds) hipLaunchKernelGGL(tiny_k<2400>, dim3(s.g), dim3(s.b), 0, 0, d);
            else hipLaunchKernelGGL(tiny_k<0>, dim3(s.g), dim3(s.b), 0, 0, d);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%5d workgroups x %4d threads%s: %.2f us per dependent launch\n", s.g, s.b, s.lds ? " + 9.6 KB LDS" : "", ms * 1e3 / K);
    }
    return 0;
}
