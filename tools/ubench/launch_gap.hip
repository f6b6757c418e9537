// Floor of a dependent kernel launch: K back-to-back launches of (a) an empty kernel, (b) a kernel of 256 x 256 threads that only
// reads the shader clock, timed by events on one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k() {}
__global__ void tiny_k(unsigned long long *out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = __builtin_readcyclecounter(); }
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int K = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0);
        (void)hipEventRecord(e0);
        for (int k = 0; k < K; ++k) {
            if (variant == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0);
            if (variant == 1) hipLaunchKernelGGL(tiny_k, dim3(256), dim3(256), 0, 0, d);
            if (variant == 2) hipLaunchKernelGGL(tiny_k, dim3(256), dim3(64), 0, 0, d);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("variant %d (%s): %.2f us per dependent launch\n", variant, variant == 0 ? "empty 1x64" : variant == 1 ? "tiny 256x256" : "tiny 256x64", ms * 1e3 / K);
    }
    return 0;
}
