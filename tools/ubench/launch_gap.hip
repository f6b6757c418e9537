// Floor of a dependent kernel launch: K back-to-back launches of (a) an empty kernel, (b) a kernel of 256 x 256 threads that only
// reads the shader clock, timed by events on one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k() {}
__global__ void tiny_k(unsigned long long *out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = __builtin_readcyclecounter(); }
// a kernel with the step kernel's memory footprint and nothing else: every env's lane reads 13 float4 rows at its start and writes 13 at its end
__global__ void state_k(float4 *S, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / 4;
    if (i >= n) return;
    float4 v[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) v[c] = S[(size_t)c * n + i];
#pragma unroll
    for (int c = 0; c < 13; ++c) { v[c].x += 1.0f; if ((threadIdx.x & 3) == 0) S[(size_t)c * n + i] = v[c]; }
}
template <int MODE>  // 0: read only (sum to one word per wave), 1: write only
__global__ void state_rw(float4 *S, float *out, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / 4;
    if (i >= n) return;
    if (MODE == 0) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 13; ++c) acc += S[(size_t)c * n + i].x;
        if (acc == 123.456f) out[0] = acc;
    } else {
#pragma unroll
        for (int c = 0; c < 13; ++c) if ((threadIdx.x & 3) == 0) S[(size_t)c * n + i] = make_float4((float)c, 1.0f, 2.0f, 3.0f);
    }
}
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int K = 2000;
    float4 *S; (void)hipMalloc(&S, (size_t)13 * 4096 * 16); (void)hipMemset(S, 0, (size_t)13 * 4096 * 16);
    float *fo; (void)hipMalloc(&fo, 64);
    for (int variant = 0; variant < 7; ++variant) {
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0);
        (void)hipEventRecord(e0);
        for (int k = 0; k < K; ++k) {
            if (variant == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0);
            if (variant == 1) hipLaunchKernelGGL(tiny_k, dim3(256), dim3(256), 0, 0, d);
            if (variant == 2) hipLaunchKernelGGL(tiny_k, dim3(256), dim3(64), 0, 0, d);
            if (variant == 3) hipLaunchKernelGGL(state_k, dim3(256), dim3(64), 0, 0, S, 4096);
            if (variant == 4) hipLaunchKernelGGL(state_k, dim3(64), dim3(256), 0, 0, S, 4096);
            if (variant == 5) hipLaunchKernelGGL(state_rw<0>, dim3(256), dim3(64), 0, 0, S, fo, 4096);
            if (variant == 6) hipLaunchKernelGGL(state_rw<1>, dim3(256), dim3(64), 0, 0, S, fo, 4096);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("variant %d (%s): %.2f us per dependent launch\n", variant, variant == 0 ? "empty 1x64" : variant == 1 ? "tiny 256x256" : variant == 2 ? "tiny 256x64" : variant == 3 ? "state r/w 256x64" : variant == 4 ? "state r/w 64x256" : variant == 5 ? "state read-only 256x64" : "state write-only 256x64", ms * 1e3 / K);
    }
    return 0;
}
