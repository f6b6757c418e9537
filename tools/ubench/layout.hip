// HBM rate of the step kernel's ACCESS PATTERN alone (no arithmetic), by state layout:
//   rows  : chunk c of env i at (c * npad + i) * 16            (the layout of round 1: 17+ streams, each 16 MB long at 1 M envs)
//   tiles : chunk c of env i at ((i / 64) * C + c) * 1024 + (i % 64) * 16   (one contiguous C-KiB block per wavefront)
// Every lane reads RD float4 chunks + 104-byte frame rows are written twice (obs, states) + WR chunks written back: the step's 329 B read /
// 381 B written per env.  One wavefront per 64 envs, 256-thread blocks, plain global_load/store_dwordx4.
//   hipcc --offload-arch=gfx950 -O2 -o layout layout.hip && ./layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int C = 33;   // 17 state chunks + 16 history rows
constexpr int RD = 19;  // 13 state + 2 target + 4 history rows
constexpr int WR = 10;  // 9 state + 1 history row

template <bool TILES, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(float4 *S, const float4 *act, float4 *obs, float4 *states, int n, int npad) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    auto at = [&](int c) -> size_t { return TILES ? ((size_t)(i >> 6) * C + c) * 64 + (i & 63) : (size_t)c * npad + i; };
    float4 v[RD];
    const float4 a = act[i];
#pragma unroll
    for (int c = 0; c < RD; ++c) v[c] = S[at(c)];
    float4 acc = a;
#pragma unroll
    for (int c = 0; c < RD; ++c) { acc.x += v[c].x; acc.y += v[c].y; acc.z += v[c].z; acc.w += v[c].w; }
#pragma unroll
    for (int c = 0; c < WR; ++c) S[at(c)] = make_float4(acc.x + c, acc.y, acc.z, acc.w);
    // frames: the wavefront's 64 rows of 104 B are one contiguous 6 656-byte range = 6.5 x 1 KiB stores
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)(i >> 6) * 416;  // float4 words per wavefront
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        const int w = it * 64 + lane;
        if (w < 416) { obs[base + w] = acc; states[base + w] = acc; }
    }
}

template <bool TILES, int BLOCK> void run(const char *name, int n, int pad = 0) {
    const int npad = (n + 63) / 64 * 64 + pad;
    float4 *S, *act, *obs, *st;
    (void)hipMalloc(&S, (size_t)C * npad * 16); (void)hipMalloc(&act, (size_t)npad * 16); (void)hipMalloc(&obs, (size_t)npad * 104); (void)hipMalloc(&st, (size_t)npad * 104);
    (void)hipMemset(S, 0, (size_t)C * npad * 16); (void)hipMemset(act, 0, (size_t)npad * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<TILES, BLOCK>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, S, act, obs, st, n, npad);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<TILES, BLOCK>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, S, act, obs, st, n, npad);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)n * (RD * 16 + 16 + WR * 16 + 208);
    printf("%-8s block=%3d n=%8d pad=%5d : %8.2f us  %6.2f TB/s (%.0f B per env)\n", name, BLOCK, n, pad, us, bytes / us / 1e6, bytes / n);
    (void)hipFree(S); (void)hipFree(act); (void)hipFree(obs); (void)hipFree(st);
}
int main() {
    for (int n : {262144, 1048576, 2000000}) {
        run<false, 256>("rows", n); run<true, 256>("tiles", n);
        run<false, 64>("rows", n); run<true, 64>("tiles", n);
    }
    for (int n : {262144, 1048576, 1000000, 524288})
        for (int pad : {0, 64, 128, 192, 256, 512, 1024, 1088, 2048, 4096, 4160, 8192, 16448, 65600}) run<false, 256>("rows", n, pad);
    return 0;
}
