// Do power-of-two tile strides alias HBM channels on the MI355X?  4 096 wavefronts each read (or write) ONE 1 KB row (16 B per lane) of their own tile,
// tile stride S bytes, the same row index in every tile -- the access pattern of the step kernel's action-history rows (16 KB per tile) and state chunks
// (13 KB per tile).  Rows per tile are walked so that every launch touches fresh lines (cold in L2: the buffer is larger than the caches).
//   hipcc --offload-arch=gfx950 -O2 -o tile_stride tile_stride.hip && ./tile_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void rd(const float4 *buf, size_t stride16, int row, float *sink) {
    const float4 v = buf[(size_t)blockIdx.x * stride16 + (size_t)row * 64 + threadIdx.x];
    if (v.x == 123.456f) *sink = v.y;
}
__global__ void wr(float4 *buf, size_t stride16, int row) {
    buf[(size_t)blockIdx.x * stride16 + (size_t)row * 64 + threadIdx.x] = float4{1.0f, 2.0f, 3.0f, (float)row};
}
int main() {
    const int tiles = 4096;
    float4 *buf; float *sink;
    const size_t max_stride = 40 * 1024;
    (void)hipMalloc(&buf, (size_t)tiles * max_stride + (1 << 20)); (void)hipMalloc(&sink, 4);
    (void)hipMemset(buf, 0, (size_t)tiles * max_stride);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (size_t stride : {13 * 1024, 16 * 1024, 17 * 1024, 32 * 1024, 33 * 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<float> us;
            for (int rep = 0; rep < 9; ++rep) {
                (void)hipEventRecord(e0);
                for (int it = 0; it < 12; ++it) {   // rows 0..11 of every tile: 12 launches x 4 MB
                    if (mode == 0) hipLaunchKernelGGL(rd, dim3(tiles), dim3(64), 0, 0, buf, stride / 16, it, sink);
                    else hipLaunchKernelGGL(wr, dim3(tiles), dim3(64), 0, 0, buf, stride / 16, it);
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                us.push_back(ms * 1e3f / 12);
            }
            std::sort(us.begin(), us.end());
            printf("tile stride %2zu KB, %s one 1 KB row of each of 4096 tiles: %.2f us per launch (median of 9 x 12)\n", stride / 1024, mode ? "write" : "read ", us[4]);
        }
    }
    return 0;
}
