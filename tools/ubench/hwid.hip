// Where do the workgroups of a 1024 x 256 launch (4 resident per CU at 128 VGPRs) land?  HW_ID per wavefront: wave slot, SIMD, CU, SE, XCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ __launch_bounds__(256) void k(unsigned* out, float* sink) {
    __shared__ float pad[2048];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = pad[(threadIdx.x * 7) & 2047];
    for (int i = 0; i < 20000; ++i) a = a * 1.0000001f + 1e-9f;   // keep every workgroup resident while the others start
    if ((threadIdx.x & 63) == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // XCC_ID [3:0]
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (a == 123.0f) *sink = a;
}
int main() {
    const int G = 1024;
    unsigned* d; float* s;
    (void)hipMalloc(&d, G * 4 * 2 * sizeof(unsigned)); (void)hipMalloc(&s, 4);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, s);
    static unsigned h[G * 4 * 2];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::map<unsigned, int> slots;
    for (int w = 0; w < G * 4; ++w) slots[h[2 * w] & 15]++;
    printf("wave_id histogram:"); for (auto& kv : slots) printf(" %u:%d", kv.first, kv.second); printf("\n");
    for (int b : {0, 1, 2, 7, 8, 9, 255, 256, 257, 512, 768, 1023}) {
        printf("block %4d:", b);
        for (int w = 0; w < 4; ++w) { unsigned hw = h[2 * (b * 4 + w)]; printf("  [slot %u simd %u cu %u sh %u se %u xcc %u]", hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, h[2 * (b * 4 + w) + 1] & 15); }
        printf("\n");
    }
    return 0;
}
