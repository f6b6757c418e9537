#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../taco_amd/csrc/taco_math.hpp"
#include "../../taco_amd/csrc/taco_step.hpp"
using namespace taco;
__global__ void k(float* in, float* out, int n) {
    rsrc_t ri = make_rsrc(in, n * 16), ro = make_rsrc(out, n * 16 * 2);
    uint32_t voff = threadIdx.x * 16;
    float4 a = buf_ld4(ri, voff, 0);
    buf_st4(ro, make_float4(a.x + 1, a.y + 2, a.z + 3, a.w + 4), voff, 0);
    buf_st4(ro, a, voff, n * 16);
}
int main() {
    const int n = 64; float h[n * 4], o[n * 8]; for (int i = 0; i < n * 4; ++i) h[i] = i;
    float *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, n); hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%g ", o[i]); printf("| "); for (int i = 0; i < 8; ++i) printf("%g ", o[n * 4 + i]); printf("\n");
}
