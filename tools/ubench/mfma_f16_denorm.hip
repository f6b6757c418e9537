// Does gfx950's v_mfma_f32_16x16x32_f16 keep SUBNORMAL f16 operands, or flush them to zero?  (The split-precision critic scales the low halves of
// its operands by 2^11 so that nothing depends on the answer; if subnormals are kept, the unscaled form -- ONE accumulator, no "am += ac / S" pass --
// would be exact enough too: tests/test_split_model.py.)  A = 2^-20 (subnormal in f16: the smallest normal is 2^-14) in every element, B = 1.0:
// D = 32 * 2^-20 if kept, 0 if flushed.  Same for bf16 with A = 2^-130 (subnormal in bf16 / f32).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f16_denorm mfma_f16_denorm.hip && ./mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, float a16, float a_bf) {
    h8 a, b; b8 c, d;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a16; b[j] = (_Float16)1.0f; c[j] = (__bf16)a_bf; d[j] = (__bf16)1.0f; }
    f4 z = {0, 0, 0, 0};
    f4 r1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
    f4 r2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, z, 0, 0, 0);
    f4 r3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, d, z, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = r1[0]; out[1] = r2[0]; out[2] = r3[0]; out[3] = (float)a[0]; }
}
int main() {
    float *d; (void)hipMalloc(&d, 64);
    for (int e : {-15, -20, -24}) {
        const float a16 = ldexpf(1.0f, e), abf = ldexpf(1.0f, -130);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a16, abf);
        float h[4]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("f16 operand 2^%d (subnormal; as converted: %g): A-side D = %g, B-side D = %g (kept: %g)   bf16 operand 2^-130: D = %g (kept: %g)\n", e, h[3], h[0], h[1],
               32.0 * a16, h[2], 32.0 * abf);
    }
    return 0;
}
