// Why does moving a frame stack cost ~0.27 us per 512-byte load+store pair per wavefront in the latency regime?  512 workgroups x 64 lanes,
// every wavefront moves 52 x 512 B inside its own 33 KB range (word w <- word w + 13, eight-byte words), variants:
//   0 in place, 8-byte words, batches of 8     1 out of place (separate destination)     2 in place, batches of 26
//   3 out of place, 16-byte words (aligned copy, same bytes)                             4 loads only (sum kept)       5 stores only
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int V> __global__ __launch_bounds__(64) void k(float* buf, float* dst, int words_per_wave, float* sink) {
    const int lane = threadIdx.x;
    f2* src = reinterpret_cast<f2*>(buf) + (size_t)blockIdx.x * (words_per_wave + 64);
    f2* out = (V == 1 || V == 3) ? reinterpret_cast<f2*>(dst) + (size_t)blockIdx.x * (words_per_wave + 64) : src;
    float acc = 0.0f;
    if (V == 3) {
        const f4* s4 = reinterpret_cast<const f4*>(src); f4* o4 = reinterpret_cast<f4*>(out);
        for (int w0 = 0; w0 < words_per_wave / 2; w0 += 8 * 64) {
            f4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { int w = w0 + j * 64 + lane; if (w < words_per_wave / 2) v[j] = s4[w]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { int w = w0 + j * 64 + lane; if (w < words_per_wave / 2) o4[w] = v[j]; }
        }
    } else {
        constexpr int NB = V == 2 ? 26 : 8;
        for (int w0 = 0; w0 < words_per_wave; w0 += NB * 64) {
            f2 v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) { int w = w0 + j * 64 + lane; v[j] = f2{1.0f, 2.0f}; if (V != 5 && w < words_per_wave) v[j] = src[w + 13]; }
#pragma unroll
            for (int j = 0; j < NB; ++j) { int w = w0 + j * 64 + lane; if (V == 4) acc += v[j].x + v[j].y; else if (w < words_per_wave) out[w] = v[j]; }
        }
    }
    if (acc == 12345.0f) *sink = acc;
}
template <int V> void run(const char* name, float* a, float* b, float* s, int grid, int wpw) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(64), 0, 0, a, b, wpw, s);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(64), 0, 0, a, b, wpw, s);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("grid %5d  %-44s %7.2f us per launch  (%.2f TB/s moved)\n", grid, name, ms * 1e3 / reps, 2.0 * grid * wpw * 8.0 / (ms * 1e-3 / reps) / 1e12);
}
int main() {
    const int wpw = 64 * 52;  // 52 words per lane
    float *a, *b, *s;
    const size_t bytes = (size_t)4096 * (wpw + 64) * 8;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&s, 4);
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    for (int grid : {256, 512, 1024, 4096}) {
        run<0>("in place, 8 B words, 8 per batch", a, b, s, grid, wpw);
        run<1>("out of place, 8 B words, 8 per batch", a, b, s, grid, wpw);
        run<2>("in place, 8 B words, 26 per batch", a, b, s, grid, wpw);
        run<3>("out of place aligned copy, 16 B words", a, b, s, grid, wpw);
        run<4>("loads only", a, b, s, grid, wpw);
        run<5>("stores only", a, b, s, grid, wpw);
    }
    return 0;
}
