// taco_rollout.hpp -- SURVEY 8f row N1: the part of PPOReplayBuffer (IsaacGymEnvs/algorithms/buffer_asymmetry.py) that sits
// directly behind step(): GAE(lambda) + returns (:93-130) and the advantage normalisation (:132).  All arrays are [H][N]
// fp32 (the reference's [H, N, 1]), env index fastest => every access of the per-env recurrence is coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace taco {

constexpr int kGaeBlock = 64;
constexpr int kNormBlocks = 256;   // partial sums: [2][kNormBlocks] doubles in the caller's workspace
constexpr int kNormThreads = 256;

// one lane per env, walking the horizon backwards; the loads do not depend on the recurrence, so the compiler keeps
// several steps of them in flight (unroll 4).  20 B of HBM traffic per (step, env).
__global__ __launch_bounds__(kGaeBlock) void gae_kernel(const float *__restrict__ rew, const float *__restrict__ done,
                                                        const float *__restrict__ value, const float *__restrict__ last_value,
                                                        int H, int N, float g, float l, float *__restrict__ adv, float *__restrict__ ret) {
    const int i = blockIdx.x * kGaeBlock + threadIdx.x;
    if (i >= N) return;
    float last = 0.0f;
    float nv = last_value[i];
#pragma unroll 4
    for (int t = H - 1; t >= 0; --t) {
        const size_t k = (size_t)t * N + i;
        const float v = value[k];
        const float nnt = 1.0f - done[k];
        const float td = rew[k] + nnt * g * nv;      // :122  rew + ((nnt * gamma) * next_values)
        const float delta = td - v;                  // :123
        last = delta + nnt * g * l * last;           // :124  delta + (((nnt * gamma) * lam) * last)
        adv[k] = last;
        ret[k] = last + v;                           // :130
        nv = v;
    }
}

// rewards_augmented[truncated] += gamma * V(s_t)  (ppo_asymmetry.py:314-324): the value of the pre-step state is value_buf[t] itself
__global__ __launch_bounds__(256) void timeout_bootstrap_kernel(float *__restrict__ rew, const float *__restrict__ value, const float *__restrict__ done,
                                                                const uint8_t *__restrict__ timeout, size_t count, float g) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < count && timeout[k] != 0 && done[k] != 0.0f) rew[k] = rew[k] + g * value[k];
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}
// fixed-shape block sum (256 threads = 4 waves): identical result for identical inputs, whatever the scheduling
__device__ __forceinline__ double block_sum(double x, double *sh) {
    x = wave_sum(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// pass 1: per-block partial sum and sum of squares, in double (each block owns a fixed slice => deterministic)
__global__ __launch_bounds__(kNormThreads) void adv_partial_kernel(const float *__restrict__ adv, size_t count, double *__restrict__ part) {
    __shared__ double sh[4];
    double s = 0.0, ss = 0.0;
    for (size_t k = (size_t)blockIdx.x * kNormThreads + threadIdx.x; k < count; k += (size_t)kNormBlocks * kNormThreads) {
        const double x = (double)adv[k];
        s += x;
        ss += x * x;
    }
    s = block_sum(s, sh);
    ss = block_sum(ss, sh);
    if (threadIdx.x == 0) { part[blockIdx.x] = s; part[kNormBlocks + blockIdx.x] = ss; }
}

// pass 2: every block re-reduces the 256 partials in the same order, then (adv - mean) / (std + 1e-8), std unbiased (:132)
__global__ __launch_bounds__(kNormThreads) void adv_apply_kernel(float *__restrict__ adv, size_t count, const double *__restrict__ part) {
    __shared__ double sh[4];
    const double s = block_sum(part[threadIdx.x], sh);
    const double ss = block_sum(part[kNormBlocks + threadIdx.x], sh);
    const double n = (double)count;
    const double mean = s / n;
    double var = (ss - s * mean) / (n - 1.0);
    var = var > 0.0 ? var : 0.0;
    const float m = (float)mean, sd = (float)sqrt(var);
    const float den = sd + 1e-8f;
    for (size_t k = (size_t)blockIdx.x * kNormThreads + threadIdx.x; k < count; k += (size_t)gridDim.x * kNormThreads) adv[k] = (adv[k] - m) / den;
}

}  // namespace taco
