// taco_policy.hpp -- SURVEY 8f row N1, second half: PPO_ActorCritic.act (IsaacGymEnvs/algorithms/nets_asymmetry.py:326-355) as ONE launch.
//
// grid = (ceil(N / 16), 2): blockIdx.y == 0 runs the actor (MLP on the observation stack, tanh head, action sampling + log-prob),
// blockIdx.y == 1 the critic (1-layer LSTM over the state stack, then the MLP).  A workgroup (8 wavefronts) owns 16 envs = the M
// dimension of v_mfma_f32_16x16x4_f32; its activations live in LDS, the weights stream from L2 (every workgroup reads the same
// packed blob, see taco_amd/policy.py::pack_state_dict), each wavefront owns output tiles tile = wave, wave + 8, ...
// Operand fragments are 16-byte loads: lane (r = lane & 15, g = lane >> 4) reads A[r][16 s + 4 g .. +3] from LDS and
// W[col][16 s + 4 g .. +3] from global (stored fragment-major, so a wavefront's load is 1 KiB contiguous) and feeds component t to
// the t-th MFMA of the block, so the accumulator is the f32 fmaf chain
//     acc = bias;  for s, t, g:  k = 16 s + 4 g + t;  acc = fma(x[k], W[col][k], acc)
// (an f32 MFMA is a k-ordered fmaf chain with one rounding per product): a definition the CPU test oracle can follow bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "taco_math.hpp"

namespace taco {

constexpr int POL_ROWS = 16;
constexpr int POL_NW = 8;             // wavefronts per workgroup: two per SIMD, so one's operand loads hide under the other's MFMAs
constexpr int POL_MAXW = 256;
constexpr int POL_LD = POL_MAXW + 4;  // LDS row stride in floats: 1040 B, so 8 consecutive rows cover all 32 banks
constexpr int POL_MAXT = 8;           // longest state stack the LSTM tile holds
constexpr int POL_XLD = 36;           // row stride of the per-timestep input tile (states_dim padded to 32, + 4)
enum : uint32_t { STREAM_POLICY = 7 };

struct PolicyParams {
    int obs_len, obs_dim, states_len, states_dim, act_dim;
    int n_actor_hidden, actor_hidden[4];
    int lstm_hidden;
    int n_critic_hidden, critic_hidden[4];
    const float *blob, *obs, *states;
    float *action, *logp, *value, *mu, *sigma;
    int n, deterministic;
    uint32_t seed_lo, seed_hi, call;
};

typedef float pf32x4 __attribute__((ext_vector_type(4)));
TD int pad16(int x) { return (x + 15) / 16 * 16; }

enum { PACT_RELU = 0, PACT_TANH = 1, PACT_NONE = 2 };

// out[16][outp] = act(in[16][inp] W^T + b); two output tiles per wavefront in flight (independent accumulators hide the 40-cycle
// dependent-MFMA latency)
TD void dense_layer(const float *in, float *out, const float *W, const float *b, int inp, int outp, int act, int wave, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = outp >> 4;
    for (int tile = wave; tile < ntiles; tile += 2 * POL_NW) {
        const int tile2 = tile + POL_NW;
        const bool two = tile2 < ntiles;  // wave-uniform
        const int col0 = tile * 16 + r, col1 = (two ? tile2 : tile) * 16 + r;
        const float b0 = b[col0], b1 = b[col1];
        pf32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
        // fragment-major weights: block s of tile T starts at ((T * ks + s) * 64 + lane) * 4 floats
        const float *w0 = W + ((size_t)tile * (inp >> 4) * 64 + lane) * 4, *w1 = W + ((size_t)(two ? tile2 : tile) * (inp >> 4) * 64 + lane) * 4;
        const float *arow = in + r * POL_LD + 4 * g;
        const int ks = inp >> 4;
        float4 a4 = *reinterpret_cast<const float4 *>(arow), x0 = *reinterpret_cast<const float4 *>(w0), x1 = *reinterpret_cast<const float4 *>(w1);
        for (int s = 0; s < ks; ++s) {
            float4 an = a4, y0 = x0, y1 = x1;
            if (s + 1 < ks) {  // next block's fragments are in flight while this block's MFMAs run
                an = *reinterpret_cast<const float4 *>(arow + 16 * (s + 1));
                y0 = *reinterpret_cast<const float4 *>(w0 + 256 * (s + 1));
                if (two) y1 = *reinterpret_cast<const float4 *>(w1 + 256 * (s + 1));
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, x0.x, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, x1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, x0.y, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, x1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, x0.z, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, x1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, x0.w, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, x1.w, acc1, 0, 0, 0);
            a4 = an; x0 = y0; x1 = y1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // C/D layout: col = lane & 15, row = 4 (lane >> 4) + i
            float v0 = acc0[i], v1 = acc1[i];
            if (act == PACT_RELU) { v0 = (v0 < 0.0f) ? 0.0f : v0; v1 = (v1 < 0.0f) ? 0.0f : v1; }
            else if (act == PACT_TANH) { v0 = tanh_own(v0); v1 = tanh_own(v1); }
            out[(4 * g + i) * POL_LD + col0] = v0;
            if (two) out[(4 * g + i) * POL_LD + col1] = v1;
        }
    }
}

__global__ __launch_bounds__(64 * POL_NW) void taco_policy_kernel(const PolicyParams P) {
    __shared__ __attribute__((aligned(16))) float bufA[POL_ROWS * POL_LD];
    __shared__ __attribute__((aligned(16))) float bufB[POL_ROWS * POL_LD];
    __shared__ __attribute__((aligned(16))) float xs[POL_MAXT * POL_ROWS * POL_XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * POL_ROWS;
    const float *w = P.blob;
    float *x = bufA, *y = bufB;

    // ---- where the actor's parameters end (both roles walk the same blob)
    int in_a = pad16(P.obs_len * P.obs_dim);
    if (blockIdx.y == 0) {
        // ------------------------------------------------------------------ actor: MLP + tanh head (nets_asymmetry.py:331-332)
        const int kin = P.obs_len * P.obs_dim;
        for (int e = tid; e < POL_ROWS * in_a; e += 64 * POL_NW) {
            const int r = e / in_a, k = e - r * in_a;
            const int env = row0 + r;
            x[r * POL_LD + k] = (env < P.n && k < kin) ? P.obs[(size_t)env * kin + k] : 0.0f;
        }
        __syncthreads();
        int in = in_a;
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const bool last = l == P.n_actor_hidden;
            const int out = pad16(last ? P.act_dim : P.actor_hidden[l]);
            dense_layer(x, y, w, w + (size_t)out * in, in, out, last ? PACT_TANH : PACT_RELU, wave, lane);
            __syncthreads();
            w += (size_t)out * in + out;
            in = out;
            float *t = x; x = y; y = t;
        }
        const float *log_std = w;
        // ---- distribution (:333-345): scale_tril = diag(exp(log_std) * exp(log_std)); one thread per env
        if (tid < POL_ROWS && row0 + tid < P.n) {
            const int env = row0 + tid;
            float lp = 0.0f, half_log_det = 0.0f;
            U4 rnd{0u, 0u, 0u, 0u};
            for (int a = 0; a < P.act_dim; ++a) {
                const float mean = x[tid * POL_LD + a];
                const float e = expf_own(log_std[a]);
                const float scale = e * e;
                float act = mean;
                if (!P.deterministic) {
                    if ((a & 3) == 0) rnd = philox(P.seed_lo, P.seed_hi, (uint32_t)env, P.call, STREAM_POLICY, (uint32_t)(a >> 2));
                    // two Box-Muller pairs per Philox block, as the env's observation noise draws them
                    const uint32_t ba = (a & 2) ? rnd.z : rnd.x, bb = (a & 2) ? rnd.w : rnd.y;
                    const float ua = 1.0f - uniform(ba), ub = uniform(bb);
                    const float rad = __builtin_sqrtf(-2.0f * log(ua));
                    float sn, cs;
                    sincos(kTwoPi * ub, sn, cs);
                    const float eps = (a & 1) ? rad * sn : rad * cs;
                    act = mean + scale * eps;
                }
                const float zz = (act - mean) / scale;
                lp = lp + zz * zz;
                half_log_det = half_log_det + log(scale);
                P.action[(size_t)env * P.act_dim + a] = act;
                P.mu[(size_t)env * P.act_dim + a] = mean;
                P.sigma[(size_t)env * P.act_dim + a] = log_std[a];
            }
            P.logp[env] = -0.5f * ((float)P.act_dim * 1.8378770664093453f + lp) - half_log_det;
        }
        return;
    }

    // ------------------------------------------------------------------ critic (:348-352): skip the actor's parameters
    for (int l = 0; l <= P.n_actor_hidden; ++l) {
        const int out = pad16(l == P.n_actor_hidden ? P.act_dim : P.actor_hidden[l]);
        w += (size_t)out * in_a + out;
        in_a = out;
    }
    w += 16;  // log_std
    int in;
    if (P.lstm_hidden > 0) {
        // LSTMEncoder (:128-136): h_T of a 1-layer LSTM over the state stack, zero initial state
        const int hp = pad16(P.lstm_hidden), ip = pad16(P.states_dim), T = P.states_len;
        const float *Wih = w, *Whh = w + (size_t)4 * hp * ip, *bs = Whh + (size_t)4 * hp * hp;
        w = bs + (size_t)4 * hp;
        for (int e = tid; e < T * POL_ROWS * ip; e += 64 * POL_NW) {
            const int t = e / (POL_ROWS * ip), rem = e - t * POL_ROWS * ip, r = rem / ip, k = rem - r * ip;
            const int env = row0 + r;
            xs[(t * POL_ROWS + r) * POL_XLD + k] = (env < P.n && k < P.states_dim) ? P.states[((size_t)env * T + t) * P.states_dim + k] : 0.0f;
        }
        for (int e = tid; e < POL_ROWS * hp; e += 64 * POL_NW) x[(e / hp) * POL_LD + (e % hp)] = 0.0f;
        __syncthreads();
        const int r = lane & 15, g = lane >> 4;
        const int ntiles = hp >> 4;
        constexpr int TPW = (POL_MAXW / 16 + POL_NW - 1) / POL_NW;  // hidden tiles per wavefront at the widest LSTM
        float cst[TPW][4];  // cell state of this wavefront's hidden tiles (tile = wave + POL_NW j), rows 4 g + i
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[j][i] = 0.0f;
        // four gate accumulators over one operand (x_t or h_{t-1}); the next block's fragments load while this block's 16 MFMAs run
        // (a deeper pipeline does not help: the operand stream, 32 B/clk per CU out of L2, is the limit, not its latency)
        auto gate_chain = [&](pf32x4 (&acc)[4], const float *arow, const float *wcol, size_t gate_stride, int ks) {
            float4 a4 = *reinterpret_cast<const float4 *>(arow), wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const float4 *>(wcol + q * gate_stride);
            for (int s = 0; s < ks; ++s) {
                float4 an = a4, wn[4] = {wq[0], wq[1], wq[2], wq[3]};
                if (s + 1 < ks) {
                    an = *reinterpret_cast<const float4 *>(arow + 16 * (s + 1));
#pragma unroll
                    for (int q = 0; q < 4; ++q) wn[q] = *reinterpret_cast<const float4 *>(wcol + q * gate_stride + 256 * (s + 1));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wq[q].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wq[q].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wq[q].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wq[q].w, acc[q], 0, 0, 0);
                a4 = an;
#pragma unroll
                for (int q = 0; q < 4; ++q) wq[q] = wn[q];
            }
        };
        for (int t = 0; t < T; ++t) {
            const float *xt = xs + (size_t)t * POL_ROWS * POL_XLD + r * POL_XLD + 4 * g;
            const float *hrow = x + r * POL_LD + 4 * g;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int tile = wave + POL_NW * j;
                if (tile < ntiles) {  // wave-uniform
                    const int col = tile * 16 + r;
                    pf32x4 acc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float b = bs[q * hp + col]; acc[q] = pf32x4{b, b, b, b}; }
                    gate_chain(acc, xt, Wih + ((size_t)tile * (ip >> 4) * 64 + lane) * 4, (size_t)hp * ip, ip >> 4);
                    gate_chain(acc, hrow, Whh + ((size_t)tile * (hp >> 4) * 64 + lane) * 4, (size_t)hp * hp, hp >> 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ig = sigmoid_own(acc[0][i]), fg = sigmoid_own(acc[1][i]), gg = tanh_own(acc[2][i]), og = sigmoid_own(acc[3][i]);
                        const float c = fg * cst[j][i] + ig * gg;
                        cst[j][i] = c;
                        y[(4 * g + i) * POL_LD + col] = og * tanh_own(c);
                    }
                }
            }
            __syncthreads();  // h_t complete in y; every wavefront is done reading h_{t-1} from x
            float *tt = x; x = y; y = tt;
        }
        in = hp;
    } else {
        const int kin = P.states_len * P.states_dim;
        in = pad16(kin);
        for (int e = tid; e < POL_ROWS * in; e += 64 * POL_NW) {
            const int r = e / in, k = e - r * in;
            const int env = row0 + r;
            x[r * POL_LD + k] = (env < P.n && k < kin) ? P.states[(size_t)env * kin + k] : 0.0f;
        }
        __syncthreads();
    }
    for (int l = 0; l <= P.n_critic_hidden; ++l) {
        const bool last = l == P.n_critic_hidden;
        const int out = pad16(last ? 1 : P.critic_hidden[l]);
        dense_layer(x, y, w, w + (size_t)out * in, in, out, last ? PACT_NONE : PACT_RELU, wave, lane);
        __syncthreads();
        w += (size_t)out * in + out;
        in = out;
        float *t = x; x = y; y = t;
    }
    if (tid < POL_ROWS && row0 + tid < P.n) P.value[row0 + tid] = x[tid * POL_LD];
}

}  // namespace taco
