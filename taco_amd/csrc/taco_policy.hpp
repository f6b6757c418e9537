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
#include <type_traits>
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
    float *action_env;  // optional [n][act_dim]: the action clipped to [act_lo, act_hi], what the env executes (ppo_asymmetry.py:310)
    float act_lo, act_hi;
    unsigned long long *stamps;  // optional: shader-clock stamps of workgroup x = 0 ([16], actor: 0..3, critic: 8..13; batched critic: [259], see its loop)
    int n, deterministic;
    uint32_t seed_lo, seed_hi, call;
    int role0;          // taco_policy_kernel: role of blockIdx.y == 0 (0: actor [+ critic at y == 1], 1: critic only)
    float *value_tail;  // batched critic: rows >= value_split write value_tail[row - value_split] (the rollout's last_value)
    int value_split;
    float *hT;          // batched critic: workspace [n][128], the LSTM's h_T between the two kernels
    int ring_n;         // critic input layout.  0: state stacks [row][states_len][states_dim].  N > 0: a frame RING [slot + k][N][states_dim] -- row
                        // = slot * N + env, its frame k is ring row `row + k * N` (the replay store of taco_rollout_run: one frame per step
                        // instead of a shifted stack per slot; include/taco_env.h taco_rollout_bufs)
    const uint32_t *clock;  // optional: the env's device-resident step word (taco_step.hpp, kCtlStep).  Non-NULL: the noise counter is
    uint32_t call_delta;    // *clock + call_delta instead of `call` -- what makes a captured rollout draw fresh noise at every replay
    uint32_t *clamped;      // split-f16 critic: counts the finite frame words beyond +-65 504 it saturated (the workspace's tail word, zeroed per call)
    uint32_t *env_status;   // ... and, inside taco_rollout_run, the env's sticky status word receives kStatusCriticClamped (taco_check reports it)
};
constexpr uint32_t kStatusCriticClampedBit = 2u;   // (= taco_step.hpp kStatusCriticClamped; this header does not include that one)

typedef float pf32x4 __attribute__((ext_vector_type(4)));
TD int pad16(int x) { return (x + 15) / 16 * 16; }

enum { PACT_RELU = 0, PACT_TANH = 1, PACT_NONE = 2 };

// out[16][outp] = act(in[16][inp] W^T + b).  A wavefront owns output tiles tile = wave, wave + POL_NW, ...; ALL operand fragments of a
// tile (KS = inp / 16 blocks, one float4 of weights from L2 and one of activations from LDS each) are requested up front, so the
// latencies are paid once per tile, not once per k block (these layers are 16 x 128 x 128: latency-bound, not throughput-bound).
// KS is a template parameter for the common widths so that the whole tile is straight-line code.
template <int KS, int NW>
TD void dense_tile(const float *in, float *out, const float *W, const float *b, int outp, int act, int wave, int lane, int ks_rt) {
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = outp >> 4;
    const int ks = KS > 0 ? KS : ks_rt;
    constexpr int MAXKS = KS > 0 ? KS : POL_MAXW / 16;
    for (int tile = wave; tile < ntiles; tile += NW) {
        const int col = tile * 16 + r;
        const float *wt = W + ((size_t)tile * ks * 64 + lane) * 4;  // fragment-major: block s of this tile at +256 s floats
        const float *arow = in + r * POL_LD + 4 * g;
        float4 wreg[MAXKS], areg[MAXKS];
#pragma unroll
        for (int s = 0; s < MAXKS; ++s)
            if (KS > 0 || s < ks) wreg[s] = *reinterpret_cast<const float4 *>(wt + 256 * s);
#pragma unroll
        for (int s = 0; s < MAXKS; ++s)
            if (KS > 0 || s < ks) areg[s] = *reinterpret_cast<const float4 *>(arow + 16 * s);
        const float b0 = b[col];
        pf32x4 acc = {b0, b0, b0, b0};
#pragma unroll
        for (int s = 0; s < MAXKS; ++s) {
            if (KS > 0 || s < ks) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s].x, wreg[s].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s].y, wreg[s].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s].z, wreg[s].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s].w, wreg[s].w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // C/D layout: col = lane & 15, row = 4 (lane >> 4) + i
            float v = acc[i];
            if (act == PACT_RELU) v = (v < 0.0f) ? 0.0f : v;
            else if (act == PACT_TANH) v = tanh_own(v);
            out[(4 * g + i) * POL_LD + col] = v;
        }
    }
}
template <int NW>
TD void dense_layer(const float *in, float *out, const float *W, const float *b, int inp, int outp, int act, int wave, int lane) {
    switch (inp >> 4) {
        case 2: dense_tile<2, NW>(in, out, W, b, outp, act, wave, lane, 2); break;
        case 4: dense_tile<4, NW>(in, out, W, b, outp, act, wave, lane, 4); break;
        case 8: dense_tile<8, NW>(in, out, W, b, outp, act, wave, lane, 8); break;
        default: dense_tile<0, NW>(in, out, W, b, outp, act, wave, lane, inp >> 4); break;
    }
}

// LSTM with the wavefront's weight fragments RESIDENT in registers (one hidden tile per wavefront: hp <= 16 POL_NW): the operand
// stream out of L2 -- the limit of the streaming form below, every workgroup pulling the same 1.7 MB per call -- happens once per
// launch instead of once per timestep, and the MFMA loop touches only LDS.  Same MFMA order, same results.  KSX / KSH = k blocks of
// the input / hidden operand.  160 of the 256 registers a wavefront may hold at two wavefronts per SIMD carry weights.
template <int KSX, int KSH, bool WIH_RES>
TD void lstm_resident(const float *Wih, const float *Whh, const float *bs, int hp, int T, const float *xs, float *&x, float *&y, int wave, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const int tile = wave;
    const bool has = tile < (hp >> 4);  // wave-uniform
    const int col = tile * 16 + r;
    constexpr int ip = KSX * 16;
    float4 wx[4][WIH_RES ? KSX : 1], wh[4][KSH];
    float bq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (has) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bq[q] = bs[q * hp + col];
            if (WIH_RES)
#pragma unroll
                for (int s = 0; s < KSX; ++s) wx[q][s] = *reinterpret_cast<const float4 *>(Wih + (size_t)q * hp * ip + ((size_t)(tile * KSX + s) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < KSH; ++s) wh[q][s] = *reinterpret_cast<const float4 *>(Whh + (size_t)q * hp * hp + ((size_t)(tile * KSH + s) * 64 + lane) * 4);
        }
    }
    float cst[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int t = 0; t < T; ++t) {
        if (has) {
            const float *xt = xs + (size_t)t * POL_ROWS * POL_XLD + r * POL_XLD + 4 * g;
            const float *hrow = x + r * POL_LD + 4 * g;
            pf32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = pf32x4{bq[q], bq[q], bq[q], bq[q]};
#pragma unroll
            for (int s = 0; s < KSX; ++s) {
                const float4 a4 = *reinterpret_cast<const float4 *>(xt + 16 * s);
                float4 wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wv[q] = WIH_RES ? wx[q][WIH_RES ? s : 0] : *reinterpret_cast<const float4 *>(Wih + (size_t)q * hp * ip + ((size_t)(tile * KSX + s) * 64 + lane) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wv[q].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wv[q].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wv[q].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wv[q].w, acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < KSH; ++s) {
                const float4 a4 = *reinterpret_cast<const float4 *>(hrow + 16 * s);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wh[q][s].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wh[q][s].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wh[q][s].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wh[q][s].w, acc[q], 0, 0, 0);
            }
            // (issuing the next step's W_ih x_{t+1} MFMAs between the VALU instructions of this epilogue was tried: no gain, the other
            // wavefront of the SIMD already keeps the matrix pipe busy there)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float c, hv;
                lstm_cell(acc[0][i], acc[1][i], acc[2][i], acc[3][i], cst[i], c, hv);
                cst[i] = c;
                y[(4 * g + i) * POL_LD + col] = hv;
            }
        }
        __syncthreads();  // h_t complete in y; every wavefront is done reading h_{t-1} from x
        float *tt = x; x = y; y = tt;
    }
}

// distribution (:333-345) for act_dim == 4: thread (er, a) owns action component a of row er of the workgroup's tile (whole quads: the
// log-prob's two sums run over the quad by DPP, in component order)
// env_lds (optional, the fused rollout kernel): [rows][4] in LDS, receives the action clipped to [act_lo, act_hi] for EVERY row of the tile
// the standard normal of (env, call, component a): counters only, so the fused rollout kernel draws it AHEAD of the MLP (its sampling wavefront idles
// between two steps) -- the same expression either way
TD float sample_eps(const PolicyParams &P, int env, int a, uint32_t call) {
    const U4 rnd = philox(P.seed_lo, P.seed_hi, (uint32_t)env, call, STREAM_POLICY, 0u);
    const uint32_t ba = (a & 2) ? rnd.z : rnd.x, bb = (a & 2) ? rnd.w : rnd.y;
    const float ua = 1.0f - uniform(ba), ub = uniform(bb);
    const float rad = __builtin_sqrtf(-2.0f * log(ua));
    float sn, cs;
    sincos(kTwoPi * ub, sn, cs);
    return (a & 1) ? rad * sn : rad * cs;
}
// eps_pre (optional): sample_eps(P, row0 + er, a, call) evaluated by the caller; scale_pre / ls_pre (with eps_pre): exp(log_std)^2 and its log
// env_seq / env_seq_val (with env_lds): the counter that tells the step wavefront the action is in env_lds -- posted as soon as it is, ahead of the log-prob
TD void sample4(const PolicyParams &P, const float *x, int ld, int row0, int er, int a, const float *log_std, uint32_t call, lds_f32 *env_lds = nullptr,
                const float *eps_pre = nullptr, float scale_pre = 0.0f, float ls_pre = 0.0f, lds_i32 *env_seq = nullptr, int env_seq_val = 0) {
    const int env = row0 + er;
    const bool live = env < P.n;
    const float mean = x[er * ld + a];
    float scale;
    if (eps_pre) scale = scale_pre;
    else { const float e = expf_own(log_std[a]); scale = e * e; }
    float act = mean;
    if (!P.deterministic) {
        const float eps = eps_pre ? *eps_pre : sample_eps(P, env, a, call);
        act = mean + scale * eps;
    }
    if (env_lds) {
        env_lds[er * 4 + a] = clampf(act, P.act_lo, P.act_hi);
        if (env_seq) {   // (one wavefront's LDS operations execute in order: the action words before the counter)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if ((threadIdx.x & 63) == 0) {   // (env_seq_val < 0: one of several posting wavefronts -- each adds one)
                if (env_seq_val >= 0) __hip_atomic_store(env_seq, env_seq_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_add(env_seq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    const float zz = (act - mean) / scale;
    const float z2 = zz * zz, ls = eps_pre ? ls_pre : log(scale);
    auto bcq = [](float v, int k) {
        const int iv = __builtin_bit_cast(int, v);
        const int o = k == 0 ? __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xf, 0xf, true) : k == 1 ? __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xf, 0xf, true)
                    : k == 2 ? __builtin_amdgcn_update_dpp(0, iv, 0xaa, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, iv, 0xff, 0xf, 0xf, true);
        return __builtin_bit_cast(float, o);
    };
    float lp = 0.0f + bcq(z2, 0); lp = lp + bcq(z2, 1); lp = lp + bcq(z2, 2); lp = lp + bcq(z2, 3);
    float hld = 0.0f + bcq(ls, 0); hld = hld + bcq(ls, 1); hld = hld + bcq(ls, 2); hld = hld + bcq(ls, 3);
    if (live) {
        P.action[(size_t)env * 4 + a] = act;
        if (P.action_env) P.action_env[(size_t)env * 4 + a] = clampf(act, P.act_lo, P.act_hi);
        P.mu[(size_t)env * 4 + a] = mean;
        P.sigma[(size_t)env * 4 + a] = log_std[a];
        if (a == 0) P.logp[env] = -0.5f * (4.0f * 1.8378770664093453f + lp) - hld;
    }
}

#define POL_STAMP(k) do { if (P.stamps && blockIdx.x == 0 && tid == 0) P.stamps[(k) + 8 * ROLE] = __builtin_readcyclecounter(); } while (0)

// ---- the actor: MLP + tanh head + action sampling / log-prob (nets_asymmetry.py:331-345) for the 16 envs of workgroup blockIdx.x
template <int NW>
TD void actor_body(const PolicyParams &P, float *bufA, float *bufB) {
    constexpr int ROLE = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t call = P.clock ? P.clock[0] + P.call_delta : P.call;
    POL_STAMP(0);
    const int row0 = blockIdx.x * POL_ROWS;
    const float *w = P.blob;
    float *x = bufA, *y = bufB;
    // ---- where the actor's parameters end (both roles walk the same blob)
    int in_a = pad16(P.obs_len * P.obs_dim);
    {
        // ------------------------------------------------------------------ actor: MLP + tanh head (nets_asymmetry.py:331-332)
        const int kin = P.obs_len * P.obs_dim;
        for (int e = tid; e < POL_ROWS * in_a; e += 64 * NW) {
            const int r = e / in_a, k = e - r * in_a;
            const int env = row0 + r;
            x[r * POL_LD + k] = (env < P.n && k < kin) ? P.obs[(size_t)env * kin + k] : 0.0f;
        }
        __syncthreads();
        POL_STAMP(1);
        int in = in_a;
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const bool last = l == P.n_actor_hidden;
            const int out = pad16(last ? P.act_dim : P.actor_hidden[l]);
            dense_layer<NW>(x, y, w, w + (size_t)out * in, in, out, last ? PACT_TANH : PACT_RELU, wave, lane);
            __syncthreads();
            w += (size_t)out * in + out;
            in = out;
            float *t = x; x = y; y = t;
        }
        POL_STAMP(2);
        const float *log_std = w;
        // ---- distribution (:333-345): scale_tril = diag(exp(log_std) * exp(log_std)).  One thread per (env, action component) when
        // act_dim == 4 (the log-prob's two sums run over the quad by DPP, in component order); one thread per env otherwise.
        if (P.act_dim == 4) {
            if (tid < 4 * POL_ROWS) sample4(P, x, POL_LD, row0, tid >> 2, tid & 3, log_std, call);
        } else if (tid < POL_ROWS && row0 + tid < P.n) {
            const int env = row0 + tid;
            float lp = 0.0f, half_log_det = 0.0f;
            U4 rnd{0u, 0u, 0u, 0u};
            for (int a = 0; a < P.act_dim; ++a) {
                const float mean = x[tid * POL_LD + a];
                const float e = expf_own(log_std[a]);
                const float scale = e * e;
                float act = mean;
                if (!P.deterministic) {
                    if ((a & 3) == 0) rnd = philox(P.seed_lo, P.seed_hi, (uint32_t)env, call, STREAM_POLICY, (uint32_t)(a >> 2));
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
                if (P.action_env) P.action_env[(size_t)env * P.act_dim + a] = clampf(act, P.act_lo, P.act_hi);
                P.mu[(size_t)env * P.act_dim + a] = mean;
                P.sigma[(size_t)env * P.act_dim + a] = log_std[a];
            }
            P.logp[env] = -0.5f * ((float)P.act_dim * 1.8378770664093453f + lp) - half_log_det;
        }
        POL_STAMP(3);
    }
}

// ---- the critic: LSTM over the state stack (or the flattened stack) + MLP (:348-352)
template <int NW, bool WIH_RES>
TD void critic_body(const PolicyParams &P, float *bufA, float *bufB, float *xs) {
    constexpr int ROLE = 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    POL_STAMP(0);
    const int row0 = blockIdx.x * POL_ROWS;
    const float *w = P.blob;
    float *x = bufA, *y = bufB;
    int in_a = pad16(P.obs_len * P.obs_dim);

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
        for (int e = tid; e < T * POL_ROWS * ip; e += 64 * NW) {
            const int t = e / (POL_ROWS * ip), rem = e - t * POL_ROWS * ip, r = rem / ip, k = rem - r * ip;
            const int env = row0 + r;
            const size_t src = P.ring_n > 0 ? ((size_t)env + (size_t)t * (size_t)P.ring_n) * P.states_dim + k : ((size_t)env * T + t) * P.states_dim + k;
            xs[(t * POL_ROWS + r) * POL_XLD + k] = (env < P.n && k < P.states_dim) ? P.states[src] : 0.0f;
        }
        for (int e = tid; e < POL_ROWS * hp; e += 64 * NW) x[(e / hp) * POL_LD + (e % hp)] = 0.0f;
        __syncthreads();
        POL_STAMP(1);
        const int ntiles = hp >> 4;
        if (ip == 32 && ntiles == 8) {
            lstm_resident<2, 8, WIH_RES>(Wih, Whh, bs, hp, T, xs, x, y, wave, lane);
        } else if (ip == 32 && ntiles == 4) {
            lstm_resident<2, 4, WIH_RES>(Wih, Whh, bs, hp, T, xs, x, y, wave, lane);
        } else {
        const int r = lane & 15, g = lane >> 4;
        constexpr int TPW = (POL_MAXW / 16 + NW - 1) / NW;  // hidden tiles per wavefront at the widest LSTM
        float cst[TPW][4];  // cell state of this wavefront's hidden tiles (tile = wave + NW j), rows 4 g + i
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
                const int tile = wave + NW * j;
                if (tile < ntiles) {  // wave-uniform
                    const int col = tile * 16 + r;
                    pf32x4 acc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float b = bs[q * hp + col]; acc[q] = pf32x4{b, b, b, b}; }
                    gate_chain(acc, xt, Wih + ((size_t)tile * (ip >> 4) * 64 + lane) * 4, (size_t)hp * ip, ip >> 4);
                    gate_chain(acc, hrow, Whh + ((size_t)tile * (hp >> 4) * 64 + lane) * 4, (size_t)hp * hp, hp >> 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float c, hv;
                        lstm_cell(acc[0][i], acc[1][i], acc[2][i], acc[3][i], cst[j][i], c, hv);
                        cst[j][i] = c;
                        y[(4 * g + i) * POL_LD + col] = hv;
                    }
                }
            }
            __syncthreads();  // h_t complete in y; every wavefront is done reading h_{t-1} from x
            float *tt = x; x = y; y = tt;
        }
        }
        POL_STAMP(2);
        in = hp;
    } else {
        const int kin = P.states_len * P.states_dim;
        in = pad16(kin);
        for (int e = tid; e < POL_ROWS * in; e += 64 * NW) {
            const int r = e / in, k = e - r * in;
            const int env = row0 + r;
            size_t src = (size_t)env * kin + k;
            if (P.ring_n > 0) { const int t = k / P.states_dim; src = ((size_t)env + (size_t)t * (size_t)P.ring_n) * P.states_dim + (k - t * P.states_dim); }
            x[r * POL_LD + k] = (env < P.n && k < kin) ? P.states[src] : 0.0f;
        }
        __syncthreads();
    }
    for (int l = 0; l <= P.n_critic_hidden; ++l) {
        const bool last = l == P.n_critic_hidden;
        const int out = pad16(last ? 1 : P.critic_hidden[l]);
        dense_layer<NW>(x, y, w, w + (size_t)out * in, in, out, last ? PACT_NONE : PACT_RELU, wave, lane);
        __syncthreads();
        w += (size_t)out * in + out;
        in = out;
        float *t = x; x = y; y = t;
    }
    POL_STAMP(3);
    if (tid < POL_ROWS && row0 + tid < P.n) {
        const int row = row0 + tid;
        if (row < P.value_split) P.value[row] = x[tid * POL_LD];
        else P.value_tail[row - P.value_split] = x[tid * POL_LD];
    }
    POL_STAMP(4);
}
#undef POL_STAMP

// One launch for both roles (taco_policy_act): grid.y = 0 actor, 1 critic
__global__ __launch_bounds__(64 * POL_NW) void taco_policy_kernel(const PolicyParams P) {
    __shared__ __attribute__((aligned(16))) float bufA[POL_ROWS * POL_LD];
    __shared__ __attribute__((aligned(16))) float bufB[POL_ROWS * POL_LD];
    __shared__ __attribute__((aligned(16))) float xs[POL_MAXT * POL_ROWS * POL_XLD];
    if (blockIdx.y + P.role0 == 0) actor_body<POL_NW>(P, bufA, bufB);
    else critic_body<POL_NW, true>(P, bufA, bufB, xs);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The critic alone over MANY rows (taco_critic_values; the rollout's H x N + N state stacks, off the per-step chain: nothing before GAE
// consumes `value`, ppo_asymmetry.py:308-342).  Same arithmetic as critic_body -- the same v_mfma_f32_16x16x4_f32 chains in the same k
// order, so the same bits -- as two kernels shaped by what tools/ubench/mfma_valu.hip measures on gfx950:
//   * an f32 MFMA occupies the SIMD's FP32 lanes for its 32 cycles: plain VALU work does NOT run beside it, neither from the same
//     wavefront nor from the SIMD's other one (MFMA + 4 v_fma = 57 cycles; the bf16 MFMA + 4 v_fma = 35), and a pending MFMA of either
//     wavefront is served before any VALU instruction.  The gate activations (5 exp + 3 div per cell) are therefore pure ADDED time,
//     and the matrix pipe's share of a timestep is MFMA / (MFMA + VALU), whatever the schedule;
//   * one wavefront alone issues a VALU instruction every ~5.6 cycles, two together one every ~2.7: the activations must be done by BOTH
//     wavefronts of a SIMD at the same time.  (A first form ran the SIMD partners one phase apart -- one's MFMAs beside the other's
//     activations -- and lost 20 % to exactly this: profiles/r02_g_critic_forms.txt.)
// taco_critic_lstm_kernel: PERSISTENT, one workgroup (8 wavefronts, two per SIMD) per CU, W_ih / W_hh fragments of hidden columns
//   16 w .. 16 w + 15 (all four gates) resident in 160 registers of wavefront w for the whole launch; the workgroup walks 32-row blocks
//   (blocks b, b + grid, ...).  Per timestep every wavefront runs its 320 MFMAs (two 16-row tiles), then its 8 cells, then ONE
//   s_barrier; h_t is double-buffered in LDS.  The next block's state stacks are staged into the other half of xs around the cells of
//   timestep 0 (loads issued before them, LDS stores after).  h_T goes to a global workspace.
//   The first timestep skips the W_hh h_{-1} chain: h_{-1} = +0, and fma(+0, w, acc) leaves every acc but a zero unchanged -- a zero acc
//   becomes +0 unless it is -0 and every weight of the column has its sign bit set, which `allneg` reproduces (finite weights assumed).
// taco_critic_mlp_kernel: the MLP on h_T, 64 rows per pass, W1 / W2 / W3 fragments register-resident, two workgroups per CU.
constexpr int CR_ROWS = 32;   // rows per LSTM block (two 16-row MFMA tiles)
constexpr int CR_LD = 132;    // h / MLP activation row stride in floats (width 128): 528 B, 8 consecutive rows cover all 32 banks
constexpr int CR_MLP_ROWS = 64;
// Conflict-free activation tiles (the pair LSTM kernel's h_t, the critic MLP's layers).  A ds_read_b128 is served in four 16-lane groups
// (MI355X_MICROARCH.md, LDS): {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- with lane = 16 g + r reading row r, 16-byte slot
// 4 s + g of an MFMA operand fragment, a group is the 16 rows r, rows 0-3 / 12-15 at slot c and rows 4-11 at slot c ^ 1 (or the other way
// round).  With the padded stride of 132 floats (33 slots) row r's slot c lies on bank slot (r + c) mod 16: rows 11 and 12 of every group met
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 44 % in rounds 2-3).  Storing slot c of row r at position c ^ b(r), b(r) = 1 for rows 4..11 (mod
// 16) and 0 otherwise, makes both halves of a group read position-consistent columns: 16 rows -> 16 different bank slots.  b() is a lane
// constant on both sides (readers: b(r); the producers' lanes write rows 4 g + i, and b(4 g + i) = b of g alone), so every address is still
// one lane register + immediates; the ds_write_b32 of the producers and the b128 row copies stay conflict-free as well.
__device__ __forceinline__ int cr_b(int row) { return ((row + 4) >> 3) & 1; }
__device__ __forceinline__ int cr_sw4(int row, int c4) { return row * CR_LD + ((c4 ^ cr_b(row)) << 2); }   // float index of (row, 4 c4) in a swizzled tile

// requirements (checked by the host, which falls back to taco_policy_kernel's critic role otherwise): pad16(states_dim) == 32,
// pad16(lstm_hidden) == 128, states_len <= POL_MAXT, critic MLP = two hidden layers padded to 128
// RING: the input is a frame ring (PolicyParams.ring_n) instead of materialised stacks -- a template parameter so that the fragment
// addressing of either layout is compile-time arithmetic (as a run-time switch it cost 23 SGPR spills and 4 more VGPR spills)
// EXACT: the cell with the oracle's operations (taco_math.hpp lstm_cell) instead of the hardware's 2^x / reciprocal (lstm_cell_fast):
// include/taco_env.h TACO_P_EXACT_CELL
template <bool RING, bool EXACT>
__global__ __launch_bounds__(64 * POL_NW) void taco_critic_lstm_kernel(const PolicyParams P) {
    __shared__ __attribute__((aligned(16))) float xs[2][POL_MAXT * CR_ROWS * 32];   // the block's state stacks as they lie in memory: [row][t][sd]
    __shared__ __attribute__((aligned(16))) float hb[3][CR_ROWS * CR_LD];   // h_t double buffer + [2] = h_T on its way to the workspace
    constexpr int KSX = 2, KSH = 8, hp = 128, ip = 32, RT = CR_ROWS / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int T = P.states_len, sd = P.states_dim;
    const int nblocks = (P.n + CR_ROWS - 1) / CR_ROWS;
    const int nb = ((int)blockIdx.x < nblocks) ? (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;  // blocks of this workgroup
    if (nb == 0) return;
    const float *w = P.blob;
    {
        int in_a = pad16(P.obs_len * P.obs_dim);
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const int out = pad16(l == P.n_actor_hidden ? P.act_dim : P.actor_hidden[l]);
            w += (size_t)out * in_a + out;
            in_a = out;
        }
        w += 16;  // log_std
    }
    const float *Wih = w, *Whh = w + (size_t)4 * hp * ip, *bs = Whh + (size_t)4 * hp * hp;

    // ---- this wavefront's weight fragments (hidden tile = wave)
    const int col = wave * 16 + r;
    float4 wx[4][KSX], wh[4][KSH];
    float bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bq[q] = bs[q * hp + col];
#pragma unroll
        for (int s = 0; s < KSX; ++s) wx[q][s] = *reinterpret_cast<const float4 *>(Wih + (size_t)q * hp * ip + ((size_t)(wave * KSX + s) * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < KSH; ++s) wh[q][s] = *reinterpret_cast<const float4 *>(Whh + (size_t)q * hp * hp + ((size_t)(wave * KSH + s) * 64 + lane) * 4);
    }
    // ---- staging of one block's state stacks: the block's bytes are contiguous in memory (32 rows x T x sd floats), and they are copied
    // AS THEY LIE into xs[half] by LDS-DMA (global_load_lds_dwordx4: no registers -- this kernel has none to spare -- 1 KiB per wavefront
    // instruction); the fragment reads below do the [row][t][sd] addressing and zero the columns beyond sd.
    // Ring input (P.ring_n = N > 0): frame k of row `row` is ring row `row + k N`, so the 32 rows' frames of ONE timestep are contiguous
    // (32 x sd floats) whatever slot boundaries the block straddles: T pieces per block, xs[half] = [t][row][sd] instead of [row][t][sd].
    constexpr bool ring = RING;
    const size_t total_bytes = (size_t)P.n * T * sd * sizeof(float);
    const int block_bytes = CR_ROWS * T * sd * (int)sizeof(float);
    const int piece_bytes = CR_ROWS * sd * (int)sizeof(float);   // ring: one timestep of the block
    const bool ring_dma_ok = ring && (((size_t)P.ring_n * sd * sizeof(float)) & 15u) == 0 && (piece_bytes & 15) == 0;   // 16-byte aligned pieces
    auto stage = [&](int blk, int half) {
        const size_t base = (size_t)blk * block_bytes;
        if constexpr (ring) {
            const size_t row0 = (size_t)blk * CR_ROWS;
            if (ring_dma_ok && row0 + CR_ROWS <= (size_t)P.n) {  // wave-uniform
                const int per_piece = (piece_bytes / 16 + 63) / 64;   // wavefront instructions per timestep piece (sd = 26: 208 x 16 B = 4)
                for (int j = wave; j < T * per_piece; j += POL_NW) {  // (wave-uniform trip count)
                    const int k = j / per_piece, off = ((j - k * per_piece) * 64 + lane) * 16;
                    const char *src = reinterpret_cast<const char *>(P.states) + (row0 + (size_t)k * (size_t)P.ring_n) * sd * sizeof(float);
                    if (off < piece_bytes)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(xs[half]) + k * piece_bytes + (j - k * per_piece) * 1024), 16, 0, 0);
                }
            } else {  // the last, partial block, or pieces that are not 16-byte aligned (odd N): plain loads, zeros for the missing rows
#pragma unroll 1
                for (int e = tid; e < T * CR_ROWS * sd; e += 64 * POL_NW) {
                    const int k = e / (CR_ROWS * sd), rem = e - k * CR_ROWS * sd, row = rem / sd;
                    xs[half][e] = row0 + row < (size_t)P.n ? P.states[(row0 + (size_t)k * (size_t)P.ring_n) * sd + rem] : 0.0f;
                }
            }
        } else if (base + block_bytes <= total_bytes) {  // (materialised stacks) wave-uniform
            const char *src = reinterpret_cast<const char *>(P.states) + base;
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // 32 rows x 8 frames x 32 floats = 2048 pieces of 16 B at most = 4 per thread
                const int piece0 = 512 * j + 64 * wave;   // this wavefront instruction's first piece
                if (piece0 * 16 < block_bytes) {            // wave-uniform
                    const int off = (piece0 + lane) * 16;
                    if (off < block_bytes)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(xs[half]) + piece0 * 16), 16, 0, 0);
                }
            }
        } else {  // the array's last, partial block (once per launch): plain loads, zeros for the missing rows
            const size_t first = base / sizeof(float), total = total_bytes / sizeof(float);
#pragma unroll 1
            for (int e = tid; e < block_bytes / (int)sizeof(float); e += 64 * POL_NW) xs[half][e] = first + e < total ? P.states[first + e] : 0.0f;
        }
    };
    stage((int)blockIdx.x, 0);
    // which of the 4 floats this lane reads in the LAST k block of the x operand exist (k = 16 (KSX - 1) + 4 g + j < sd)
    bool xvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xvalid[j] = 16 * (KSX - 1) + 4 * g + j < sd;
    const bool sd_even = (sd & 1) == 0;
    // ---- sign bookkeeping of the skipped first h chain (see above): does every W_hh weight of (gate q, column col) have its sign bit set?
    bool allneg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t m = 0x80000000u;
#pragma unroll
        for (int s = 0; s < KSH; ++s)
            m &= __builtin_bit_cast(uint32_t, wh[q][s].x) & __builtin_bit_cast(uint32_t, wh[q][s].y) & __builtin_bit_cast(uint32_t, wh[q][s].z) &
                 __builtin_bit_cast(uint32_t, wh[q][s].w);
        m &= (uint32_t)__shfl_xor((int)m, 16, 64);
        m &= (uint32_t)__shfl_xor((int)m, 32, 64);
        allneg[q] = m != 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first block's stacks (LDS-DMA) have landed
    __syncthreads();

    pf32x4 acc[RT][4];  // [row tile][gate], rows 16 rt + 4 g + i, column col
    float cst[RT][4];   // cell state
    auto mfma_x = [&](const float *xh, int t) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[rt][q] = pf32x4{bq[q], bq[q], bq[q], bq[q]};
        const float *xt = xh + (ring ? (t * CR_ROWS + r) * sd : (r * T + t) * sd) + 4 * g;
        const int tile_stride = ring ? 16 * sd : 16 * T * sd;   // floats between the two 16-row tiles of the block
#pragma unroll
        for (int s = 0; s < KSX; ++s) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float *at = xt + rt * tile_stride + 16 * s;
                float4 a4;
                if (sd_even) {  // frames start 8-byte aligned
                    const float2 lo = *reinterpret_cast<const float2 *>(at), hi = *reinterpret_cast<const float2 *>(at + 2);
                    a4 = float4{lo.x, lo.y, hi.x, hi.y};
                } else {
                    a4 = float4{at[0], at[1], at[2], at[3]};
                }
                if (s == KSX - 1) {  // beyond the frame: what lies there is the next frame, not the zero padding of the weights' k range
                    a4.x = xvalid[0] ? a4.x : 0.0f; a4.y = xvalid[1] ? a4.y : 0.0f; a4.z = xvalid[2] ? a4.z : 0.0f; a4.w = xvalid[3] ? a4.w : 0.0f;
                } else if (16 * (s + 1) > sd) {
                    a4.x = 16 * s + 4 * g + 0 < sd ? a4.x : 0.0f; a4.y = 16 * s + 4 * g + 1 < sd ? a4.y : 0.0f;
                    a4.z = 16 * s + 4 * g + 2 < sd ? a4.z : 0.0f; a4.w = 16 * s + 4 * g + 3 < sd ? a4.w : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wx[q][s].x, acc[rt][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wx[q][s].y, acc[rt][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wx[q][s].z, acc[rt][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wx[q][s].w, acc[rt][q], 0, 0, 0);
            }
        }
    };
    auto mfma_h = [&](const float *hprev) {
        const float *hrow = hprev + r * CR_LD + 4 * g;
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 a4 = *reinterpret_cast<const float4 *>(hrow + 16 * rt * CR_LD + 16 * s);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wh[q][s].x, acc[rt][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wh[q][s].y, acc[rt][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wh[q][s].z, acc[rt][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wh[q][s].w, acc[rt][q], 0, 0, 0);
            }
        }
    };
    // the gate activations of timestep t; h_t to LDS, or (last timestep) h_T to the workspace rows of the block
    auto cells = [&](int t, int blk, bool first) {
        float hv[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float cn;
                lstm_cell_batched<EXACT>(acc[rt][0][i], acc[rt][1][i], acc[rt][2][i], acc[rt][3][i], first ? 0.0f : cst[rt][i], cn, hv[rt][i]);
                cst[rt][i] = cn;
            }
        float *hout = hb[t + 1 < T ? (t & 1) : 2] + (4 * g) * CR_LD + col;   // (the last timestep's h goes out through hb[2], see flush_hT)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) hout[(16 * rt + i) * CR_LD] = hv[rt][i];
    };
    // h_T of block `blk` (complete in hb[2] after the barrier of its last timestep) -> workspace rows, 16 bytes per lane, coalesced; hb[2] is
    // not written again before the last timestep of the next block, many barriers later
    auto flush_hT = [&](int blk) {
        const size_t row0 = (size_t)blk * CR_ROWS;
#pragma unroll
        for (int j = 0; j < CR_ROWS * (hp / 4) / (64 * POL_NW); ++j) {   // 32 rows x 32 float4 = 2 per thread
            const int e = tid + 64 * POL_NW * j, row = e >> 5, k4 = e & 31;
            if (row0 + row < (size_t)P.n) *reinterpret_cast<float4 *>(P.hT + (row0 + row) * hp + 4 * k4) = *reinterpret_cast<const float4 *>(hb[2] + row * CR_LD + 4 * k4);
        }
    };
    // profiling (P.stamps): workgroup 0, wavefronts 0 and 4: shader clock at the start of a timestep, after its MFMAs, after its cells, for the
    // first 21 timesteps ([wave >> 2][21][3] uint64); [256] = the 100 MHz wall clock at entry, [257] at exit, [258] shader clock at exit
    const bool stamping = P.stamps != nullptr && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0;
    if (stamping && wave == 0) P.stamps[256] = wall_clock64();
    int step = 0;
    for (int k = 0; k < nb; ++k) {
        const int blk = (int)blockIdx.x + k * (int)gridDim.x;
        const float *xh = xs[k & 1];
        // the next block's stacks: the DMA lands during timestep 0 (the barrier that ends it waits for it; xs[(k + 1) & 1] was last read
        // in block k - 1)
        if (k + 1 < nb) stage(blk + (int)gridDim.x, (k + 1) & 1);
        if (k > 0) flush_hT(blk - (int)gridDim.x);
        // ---- timestep 0: the x chain only (+ the sign rule)
        if (stamping && step < 21) P.stamps[(wave >> 2) * 64 + 3 * step] = __builtin_readcyclecounter();
        mfma_x(xh, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[rt][q][i];
                    acc[rt][q][i] = (v == 0.0f && !allneg[q]) ? 0.0f : v;
                }
        if (stamping && step < 21) P.stamps[(wave >> 2) * 64 + 3 * step + 1] = __builtin_readcyclecounter();
        cells(0, blk, true);
        if (stamping && step < 21) P.stamps[(wave >> 2) * 64 + 3 * step + 2] = __builtin_readcyclecounter();
        ++step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next block's stacks have landed (the DMA was issued a whole timestep ago)
        __syncthreads();
        for (int t = 1; t < T; ++t) {
            if (stamping && step < 21) P.stamps[(wave >> 2) * 64 + 3 * step] = __builtin_readcyclecounter();
            mfma_x(xh, t);
            mfma_h(hb[(t - 1) & 1]);
            if (stamping && step < 21) P.stamps[(wave >> 2) * 64 + 3 * step + 1] = __builtin_readcyclecounter();
            cells(t, blk, false);
            if (stamping && step < 21) P.stamps[(wave >> 2) * 64 + 3 * step + 2] = __builtin_readcyclecounter();
            ++step;
            __syncthreads();
        }
    }
    flush_hT((int)blockIdx.x + (nb - 1) * (int)gridDim.x);
    if (stamping && wave == 0) { P.stamps[257] = wall_clock64(); P.stamps[258] = __builtin_readcyclecounter(); }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same LSTM over a frame ring with the input projection SHARED between consecutive slots ("W_ih x once per
// frame", as far as registers allow).  A block is 16 envs x TWO consecutive slots (tile A = slot 2p, tile B = slot 2p + 1): at timestep
// k tile A consumes frame 2p + k and tile B frame 2p + k + 1 -- which is the frame tile A consumes at timestep k + 1.  So tile B's
// accumulators after its x chain (bias + W_ih x, BEFORE the h chain) are kept (16 registers) and become tile A's starting accumulators
// of the next timestep: bias, then the x terms, then the h terms -- the same fma chain in the same order, the same bits -- and tile A
// runs an x chain only at timestep 0.  6 x chains per block instead of 10: 1 216 MFMAs per wavefront and block instead of 1 344 (-9.5 %).
// (The whole T-frame window of projections, 5 x 32 rows x 512 floats, fits neither the LDS nor the registers left beside the resident
// weights: one shared neighbour is what fits.)  The two tiles run one after the other (one set of accumulators: the four gate chains of a
// tile already cover the MFMA latency), everything else -- persistent workgroups, resident W_ih / W_hh fragments, LDS-DMA staging of the
// next block, one barrier per timestep, h_T through the workspace -- is taco_critic_lstm_kernel's.
// Rows: row = slot * N + env as everywhere; S = rows / N slots; an odd S leaves the last block's tile B without a slot (computed on zero
// frames, written nowhere); env groups of 16 with a ragged last one.
template <bool EXACT>
__global__ __launch_bounds__(64 * POL_NW) void taco_critic_lstm_pair_kernel(const PolicyParams P) {
    constexpr int FR = POL_MAXT + 1;   // frames a block reads: T + 1
    __shared__ __attribute__((aligned(16))) float xs[2][FR * 16 * 32];   // [frame][env][sd] as they lie in the ring
    __shared__ __attribute__((aligned(16))) float hb[3][CR_ROWS * CR_LD];   // h_t double buffer (rows 0..15 tile A, 16..31 tile B) + [2] = h_T on its way out
    constexpr int KSX = 2, KSH = 8, hp = 128, ip = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int T = P.states_len, sd = P.states_dim, N = P.ring_n;
    const int S = P.n / N;                                  // slots
    const int groups = (N + 15) / 16, pairs = (S + 1) / 2, nblocks = groups * pairs;
    const int nb = ((int)blockIdx.x < nblocks) ? (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (nb == 0) return;
    const float *w = P.blob;
    {
        int in_a = pad16(P.obs_len * P.obs_dim);
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const int out = pad16(l == P.n_actor_hidden ? P.act_dim : P.actor_hidden[l]);
            w += (size_t)out * in_a + out;
            in_a = out;
        }
        w += 16;  // log_std
    }
    const float *Wih = w, *Whh = w + (size_t)4 * hp * ip, *bs = Whh + (size_t)4 * hp * hp;
    const int col = wave * 16 + r;
    float4 wx[4][KSX], wh[4][KSH];
    float bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bq[q] = bs[q * hp + col];
#pragma unroll
        for (int s = 0; s < KSX; ++s) wx[q][s] = *reinterpret_cast<const float4 *>(Wih + (size_t)q * hp * ip + ((size_t)(wave * KSX + s) * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < KSH; ++s) wh[q][s] = *reinterpret_cast<const float4 *>(Whh + (size_t)q * hp * hp + ((size_t)(wave * KSH + s) * 64 + lane) * 4);
    }
    const int ring_rows = S + T - 1;                        // frames in the ring
    const int piece_bytes = 16 * sd * (int)sizeof(float);   // one frame of a full env group
    const bool dma_ok = (((size_t)N * sd * sizeof(float)) & 15u) == 0 && (piece_bytes & 15) == 0;
    auto stage = [&](int blk, int half) {
        const int pair = blk / groups, e0 = (blk - pair * groups) * 16, slot0 = 2 * pair;
        const int per_piece = (piece_bytes / 16 + 63) / 64;   // wavefront instructions per frame (sd = 26: 104 x 16 B = 2)
        if (dma_ok && e0 + 16 <= N && slot0 + T < ring_rows) {   // (wave-uniform) every frame of the block exists and is whole
            for (int j = wave; j < (T + 1) * per_piece; j += POL_NW) {
                const int f = j / per_piece, off = ((j - f * per_piece) * 64 + lane) * 16;
                const char *src = reinterpret_cast<const char *>(P.states) + ((size_t)(slot0 + f) * N + e0) * sd * sizeof(float);
                if (off < piece_bytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                     (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(xs[half]) + f * piece_bytes + (j - f * per_piece) * 1024), 16, 0, 0);
            }
        } else {  // ragged env group, a frame beyond the ring (odd S) or unaligned pieces: plain loads, zeros for what does not exist
#pragma unroll 1
            for (int e = tid; e < (T + 1) * 16 * sd; e += 64 * POL_NW) {
                const int f = e / (16 * sd), rem = e - f * 16 * sd, env = rem / sd;
                xs[half][e] = (slot0 + f < ring_rows && e0 + env < N) ? P.states[((size_t)(slot0 + f) * N + e0) * sd + rem] : 0.0f;
            }
        }
    };
    stage((int)blockIdx.x, 0);
    bool xvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xvalid[j] = 16 * (KSX - 1) + 4 * g + j < sd;
    const bool sd_even = (sd & 1) == 0;
    bool allneg[4];   // sign bookkeeping of the skipped first h chain, see taco_critic_lstm_kernel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t m = 0x80000000u;
#pragma unroll
        for (int s = 0; s < KSH; ++s)
            m &= __builtin_bit_cast(uint32_t, wh[q][s].x) & __builtin_bit_cast(uint32_t, wh[q][s].y) & __builtin_bit_cast(uint32_t, wh[q][s].z) &
                 __builtin_bit_cast(uint32_t, wh[q][s].w);
        m &= (uint32_t)__shfl_xor((int)m, 16, 64);
        m &= (uint32_t)__shfl_xor((int)m, 32, 64);
        allneg[q] = m != 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    pf32x4 acc[4], keep[4];   // the tile in flight; tile B's accumulators after its x chain = tile A's start of the next timestep
    float cstA[4], cstB[4];
    auto x_chain = [&](const float *xh, int f) __attribute__((always_inline)) {   // acc = bias + W_ih x(frame f), rows = the block's 16 envs
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = pf32x4{bq[q], bq[q], bq[q], bq[q]};
        const float *xt = xh + (f * 16 + r) * sd + 4 * g;
#pragma unroll
        for (int s = 0; s < KSX; ++s) {
            const float *at = xt + 16 * s;
            float4 a4;
            if (sd_even) {
                const float2 lo = *reinterpret_cast<const float2 *>(at), hi = *reinterpret_cast<const float2 *>(at + 2);
                a4 = float4{lo.x, lo.y, hi.x, hi.y};
            } else {
                a4 = float4{at[0], at[1], at[2], at[3]};
            }
            if (s == KSX - 1) {
                a4.x = xvalid[0] ? a4.x : 0.0f; a4.y = xvalid[1] ? a4.y : 0.0f; a4.z = xvalid[2] ? a4.z : 0.0f; a4.w = xvalid[3] ? a4.w : 0.0f;
            } else if (16 * (s + 1) > sd) {
                a4.x = 16 * s + 4 * g + 0 < sd ? a4.x : 0.0f; a4.y = 16 * s + 4 * g + 1 < sd ? a4.y : 0.0f;
                a4.z = 16 * s + 4 * g + 2 < sd ? a4.z : 0.0f; a4.w = 16 * s + 4 * g + 3 < sd ? a4.w : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wx[q][s].x, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wx[q][s].y, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wx[q][s].z, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wx[q][s].w, acc[q], 0, 0, 0);
        }
    };
    auto h_chain = [&](const float *hprev, int tile) __attribute__((always_inline)) {
        const float *hrow = hprev + (16 * tile + r) * CR_LD + 4 * (g ^ cr_b(r));   // (swizzled tile: cr_sw4)
        float4 nx = *reinterpret_cast<const float4 *>(hrow);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const float4 a4 = nx;
            if (s + 1 < KSH) nx = *reinterpret_cast<const float4 *>(hrow + 16 * (s + 1));   // on its way while this block's 16 MFMAs issue
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wh[q][s].x, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wh[q][s].y, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wh[q][s].z, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wh[q][s].w, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // this fragment's 16 MFMAs, then ONE more LDS read: two fragments ahead of the
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // MFMAs at any time (the two reads in front of the loop + one per block)
        }
    };
    auto zero_rule = [&]() __attribute__((always_inline)) {   // timestep 0: the skipped W_hh h_{-1} chain would have turned a -0 into +0 unless every weight is negative
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = acc[q][i];
                acc[q][i] = (v == 0.0f && !allneg[q]) ? 0.0f : v;
            }
    };
    auto cells = [&](int t, int tile, float (&cst)[4], bool first) __attribute__((always_inline)) {
        float *hout = hb[t + 1 < T ? (t & 1) : 2] + (16 * tile + 4 * g) * CR_LD + (col ^ (cr_b(4 * g) << 2));   // rows 4 g + i: b(row) = b(4 g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float cn, hv;
            lstm_cell_batched<EXACT>(acc[0][i], acc[1][i], acc[2][i], acc[3][i], first ? 0.0f : cst[i], cn, hv);
            cst[i] = cn;
            hout[i * CR_LD] = hv;
        }
    };
    // h_T of block `blk` (complete in hb[2] after the barrier of its last timestep) -> workspace rows slot * N + env
    auto flush_hT = [&](int blk) {
        const int pair = blk / groups, e0 = (blk - pair * groups) * 16;
#pragma unroll
        for (int j = 0; j < CR_ROWS * (hp / 4) / (64 * POL_NW); ++j) {   // 32 rows x 32 float4 = 2 per thread
            const int e = tid + 64 * POL_NW * j, row = e >> 5, k4 = e & 31;
            const int slot = 2 * pair + (row >> 4), env = e0 + (row & 15);
            if (slot < S && env < N)
                *reinterpret_cast<float4 *>(P.hT + ((size_t)slot * N + env) * hp + 4 * k4) = *reinterpret_cast<const float4 *>(hb[2] + cr_sw4(row, k4));
        }
    };
    for (int k = 0; k < nb; ++k) {
        const int blk = (int)blockIdx.x + k * (int)gridDim.x;
        const float *xh = xs[k & 1];
        if (k + 1 < nb) stage(blk + (int)gridDim.x, (k + 1) & 1);
        if (k > 0) flush_hT(blk - (int)gridDim.x);
        // ---- timestep 0: x chains only
        x_chain(xh, 0);
        zero_rule();
        cells(0, 0, cstA, true);
        x_chain(xh, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] = acc[q];
        zero_rule();
        cells(0, 1, cstB, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next block's frames have landed
        __syncthreads();
        for (int t = 1; t < T; ++t) {
            const float *hprev = hb[(t - 1) & 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = keep[q];      // tile A: bias + W_ih x(frame t), computed for tile B a timestep ago
            h_chain(hprev, 0);
            cells(t, 0, cstA, false);
            x_chain(xh, t + 1);                                 // tile B: frame t + 1
#pragma unroll
            for (int q = 0; q < 4; ++q) keep[q] = acc[q];
            h_chain(hprev, 1);
            cells(t, 1, cstB, false);
            __syncthreads();
        }
    }
    flush_hT((int)blockIdx.x + (nb - 1) * (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The pair kernel on the 16-BIT MATRIX PIPE with SPLIT operands (round 5; include/taco_env.h TACO_P_SPLIT_F16 / TACO_P_SPLIT_BF16: a flag at the C
// boundary, which the Python host layer sets by default -- taco_amd/policy.py critic_split).
// taco_critic_lstm_pair_kernel is MFMA time + the cells' VALU time, and its MFMA time is the f32 pipe's: v_mfma_f32_16x16x4_f32 delivers
// 2 048 flop in 32 cycles, v_mfma_f32_16x16x32_{f16,bf16} 16 384 in 16 (MI355X_MICROARCH.md) -- 16 x per cycle.  Every f32 operand v is
// split into two 16-bit halves, v = v1 + v2 with v1 = RN16(v) and v2 = RN16(v - v1) (v - v1 is exact in f32; the low half of a typical f16
// operand is SUBNORMAL, which gfx950's 16-bit MFMAs keep: tools/ubench/mfma_f16_denorm -- the first two forms of this kernel scaled the low
// halves by 2^11 into the normal range and carried the cross terms in accumulators of their own), and a product chain is THREE MFMA chains,
//     acc = bias + sum x1 w1 + sum x1 w2 + sum x2 w1                      (f32 accumulation inside the MFMA; x2 w2 is dropped)
// per-product relative error <= 3 * 2^-22 (f16: 11 + 11 significant bits) or 3 * 2^-16 (bf16: 8 + 8) instead of one f32 rounding.
// MEASURED against the exact f32 critic (tests/test_policy_gpu.py, tools/cell_ab.py; a numpy model of the same arithmetic predicted the ranking):
// f16 halves: |value difference| 3e-7 ... 1e-6 on O(1) values (the hardware cell alone: 7e-7) -- inside the 2e-6 bar of the fast cell; bf16
// halves with three terms: 1e-5 ... 3e-5, OUTSIDE that bar (six terms -- a three-way split -- would meet it at 1.5 x the weight registers,
// which this kernel does not have).
// Operand range of the f16 form: finite frame words beyond +-65 504 saturate there AND ARE COUNTED (frames are O(1): normalised positions, rates,
// voltages -- the clamp keeps a huge finite word from turning into inf - inf; the gates it feeds are saturated either way); +-inf becomes NaN, NaN
// stays NaN (a diverged env poisons its value as in the f32 kernel); h_t lies in [-1, 1]; bf16 has f32's range.
// Layout (cdna_hip_programming.md section 3): lane (r = lane & 15, g = lane >> 4) holds A[row r][k = 32 S + 8 g + j] and
// B[k = 32 S + 8 g + j][col r], j = 0 .. 7 (four VGPRs each); D: row 4 g + i, col r.  The product is taken TRANSPOSED (D = W x^T: the weights
// are the A operand, the activations the B operand -- see the kernel).  Weights: both halves of the wavefront's W_ih / W_hh columns resident
// (2 x 16 bit per weight: the 160 registers of the f32 form).  The frames are split ONCE per block into LDS planes; h_t travels through LDS as
// two 16-bit planes (one 8-byte store per plane and lane), read back as ds_read_b128 per plane and K block, row stride 272 B = 17 slots of
// 16 B -- row r's slot c on bank slot (r + c) mod 16 like the f32 tiles, so the same swizzle c ^ b(r) keeps the reads conflict-free.  The
// bias rides in the weights' column sd against a 1.0 in the frames.  The last timestep's h_T leaves as f32 (taco_critic_mlp_split_kernel splits it again: same bytes as two 16-bit planes would be).
// Everything else -- persistent workgroups, pairs of consecutive slots sharing the input projection, LDS-DMA staging -- is the pair kernel's.
template <int KIND> struct SplitKind;
template <> struct SplitKind<1> {
    typedef _Float16 T;
    typedef _Float16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ pf32x4 mfma(V8 a, V8 b, pf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct SplitKind<2> {
    typedef __bf16 T;
    typedef __bf16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ pf32x4 mfma(V8 a, V8 b, pf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
constexpr int CR_HLD = 272;   // bytes per row of a 16-bit h plane (128 columns + one 16-byte slot of padding)
// The two halves of an f32 value: hi = RN16(v), lo = RN16(v - hi).  The value is PINNED first: left to itself hipcc folds "(half)(a * b)" into
// v_fma_mixlo_f16 -- which rounds the EXACT product once, where a conversion of the f32 product rounds twice: one f16 ulp apart in 5e-5 of the
// cases, tools/ubench/mixlo_rounding -- and it does so at some uses of the value and not at others (also under -ffp-contract=off).  Two builds of
// this kernel stored a "hi" converted from the f32 value and subtracted a "hi" folded into the producing multiply: a rare one-ulp mismatch
// between an operand's halves, 2e-5 on the critic's value, with f16 only (bf16 has no such instruction).  Behind the pin every use sees ONE value.
template <typename HT> __device__ __forceinline__ void split16(float v, HT &hi, HT &lo) {
    asm volatile("" : "+v"(v));
    hi = (HT)v;
    lo = (HT)(v - (float)hi);
}

constexpr int CR_XROWS = (POL_MAXT + 1) * 16;   // rows of a block's frame planes: (T + 1) frames x 16 envs (a multiple of 16: the [k block][row] layout below is conflict-free)

// TACO_LSTM_SYNC (round 6, an A/B switch; the product builds 0): how the eight wavefronts of the split LSTM kernel meet between timesteps.
//   0  one workgroup barrier per timestep: every wavefront is in its MFMA chains, then in its cells, at the same time -- the two wavefronts of a SIMD run
//      at roughly the SUM of both phases (2.9 us per timestep against 2.4 priced serially and 1.5 if both pipes were always busy);
//   1  per-tile LDS counters instead (h_t of tile X is complete when its counter says so): no barrier inside the timestep loop;
//   2  = 1 + the cooperative VALU work of a block (splitting the next block's frames, flushing h_T) done by wavefronts 4-7 alone, which puts them one
//      segment BEHIND wavefronts 0-3, so that a SIMD's two wavefronts are in opposite phases.
// Why it was tried: tools/ubench/mfma16_cross_wave (profiles/r06_f_ubench_mfma16_cross_wave.txt) -- unlike f32 MFMAs, 1 024 v_mfma_f32_16x16x32_f16 of one
// wavefront and 4 096 v_fma of its SIMD partner finish in 29.1 k cycles, not in the 40.0 k of their sum (17.4 + 22.6): the partner's VALU issues under the
// 16-bit MFMAs at ~60 % of its own rate.  What the kernel did with it: bit-identical values, 0.287 (1) and 0.291 ms (2) per values_ring(37 x 4 096) against
// 0.278 (0), 1.195 / 1.235 against 1.163 ms at 557 056 rows (profiles/r06_g_ab_lstm_sync.txt): the counters' polling and the lost lock-step of the LDS
// traffic cost more than the overlap returns -- round 5's two schedules (barrier-separated ping-pong, in-wave interleave) had lost the same way.
#ifndef TACO_LSTM_SYNC
#define TACO_LSTM_SYNC 0
#endif
template <int KIND>
__global__ __launch_bounds__(64 * POL_NW) void taco_critic_lstm_pair_split_kernel(const PolicyParams P) {
    typedef SplitKind<KIND> SK;
    typedef typename SK::T HT;
    typedef typename SK::V8 V8;
    typedef typename SK::T V4 __attribute__((ext_vector_type(4)));
    typedef typename SK::T V2 __attribute__((ext_vector_type(2)));
    constexpr int FR = POL_MAXT + 1;
    // Frames: staged as f32 (LDS-DMA, as they lie in the ring) and split ONCE per block by the whole workgroup into two 16-bit planes -- in the
    // first form every wavefront split every frame it consumed, eight-fold redundant: 46 of the kernel's 277 us (profiles/r05_e_critic_split_probe.txt).
    // Plane layout [k block g][row][8 halves]: lane (r, g) of a B-operand read takes 16 B at slot g * CR_XROWS + row -- the 16 rows of a 16-lane
    // service group are 16 consecutive slots, and the groups' other k block lies a multiple of 16 slots away on the complementary rows: no conflicts.
    // Column sd of every row holds 1.0: the bias rides in the weights' column sd (sd <= 31; the host sends sd = 32 to the f32 kernel).
    __shared__ __attribute__((aligned(16))) float xst[FR * 16 * 32];                   // staging: [frame][env][sd] f32
    __shared__ __attribute__((aligned(16))) unsigned char xp[2][2][4 * CR_XROWS * 16]; // [block parity][high / low half][k block][row][8 x 16 bit]
    __shared__ __attribute__((aligned(16))) unsigned char hs[2][2][CR_ROWS * CR_HLD];  // h_t: [buffer][high / low half][row][k], 16-bit
    __shared__ __attribute__((aligned(16))) float hT[CR_ROWS * CR_LD];                 // h_T (f32, swizzled like the MLP's tiles) on its way out
    __shared__ int tile_cnt[2];   // TACO_LSTM_SYNC >= 1: cells() calls completed on tile A / B, summed over the wavefronts (monotonic over the kernel)
    constexpr int KSX = 2, KSH = 8, hp = 128, ip = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int T = P.states_len, sd = P.states_dim, N = P.ring_n;
    if (tid < 2) tile_cnt[tid] = 0;   // (ordered ahead of every use by the barriers of the first block's staging)
#ifdef TACO_LSTM_STAMPS   // A/B probe builds only: segment boundaries of wavefronts 0 and 4 of workgroup 0 (P.stamps: [2][128] shader-clock words)
    int stamp_no = 0;
    const bool stamping = P.stamps != nullptr && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0;
#define LSTM_STAMP() do { if (stamping && stamp_no < 128) P.stamps[(wave >> 2) * 128 + stamp_no++] = __builtin_readcyclecounter(); } while (0)
#else
#define LSTM_STAMP() do { } while (0)
#endif
    // One wavefront's LDS operations execute in order (h_t before the increment on the writer's side, the counter before h_t on the reader's), so the
    // fences only have to stop the compiler.  Bounded: a protocol bug ends in wrong values, never in a hang.
    auto post_tile = [&](int tile) __attribute__((always_inline)) {
        if constexpr (TACO_LSTM_SYNC >= 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_fetch_add(&tile_cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto wait_tile = [&](int tile, int want) __attribute__((always_inline)) {
        if constexpr (TACO_LSTM_SYNC >= 1) {
            int spins = 0;
            while (__hip_atomic_load(&tile_cnt[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    // the threads that do a block's cooperative work (frame split, h_T flush): all of them, or wavefronts 4-7 (SYNC 2)
    constexpr int COOP_T = TACO_LSTM_SYNC == 2 ? 64 * POL_NW / 2 : 64 * POL_NW;
    const bool coop = TACO_LSTM_SYNC == 2 ? wave >= POL_NW / 2 : true;
    const int ctid = TACO_LSTM_SYNC == 2 ? tid - 64 * POL_NW / 2 : tid;
    const int S = P.n / N;
    const int groups = (N + 15) / 16, pairs = (S + 1) / 2, nblocks = groups * pairs;
    const int nb = ((int)blockIdx.x < nblocks) ? (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (nb == 0) return;
    const float *w = P.blob;
    {
        int in_a = pad16(P.obs_len * P.obs_dim);
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const int out = pad16(l == P.n_actor_hidden ? P.act_dim : P.actor_hidden[l]);
            w += (size_t)out * in_a + out;
            in_a = out;
        }
        w += 16;  // log_std
    }
    const float *Wih = w, *Whh = w + (size_t)4 * hp * ip, *bs = Whh + (size_t)4 * hp * hp;
    const int col = wave * 16 + r;
    // The product is taken TRANSPOSED, D = W x^T: the weights are the A operand (lane (r, g): W[col 16 wave + r][k = 32 Sb + 8 g + j]), the
    // activations the B operand (lane (r, g): x[env row r][k = 32 Sb + 8 g + j]), and lane (r, g) of D holds hidden columns 16 wave + 4 g + i,
    // i = 0 .. 3, of env row r: a lane's four h values are ADJACENT in a row of h_t and leave as one 8-byte store per plane (the first form,
    // D = x W^T, had them in four rows: eight 2-byte stores).
    // Both halves of this wavefront's weights; the blob is fragment-major for the f32 form: float4 ((wave KS + s) 64 + 16 g' + r) holds
    // W[col][16 s + 4 g' .. + 3]; this lane's k = 32 Sb + 8 g + j lies in fragments s = 2 Sb + (g >> 1), g' = 2 (g & 1) + (j >> 2)
    V8 wxh[4], wxl[4], whh[4][4], whl[4][4];
    auto halves = [](const float (&v)[8], V8 &hi, V8 &lo) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { HT h, l; split16(v[j], h, l); hi[j] = h; lo[j] = l; }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        {
            const float4 *f = reinterpret_cast<const float4 *>(Wih + (size_t)q * hp * ip) + (size_t)(wave * KSX + (g >> 1)) * 64 + 32 * (g & 1) + r;
            const float4 a = f[0], b = f[16];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const float bias = bs[q * hp + col];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (8 * g + j == sd) ? bias : v[j];   // the bias column (the frames' column sd is 1.0)
            halves(v, wxh[q], wxl[q]);
        }
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const float4 *f = reinterpret_cast<const float4 *>(Whh + (size_t)q * hp * hp) + (size_t)(wave * KSH + 2 * sb + (g >> 1)) * 64 + 32 * (g & 1) + r;
            const float4 a = f[0], b = f[16];
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            halves(v, whh[q][sb], whl[q][sb]);
        }
    }
    const int ring_rows = S + T - 1;
    const int piece_bytes = 16 * sd * (int)sizeof(float);
    const bool dma_ok = (((size_t)N * sd * sizeof(float)) & 15u) == 0 && (piece_bytes & 15) == 0;
    auto stage = [&](int blk) {   // the block's T + 1 frames -> xst (f32)
        const int pair = blk / groups, e0 = (blk - pair * groups) * 16, slot0 = 2 * pair;
        const int per_piece = (piece_bytes / 16 + 63) / 64;
        if (dma_ok && e0 + 16 <= N && slot0 + T < ring_rows) {
            for (int j = wave; j < (T + 1) * per_piece; j += POL_NW) {
                const int f = j / per_piece, off = ((j - f * per_piece) * 64 + lane) * 16;
                const char *src = reinterpret_cast<const char *>(P.states) + ((size_t)(slot0 + f) * N + e0) * sd * sizeof(float);
                if (off < piece_bytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                     (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(xst) + f * piece_bytes + (j - f * per_piece) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int e = tid; e < (T + 1) * 16 * sd; e += 64 * POL_NW) {
                const int f = e / (16 * sd), rem = e - f * 16 * sd, env = rem / sd;
                xst[e] = (slot0 + f < ring_rows && e0 + env < N) ? P.states[((size_t)(slot0 + f) * N + e0) * sd + rem] : 0.0f;
            }
        }
    };
    auto split_frames = [&](int half, bool everyone) {   // xst -> the two planes of xp[half]: every thread pairs of adjacent columns (one 4-byte store per plane)
        if (!everyone && !coop) return;
        const int e0_ = everyone ? tid : ctid, stride_ = everyone ? 64 * POL_NW : COOP_T;
#pragma unroll 1
        for (int e = e0_; e < (T + 1) * 16 * 16; e += stride_) {
            const int row = e >> 4, k = (e & 15) * 2;
            float v0 = k < sd ? xst[row * sd + k] : (k == sd ? 1.0f : 0.0f);
            float v1 = k + 1 < sd ? xst[row * sd + k + 1] : (k + 1 == sd ? 1.0f : 0.0f);
            if constexpr (KIND == 1) {
                // f16's range.  NaN stays NaN and +-inf BECOMES NaN (the f32 kernel propagates both: a diverged env must not come out with a
                // plausible value -- round 5's advisor); a huge FINITE word saturates at +-65 504 (it would otherwise turn into inf - inf) and
                // is COUNTED: the workspace's tail word (taco_critic_clamped_words) and, in a rollout, the env's status word (taco_check)
                const float a0 = __builtin_fabsf(v0), a1 = __builtin_fabsf(v1);
                if (a0 > 65504.0f || a1 > 65504.0f) {   // (false for NaN)
                    const uint32_t n0 = a0 > 65504.0f && a0 < __builtin_inff(), n1 = a1 > 65504.0f && a1 < __builtin_inff();
                    if (a0 > 65504.0f) v0 = n0 ? __builtin_copysignf(65504.0f, v0) : __builtin_nanf("");
                    if (a1 > 65504.0f) v1 = n1 ? __builtin_copysignf(65504.0f, v1) : __builtin_nanf("");
                    if (n0 | n1) {   // (clamp EVENTS: a frame is staged by every block of two slots whose stacks hold it, up to (T + 2) / 2 of them)
                        if (P.clamped) atomicAdd(P.clamped, n0 + n1);
                        if (P.env_status) atomicOr(P.env_status, kStatusCriticClampedBit);
                    }
                }
            }
            HT h0, h1, l0, l1;
            split16(v0, h0, l0);
            split16(v1, h1, l1);
            const int at = (((k >> 3) * CR_XROWS + row) << 4) + (k & 7) * 2;
            *reinterpret_cast<V2 *>(xp[half][0] + at) = V2{h0, h1};
            *reinterpret_cast<V2 *>(xp[half][1] + at) = V2{l0, l1};
        }
    };
    stage((int)blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    split_frames(0, true);
    __syncthreads();

    // ONE accumulator set for the three chains of a product chain (third form, same round: gfx950's 16-bit MFMAs KEEP subnormal operands --
    // tools/ubench/mfma_f16_denorm, profiles/r05_l_ubench_mfma_f16_denorm.txt -- so the low halves need no scale, the cross terms no accumulators of
    // their own and no "am += ac / S" pass: 32 VALU instructions per tile and timestep and 16 registers less, which is where tile B's x part
    // -- tile A's start of the next timestep -- lives again instead of being parked in LDS).
    pf32x4 am[4], keep[4];
    float cstA[4], cstB[4];
    // (fetching the frame's planes a cells() call ahead of this chain was tried with the h chain's prefetch below: no gain, profiles/r06_s_ab_hchain_prefetch.txt)
    auto x_chain = [&](const unsigned char *xh, int f) __attribute__((always_inline)) {   // am = bias + W_ih x(frame f) (combined), rows = the block's 16 envs
        const int at = (g * CR_XROWS + f * 16 + r) << 4;
        const V8 ah = *reinterpret_cast<const V8 *>(xh + at), al = *reinterpret_cast<const V8 *>(xh + 4 * CR_XROWS * 16 + at);
#pragma unroll
        for (int q = 0; q < 4; ++q) am[q] = SK::mfma(wxl[q], ah, pf32x4{0.0f, 0.0f, 0.0f, 0.0f});   // (the small terms first)
#pragma unroll
        for (int q = 0; q < 4; ++q) am[q] = SK::mfma(wxh[q], al, am[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) am[q] = SK::mfma(wxh[q], ah, am[q]);
    };
    auto h_chain = [&](const unsigned char *hprev, int tile) __attribute__((always_inline)) {   // am (= the x part) += W_hh h_{t-1}
        const unsigned char *rowh = hprev + (16 * tile + r) * CR_HLD, *rowl = rowh + CR_ROWS * CR_HLD;
#ifndef TACO_AB_HCHAIN_NOPREFETCH
        // the NEXT k block's two planes are on their way while this block's twelve MFMAs issue (round 6: left to itself the compiler read a block's
        // operands right ahead of its MFMAs and waited -- ~80 clocks of LDS latency per k block on the chain's path: 1 090 -> 970 clocks for the 48 MFMAs, the timestep
        // 6 180 -> 5 970; values_ring(37 x 4 096) 0.271 -> 0.268 ms, the critic at 557 056 rows 1.165 -> 1.135 ms, bit-identical: profiles/r06_s_ab_hchain_prefetch.txt)
        const int pos0 = (g ^ cr_b(r)) << 4;
        V8 nh = *reinterpret_cast<const V8 *>(rowh + pos0), nl = *reinterpret_cast<const V8 *>(rowl + pos0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const V8 ah = nh, al = nl;
            if (sb + 1 < 4) {
                const int pos = ((4 * (sb + 1) + g) ^ cr_b(r)) << 4;
                nh = *reinterpret_cast<const V8 *>(rowh + pos); nl = *reinterpret_cast<const V8 *>(rowl + pos);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = SK::mfma(whl[q][sb], ah, am[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = SK::mfma(whh[q][sb], al, am[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = SK::mfma(whh[q][sb], ah, am[q]);
            if (sb + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the next block's two reads FIRST ...
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                  // ... then this block's twelve MFMAs
        }
#else
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const int pos = ((4 * sb + g) ^ cr_b(r)) << 4;
            const V8 ah = *reinterpret_cast<const V8 *>(rowh + pos), al = *reinterpret_cast<const V8 *>(rowl + pos);
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = SK::mfma(whl[q][sb], ah, am[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = SK::mfma(whh[q][sb], al, am[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = SK::mfma(whh[q][sb], ah, am[q]);
        }
#endif
    };
    // this lane's cells: env row 16 tile + r, hidden columns c0 .. c0 + 3
    const int c0 = 16 * wave + 4 * g;
    auto cells = [&](int t, int tile, float (&cst)[4], bool first) __attribute__((always_inline)) {
        const bool last = t + 1 >= T;
        const int row = 16 * tile + r;
        float hv[4];
        // (two cells per call on gfx950's packed f32 multiply / add / fma -- 73 VALU instructions fewer per timestep, bit-identical -- bought nothing:
        // 0.278 vs 0.278 ms, profiles/r06_n_ab_packed_cell.txt; v_pk_*_f32 issue at half rate, tools/ubench/mfma_valu "v_pk_fma_f32 | same")
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float cn;
            lstm_cell_fast(am[0][i], am[1][i], am[2][i], am[3][i], first ? 0.0f : cst[i], cn, hv[i]);
            cst[i] = cn;
        }
        if (last) {   // (ONE wave-uniform branch per call)
            *reinterpret_cast<float4 *>(hT + cr_sw4(row, c0 >> 2)) = float4{hv[0], hv[1], hv[2], hv[3]};
        } else {
            unsigned char *oh = hs[t & 1][0] + row * CR_HLD + (((c0 >> 3) ^ cr_b(row)) << 4) + (c0 & 7) * 2;
            V4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { HT h, l; split16(hv[i], h, l); hi[i] = h; lo[i] = l; }
            *reinterpret_cast<V4 *>(oh) = hi;
            *reinterpret_cast<V4 *>(oh + CR_ROWS * CR_HLD) = lo;
        }
    };
    auto flush_hT = [&](int blk, bool everyone) {
        if (!everyone && !coop) return;
        const int pair = blk / groups, e0 = (blk - pair * groups) * 16;
        const int e0_ = everyone ? tid : ctid, stride_ = everyone ? 64 * POL_NW : COOP_T;
#pragma unroll 1
        for (int e = e0_; e < CR_ROWS * (hp / 4); e += stride_) {
            const int row = e >> 5, k4 = e & 31;
            const int slot = 2 * pair + (row >> 4), env = e0 + (row & 15);
            if (slot < S && env < N)
                *reinterpret_cast<float4 *>(P.hT + ((size_t)slot * N + env) * hp + 4 * k4) = *reinterpret_cast<const float4 *>(hT + cr_sw4(row, k4));
        }
    };
    for (int k = 0; k < nb; ++k) {
        const int blk = (int)blockIdx.x + k * (int)gridDim.x;
        const unsigned char *xh = xp[k & 1][0];
        if (k + 1 < nb) stage(blk + (int)gridDim.x);   // (xst is free: its frames were split into xp[k & 1] a block ago)
        if constexpr (TACO_LSTM_SYNC >= 1) {
            // the previous block is COMPLETE in every wavefront (its last cells wrote h_T, and its last chains read the h planes timestep 0 is about to
            // overwrite): what the barrier behind its last timestep said
            if (k > 0) { wait_tile(0, 8 * k * T); wait_tile(1, 8 * k * T); }
        }
        if (k > 0) flush_hT(blk - (int)gridDim.x, TACO_LSTM_SYNC != 2);
        // ---- timestep 0: x chains only (h_{-1} = 0)
        x_chain(xh, 0);
        cells(0, 0, cstA, true);
        post_tile(0);
        x_chain(xh, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] = am[q];
        cells(0, 1, cstB, true);
        post_tile(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next block's frames have landed
        __syncthreads();
        if (k + 1 < nb) split_frames((k + 1) & 1, TACO_LSTM_SYNC != 2);   // ... and become its planes (read from the next block on: T - 1 >= 1 barriers / counter waits lie in between)
        for (int t = 1; t < T; ++t) {
            const unsigned char *hprev = hs[(t - 1) & 1][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = keep[q];        // tile A: bias + W_ih x(frame t), computed for tile B a timestep ago
            wait_tile(0, 8 * (k * T + t));                       // h_{t-1} of tile A is complete (all eight wavefronts' columns)
            LSTM_STAMP();
            h_chain(hprev, 0);
            LSTM_STAMP();
            cells(t, 0, cstA, false);
            post_tile(0);
            LSTM_STAMP();
            x_chain(xh, t + 1);                                // tile B: frame t + 1
#pragma unroll
            for (int q = 0; q < 4; ++q) keep[q] = am[q];
            wait_tile(1, 8 * (k * T + t));
            LSTM_STAMP();
            h_chain(hprev, 1);
            LSTM_STAMP();
            cells(t, 1, cstB, false);
            post_tile(1);
            LSTM_STAMP();
            if constexpr (TACO_LSTM_SYNC == 0) __syncthreads();
            LSTM_STAMP();
        }
    }
    if constexpr (TACO_LSTM_SYNC >= 1) { wait_tile(0, 8 * nb * T); wait_tile(1, 8 * nb * T); }
    flush_hT((int)blockIdx.x + (nb - 1) * (int)gridDim.x, true);
}

// value = W3 relu(W2 relu(W1 h_T + b1) + b2) + b3 over the workspace rows (critic_body's dense layers; same chains, same bits)
__global__ __launch_bounds__(64 * POL_NW, 4) void taco_critic_mlp_kernel(const PolicyParams P) {
    __shared__ __attribute__((aligned(16))) float xb[CR_MLP_ROWS * CR_LD];
    __shared__ __attribute__((aligned(16))) float yb[CR_MLP_ROWS * CR_LD];
    constexpr int hp = 128, ip = 32, KS = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const float *w = P.blob;
    {
        int in_a = pad16(P.obs_len * P.obs_dim);
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const int out = pad16(l == P.n_actor_hidden ? P.act_dim : P.actor_hidden[l]);
            w += (size_t)out * in_a + out;
            in_a = out;
        }
        w += 16 + (size_t)4 * hp * ip + (size_t)4 * hp * hp + (size_t)4 * hp;
    }
    const float *W1 = w, *b1 = W1 + hp * hp, *W2 = b1 + hp, *b2 = W2 + hp * hp, *W3 = b2 + hp, *b3 = W3 + 16 * hp;
    const int col = wave * 16 + r;
    float4 w1[KS], w2[KS], w3[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        w1[s] = *reinterpret_cast<const float4 *>(W1 + ((size_t)(wave * KS + s) * 64 + lane) * 4);
        w2[s] = *reinterpret_cast<const float4 *>(W2 + ((size_t)(wave * KS + s) * 64 + lane) * 4);
        w3[s] = *reinterpret_cast<const float4 *>(W3 + ((size_t)s * 64 + lane) * 4);
    }
    const float c1 = b1[col], c2 = b2[col], c3 = b3[r];
    auto layer = [&](const float *in, float *out, const float4 (&wf)[KS], float bias) {
#pragma unroll
        for (int rt = 0; rt < CR_MLP_ROWS / 16; ++rt) {
            const float *arow = in + (16 * rt + r) * CR_LD + 4 * (g ^ cr_b(r));   // (swizzled tiles: cr_sw4)
            pf32x4 acc = {bias, bias, bias, bias};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * s);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wf[s].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wf[s].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wf[s].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wf[s].w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(16 * rt + 4 * g + i) * CR_LD + (col ^ (cr_b(4 * g) << 2))] = acc[i] < 0.0f ? 0.0f : acc[i];
        }
    };
    const int nchunks = (P.n + CR_MLP_ROWS - 1) / CR_MLP_ROWS;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t row0 = (size_t)c * CR_MLP_ROWS;
        for (int e = tid; e < CR_MLP_ROWS * (hp / 4); e += 64 * POL_NW) {  // 64 rows x 32 float4
            const int row = e >> 5, k4 = e & 31;
            const float4 v = (row0 + row < (size_t)P.n) ? *reinterpret_cast<const float4 *>(P.hT + (row0 + row) * hp + 4 * k4) : float4{0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<float4 *>(xb + cr_sw4(row, k4)) = v;
        }
        __syncthreads();
        layer(xb, yb, w1, c1);
        __syncthreads();
        layer(yb, xb, w2, c2);
        __syncthreads();
        if (wave < CR_MLP_ROWS / 16) {  // the 128 -> 1 head (padded to a 16-column tile): wavefront w takes row tile w
            const float *arow = xb + (16 * wave + r) * CR_LD + 4 * (g ^ cr_b(r));
            pf32x4 acc = {c3, c3, c3, c3};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * s);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w3[s].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w3[s].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w3[s].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w3[s].w, acc, 0, 0, 0);
            }
            if (r == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t row = row0 + 16 * wave + 4 * g + i;
                    if (row < (size_t)P.n) {
                        if (row < (size_t)P.value_split) P.value[row] = acc[i];
                        else P.value_tail[row - (size_t)P.value_split] = acc[i];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// The same MLP on the 16-bit matrix pipe with split operands (TACO_P_SPLIT_F16 / _BF16: the split LSTM kernel's companion).  After the LSTM moved to
// v_mfma_f32_16x16x32_f16 the f32 MLP above was a quarter of the critic's time (88 of 330 us at 135 168 rows, 101 TFLOP/s of f32 MFMA); here every
// product chain of the two hidden layers is three 16-bit chains into one f32 accumulator, as in taco_critic_lstm_pair_split_kernel: 12 MFMAs of 16
// cycles per 16-row tile and layer where the f32 form issues 32 of 32 cycles.  The product is taken TRANSPOSED (D = W x^T: weights = A operand,
// activations = B operand), so a lane's four outputs are adjacent columns of one row and leave as one 8-byte store per plane; activations travel
// from layer 1 to layer 2 as two 16-bit planes laid out like the LSTM's h_t (row stride CR_HLD, slot c of row r at c ^ b(r): conflict-free
// ds_read_b128).  Every value is split behind split16's pin.  ReLU outputs are clamped to f16's largest finite value in the f16 form (they are O(1);
// the clamp only keeps inf - inf out of the low half).  The 128 -> 1 head stays in f32 on the VALU, straight from layer 2's accumulators (a lane holds
// four columns of a row: four fma, two cross-lane adds, one partial sum per wavefront and row through LDS) -- layer 2's output is neither split nor
// stored, and the head costs no MFMA tile of which one row in sixteen is real.
#ifndef TACO_AB_MLP_MINW
#define TACO_AB_MLP_MINW 4
#endif
template <int KIND>
__global__ __launch_bounds__(64 * POL_NW, TACO_AB_MLP_MINW) void taco_critic_mlp_split_kernel(const PolicyParams P) {
    typedef SplitKind<KIND> SK;
    typedef typename SK::T HT;
    typedef typename SK::V8 V8;
    typedef typename SK::T V4 __attribute__((ext_vector_type(4)));
    constexpr int hp = 128, ip = 32, ROWS = CR_MLP_ROWS, PLANE = ROWS * CR_HLD;
    __shared__ __attribute__((aligned(16))) unsigned char ap[2][2][PLANE];   // [h_T / layer 1's output][high / low half][row][k], 16-bit
    __shared__ float part[POL_NW][ROWS];                                      // the head's partial sums: [wavefront = 16 columns][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const float *w = P.blob;
    {
        int in_a = pad16(P.obs_len * P.obs_dim);
        for (int l = 0; l <= P.n_actor_hidden; ++l) {
            const int out = pad16(l == P.n_actor_hidden ? P.act_dim : P.actor_hidden[l]);
            w += (size_t)out * in_a + out;
            in_a = out;
        }
        w += 16 + (size_t)4 * hp * ip + (size_t)4 * hp * hp + (size_t)4 * hp;
    }
    const float *W1 = w, *b1 = W1 + hp * hp, *W2 = b1 + hp, *b2 = W2 + hp * hp, *W3 = b2 + hp, *b3 = W3 + 16 * hp;
    // this wavefront's weights as A operands: lane (r, g) holds W[out 16 wave + r][k = 32 sb + 8 g + j] (the blob is fragment-major for the f32 form:
    // float4 ((tile 8 + s) 64 + 16 g' + r) = W[16 s + 4 g' .. + 3][out 16 tile + r]; see the split LSTM kernel)
    V8 w1h[4], w1l[4], w2h[4], w2l[4];
    auto frag = [&](const float *W, int tile, int sb, V8 &hi, V8 &lo) {
        const float4 *f = reinterpret_cast<const float4 *>(W) + (size_t)(tile * 8 + 2 * sb + (g >> 1)) * 64 + 32 * (g & 1) + r;
        const float4 a = f[0], b = f[16];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { HT h, l; split16(v[j], h, l); hi[j] = h; lo[j] = l; }
    };
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) { frag(W1, wave, sb, w1h[sb], w1l[sb]); frag(W2, wave, sb, w2h[sb], w2l[sb]); }
    const int c0 = 16 * wave + 4 * g;   // this lane's output columns c0 .. c0 + 3 (of env row r of a tile)
    float c1[4], c2[4], hw[4];          // biases of its columns; the head's weights W3[k = c0 + i][column 0] (fragment s = c0 / 16 = wave, g' = g, r = 0)
    {
        const float4 h4 = reinterpret_cast<const float4 *>(W3)[(size_t)wave * 64 + 16 * g];
        hw[0] = h4.x; hw[1] = h4.y; hw[2] = h4.z; hw[3] = h4.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { c1[i] = b1[c0 + i]; c2[i] = b2[c0 + i]; }
    const float hb = b3[0];
    // HEAD false: relu, split, store the two planes of the next layer's input.  HEAD true: relu, this lane's share of the head, reduced over the
    // wavefront's sixteen columns (lanes r, 16 + r, 32 + r, 48 + r), one partial per row
    auto layer = [&](auto head_c, const unsigned char *in, unsigned char *out, const V8 (&wh)[4], const V8 (&wl)[4], const float (&bias)[4]) __attribute__((always_inline)) {
        constexpr bool HEAD = decltype(head_c)::value;
        constexpr int TG = 2;   // row tiles in flight per wavefront (independent accumulator chains)
#pragma unroll
        for (int r0 = 0; r0 < ROWS / 16; r0 += TG) {
            pf32x4 acc[TG];
#pragma unroll
            for (int u = 0; u < TG; ++u) acc[u] = pf32x4{bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                const int pos = ((4 * sb + g) ^ cr_b(r)) << 4;
                V8 ah[TG], al[TG];
#pragma unroll
                for (int u = 0; u < TG; ++u) {
                    ah[u] = *reinterpret_cast<const V8 *>(in + (16 * (r0 + u) + r) * CR_HLD + pos);
                    al[u] = *reinterpret_cast<const V8 *>(in + PLANE + (16 * (r0 + u) + r) * CR_HLD + pos);
                }
#pragma unroll
                for (int u = 0; u < TG; ++u) acc[u] = SK::mfma(wl[sb], ah[u], acc[u]);   // (the small terms first)
#pragma unroll
                for (int u = 0; u < TG; ++u) acc[u] = SK::mfma(wh[sb], al[u], acc[u]);
#pragma unroll
                for (int u = 0; u < TG; ++u) acc[u] = SK::mfma(wh[sb], ah[u], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < TG; ++u) {
                const int row = 16 * (r0 + u) + r;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[u][i] < 0.0f ? 0.0f : acc[u][i];
                if constexpr (HEAD) {
                    float ps = v[0] * hw[0];
                    ps = fma(v[1], hw[1], ps); ps = fma(v[2], hw[2], ps); ps = fma(v[3], hw[3], ps);
                    ps += __shfl_xor(ps, 16);
                    ps += __shfl_xor(ps, 32);
                    if (g == 0) part[wave][row] = ps;
                } else {
                    unsigned char *oh = out + row * CR_HLD + (((c0 >> 3) ^ cr_b(row)) << 4) + (c0 & 7) * 2;
                    V4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float x = v[i];
                        if constexpr (KIND == 1) x = x > 65504.0f ? 65504.0f : x;
                        HT h, l;
                        split16(x, h, l);
                        hi[i] = h; lo[i] = l;
                    }
                    *reinterpret_cast<V4 *>(oh) = hi;
                    *reinterpret_cast<V4 *>(oh + PLANE) = lo;
                }
            }
        }
    };
    // a pass = 64 rows of h_T: every thread two pieces of eight columns (32 coalesced bytes each), fetched a pass ahead into registers
    const int nchunks = (P.n + ROWS - 1) / ROWS;
    float4 pre[4];
    auto fetch = [&](int c) {
        const size_t row0 = (size_t)c * ROWS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 64 * POL_NW * j, row = e >> 4, slot = e & 15;
            const bool ok = row0 + row < (size_t)P.n;
            const float4 *src = reinterpret_cast<const float4 *>(P.hT + (row0 + row) * hp + 8 * slot);
            pre[2 * j] = ok ? src[0] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            pre[2 * j + 1] = ok ? src[1] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 64 * POL_NW * j, row = e >> 4, slot = e & 15;
            const float v[8] = {pre[2 * j].x, pre[2 * j].y, pre[2 * j].z, pre[2 * j].w, pre[2 * j + 1].x, pre[2 * j + 1].y, pre[2 * j + 1].z, pre[2 * j + 1].w};
            V8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) { HT h, l; split16(v[q], h, l); hi[q] = h; lo[q] = l; }
            unsigned char *at = ap[0][0] + row * CR_HLD + ((slot ^ cr_b(row)) << 4);
            *reinterpret_cast<V8 *>(at) = hi;
            *reinterpret_cast<V8 *>(at + PLANE) = lo;
        }
    };
    if ((int)blockIdx.x < nchunks) fetch((int)blockIdx.x);
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t row0 = (size_t)c * ROWS;
        put();            // (ap[0] was last read by layer 1 of the previous pass: two barriers ago)
        __syncthreads();
        if (c + (int)gridDim.x < nchunks) fetch(c + (int)gridDim.x);
        layer(std::false_type{}, ap[0][0], ap[1][0], w1h, w1l, c1);
        __syncthreads();
        layer(std::true_type{}, ap[1][0], nullptr, w2h, w2l, c2);
        __syncthreads();
        if (tid < ROWS) {   // value = b3 + the eight wavefronts' partial sums, in wavefront order (`part` is next written behind the next pass's second barrier)
            float val = hb;
#pragma unroll
            for (int wv_ = 0; wv_ < POL_NW; ++wv_) val += part[wv_][tid];
            const size_t row = row0 + tid;
            if (row < (size_t)P.n) {
                if (row < (size_t)P.value_split) P.value[row] = val;
                else P.value_tail[row - (size_t)P.value_split] = val;
            }
        }
    }
}


// The actor alone at LARGE N (taco_policy_act / taco_rollout_run above 8 192 envs): the 16-row form above is built for latency (256
// workgroups at 4 096 envs, one per CU: 9.6 us) and runs at 25 % of the f32 MFMA peak from 16 384 envs on; this form gives a wavefront
// 16 output columns of every layer (tile = wave; fragments of 2 + 8 + 8 + 8 float4), takes 64 rows per pass (four 16-row MFMA tiles per
// weight fragment) and walks passes b, b + grid, ...  Same chains in the same k order as dense_tile: same bits.
// requirements (host): the documented actor -- pad16(obs_len * obs_dim) == 32, three hidden layers padded to 128, act_dim == 4
// (End of round 5: at most 128 registers, so that TWO workgroups share a CU and one's staging / sampling / barrier waits lie under the other's MFMAs.  The
// first build kept all four layers' fragments in registers -- 104 + working set = 193, one workgroup per CU -- although its launches give a workgroup one
// or two passes: 33.1 us at 32 768 rows = 49 % of the f32 MFMA peak, a quarter of config 5's rollout at its per-rank shape.  Here a layer's fragments are
// fetched (from L2: 26 KB per wavefront and pass) behind the layer before it, ahead of the barrier: the other workgroup's MFMAs cover the fetch.)
constexpr int ACT_ROWS = CR_MLP_ROWS;   // rows per pass of taco_actor_kernel (32-row passes, four workgroups per CU wanted, two resident at 128 registers: 37.6 us against 31.2)
#ifndef TACO_AB_ACTOR_MINW
#define TACO_AB_ACTOR_MINW 4
#endif
__global__ __launch_bounds__(64 * POL_NW, TACO_AB_ACTOR_MINW) void taco_actor_kernel(const PolicyParams P) {
    __shared__ __attribute__((aligned(16))) float xb[ACT_ROWS * CR_LD];
    __shared__ __attribute__((aligned(16))) float yb[ACT_ROWS * CR_LD];
    constexpr int hp = 128, ip = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const uint32_t call = P.clock ? P.clock[0] + P.call_delta : P.call;
    const float *W0 = P.blob, *b0 = W0 + hp * ip, *W1 = b0 + hp, *b1 = W1 + hp * hp, *W2 = b1 + hp, *b2 = W2 + hp * hp, *W3 = b2 + hp, *b3 = W3 + 16 * hp;
    const float *log_std = b3 + 16;
    const int col = wave * 16 + r;
    float4 w0[2], wA[8];
    auto fetch8 = [&](float4 (&w)[8], const float *W, int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) w[s] = *reinterpret_cast<const float4 *>(W + ((size_t)(tile * 8 + s) * 64 + lane) * 4);
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) w0[s] = *reinterpret_cast<const float4 *>(W0 + ((size_t)(wave * 2 + s) * 64 + lane) * 4);
    const float c0 = b0[col], c1 = b1[col], c2 = b2[col], c3 = b3[r];
    auto layer = [&](const float *in, float *out, const float4 *wf, int ks, float bias) {
#pragma unroll
        for (int rt = 0; rt < ACT_ROWS / 16; ++rt) {
            const float *arow = in + (16 * rt + r) * CR_LD + 4 * (g ^ cr_b(r));   // (swizzled tiles: cr_sw4)
            pf32x4 acc = {bias, bias, bias, bias};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < ks) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * s);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wf[s].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wf[s].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wf[s].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wf[s].w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(16 * rt + 4 * g + i) * CR_LD + (col ^ (cr_b(4 * g) << 2))] = acc[i] < 0.0f ? 0.0f : acc[i];
        }
    };
    const int kin = P.obs_len * P.obs_dim;
    const int npass = (P.n + ACT_ROWS - 1) / ACT_ROWS;
    for (int c = blockIdx.x; c < npass; c += gridDim.x) {
        const int row0 = c * ACT_ROWS;
        for (int e = tid; e < ACT_ROWS * ip; e += 64 * POL_NW) {  // obs [row][kin] -> xb[row][32], zero beyond kin and beyond the last row
            const int row = e >> 5, k = e & 31;
            xb[cr_sw4(row, k >> 2) + (k & 3)] = (row0 + row < P.n && k < kin) ? P.obs[(size_t)(row0 + row) * kin + k] : 0.0f;
        }
        fetch8(wA, W1, wave);
        __syncthreads();
        layer(xb, yb, w0, 2, c0);
        __syncthreads();
        layer(yb, xb, wA, 8, c1);
        fetch8(wA, W2, wave);
        __syncthreads();
        layer(xb, yb, wA, 8, c2);
        fetch8(wA, W3, 0);
        __syncthreads();
        const float4 (&w3)[8] = wA;
        if (wave < ACT_ROWS / 16) {  // the 128 -> 4 head (one 16-column tile) with its tanh: wavefront w takes row tile w
            const float *arow = yb + (16 * wave + r) * CR_LD + 4 * (g ^ cr_b(r));
            pf32x4 acc = {c3, c3, c3, c3};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * s);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w3[s].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w3[s].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w3[s].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w3[s].w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) xb[(16 * wave + 4 * g + i) * CR_LD + r] = tanh_own(acc[i]);   // (the head's output, read by sample4: plain rows)
        }
        __syncthreads();
        if (tid < 4 * ACT_ROWS) sample4(P, xb, CR_LD, row0, tid >> 2, tid & 3, log_std, call);
        __syncthreads();  // xb is staged again by the next pass
    }
}

}  // namespace taco
