// taco_fused.hpp -- SURVEY 8f row N1 "fused with env step on-device": the per-step chain of one PPO rollout (ppo_asymmetry.py:308-342:
// agent.act -> clip -> env.step -> store) as ONE PERSISTENT kernel.
//
// A workgroup owns 16 envs for all `horizon` steps (envs never interact and every random stream is counter-based, so no grid-wide
// synchronisation exists anywhere).  Eight wavefronts, two per SIMD of the CU the workgroup has to itself:
//     wave 0  SIMD 0   the step wavefront          (step_core role 0: reset, delay line, the ten substeps, state stores)
//     wave 1  SIMD 1   battery server, then reward / done                                           (role 1)
//     wave 4  SIMD 0   states-frame role: the newest frame -> the replay store's frame ring         (role 3; asleep during the substeps)
//     wave 5  SIMD 1   obs-frame role: observation noise, frame -> obs_store[t + 1] AND the actor's input rows in LDS   (role 2)
//     waves 2, 6 (SIMD 2) and 3, 7 (SIMD 3)   the ACTOR: MLP 26-128-128-128-4 on f32 MFMA, weights of two 16-column tiles per wavefront
//                      register-resident for the whole rollout, tanh head + sampling + log-prob by the first of them.
// f32 MFMAs and VALU instructions of one SIMD do not overlap (tools/ubench/mfma_valu), so the actor keeps off the SIMDs of the step wavefront
// and of the battery server, whose instruction streams ARE the step's critical path.
// What makes the fusion pay: the action of step t reaches the dynamics only `delay_time` (>= 10 ms in every documented configuration) later,
// so the ten substeps of step t consume OLDER actions and run CONCURRENTLY with the actor computing a_t from obs_t; a_t is needed only for the
// bookkeeping after the loop (history row, frame words 19..22).  Per step the workgroup executes two barriers (the two of the four-role
// form); the actor's wavefronts synchronise among themselves through an LDS counter.  Envs with fewer than ten pending slots, or in the
// overflow regime of the delay line, make their step wavefront wait for the actor first (step_core `fetch_action`).
// Same arithmetic as the launch-per-step path -- step_core IS the step kernel's body, the MFMA chains are dense_tile's in the same k order,
// sample4 is shared -- so a rollout is bit-identical to taco_policy_act / taco_step_rollout per step (tests/test_rollout_gpu.py).
//
// Round 6, LPE = 1 (`taco_rollout_kernel<1>`; opt-in: taco_set_rollout_fusion(env, 3)): the same kernel with ONE LANE PER ENV -- a workgroup owns 64 envs, its
// step wavefront is the one-lane four-role form's (step_core<256, 1, true, ..., FUSED>: battery inline, rotor-noise table and euler angles served by the role
// wavefronts), and the actor runs FOUR 16-row tiles per layer against the same register-resident weights; each actor wavefront samples one row tile (the
// 128 -> 4 head's weights are resident in all four), so the action mailbox is complete after four posts.  Built for launches above 8 192 envs (512 workgroups
// at 32 768 envs: two rounds of the chip instead of the quad form's eight), bit-identical to the other paths -- and measured SLOWER than launch-per-step there
// (2.40 vs 2.05 ms per rollout at 32 768 x 16): a step takes 71 k clocks because 64 rows of f32 MFMAs (77.8 k matrix-pipe cycles) share the two SIMDs the
// step and battery wavefronts leave free.  Not the host's choice anywhere; kept for the A/B record (profiles/r06_b_rollout_ab_lane.txt, tools/rollout_ab.py).
#pragma once
#include "taco_policy.hpp"
#include "taco_step.hpp"

namespace taco {

constexpr int FU_LD = 132;      // activation row stride in floats (width 128 + 4: 8 consecutive rows cover all 32 banks)
constexpr int FU_XLD = 36;      // input row stride (26 obs words padded to 32, + 4)
constexpr int FU_THREADS = 512;

struct RolloutParams {
    StepParams S;          // the env's parameters as launch_step prepares them (state / history / ring pointers, flags, constants); the
                           // per-step output pointers below replace its obs / states / rew / ... members
    const float *blob;     // packed policy weights (taco_amd/policy.py::pack_state_dict): the documented actor at its head
    uint32_t pseed_lo, pseed_hi, call0;
    uint32_t call_delta;   // captured launch (S.use_ctl): the actor's noise counter of step t is (the device-resident step word) + call_delta + t
    float act_lo, act_hi;
    float *obs_store;      // [horizon + 1][n][26]
    float *frames;         // [horizon + len_states][n][26]   the states frame ring
    float *act_buf, *mu_buf, *sigma_buf;   // [horizon][n][4]
    float *logp_buf, *rew_buf, *done_buf;  // [horizon][n]
    uint8_t *timeout_buf;                  // [horizon][n]
    int horizon, len_states;
    unsigned long long *stamps;  // optional [8 + 2 * 64 + workgroups]: workgroup 0: [w] = SIMD of wavefront w; [8 + 2 t], [9 + 2 t] = shader clock of the step
                                 // wavefront at the start / end of step t (t < 64); [136 + b] = clocks workgroup b's step wavefront spent in its loop
};

template <int LPE>
__global__ __launch_bounds__(FU_THREADS) void taco_rollout_kernel(const RolloutParams R) {
    static_assert(LPE == 4 || LPE == 1, "four lanes per env (16 envs per workgroup) or one (64 envs)");
    constexpr int ROWS = 64 / LPE, RT = ROWS / 16;   // envs per workgroup, 16-row tiles of the actor
    __shared__ __attribute__((aligned(16))) float xin[ROWS * FU_XLD];   // the actor's input: obs_t of the workgroup's envs
    __shared__ __attribute__((aligned(16))) float xa[ROWS * FU_LD];
    __shared__ __attribute__((aligned(16))) float xb[ROWS * FU_LD];
    __shared__ float act_lds[ROWS * 4];
    __shared__ int act_seq, a_cnt, reset_seq, reset_lds[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = R.S.n, row0 = blockIdx.x * ROWS;
    if (row0 >= n) return;
    const int role = wave == 0 ? 0 : wave == 1 ? 1 : wave == 5 ? 2 : wave == 4 ? 3 : -1;          // step_core role of this wavefront, or
    const int actor = wave == 2 ? 0 : wave == 6 ? 1 : wave == 3 ? 2 : wave == 7 ? 3 : -1;        // which of the four actor wavefronts it is
    if (R.stamps && blockIdx.x == 0 && lane == 0) R.stamps[wave] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3u;   // HW_ID.SIMD_ID
    // the step clock, as the step kernel takes it: kernel arguments, or the device-resident copy when this launch was captured into a graph
    // (the workgroup's own pair of the per-16-env clock, taco_step.hpp "wclk": read here, rewritten by this workgroup when the rollout is done)
    const __attribute__((address_space(4))) uint32_t *wc =
        (const __attribute__((address_space(4))) uint32_t *)(R.S.wclk + (size_t)blockIdx.x * RT * kWclkWordsPerGroup);   // (one pair per 16 envs: the first of this workgroup's)
    const bool from_ctl = R.S.use_ctl != 0;
    const uint32_t c_step = wc[0], c_aux = wc[1];
    const uint32_t step0 = from_ctl ? c_step : R.S.step;
    const int head0 = from_ctl ? wclk_head(c_aux) : R.S.head, hh0 = from_ctl ? wclk_hh(c_aux) : R.S.hh;
    const int phase0 = from_ctl ? wclk_phase(c_aux) : R.S.st_phase;
    const int H = R.horizon;
    const size_t frame = (size_t)n * 26;
    // the starting observation -> the actor's input rows (columns 26..31 stay zero: the weights' k range is padded to 32)
    for (int e = tid; e < ROWS * 32; e += FU_THREADS) {
        const int r = e >> 5, k = e & 31;
        xin[r * FU_XLD + k] = (row0 + r < n && k < 26) ? R.obs_store[(size_t)(row0 + r) * 26 + k] : 0.0f;
    }
    if (tid == 0) { act_seq = 0; a_cnt = 0; reset_seq = 0; }
    __syncthreads();

    if (actor >= 0) {
        // ---------------------------------------------------------------------------------------------------------------- the actor
        constexpr int hp = 128, ip = 32;
        const int r = lane & 15, g = lane >> 4;
        const float *W0 = R.blob, *b0 = W0 + hp * ip, *W1 = b0 + hp, *b1 = W1 + hp * hp, *W2 = b1 + hp, *b2 = W2 + hp * hp, *W3 = b2 + hp, *b3 = W3 + 16 * hp;
        const float *log_std = b3 + 16;
        const int tl[2] = {actor, actor + 4};   // this wavefront's two 16-column tiles of every 128-wide layer
        float4 w0[2][2], w1[2][8], w2[2][8], w3[8];
        float c0[2], c1[2], c2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int s = 0; s < 2; ++s) w0[j][s] = *reinterpret_cast<const float4 *>(W0 + ((size_t)(tl[j] * 2 + s) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                w1[j][s] = *reinterpret_cast<const float4 *>(W1 + ((size_t)(tl[j] * 8 + s) * 64 + lane) * 4);
                w2[j][s] = *reinterpret_cast<const float4 *>(W2 + ((size_t)(tl[j] * 8 + s) * 64 + lane) * 4);
            }
            c0[j] = b0[tl[j] * 16 + r]; c1[j] = b1[tl[j] * 16 + r]; c2[j] = b2[tl[j] * 16 + r];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) w3[s] = *reinterpret_cast<const float4 *>(W3 + ((size_t)s * 64 + lane) * 4);   // (used by actor 0 only)
        const float c3 = b3[r];
        // out[16][128] = relu(in[16][K] W^T + b) for this wavefront's two tiles: dense_tile's chain (bias, then k ascending), two independent
        // accumulators interleaved so that one's MFMA latency hides under the other's
        auto layer = [&](const float *in, int ld, float *out, const float4 *wa, const float4 *wb, int ks, float ba, float bb) __attribute__((always_inline)) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {   // (LPE = 1: four 16-row tiles against the same resident weights)
                const float *arow = in + (rt * 16 + r) * ld + 4 * g;
                pf32x4 acc_a = {ba, ba, ba, ba}, acc_b = {bb, bb, bb, bb};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s < ks) {
                        const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * s);
                        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wa[s].x, acc_a, 0, 0, 0);
                        acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wb[s].x, acc_b, 0, 0, 0);
                        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wa[s].y, acc_a, 0, 0, 0);
                        acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wb[s].y, acc_b, 0, 0, 0);
                        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wa[s].z, acc_a, 0, 0, 0);
                        acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wb[s].z, acc_b, 0, 0, 0);
                        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wa[s].w, acc_a, 0, 0, 0);
                        acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wb[s].w, acc_b, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // C/D layout: col = lane & 15, row = 4 (lane >> 4) + i
                    out[(rt * 16 + 4 * g + i) * FU_LD + tl[0] * 16 + r] = acc_a[i] < 0.0f ? 0.0f : acc_a[i];
                    out[(rt * 16 + 4 * g + i) * FU_LD + tl[1] * 16 + r] = acc_b[i] < 0.0f ? 0.0f : acc_b[i];
                }
            }
        };
        // the four actor wavefronts meet at an LDS counter (the workgroup's barrier belongs to the step: its other wavefronts are in the middle of
        // the substeps).  One wavefront's LDS operations execute in order -- outputs before the increment, the counter before the next
        // layer's reads -- so only the compiler needs fences.  Bounded: a protocol bug ends in wrong numbers and a status bit, never in a hang.
        int sync_no = 0;
        bool stuck = false;
        auto meet = [&]() __attribute__((always_inline)) {
            sync_no += 4;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_fetch_add(&a_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;
            while (__hip_atomic_load(&a_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < sync_no && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            stuck |= spins >= (1 << 22);
        };
        PolicyParams PP{};   // what sample4 reads
        PP.n = n; PP.deterministic = 0; PP.seed_lo = R.pseed_lo; PP.seed_hi = R.pseed_hi; PP.act_lo = R.act_lo; PP.act_hi = R.act_hi; PP.action_env = nullptr;
        const uint32_t call_base = from_ctl ? step0 + R.call_delta : R.call0;   // (the workgroup's own clock pair: the control block is rewritten by workgroup 0 when IT is done)
        // what the sampling does not need the MLP for: the distribution's scale (step-invariant) and, per step, the standard normal of (env, call) --
        // drawn by the sampling wavefront AHEAD of barrier 1, where it idles (Philox, log, sqrt, sincos: ~1 000 clocks off the path between the
        // head's last MFMA and the action a step wavefront with fewer than ten pending slots is waiting for: the tail of the kernel)
        // (LPE = 1: every actor wavefront samples ONE of the four row tiles -- the head's weights are resident in all of them)
        const bool sampler = RT == 1 ? actor == 0 : true;
        const int stile = RT == 1 ? 0 : actor;           // the row tile this wavefront's head + sampling works on
        float scale_pre = 0.0f, ls_pre = 0.0f;
        if (sampler) { const float e = expf_own(log_std[lane & 3]); scale_pre = e * e; ls_pre = log(scale_pre); }
#pragma unroll 1
        for (int t = 0; t < H; ++t) {
            float eps_pre = 0.0f;
            if (sampler) eps_pre = sample_eps(PP, row0 + stile * 16 + (lane >> 2), lane & 3, call_base + (uint32_t)t);
            __syncthreads();   // barrier 1 of the step: obs_t is in xin
            layer(xin, FU_XLD, xa, w0[0], w0[1], 2, c0[0], c0[1]);
            meet();
            layer(xa, FU_LD, xb, w1[0], w1[1], 8, c1[0], c1[1]);
            meet();
            layer(xb, FU_LD, xa, w2[0], w2[1], 8, c2[0], c2[1]);
            meet();
            if (sampler) {   // the 128 -> 4 head (one 16-column tile) with its tanh, then the distribution (nets_asymmetry.py:333-345)
                const float *arow = xa + (stile * 16 + r) * FU_LD + 4 * g;
                pf32x4 acc = {c3, c3, c3, c3};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * s);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w3[s].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w3[s].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w3[s].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w3[s].w, acc, 0, 0, 0);
                }
                // (xb: every actor wavefront is past its last read of it -- with four samplers each writes ITS tile's rows only, and the others' reads of xb
                // ended at the meet() above)
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[(stile * 16 + 4 * g + i) * FU_LD + r] = tanh_own(acc[i]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // same wavefront writes and reads: in order
                PP.action = R.act_buf + (size_t)t * n * 4; PP.mu = R.mu_buf + (size_t)t * n * 4; PP.sigma = R.sigma_buf + (size_t)t * n * 4;
                PP.logp = R.logp_buf + (size_t)t * n;
                // posts the action mailbox as soon as the tile's actions are in act_lds, ahead of the log-prob and the stores: RT == 1: act_seq = t + 1;
                // RT == 4: one increment per tile, complete at 4 (t + 1)
                sample4(PP, xb + stile * 16 * FU_LD, FU_LD, row0 + stile * 16, lane >> 2, lane & 3, log_std, call_base + (uint32_t)t, (lds_f32 *)act_lds + stile * 64, &eps_pre,
                        scale_pre, ls_pre, (lds_i32 *)&act_seq, RT == 1 ? t + 1 : -1);
            }
            __syncthreads();   // barrier 2 of the step
        }
        if (stuck && lane == 0) atomicOr(&R.S.ctl[kCtlStatus], kStatusMailboxTimeout);
    } else {
        // ---------------------------------------------------------------------------------------------------------------- the env step
        StepParams Q = R.S;
        Q.len_obs = 1; Q.len_states = 1;   // one frame per step and stack: the replay store keeps the stacks as a frame ring
        Q.obs_bytes = Q.states_bytes = (uint32_t)(frame * sizeof(float));
        Q.gather = nullptr; Q.stamps = nullptr; Q.obs_out = nullptr; Q.states_out = nullptr; Q.act_in = nullptr;
        const unsigned long long loop_t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int t = 0; t < H; ++t) {
            if (R.stamps && blockIdx.x == 0 && tid == 0 && t < 64) R.stamps[8 + 2 * t] = __builtin_readcyclecounter();
            Q.obs_prev = R.obs_store + (size_t)t * frame; Q.obs = R.obs_store + (size_t)(t + 1) * frame;
            Q.states = R.frames + (size_t)(t + R.len_states) * frame; Q.states_prev = Q.states;
            Q.rew = R.rew_buf + (size_t)t * n; Q.done_f32 = R.done_buf + (size_t)t * n; Q.timeout = R.timeout_buf + (size_t)t * n;
            FusedCtx FX;
            FX.role = role; FX.step = step0 + (uint32_t)t; FX.head = (head0 + 10 * t) % TACO_RING_SLOTS; FX.hh = (hh0 + t) % HIST_ROWS;
            FX.act_lds = (lds_f32 *)act_lds; FX.act_seq = (lds_i32 *)&act_seq; FX.act_want = RT * (t + 1); FX.xin = (lds_f32 *)xin; FX.xin_ld = FU_XLD;
            FX.reset_lds = (lds_i32 *)reset_lds; FX.reset_seq = (lds_i32 *)&reset_seq; FX.reset_want = t;
            if constexpr (LPE == 4) step_core<256, 4, true, false, false, false, false, true, true>(Q, FX);   // (its two barriers are the step's two)
            else step_core<256, 1, true, false, false, false, true, true, false>(Q, FX);                       // (the one-lane four-role form, no register cap)
            if (R.stamps && blockIdx.x == 0 && tid == 0 && t < 64) R.stamps[9 + 2 * t] = __builtin_readcyclecounter();
        }
        if (R.stamps && tid == 0) R.stamps[136 + blockIdx.x] = __builtin_readcyclecounter() - loop_t0;
    }
    // leave the clock after `horizon` steps in this workgroup's pair and (workgroup 0) in the control block, as every step launch does: no
    // wavefront of this workgroup reads the pair any more (they took it H steps of barriers ago), nobody reads the control block in a launch
    if (wave == 0 && lane == 0) {
        const uint32_t nstep = step0 + (uint32_t)H;
        const int nhead = (head0 + 10 * H) % TACO_RING_SLOTS, nhh = (hh0 + H) % HIST_ROWS;
        uint32_t *w = R.S.wclk + (size_t)blockIdx.x * RT * kWclkWordsPerGroup;
#pragma unroll
        for (int gq = 0; gq < RT; ++gq) { w[2 * gq] = nstep; w[2 * gq + 1] = wclk_aux(nhead, nhh, phase0); }
        if (blockIdx.x == 0) { R.S.ctl[kCtlStep] = nstep; R.S.ctl[kCtlHead] = (uint32_t)nhead; R.S.ctl[kCtlHh] = (uint32_t)nhh; if (R.S.use_ctl) atomicAdd(&R.S.ctl[kCtlDevSteps], 1u); }
    }
}

}  // namespace taco
