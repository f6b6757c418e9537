/* CPU ORACLE for SURVEY.md section 8f row N1, second half: PPO_ActorCritic.act (IsaacGymEnvs/algorithms/nets_asymmetry.py:326-355)
 * for the documented training configuration (README.md:60-66: actor = MLP on obs, critic = 1-layer LSTM over the state stack
 * + MLP).  TEST INFRASTRUCTURE ONLY (see oracle/taco_oracle.h).
 *
 * The arithmetic is DEFINED here and followed bit for bit by the HIP kernel (taco_amd/csrc/taco_policy.hpp):
 *   linear:  acc = bias[o];  for s, t, g:  k = 16 s + 4 g + t;  acc = fmaf(x[k], W[o][k], acc)
 *            (the order in which the 16x16x4 f32 MFMA consumes 16-byte operand fragments; an f32 MFMA is a k-ordered fmaf chain)
 *   LSTM:    gates start from b_ih + b_hh, run the chain over x_t, then over h_{t-1}; then the fused cell lstm_cell() below
 *            (c = f * c + i * g, h = o * tanh(c) with three divisions for the five activations)   (nets_asymmetry.py:128-136, torch.nn.LSTM)
 *   actor head: tanh; log_std / scale_tril quirk of :334-335 (scale = exp(log_std) * exp(log_std)); MultivariateNormal log_prob.
 * Parity with the reference (which uses the platform BLAS / vectorised transcendentals) is a tolerance, 1e-5, pinned by
 * tests/golden/policy.npz generated from the reference module on CPU. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "taco_oracle.h"

static inline uint32_t pf2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float pu2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_expf(float x) {
    const float xc = fminf(fmaxf(x, -87.33654475055310898657f), 88.72283905206835f);
    const float k = rintf(xc * 1.44269504088896341f);
    float r = fmaf(-k, 0.693359375f, xc);
    r = fmaf(-k, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r) + 1.0f;
    const float e = ldexpf(p, (int)k);   /* exact: the clamp keeps p * 2^k normal */
    return (x != x) ? x : e;
}
float orc_sigmoidf(float x) { return 1.0f / (1.0f + orc_expf(-x)); }
float orc_tanhf(float x) {
    const float z = fabsf(x);
    const float zc = fminf(z, 44.0f);
    const float bigm = 1.0f - 2.0f / (orc_expf(zc + zc) + 1.0f);
    const float big = pu2f((pf2u(bigm) & 0x7fffffffu) | (pf2u(x) & 0x80000000u));
    const float z2 = x * x;
    float p = -5.70498872745e-3f;
    p = fmaf(p, z2, 2.06390887954e-2f);
    p = fmaf(p, z2, -5.37397155531e-2f);
    p = fmaf(p, z2, 1.33314422036e-1f);
    p = fmaf(p, z2, -3.33332819422e-1f);
    const float small = fmaf(p * z2, x, x);
    const float t = (z >= 0.625f) ? big : small;
    return (x != x) ? x : t;
}

/* the fused LSTM cell of taco_math.hpp::lstm_cell, operation for operation */
static inline void lstm_cell(float ai, float af, float ag, float ao, float c_old, float *c_new, float *h) {
    const float ei = orc_expf(-ai), ef = orc_expf(-af), eo = orc_expf(-ao);
    const float zg = fabsf(ag);
    const float tg = orc_expf(-(zg + zg));
    const float ng = pu2f((pf2u(1.0f - tg) & 0x7fffffffu) | (pf2u(ag) & 0x80000000u));
    const float ig = ng / ((1.0f + ei) * (1.0f + tg));
    const float c = c_old / (1.0f + ef) + ig;
    const float zc = fabsf(c);
    const float tc = orc_expf(-(zc + zc));
    const float nc = pu2f((pf2u(1.0f - tc) & 0x7fffffffu) | (pf2u(c) & 0x80000000u));
    *c_new = c;
    *h = nc / ((1.0f + eo) * (1.0f + tc));
}

static inline int pad16(int x) { return (x + 15) / 16 * 16; }

/* Weight matrices are stored FRAGMENT-MAJOR: [out tile of 16][k block of 16][lane = 16 g + r][t], i.e. the 16 bytes lane (r, g) of
 * a wavefront feeds to the four MFMAs of k block s sit at ((tile * S + s) * 64 + 16 g + r) * 4 -- one wavefront load = 1 KiB contiguous.
 * chain(): acc over one padded operand row for output column o, in the MFMA's k order (k = 16 s + 4 g + t for s, t, g). */
static inline float chain(const float *x, const float *Wf, int o, int kp, float acc) {
    const int S = kp / 16, tile = o >> 4, r = o & 15;
    for (int s = 0; s < S; ++s)
        for (int t = 0; t < 4; ++t)
            for (int g = 0; g < 4; ++g) {
                const int k = 16 * s + 4 * g + t;
                acc = fmaf(x[k], Wf[(((size_t)tile * S + s) * 64 + 16 * g + r) * 4 + t], acc);
            }
    return acc;
}

enum { ACT_RELU = 0, ACT_TANH = 1, ACT_NONE = 2 };
/* y[outp] = act(W[outp][inp] x + b), all padded */
static void linear(const float *x, const float *W, const float *b, int inp, int outp, int act, float *y) {
    for (int o = 0; o < outp; ++o) {
        float a = chain(x, W, o, inp, b[o]);
        if (act == ACT_RELU) a = (a < 0.0f) ? 0.0f : a;   /* a NaN stays NaN, like torch.relu */
        else if (act == ACT_TANH) a = orc_tanhf(a);
        y[o] = a;
    }
}

size_t orc_policy_blob_floats(const orc_policy_cfg *c) {
    size_t n = 0;
    int in = pad16(c->obs_len * c->obs_dim);
    for (int l = 0; l <= c->n_actor_hidden; ++l) {
        const int out = pad16(l < c->n_actor_hidden ? c->actor_hidden[l] : c->act_dim);
        n += (size_t)out * in + out;
        in = out;
    }
    n += 16; /* log_std */
    if (c->lstm_hidden > 0) {
        const int hp = pad16(c->lstm_hidden), ip = pad16(c->states_dim);
        n += (size_t)4 * hp * ip + (size_t)4 * hp * hp + (size_t)4 * hp;
        in = hp;
    } else {
        in = pad16(c->states_len * c->states_dim);
    }
    for (int l = 0; l <= c->n_critic_hidden; ++l) {
        const int out = pad16(l < c->n_critic_hidden ? c->critic_hidden[l] : 1);
        n += (size_t)out * in + out;
        in = out;
    }
    return n;
}

/* One env.  obs [obs_len*obs_dim], states [states_len][states_dim]; eps [act_dim] = the standard-normal draw (ignored when
 * deterministic).  Outputs: action, mu, sigma [act_dim]; logp, value scalars. */
static void act_one(const orc_policy_cfg *c, const float *blob, const float *obs, const float *states, const float *eps, int deterministic,
                    float *action, float *logp, float *value, float *mu, float *sigma) {
    float bufa[ORC_POLICY_MAXW], bufb[ORC_POLICY_MAXW];
    const float *w = blob;
    /* ---- actor MLP */
    int in = pad16(c->obs_len * c->obs_dim);
    memset(bufa, 0, sizeof(bufa));
    memcpy(bufa, obs, sizeof(float) * (size_t)(c->obs_len * c->obs_dim));
    float *x = bufa, *y = bufb;
    for (int l = 0; l <= c->n_actor_hidden; ++l) {
        const int last = l == c->n_actor_hidden;
        const int out = pad16(last ? c->act_dim : c->actor_hidden[l]);
        linear(x, w, w + (size_t)out * in, in, out, last ? ACT_TANH : ACT_RELU, y);
        w += (size_t)out * in + out;
        in = out;
        float *t = x; x = y; y = t;
    }
    const float *log_std = w;
    w += 16;
    /* distribution: covariance = diag(exp(log_std) * exp(log_std)) passed as scale_tril (nets_asymmetry.py:334-335) */
    float lp = 0.0f, half_log_det = 0.0f;
    for (int a = 0; a < c->act_dim; ++a) {
        const float e = orc_expf(log_std[a]);
        const float scale = e * e;
        mu[a] = x[a];
        sigma[a] = log_std[a];                       /* the reference returns log_std.repeat(N, 1) as "sigma" (:354) */
        action[a] = deterministic ? x[a] : x[a] + scale * eps[a];
        const float zz = (action[a] - x[a]) / scale; /* MultivariateNormal.log_prob: M = |L^-1 diff|^2, half_log_det = sum log diag(L) */
        lp = lp + zz * zz;
        half_log_det = half_log_det + orc_logf(scale);
    }
    *logp = -0.5f * ((float)c->act_dim * 1.8378770664093453f + lp) - half_log_det;
    /* ---- critic */
    if (c->lstm_hidden > 0) {
        const int hp = pad16(c->lstm_hidden), ip = pad16(c->states_dim);
        const float *Wih = w, *Whh = w + (size_t)4 * hp * ip, *bs = Whh + (size_t)4 * hp * hp;
        w = bs + (size_t)4 * hp;
        float h[ORC_POLICY_MAXW], hn[ORC_POLICY_MAXW], cst[ORC_POLICY_MAXW], xt[ORC_POLICY_MAXW];
        memset(h, 0, sizeof(h)); memset(cst, 0, sizeof(cst));
        for (int t = 0; t < c->states_len; ++t) {
            memset(xt, 0, sizeof(float) * (size_t)ip);
            memcpy(xt, states + (size_t)t * c->states_dim, sizeof(float) * (size_t)c->states_dim);
            for (int j = 0; j < hp; ++j) {
                float gate[4];
                for (int q = 0; q < 4; ++q) {
                    float a = bs[q * hp + j];
                    a = chain(xt, Wih + (size_t)q * hp * ip, j, ip, a);
                    a = chain(h, Whh + (size_t)q * hp * hp, j, hp, a);
                    gate[q] = a;
                }
                lstm_cell(gate[0], gate[1], gate[2], gate[3], cst[j], &cst[j], &hn[j]);
            }
            memcpy(h, hn, sizeof(float) * (size_t)hp);
        }
        memset(bufa, 0, sizeof(bufa));
        memcpy(bufa, h, sizeof(float) * (size_t)hp);
        in = hp;
    } else {
        in = pad16(c->states_len * c->states_dim);
        memset(bufa, 0, sizeof(bufa));
        memcpy(bufa, states, sizeof(float) * (size_t)(c->states_len * c->states_dim));
    }
    x = bufa; y = bufb;
    for (int l = 0; l <= c->n_critic_hidden; ++l) {
        const int last = l == c->n_critic_hidden;
        const int out = pad16(last ? 1 : c->critic_hidden[l]);
        linear(x, w, w + (size_t)out * in, in, out, last ? ACT_NONE : ACT_RELU, y);
        w += (size_t)out * in + out;
        in = out;
        float *t = x; x = y; y = t;
    }
    *value = x[0];
}

int orc_policy_act(const orc_policy_cfg *c, const float *blob, int n, const float *obs, const float *states, const float *eps, int deterministic,
                   float *action, float *logp, float *value, float *mu, float *sigma) {
    if (c->act_dim < 1 || c->act_dim > 16 || c->n_actor_hidden < 0 || c->n_actor_hidden > 4 || c->n_critic_hidden < 0 || c->n_critic_hidden > 4) return -1;
    const size_t od = (size_t)c->obs_len * c->obs_dim, sd = (size_t)c->states_len * c->states_dim;
    if (pad16((int)od) > ORC_POLICY_MAXW || pad16((int)sd) > ORC_POLICY_MAXW || pad16(c->lstm_hidden) > ORC_POLICY_MAXW) return -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        act_one(c, blob, obs + i * od, states + i * sd, eps ? eps + (size_t)i * c->act_dim : NULL, deterministic || !eps,
                action + (size_t)i * c->act_dim, logp + i, value + i, mu + (size_t)i * c->act_dim, sigma + (size_t)i * c->act_dim);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ one PPO rollout, end to end
 * PPO.run's horizon loop (IsaacGymEnvs/algorithms/ppo_asymmetry.py:308-342) with PPOReplayBuffer.store (buffer_asymmetry.py:49-68) on the
 * oracle env: for t < H { (action, log_prob, value, mu, sigma) = agent.act(obs, states) (:309; the noise of step t = orc_policy_noise(pseed,
 * call0 + t)); actions_clipped = clamp(action, act_lo, act_hi) (:310); env.step (:311) -> next stacks, reward, done, time-outs;
 * rewards_augmented[truncated] += gamma * value(pre-step state) (:314-324: the value of the state the step started from is value[t] itself);
 * store (:326-329) }, then last_values = agent.act(obs, states)[2] (:341).  What taco_rollout_run / RolloutBuffer.run() compute on the GPU, in
 * the reference's order (the GPU batches the critic after the loop: nothing before GAE reads a value).
 * Arrays: obs_store [H + 1][n][obs_len * 26], states_store [H + 1][n][states_len * 26] (slot 0 = the stacks to start from, filled by the
 * caller; slot t + 1 = what step t leaves: MATERIALISED stacks, the reference's layout); act_buf / mu_buf / sigma_buf [H][n][4] (act_buf = the
 * un-clipped sample, :326); rew_buf / done_buf / value_buf / logp_buf [H][n]; timeout_buf [H][n] u8; last_value [n].
 * GAE is orc_gae (buffer_asymmetry.py:93-132) on these arrays. */
int orc_rollout(orc_env *e, const orc_policy_cfg *c, const float *blob, int H, uint64_t pseed, uint32_t call0, double gamma, double act_lo,
                double act_hi, int64_t *reset_buf, float *obs_store, float *states_store, float *act_buf, float *rew_buf, float *done_buf,
                float *value_buf, float *logp_buf, float *mu_buf, float *sigma_buf, uint8_t *timeout_buf, float *last_value) {
    if (!e || !c || c->act_dim != 4 || c->obs_dim != 26 || c->states_dim != 26 || H < 1) return -1;
    const int n = orc_num_envs(e);
    const size_t od = (size_t)c->obs_len * 26, sd = (size_t)c->states_len * 26;
    const float lo = (float)act_lo, hi = (float)act_hi, g = (float)gamma;
    float *eps = (float *)malloc(sizeof(float) * (size_t)n * 4), *aenv = (float *)malloc(sizeof(float) * (size_t)n * 4);
    if (!eps || !aenv) { free(eps); free(aenv); return -1; }
    int rc = 0;
    for (int t = 0; t < H && rc == 0; ++t) {
        float *obs_t = obs_store + (size_t)t * n * od, *st_t = states_store + (size_t)t * n * sd;
        orc_policy_noise(pseed, call0 + (uint32_t)t, n, 4, eps);
        rc = orc_policy_act(c, blob, n, obs_t, st_t, eps, 0, act_buf + (size_t)t * n * 4, logp_buf + (size_t)t * n, value_buf + (size_t)t * n,
                            mu_buf + (size_t)t * n * 4, sigma_buf + (size_t)t * n * 4);
        if (rc != 0) break;
        for (size_t k = 0; k < (size_t)n * 4; ++k) {   /* torch.clip(actions, lo, hi): a NaN stays a NaN */
            const float a = act_buf[(size_t)t * n * 4 + k];
            aenv[k] = (a != a) ? a : (a < lo ? lo : (a > hi ? hi : a));
        }
        /* env.step works in place on its buffers (the stacks are shifted by one frame): slot t + 1 starts as a copy of slot t */
        memcpy(obs_t + (size_t)n * od, obs_t, sizeof(float) * (size_t)n * od);
        memcpy(st_t + (size_t)n * sd, st_t, sizeof(float) * (size_t)n * sd);
        rc = orc_step(e, aenv, obs_t + (size_t)n * od, st_t + (size_t)n * sd, rew_buf + (size_t)t * n, reset_buf, timeout_buf + (size_t)t * n);
        for (int i = 0; i < n; ++i) {
            const size_t k = (size_t)t * n + i;
            done_buf[k] = (float)reset_buf[i];
            if (timeout_buf[k] != 0 && reset_buf[i] != 0) rew_buf[k] = rew_buf[k] + g * value_buf[k];   /* :320-324 */
        }
    }
    if (rc == 0) {   /* :341 (only the value is kept) */
        float *scr = (float *)malloc(sizeof(float) * (size_t)n * 13);
        if (!scr) rc = -1;
        else {
            rc = orc_policy_act(c, blob, n, obs_store + (size_t)H * n * od, states_store + (size_t)H * n * sd, NULL, 1, scr, scr + (size_t)n * 4, last_value,
                                scr + (size_t)n * 5, scr + (size_t)n * 9);
            free(scr);
        }
    }
    free(eps); free(aenv);
    return rc;
}
