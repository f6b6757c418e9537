// taco_capi.hip -- C ABI of libtaco_env.so (declared in include/taco_env.h) over the fused step kernel.
//
// The library never allocates device memory, never synchronises the host and never calls exit(): the caller owns every
// buffer (PyTorch-ROCm tensors in the Python host layer), the handle only keeps the configuration, the derived fp32
// constants and two integers that advance with every step (step counter, ring head).
#include <hip/hip_runtime.h>
#include <limits.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "taco_step.hpp"
// the kernel derives the history and device-clock pointers from the state pointer (StepKernelArgs): hold the layout of taco_create to it
static bool step_layout_ok(const taco::StepParams &P) {
    const size_t npad4 = (size_t)((P.n + 63) & ~63) * 4;
    return P.hist == P.S + (size_t)taco::NUM_CHUNKS * npad4 &&
           (const void *)P.wclk == (const void *)(P.S + (size_t)(taco::NUM_CHUNKS + taco::HIST_ROWS + TACO_RING_SLOTS) * npad4);
}
// the argument list of taco_step_kernel (taco_step.hpp StepKernelArgs: the leading scalars arrive preloaded in registers)
#define STEP_ARGS(P) uint32_t early_ = taco::pack_early((P).hh, (P).hw_rows, (P).stamps != nullptr, (P).use_ctl); \
    void *args[] = {&(P).S, &(P).reset, &(P).act_in, &(P).seed_lo, &(P).seed_hi, &(P).step, &(P).task_mode, &(P).n, &(P).flags, &early_, &(P).env_offset, &(P)}

#include "taco_rollout.hpp"
#include "taco_policy.hpp"
#include "taco_fused.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    std::snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return TACO_ERR_HIP;
}

constexpr int kBlockSmall = 64;   // one wavefront per workgroup
constexpr int kBlockPair = 128;   // the served quad form: one step wavefront + one serving wavefront
constexpr int kBlockLarge = 256;  // the four-role forms: one step wavefront + three role wavefronts

}  // namespace

struct taco_env {
    taco_cfg cfg;
    int device;
    int npad;
    float *S;
    float *hist;
    float *ring;
    int64_t step_count;
    int head;  // physical ring slot of logical slot 0; advances by 10 per step (mod 100)
    int hh;    // action-history row the next step writes; advances by 1 per step (mod 16)
#ifdef TACO_TEST_HOOKS
    int slow_server;
#endif
    int form;  // TACO_FORM_* the step launches (never TACO_FORM_AUTO here: resolved by choose_form)
    int form_ring;    // ... and the form of a launch that writes only the newest states frame (taco_rollout_io.states_newest_only)
    int form_pinned;  // taco_set_kernel_form chose `form`: every launch uses it
    int fusion_force; // taco_set_rollout_fusion(env, 2 / 3): the persistent kernel's quad / one-lane form whatever the env count (A/B: the quad form's workgroups queue, eight rounds of the chip at 32 768 envs)
    int fusion_off;   // taco_set_rollout_fusion(env, 0): taco_rollout_run launches the actor and the step per step even where the persistent kernel applies
    unsigned long long *rollout_stamps;  // optional profiling buffer of the persistent rollout kernel (taco_bind_rollout_stamps)
    uint32_t *ctl;        // control block (last 256 bytes of the workspace): device-resident step clock + sticky status word
    uint32_t *wclk;       // the same clock per 16 envs, where captured kernels take it from (taco_step.hpp "wclk"): in front of the control block
    int clock_on_device;  // a launch of this handle has been captured into a HIP graph (sticky): replays advance the device clock only, so from
                          // then on every launch reads the device clock and every host-side use of the fields above re-reads them first
    uint32_t dev_steps_seen;  // ctl[kCtlDevSteps] as of the last refresh_clock (latched for good by taco_release_graphs): see taco_check
    void *capture_stream; // the stream a launch of this handle was last seen CAPTURING on (refresh_clock refuses to synchronise while it still is)
    int params_imported;  // taco_set_state has been called: rotor / aero parameters may differ per env from now on (see kUniformParams)
    float *gather;  // optional per-rank all-gather block, see taco_bind_gather_block
    float *st_ring;   // optional frame ring behind the state stacks, see taco_bind_states_ring: [st_rows][num_envs][26]
    int st_rows, st_period, st_phase;   // st_period = st_rows - (len_states - 1); st_phase: ring phase of the NEXT step (part of the step clock)
    unsigned long long *stamps;  // optional phase stamps, see taco_bind_phase_stamps
    taco::StepParams P;
};

namespace {

int round_up(int x, int m) { return (x + m - 1) / m * m; }

bool is_capturing(void *stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return stream != nullptr && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
// a launch of this handle is being captured into a HIP graph: from now on the step clock is the device-resident one (see launch_step)
void note_capture(taco_env *e, void *stream) {
    if (is_capturing(stream)) { e->clock_on_device = 1; e->capture_stream = stream; }
}

// x / c via q = x * RN(1/c) + one fma correction (taco_math.hpp div_const) is used for the run-time divisor dt only if it is
// bit-identical to the IEEE quotient for this dt: checked exhaustively over one binade of x, both signs (the three
// operations are scale-invariant, so this covers the normal range).  ~10 ms on the host, once per taco_create / difficulty change.
bool div_const_is_exact(float c) {
    static std::mutex mu;  // handles may be created from several host threads (one per device)
    static float cached_c = 0.0f;
    static bool cached_ok = false;
    std::lock_guard<std::mutex> lock(mu);
    if (c == cached_c) return cached_ok;
    const float rc = 1.0f / c;
    bool ok = true;
    for (uint32_t m = 0; m < (1u << 23) && ok; ++m) {
        uint32_t u = 0x3f800000u | m;
        float x;
        std::memcpy(&x, &u, 4);
        const float q = x * rc, r = std::fmaf(-q, c, x), d = std::fmaf(r, rc, q);
        ok = (d == x / c);
    }
    cached_c = c;
    cached_ok = ok;
    return ok;
}

bool cfg_ok(const taco_cfg *c) {
    if (!c) return fail(0, "cfg is null"), false;
    if (c->num_envs <= 0) return fail(0, "num_envs must be > 0"), false;
    if (c->num_envs > 2000000) return fail(0, "num_envs must be <= 2 000 000 per handle (32-bit buffer offsets); shard larger jobs"), false;
    if (c->env_offset < 0 || c->num_envs_global < c->env_offset + c->num_envs)
        return fail(0, "env_offset + num_envs exceeds num_envs_global"), false;
    if (c->task_mode < TACO_TASK_POS || c->task_mode > TACO_TASK_MIX) return fail(0, "task_mode out of range"), false;
    if (c->len_obs < 1 || c->len_states < 1) return fail(0, "len_obs / len_states must be >= 1"), false;
    if ((size_t)c->num_envs * (size_t)(c->len_obs > c->len_states ? c->len_obs : c->len_states) * 104 >= 0xffffffffull)
        return fail(0, "num_envs * len * 104 bytes must stay below 4 GiB (32-bit buffer offsets); shard the job"), false;
    if (c->control_freq_inv != 10) return fail(0, "control_freq_inv must be 10 (the delay line holds 10 one-ms slots per action)"), false;
    if (c->substeps < 1 || c->substeps > 8) return fail(0, "substeps must be in 1..8"), false;
    if (c->delay_time < 0 || c->delay_time > 90) return fail(0, "delay_time must be in 0..90 ms"), false;
    if (c->max_episode_length < 2 || c->max_episode_length > taco::kProgressMax)
        return fail(0, "max_episode_length must be in 2..65535 (progress_buf is kept in 16 bits)"), false;
    if (c->dt > 0 && !div_const_is_exact((float)c->dt))
        return fail(0, "this dt is not supported: x / dt cannot be evaluated exactly as a multiply-and-correct (use dt = 0.001, the rotor model's fixed sample time)"), false;
    if (!(c->dt > 0) || !(c->mass > 0) || !(c->inertia[0] > 0) || !(c->inertia[1] > 0) || !(c->inertia[2] > 0))
        return fail(0, "dt, mass and inertia must be positive"), false;
    return true;
}

// fp32 images of the Python-double expressions of the reference, computed in double exactly where Python would
void derive(taco_env *e) {
    const taco_cfg &c = e->cfg;
    taco::StepParams &P = e->P;
    const double d = c.difficulty;
    P.n = c.num_envs; P.npad = e->npad; P.env_offset = c.env_offset; P.task_mode = c.task_mode;
    P.mix_n1 = (int)((double)c.num_envs_global / 3 * 1);  // fpv_asymmetry.py:924
    P.mix_n2 = (int)((double)c.num_envs_global / 3 * 2);  // fpv_asymmetry.py:925
    P.len_obs = c.len_obs; P.len_states = c.len_states; P.substeps = c.substeps; P.max_len = c.max_episode_length;
    P.delay_time = c.delay_time; P.flags = c.flags;
#ifdef TACO_TEST_HOOKS
    if (e->slow_server) P.flags |= taco::kDebugSlowServer;  // see taco_test_slow_battery_server
#endif
    P.seed_lo = (uint32_t)c.seed; P.seed_hi = (uint32_t)(c.seed >> 32);
    P.dt = (float)c.dt; P.clip_act = (float)c.clip_actions; P.df = (float)d;
    P.clip_obs = (float)c.clip_obs; P.clip_states = (float)c.clip_states;
    P.h = (float)(c.dt / (double)c.substeps);
    P.half_h = (float)(0.5 * (c.dt / (double)c.substeps));
    P.inv_m = (float)(1.0 / c.mass); P.g = (float)c.gravity_z;
    P.J0 = (float)c.inertia[0]; P.J1 = (float)c.inertia[1]; P.J2 = (float)c.inertia[2];
    P.hJi0 = (float)((c.dt / (double)c.substeps) / c.inertia[0]);
    P.hJi1 = (float)((c.dt / (double)c.substeps) / c.inertia[1]);
    P.hJi2 = (float)((c.dt / (double)c.substeps) / c.inertia[2]);
    P.rdt = 1.0f / P.dt;
    P.arm_x = (float)c.arm_x; P.arm_y = (float)c.arm_y;
    // torch_rand_float(lower, upper): (upper - lower) * u + lower
    P.flip_xy_sc = (float)((0.5 + 1.5 * d) - (-0.5 - 1.5 * d)); P.flip_xy_lo = (float)(-0.5 - 1.5 * d);  // fpv_asymmetry.py:856
    P.flip_v_sc = (float)((3 * d) - (-3 * d)); P.flip_v_lo = (float)(-3 * d);                             // :870
    P.dr_sc = (float)((1 + 0.05 * d) - (1 - 0.05 * d)); P.dr_lo = (float)(1 - 0.05 * d);                  // thrust_dynamics.py:119, :207
    P.tau_sc = (float)((c.rotor_response_time + 0.001) - (c.rotor_response_time - 0.001));               // thrust_dynamics.py:137
    P.tau_lo = (float)(c.rotor_response_time - 0.001);
    P.tau_fixed = (float)c.rotor_response_time;
    P.nq_sc = (float)((d * 0.05) - (-(d * 0.05))); P.nq_lo = (float)(-(d * 0.05));                        // fpv_asymmetry.py:405
    {   // how many history rows the up-front loads fetch = the deepest run queue this configuration reaches before a push: with the deploy
        // length fixed at 10 slots an env that starts from L pending slots never queues more than max(1, ceil(L / 10)) runs (brute-forced
        // over L = 0..40 in DESIGN.md section 3); a random deploy length lets the queue grow until the env turns DENSE
        const int l_max = c.delay_time + ((c.flags & TACO_F_RANDOM_DELAY_TIME) ? 3 : 0);
        int rows = (c.flags & TACO_F_RANDOM_DEPLOY_TIME) ? 4 : (l_max + 9) / 10;
        P.hw_rows = rows < 1 ? 1 : (rows > 4 ? 4 : rows);
        if (e->params_imported) P.hw_rows = 4;  // an imported blob may hold any queue
        // rotor / aero parameters identical in every env: init_state_kernel and reset_env write the same nominal values when nothing
        // randomises them (reset_env writes tau = rotor_response_time only with ROTOR_RESPONSE on) and no blob has been imported
        const uint32_t rnd = TACO_F_RANDOM_ROTORDYNAMIC_COE | TACO_F_RANDOM_AERODYNAMIC_COE | TACO_F_RANDOM_ROTOR_RESPONSE;
        if (!(c.flags & rnd) && (c.flags & TACO_F_ROTOR_RESPONSE) && !e->params_imported) P.flags |= taco::kUniformParams;
    }
    {   // the PLAIN loop forms of the one-lane kernels carry the default airframe's constants as literals (taco_step.hpp kDefaultConsts)
        const taco::Consts mine{P.dt, P.rdt, P.h, P.half_h, P.inv_m, P.g, P.J0, P.J1, P.J2, P.hJi0, P.hJi1, P.hJi2, P.arm_x, P.arm_y};
        if (std::memcmp(&mine, &taco::kDefaultConsts, sizeof(mine)) == 0) P.flags |= taco::kPlainConsts;
    }
}

// Initial state = what the reference holds after construction, before the first step (fpv_asymmetry.py:124-200,
// sub-model constructors): actors at (0,0,4) with identity attitude, nominal rotor / aero parameters, zero everything else.
__device__ __forceinline__ float &word(float *S, int npad, int i, int field) {
    (void)npad;
    const int sl = taco::field_slot(field);
    return S[taco::tile_word(taco::NUM_CHUNKS, sl >> 2, i) + (sl & 3)];
}
// host-initiated changes of the device-resident clock (taco_set_step_count, taco_bind_states_ring): every per-16-env pair + the control block
__global__ void set_clock_kernel(uint32_t *wclk, uint32_t *ctl, int groups, int set_step, uint32_t step, int clear_phase) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < groups) {
        if (set_step) wclk[2 * i] = step;
        if (clear_phase) wclk[2 * i + 1] &= 0xffffu;
    }
    if (i == 0) {
        if (set_step) ctl[taco::kCtlStep] = step;
        if (clear_phase) ctl[taco::kCtlPhase] = 0u;
    }
}
__global__ void init_state_kernel(float *S, float *hist, float *ring, uint32_t *ctl, uint32_t *wclk, int npad, float tau0, int delay_time) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int)(taco::kCtlBytes / 4)) ctl[i] = 0u;
    if (i < (int)taco::wclk_words(npad)) wclk[i] = 0u;
    if (i >= npad) return;
    for (int c = 0; c < (int)taco::NUM_CHUNKS; ++c)
        for (int k = 0; k < 4; ++k) S[taco::tile_word(taco::NUM_CHUNKS, c, i) + k] = 0.0f;
    for (int r = 0; r < taco::HIST_ROWS; ++r)
        for (int k = 0; k < 4; ++k) hist[taco::tile_word(taco::HIST_ROWS, r, i) + k] = 0.0f;
    for (int r = 0; r < TACO_RING_SLOTS; ++r)
        for (int k = 0; k < 4; ++k) ring[taco::tile_word(TACO_RING_SLOTS, r, i) + k] = 0.0f;
    word(S, npad, i, TACO_S_POS + 2) = 4.0f;
    word(S, npad, i, TACO_S_QUAT + 3) = 1.0f;
    word(S, npad, i, TACO_S_TGT_POS + 2) = 4.0f;
    word(S, npad, i, TACO_S_TGT_QUAT + 3) = 1.0f;
    for (int k = 0; k < 4; ++k) word(S, npad, i, TACO_S_TAU + k) = tau0;
    const float opara[5] = {0.0f, 12.9466f, 0.1872f, -5.1220f, 0.5906f};
    for (int k = 0; k < 5; ++k) word(S, npad, i, TACO_S_OPARA + k) = opara[k];
    word(S, npad, i, TACO_S_CF) = 1.13e-05f;
    word(S, npad, i, TACO_S_CT) = 0.05f;
    word(S, npad, i, TACO_S_DX) = -0.386f;
    word(S, npad, i, TACO_S_DY) = -0.53f;
    word(S, npad, i, TACO_S_KT) = 0.009f;
    word(S, npad, i, TACO_S_DELAY_LEN) = __builtin_bit_cast(float, (int)taco::make_dw(delay_time, delay_time, 0, 0, 0, false));  // Z = L: all implied zeros
    word(S, npad, i, TACO_S_PROGRESS) = __builtin_bit_cast(float, (int)taco::make_pw(0, 0));
}

// `actions` (blob rows 40..43) = the action of the last step: row hh - 1 of the action history; 0 while no step has run since the env's reset
// (only taco_reset_done / reset_idx leave an env there, FA:572).  Shared by the blob export and the one-row export.
__device__ __forceinline__ float last_action(float *S, const float *hist, int npad, int i, int hh, int k) {
    const uint32_t pw = __builtin_bit_cast(uint32_t, word(S, npad, i, TACO_S_PROGRESS));
    return taco::pw_progress(pw) == 0 ? 0.0f : hist[taco::tile_word(taco::HIST_ROWS, (hh - 1) & (taco::HIST_ROWS - 1), i) + k];
}

// blob (include/taco_env.h layout: field-major words, logical dense delay line) <-> workspace (float4 chunks; pending
// actions as run queue over the action history, or as the physical slot ring once DENSE)
__global__ void export_state_kernel(const float *S, const float *hist, const float *ring, uint32_t *blob, int n, int npad, int head, int hh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (i >= n) return;
    float *Sm = const_cast<float *>(S);
    const uint32_t dl = __builtin_bit_cast(uint32_t, word(Sm, npad, i, TACO_S_DELAY_LEN));  // taco_step.hpp "queue words"
    const uint32_t pw = __builtin_bit_cast(uint32_t, word(Sm, npad, i, TACO_S_PROGRESS));
    const int L = taco::dw_L(dl), Z = taco::dw_Z(dl);
    const bool dense = taco::dw_dense(dl);
    float val;
    if (row == TACO_S_DELAY_LEN) {
        val = __builtin_bit_cast(float, L);
    } else if (row >= TACO_S_ACT && row < TACO_S_ACT + 4) {            // actions = the action of the last step
        val = last_action(Sm, hist, npad, i, hh, row - TACO_S_ACT);
    } else if (row >= TACO_S_ACT_OLD && row < TACO_S_ACT_OLD + 4) {    // actions_old = the one before, 0 right after a reset (FA:572-573)
        const int progress = taco::pw_progress(pw);
        val = (progress <= 1) ? 0.0f : hist[taco::tile_word(taco::HIST_ROWS, (hh - 2) & (taco::HIST_ROWS - 1), i) + (row - TACO_S_ACT_OLD)];
    } else if (row == TACO_S_PROGRESS) {
        val = __builtin_bit_cast(float, taco::pw_progress(pw));
    } else if (row < TACO_NUM_FIELDS) {
        val = word(Sm, npad, i, row);
    } else {
        const int r = row - TACO_NUM_FIELDS, slot = r >> 2, ch = r & 3;
        if (dense) {
            val = ring[taco::tile_word(TACO_RING_SLOTS, (head + slot) % TACO_RING_SLOTS, i) + ch];
        } else {  // expand [Z zeros | run_0 (rem0) | run_1 | ... ] ; beyond L: zeros
            const int lens = taco::qw_lens(dl, pw), m = taco::dw_m(dl), rem0 = taco::dw_rem0(dl);
            val = 0.0f;
            int pos = Z;
            for (int j = 0; j < m && slot >= pos; ++j) {
                const int len = (j == 0) ? rem0 : taco::run_len(lens, j);
                if (slot < pos + len) val = hist[taco::tile_word(taco::HIST_ROWS, (hh - (m - j)) & (taco::HIST_ROWS - 1), i) + ch];
                pos += len;
            }
        }
    }
    blob[(size_t)row * n + i] = __builtin_bit_cast(uint32_t, val);
}
__global__ void import_state_kernel(float *S, float *hist, float *ring, const uint32_t *blob, int n, int npad, int head, int hh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (i >= n) return;
    float val = __builtin_bit_cast(float, blob[(size_t)row * n + i]);
    if (row == TACO_S_DELAY_LEN) {
        // an imported delay line is arbitrary content: the env becomes DENSE (literal slot ring), its run queue empty
        const int L = (int)(__builtin_bit_cast(uint32_t, val) & 0x7fu);
        val = __builtin_bit_cast(float, (int)taco::make_dw(L > TACO_RING_SLOTS ? TACO_RING_SLOTS : L, 0, 0, 0, 0, true));
    }
    if (row == TACO_S_PROGRESS) val = __builtin_bit_cast(float, (int)taco::make_pw(__builtin_bit_cast(int, val) < 0 ? 0 : __builtin_bit_cast(int, val), 0));
    if (row >= TACO_S_ACT && row < TACO_S_ACT + 4) {
        hist[taco::tile_word(taco::HIST_ROWS, (hh - 1) & (taco::HIST_ROWS - 1), i) + (row - TACO_S_ACT)] = val;
    } else if (row >= TACO_S_ACT_OLD && row < TACO_S_ACT_OLD + 4) {
        hist[taco::tile_word(taco::HIST_ROWS, (hh - 2) & (taco::HIST_ROWS - 1), i) + (row - TACO_S_ACT_OLD)] = val;
    } else if (row < TACO_NUM_FIELDS) {
        word(S, npad, i, row) = val;
    } else {
        const int r = row - TACO_NUM_FIELDS, slot = r >> 2, ch = r & 3;
        ring[taco::tile_word(TACO_RING_SLOTS, (head + slot) % TACO_RING_SLOTS, i) + ch] = val;
    }
}

// The instantiation of the step kernel a handle launches (include/taco_env.h TACO_FORM_*), chosen when the handle is created from the
// env count (taco_set_kernel_form overrides it).  4 lanes per env (quad layout) while one lane per env would leave most SIMDs without a
// wavefront (measured: 19.8 us vs 21.0 us at 16 384 envs, 24.6 us vs 22 us at 20 480); the 128-VGPR throughput form from 65 536 envs.
constexpr int kQuadMaxEnvs = 16384;
constexpr int kSplitLaneMaxEnvs = 65536;  // 1 024 workgroups (31 KB of LDS, 111 VGPRs): four resident per CU
constexpr int kSplitMaxEnvs = 8192;  // measured: 16.8 vs 17.6 us at 4 096 envs, 18.4 vs 18.8 at 8 192, no gain at 16 384
constexpr int kThroughputMinEnvs = 65536;
struct FormInfo { int lpe, block, split; const void *fn; };
template <bool OUT> const void *kernel_of(int form) {
    switch (form) {
        case TACO_FORM_QUAD_ROLES: return (const void *)taco::taco_step_kernel<kBlockLarge, 4, true, false, OUT>;
        case TACO_FORM_LANE_ROLES: return (const void *)taco::taco_step_kernel<kBlockLarge, 1, true, false, OUT>;
        case TACO_FORM_QUAD: return (const void *)taco::taco_step_kernel<kBlockSmall, 4, false, false, OUT>;
        case TACO_FORM_QUAD_SERVED: return (const void *)taco::taco_step_kernel<kBlockPair, 4, true, false, OUT>;
        case TACO_FORM_LANE_THROUGHPUT: return (const void *)taco::taco_step_kernel<kBlockSmall, 1, false, true, OUT>;
        default: return (const void *)taco::taco_step_kernel<kBlockSmall, 1, false, false, OUT>;
    }
}
// `out`: the instantiation that also writes the clamped obs / states copies (taco_rollout_io.obs_out / states_out)
// `wide`: the two-wavefronts-per-SIMD build of the four-role quad form (taco_step.hpp, WIDE), see wide_form()
// `lin`: the build of the four-role quad form whose obs role wavefront runs the linear chain (taco_step.hpp, LIN), see lin_form(); WIDE implies it
FormInfo form_info(int form, bool out = false, bool wide = false, bool lin = false) {
    const void *fn = out ? kernel_of<true>(form) : kernel_of<false>(form);
    if (lin && !wide && form == TACO_FORM_QUAD_ROLES)
        fn = out ? (const void *)taco::taco_step_kernel<kBlockLarge, 4, true, false, true, false, false, true>
                 : (const void *)taco::taco_step_kernel<kBlockLarge, 4, true, false, false, false, false, true>;
    if (lin && form == TACO_FORM_QUAD_SERVED)
        fn = out ? (const void *)taco::taco_step_kernel<kBlockPair, 4, true, false, true, false, false, true>
                 : (const void *)taco::taco_step_kernel<kBlockPair, 4, true, false, false, false, false, true>;
    if (wide && form == TACO_FORM_QUAD_ROLES)
        fn = out ? (const void *)taco::taco_step_kernel<kBlockLarge, 4, true, false, true, false, true, true>
                 : (const void *)taco::taco_step_kernel<kBlockLarge, 4, true, false, false, false, true, true>;
    if (wide && form == TACO_FORM_LANE_ROLES)
        fn = out ? (const void *)taco::taco_step_kernel<kBlockLarge, 1, true, false, true, false, true>
                 : (const void *)taco::taco_step_kernel<kBlockLarge, 1, true, false, false, false, true>;
    switch (form) {
        case TACO_FORM_QUAD_ROLES: return {4, kBlockLarge, 1, fn};
        case TACO_FORM_LANE_ROLES: return {1, kBlockLarge, 1, fn};
        case TACO_FORM_QUAD: return {4, kBlockSmall, 0, fn};
        case TACO_FORM_QUAD_SERVED: return {4, kBlockPair, 1, fn};
        default: return {1, kBlockSmall, 0, fn};
    }
}
// every wavefront of the launch has a SIMD to itself (256 workgroups of 4) and no stack history to move: the register cap buys nothing
// ... and nothing for the served tables to do (the WIDE build carries neither the euler nor the rotor-noise server: flip envs, record_flag and
// rotor noise take the general build, where those leave the step wavefront's instruction stream)
// The one-lane role form: its launch puts at most two wavefronts on a SIMD up to 32 768 envs (512 workgroups of four).
bool wide_form(const taco_cfg &c, int form) {
    if (form == TACO_FORM_LANE_ROLES) return c.num_envs <= 32768;
    return c.num_envs <= 4096 && c.len_obs == 1 && c.len_states == 1 && (c.task_mode == TACO_TASK_POS || c.task_mode == TACO_TASK_ROTATE) &&
           !(c.flags & (TACO_F_ROTOR_NOISE | TACO_F_TRACK_RPY));
}
// the four-role quad form with the linear chain served by the obs role wavefront: while at most two wavefronts share a SIMD (taco_step.hpp "LIN")
// ... and of the served pair (QUAD_SERVED) for launches without flip envs: its one serving wavefront then runs the linear chain (with flip envs it serves the euler angles)
bool lin_form(const taco_cfg &c, int form) {
    const bool euler = c.task_mode == TACO_TASK_FLIP || c.task_mode == TACO_TASK_MIX || (c.flags & TACO_F_TRACK_RPY) != 0;
    return (form == TACO_FORM_QUAD_ROLES && c.num_envs <= 8192) || (form == TACO_FORM_QUAD_SERVED && !euler);
}
int choose_form(const taco_cfg &c) {
    const bool stacks = c.len_obs > 1 || c.len_states > 1;
    if (c.num_envs <= kQuadMaxEnvs) {
        // flip envs (the euler server) keep the role form up to the quad limit: 18.1 vs 19.1 us at 16 384 flip envs (profiles/r03_f_ab_step_servers.txt)
        const bool euler_served = c.task_mode == TACO_TASK_FLIP || c.task_mode == TACO_TASK_MIX || (c.flags & TACO_F_TRACK_RPY) != 0;
        // ... above 12 288 envs without frame stacks as the SERVED pair (a step wavefront that runs the whole step + one serving wavefront: fewer
        // wavefronts to dispatch, one barrier; flip 16 384: 17.8 vs 18.3 us; a tie at 12 288), with stacks the roles also move the history
        if (euler_served) return (!stacks && c.num_envs > 12288) ? TACO_FORM_QUAD_SERVED : TACO_FORM_QUAD_ROLES;
        // three helper wavefronts per 16 envs take over the post-phase: pays while every wavefront still has a SIMD to itself.  With
        // frame stacks the role wavefronts also move the stack history under the substeps, which pays up to the quad limit
        // (16 384 envs, 5 state frames: 19.2 us vs 22.1 us; 5 + 5 frames: 19.9 us vs 26.0 us; without stacks 18.0 us vs 17.5 us).
        // (round 5: without stacks above 8 192 envs the SERVED PAIR, whose serving wavefront runs the linear chain (taco_step.hpp LIN, lin_form):
        // rotate 16 384 14.66 -> 14.51 us, pos 16 384 14.20 -> 14.08, 12 288 envs 14.46 -> 14.31 / 14.05 -> 13.82 against the one-wavefront quad form,
        // profiles/r05_g_ab_pair_lin.txt -- 1 %: at two wavefronts per SIMD the server's issue slots come out of a step wavefront next to it)
        return c.num_envs <= (stacks ? kQuadMaxEnvs : kSplitMaxEnvs) ? TACO_FORM_QUAD_ROLES : TACO_FORM_QUAD_SERVED;
    }
    // One lane per env, 16 385 ... 65 536 envs with frame stacks: the same four-role form (64 envs per workgroup) hides the stack
    // history, which a lone wavefront moves at only ~5 GB/s (tools/ubench/shift), under the substeps.  Likewise -- stacks or not -- when the
    // role wavefronts have something to SERVE: the rotor-noise table (config 5's flags on the states frame ring, one frame written:
    // 23.5 vs 31.4 us at 32 768 envs, 28.9 vs 33.3 at 65 536; profiles/r04_d_ring_form_ab.txt) or the euler angles of flip envs.
    const bool served = (c.flags & TACO_F_ROTOR_NOISE) != 0 || c.task_mode == TACO_TASK_FLIP || c.task_mode == TACO_TASK_MIX || (c.flags & TACO_F_TRACK_RPY) != 0;
    if ((stacks || served) && c.num_envs <= kSplitLaneMaxEnvs) return TACO_FORM_LANE_ROLES;
    return c.num_envs >= kThroughputMinEnvs ? TACO_FORM_LANE_THROUGHPUT : TACO_FORM_LANE;
}
void grid_of_form(const taco_env *e, int form, int *grid, int *block) {
    const FormInfo f = form_info(form);
    *block = f.block;
    *grid = f.split ? (e->cfg.num_envs * f.lpe + 63) / 64 : (e->cfg.num_envs * f.lpe + f.block - 1) / f.block;
}
void grid_of(const taco_env *e, int *grid, int *block) { grid_of_form(e, e->form, grid, block); }

}  // namespace

extern "C" {

int taco_gather_row_floats(int len_obs);

#ifndef TACO_SOURCE_HASH
#define TACO_SOURCE_HASH "unhashed-build-0"
#endif
int taco_abi_version(void) { return TACO_ABI_VERSION; }
// hash of the sources this binary was built from (taco_amd/build.py embeds it; _lib.load() refuses a binary older than csrc/)
const char *taco_source_hash(void) { static const char tag[] = "taco-src-hash:" TACO_SOURCE_HASH; return tag + 14; }
const char *taco_last_error(void) { return g_err; }
const char *taco_step_kernel_name(void) { return "taco_step_kernel"; }

size_t taco_workspace_bytes(const taco_cfg *cfg) {
    if (!cfg || cfg->num_envs <= 0) return 0;
    const size_t npad = (size_t)round_up(cfg->num_envs, 64);
    return (size_t)(taco::NUM_CHUNKS + taco::HIST_ROWS + TACO_RING_SLOTS) * npad * 4 * sizeof(float) +
           (taco::wclk_words((int)npad) * sizeof(uint32_t) + 255) / 256 * 256 + taco::kCtlBytes;
}

int taco_create(const taco_cfg *cfg, int device, void *workspace, size_t workspace_bytes, void *stream, taco_env **out) {
    g_err[0] = 0;
    if (!out) return fail(TACO_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!cfg_ok(cfg)) return TACO_ERR_INVALID_ARG;
    if (!workspace) return fail(TACO_ERR_WORKSPACE, "workspace is null");
    if (workspace_bytes < taco_workspace_bytes(cfg)) return fail(TACO_ERR_WORKSPACE, "workspace smaller than taco_workspace_bytes(cfg)");
    if (((uintptr_t)workspace & 255u) != 0) return fail(TACO_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return hip_fail(he, "hipSetDevice");
    taco_env *e = new (std::nothrow) taco_env();
    if (!e) return fail(TACO_ERR_INVALID_ARG, "out of host memory");
    e->cfg = *cfg;
    e->device = device;
    e->npad = round_up(cfg->num_envs, 64);
    e->S = (float *)workspace;
    e->hist = e->S + (size_t)taco::NUM_CHUNKS * e->npad * 4;
    e->ring = e->hist + (size_t)taco::HIST_ROWS * e->npad * 4;
    e->wclk = (uint32_t *)(e->ring + (size_t)TACO_RING_SLOTS * e->npad * 4);
    e->ctl = (uint32_t *)((char *)e->wclk + (taco::wclk_words(e->npad) * sizeof(uint32_t) + 255) / 256 * 256);
    e->clock_on_device = 0;
    e->dev_steps_seen = 0;
    e->capture_stream = nullptr;
    e->step_count = 0;
    e->head = 0;
    e->hh = 0;
    e->gather = nullptr;
    e->st_ring = nullptr; e->st_rows = 0; e->st_period = 0; e->st_phase = 0;
    e->stamps = nullptr;
    e->params_imported = 0;
    e->form = choose_form(e->cfg);
    { taco_cfg one = e->cfg; one.len_states = 1; e->form_ring = choose_form(one); }
    e->form_pinned = 0;
    e->fusion_off = 0;
    e->fusion_force = 0;
    e->rollout_stamps = nullptr;
    std::memset(&e->P, 0, sizeof(e->P));
    derive(e);
    hipLaunchKernelGGL(init_state_kernel, dim3((e->npad + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->S, e->hist, e->ring, e->ctl, e->wclk, e->npad,
                       (float)cfg->rotor_response_time, cfg->delay_time);
    he = hipGetLastError();
    if (he != hipSuccess) { delete e; return hip_fail(he, "init_state_kernel launch"); }
    *out = e;
    return TACO_OK;
}

void taco_destroy(taco_env *env) { delete env; }

namespace {
int launch_step(taco_env *e, const taco_rollout_io *io, void *stream) {
    // states_next == NULL with a frame ring bound (taco_bind_states_ring): the launch writes ONE states frame into the ring row of its phase
    const bool ring_mode = io->states_next == nullptr && e->st_ring != nullptr && e->cfg.len_states > 1;
    if (!io->actions || !io->obs_next || (!io->states_next && !ring_mode) || !io->rew || !io->reset_buf || !io->timeout_buf)
        return fail(TACO_ERR_INVALID_ARG, "taco_step: null buffer pointer");
    if (ring_mode && (io->states_prev || io->states_newest_only))
        return fail(TACO_ERR_INVALID_ARG, "taco_step_rollout on the bound states ring: states_prev must be NULL and states_newest_only 0");
    if (((uintptr_t)io->actions & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "actions must be 16-byte aligned");
    // The step counter, the ring head and the history row travel in the kernel arguments and advance on the host with every call; a launch
    // that is being CAPTURED into a HIP graph would replay one step index for ever, so it reads the device-resident copy of the clock
    // instead (every eager launch leaves the next values there) and is followed by a one-thread kernel that advances it.  Once a capture
    // has happened the host's copy may be stale (replays advance the device clock only): every launch then takes the device path until
    // taco_get_step_count has re-read it.
    note_capture(e, stream);
    const size_t n_envs = (size_t)e->cfg.num_envs;
    taco::StepParams P = e->P;
    P.S = e->S; P.ring = e->ring; P.hist = e->hist;
    P.act_in = io->actions; P.obs = io->obs_next; P.states = io->states_next; P.rew = io->rew;
    P.obs_prev = io->obs_prev ? io->obs_prev : io->obs_next;
    P.states_prev = io->states_prev ? io->states_prev : io->states_next;
    if ((((uintptr_t)P.obs_prev | (uintptr_t)P.states_prev | (uintptr_t)P.obs | (uintptr_t)P.states) & 3u) != 0)
        return fail(TACO_ERR_INVALID_ARG, "obs / states buffers must be 4-byte aligned");
    P.done_f32 = io->done_f32;
    P.reset = (long long *)io->reset_buf; P.timeout = io->timeout_buf;
    P.gather = e->gather;
    P.stamps = e->stamps;
    P.obs_out = io->obs_out; P.states_out = io->states_out;
    if ((((uintptr_t)P.obs_out | (uintptr_t)P.states_out) & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "obs_out / states_out must be 16-byte aligned");
    P.ctl = e->ctl; P.wclk = e->wclk;
    P.use_ctl = e->clock_on_device;
    P.step = (uint32_t)e->step_count;
    P.obs_bytes = (uint32_t)((size_t)n_envs * e->cfg.len_obs * 26 * sizeof(float));
    P.states_bytes = (uint32_t)((size_t)n_envs * e->cfg.len_states * 26 * sizeof(float));
    // states_newest_only: `states_next` is ONE frame per env ([num_envs][26], a row of the replay store's frame ring): the launch runs as a
    // len_states = 1 step -- nothing is shifted, nothing of an older stack is read
    const bool newest_only = (io->states_newest_only != 0 && e->cfg.len_states > 1) || ring_mode;
    if (newest_only) {
        // (ring mode: states_out = a second ring of the same geometry for the clamped frames)
        if (io->states_out && !ring_mode) return fail(TACO_ERR_INVALID_ARG, "states_newest_only writes no stack: states_out must be NULL");
        P.len_states = 1; P.states_bytes = (uint32_t)(n_envs * 26 * sizeof(float)); P.states_prev = P.states;
    }
    P.st_period = 0; P.st_front = 0;
    P.st_phase = e->st_phase;   // (every launch republishes the clock, the ring phase included: a launch that does not write the ring hands it on unchanged)
    if (ring_mode) {
        P.states = e->st_ring; P.states_prev = e->st_ring;
        P.st_period = e->st_period; P.st_front = e->cfg.len_states - 1;
    }
    P.gather_row = (uint32_t)taco_gather_row_floats(e->cfg.len_obs);
    P.gather_bytes = (uint32_t)((size_t)n_envs * P.gather_row * sizeof(float));
    P.head = e->head;
    P.hh = e->hh;
    int grid, block;
    P.stamps_on = P.stamps != nullptr;
    if (!step_layout_ok(P)) return fail(TACO_ERR_WORKSPACE, "taco_step: the handle's workspace layout is not the one the step kernel derives its history / clock pointers from");
    STEP_ARGS(P);
    // (a handle created with a state stack launches the form chosen for stacks; the newest-frame-only launch of the same handle takes the
    // form its env count would get WITHOUT one -- the role wavefronts have no history to move -- unless the caller pinned a form)
    const int form = newest_only && !e->form_pinned ? e->form_ring : e->form;
    grid_of_form(e, form, &grid, &block);
    taco_cfg eff = e->cfg;
    if (newest_only) eff.len_states = 1;
    hipError_t he = hipLaunchKernel(form_info(form, P.obs_out != nullptr || P.states_out != nullptr, wide_form(eff, form), lin_form(eff, form)).fn, dim3(grid), dim3(block), args, 0,
                                    (hipStream_t)stream);
    if (he == hipSuccess) he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "taco_step_kernel launch");
    // (every launch leaves the next clock on the device itself -- taco_step.hpp "wclk" --, captured or not)
    e->step_count += 1;
    e->head = (e->head + 10) % TACO_RING_SLOTS;
    e->hh = (e->hh + 1) % taco::HIST_ROWS;
    if (ring_mode) e->st_phase = (e->st_phase + 1) % e->st_period;
    return TACO_OK;
}
}  // namespace

int taco_step(taco_env *e, const float *actions, float *obs_buf, float *states_buf, float *rew_buf, int64_t *reset_buf,
              uint8_t *timeout_buf, void *stream) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    taco_rollout_io io{};
    io.actions = actions; io.obs_next = obs_buf; io.states_next = states_buf; io.rew = rew_buf;
    io.reset_buf = reset_buf; io.timeout_buf = timeout_buf;
    return launch_step(e, &io, stream);
}

int taco_reset_done(taco_env *e, int64_t *reset_buf, void *stream) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (!reset_buf) return fail(TACO_ERR_INVALID_ARG, "taco_reset_done: reset_buf is null");
    note_capture(e, stream);
    taco::StepParams P = e->P;
    P.S = e->S; P.ring = e->ring; P.hist = e->hist;
    P.reset = (long long *)reset_buf;
    P.ctl = e->ctl; P.wclk = e->wclk; P.use_ctl = e->clock_on_device;
    P.step = (uint32_t)e->step_count; P.head = e->head; P.hh = e->hh;   // the clock of the NEXT step: it does not advance here
    P.stamps = nullptr; P.stamps_on = 0;
    if (!step_layout_ok(P)) return fail(TACO_ERR_WORKSPACE, "taco_reset_done: the handle's workspace layout is not the one the step kernel derives its history / clock pointers from");
    STEP_ARGS(P);
    const void *fn = (const void *)taco::taco_step_kernel<64, 1, false, false, false, true>;
    hipError_t he = hipLaunchKernel(fn, dim3((unsigned)((e->cfg.num_envs + 63) / 64)), dim3(64), args, 0, (hipStream_t)stream);
    if (he == hipSuccess) he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "taco_reset_done launch");
    return TACO_OK;
}

int taco_step_rollout(taco_env *e, const taco_rollout_io *io, void *stream) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (!io) return fail(TACO_ERR_INVALID_ARG, "taco_step_rollout: io is null");
    return launch_step(e, io, stream);
}

size_t taco_gae_workspace_bytes(void) { return (size_t)2 * taco::kNormBlocks * sizeof(double); }

int taco_gae(const float *rew, const float *done, const float *value, const float *last_value, int horizon, int num_envs,
             double gamma, double lam, float *adv, float *ret, int normalize, void *workspace, void *stream) {
    if (!rew || !done || !value || !last_value || !adv || !ret) return fail(TACO_ERR_INVALID_ARG, "taco_gae: null buffer pointer");
    if (horizon < 1 || num_envs < 1) return fail(TACO_ERR_INVALID_ARG, "taco_gae: horizon and num_envs must be >= 1");
    hipLaunchKernelGGL(taco::gae_kernel, dim3((num_envs + taco::kGaeBlock - 1) / taco::kGaeBlock), dim3(taco::kGaeBlock), 0, (hipStream_t)stream,
                       rew, done, value, last_value, horizon, num_envs, (float)gamma, (float)lam, adv, ret);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "gae_kernel launch");
    if (normalize) {
        if (!workspace || ((uintptr_t)workspace & 7u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_gae: normalize needs an 8-byte aligned workspace");
        const size_t count = (size_t)horizon * (size_t)num_envs;
        if (count < 2) return fail(TACO_ERR_INVALID_ARG, "taco_gae: normalize needs at least two samples");
        double *part = (double *)workspace;
        hipLaunchKernelGGL(taco::adv_partial_kernel, dim3(taco::kNormBlocks), dim3(taco::kNormThreads), 0, (hipStream_t)stream, adv, count, part);
        size_t blocks = (count + taco::kNormThreads - 1) / taco::kNormThreads;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(taco::adv_apply_kernel, dim3((unsigned)blocks), dim3(taco::kNormThreads), 0, (hipStream_t)stream, adv, count, (const double *)part);
        he = hipGetLastError();
        if (he != hipSuccess) return hip_fail(he, "advantage normalisation launch");
    }
    return TACO_OK;
}

int taco_set_difficulty(taco_env *e, double difficulty) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    e->cfg.difficulty = difficulty;
    derive(e);
    return TACO_OK;
}

int taco_gather_row_floats(int len_obs) { return (len_obs * 26 + 3 + 31) / 32 * 32; }

int taco_bind_gather_block(taco_env *e, float *block) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (block && (size_t)e->cfg.num_envs * (size_t)taco_gather_row_floats(e->cfg.len_obs) * sizeof(float) >= 0xffffffffull)
        return fail(TACO_ERR_INVALID_ARG, "gather block of num_envs rows must stay below 4 GiB (32-bit buffer offsets); shard the job");
    if (((uintptr_t)block & 7u) != 0) return fail(TACO_ERR_INVALID_ARG, "gather block must be 8-byte aligned");
    e->gather = block;
    return TACO_OK;
}

namespace {
// re-read the clock the graph replays advanced (blocking, device-wide): afterwards the host's copy is current again and launches go back to
// the eager path.
// Synchronising while a stream is capturing is illegal (it would invalidate the capture): TACO_ERR_STATE instead.
int refresh_clock(taco_env *e, void *stream = nullptr) {
    if (e->capture_stream != nullptr) {
        if (is_capturing(e->capture_stream) || is_capturing(stream))
            return fail(TACO_ERR_STATE, "the step clock cannot be read back while a stream is capturing launches of this handle: end the capture first "
                                        "(taco_get_step_count / taco_set_step_count / taco_get_state / taco_set_state / taco_get_field / an eager taco_rollout_run synchronise)");
        e->capture_stream = nullptr;
    }
    if (!e->clock_on_device) return TACO_OK;
    uint32_t c[8];
    hipError_t he;
    // (the whole device, whatever stream the caller named: the replays that advanced the clock may have run on ANOTHER stream the caller's has
    // not waited for -- adopting a stale clock would corrupt the Philox counters and the delay ring's head silently; this path is rare)
    (void)stream;
    he = hipDeviceSynchronize();
    if (he == hipSuccess) he = hipMemcpy(c, e->ctl, sizeof(c), hipMemcpyDeviceToHost);
    if (he != hipSuccess) return hip_fail(he, "reading the device-resident step clock");
    e->step_count = (e->step_count & ~(int64_t)0xffffffff) | c[taco::kCtlStep];  // (the kernel's Philox counter is the low 32 bits)
    e->head = (int)c[taco::kCtlHead];
    e->hh = (int)c[taco::kCtlHh];
    if (e->st_period > 0) e->st_phase = (int)(c[taco::kCtlPhase] % (uint32_t)e->st_period);
    e->dev_steps_seen = c[taco::kCtlDevSteps];
    // clock_on_device stays set: a graph captured from this handle may be replayed again at any time, behind the host's back -- from the first
    // capture on every launch takes its clock from the device and every host-side read of it comes here (rounds 2-3 went back to the host's
    // copy after one re-read: a replay after that left it stale, silently)
    return TACO_OK;
}
}  // namespace

int64_t taco_get_step_count(taco_env *e) {
    if (!e) return -1;
    if (refresh_clock(e) != TACO_OK) return -1;
    return e->step_count;
}
int64_t taco_peek_step_count(const taco_env *e) { return e ? e->step_count : -1; }
// Graph mode (clock_on_device): from the first captured launch every host-side use of the clock blocks (refresh_clock) -- a graph may have
// been replayed behind the host's back, and no non-blocking scheme can know about a replay that is enqueued but has not run yet.  The way out
// is the caller's word: taco_release_graphs = "no graph holding launches of this handle will be replayed from now on" (one last re-read, then
// the handle is back on the eager path: kernel arguments, the host's own copy, no synchronisation anywhere).  A replay after that still
// steps the envs correctly as far as IT goes (graphs read the device copy, which eager launches keep current) but leaves the host's copy
// behind: taco_check reports exactly that, and keeps reporting it -- every launch that takes its clock from the device counts itself in
// ctl[kCtlDevSteps], a word eager launches outside graph mode never write, and the release latches its value: a later eager step overwrites
// kCtlStep (so "device clock ahead of the host's" alone would go quiet again, round 5's advisor), never the count.
int taco_graph_mode(const taco_env *e) { return e ? e->clock_on_device : TACO_ERR_INVALID_ARG; }
int taco_release_graphs(taco_env *e) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    const int rc = refresh_clock(e);   // (refuses while a stream is still capturing launches of this handle)
    if (rc != TACO_OK) return rc;
    e->clock_on_device = 0;
    return TACO_OK;
}
// The device-resident copy of the clock is set as well: a capture may be the very next launch (checkpoint restore -> capture -> replay), and a
// captured step reads its Philox counter from there.  (Ring head and history row are not part of a checkpoint -- the blob is in logical
// order -- and their device copies already equal the host's: every eager launch mirrors them.)
int taco_set_step_count(taco_env *e, int64_t n) {
    if (!e || n < 0) return fail(TACO_ERR_INVALID_ARG, "taco_set_step_count: bad argument");
    const int rc = refresh_clock(e);
    if (rc != TACO_OK) return rc;
    hipError_t he = hipDeviceSynchronize();   // an eager step still in flight would overwrite the words with ITS successor
    if (he == hipSuccess) {
        const int groups = e->npad / 16;
        hipLaunchKernelGGL(set_clock_kernel, dim3((groups + 255) / 256), dim3(256), 0, (hipStream_t) nullptr, e->wclk, e->ctl, groups, 1, (uint32_t)n, 0);
        he = hipGetLastError();
        if (he == hipSuccess) he = hipDeviceSynchronize();
    }
    if (he != hipSuccess) return hip_fail(he, "writing the device-resident step clock");
    e->step_count = n;
    return TACO_OK;
}

int taco_check(taco_env *e, void *stream) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t he = hipMemcpyAsync(c, e->ctl, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "taco_check");
    const uint32_t status = c[taco::kCtlStatus], dev_step = c[taco::kCtlStep];
    // outside graph mode (a) the host's copy of the clock leads the device's by the launches still in flight, never the other way round, and
    // (b) no launch takes its clock from the device, so ctl[kCtlDevSteps] stands where taco_release_graphs found it.  Either one broken means
    // a captured graph was replayed after the release (or by another handle on this workspace); (b) stays broken whatever eager steps follow
    if (!e->clock_on_device && e->capture_stream == nullptr && ((int32_t)(dev_step - (uint32_t)e->step_count) > 0 || c[taco::kCtlDevSteps] != e->dev_steps_seen))
        return fail(TACO_ERR_STATE, "the device-resident step clock is ahead of the host's copy: a captured graph holding launches of this handle was "
                                    "replayed after taco_release_graphs; eager steps since then used stale random-stream counters");
    if (status & taco::kStatusCriticClamped)
        return fail(TACO_ERR_STATE, "the split-f16 critic of a taco_rollout_run saturated finite frame words beyond +-65 504 (values of those rows are approximate): "
                                    "normalise the states, or run the critic in f32 (taco_policy_cfg.flags without TACO_P_SPLIT_F16; ActorCritic(critic_split=None))");
    if (status & taco::kStatusMailboxTimeout)
        return fail(TACO_ERR_STATE, "a step kernel gave up waiting on its battery mailbox (role wavefronts out of step): the affected envs' voltage "
                                    "was set to NaN; results since the last clean taco_check are suspect");
    return TACO_OK;
}

namespace {
__global__ void export_field_kernel(const float *S, const float *hist, uint32_t *out, int field, int n, int npad, int hh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float *Sm = const_cast<float *>(S);
    float val;
    if (field == TACO_S_PROGRESS) val = __builtin_bit_cast(float, taco::pw_progress(__builtin_bit_cast(uint32_t, word(Sm, npad, i, TACO_S_PROGRESS))));
    else if (field == TACO_S_DELAY_LEN) val = __builtin_bit_cast(float, taco::dw_L(__builtin_bit_cast(uint32_t, word(Sm, npad, i, TACO_S_DELAY_LEN))));
    else if (field >= TACO_S_ACT && field < TACO_S_ACT + 4) val = last_action(Sm, hist, npad, i, hh, field - TACO_S_ACT);
    else val = word(Sm, npad, i, field);
    out[i] = __builtin_bit_cast(uint32_t, val);
}
}  // namespace

int taco_get_field(taco_env *e, int field, uint32_t *out, void *stream) {
    if (!e || !out) return fail(TACO_ERR_INVALID_ARG, "taco_get_field: null argument");
    if (field < 0 || field >= TACO_NUM_FIELDS || (field >= TACO_S_ACT_OLD && field < TACO_S_ACT_OLD + 4))
        return fail(TACO_ERR_INVALID_ARG, "taco_get_field: field must be a TACO_S_* row other than actions_old (use taco_get_state for that)");
    const int rc = refresh_clock(e, stream);
    if (rc != TACO_OK) return rc;
    const int n = e->cfg.num_envs;
    hipLaunchKernelGGL(export_field_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->S, e->hist, out, field, n, e->npad, e->hh);
    hipError_t he = hipGetLastError();
    return he == hipSuccess ? TACO_OK : hip_fail(he, "export_field_kernel launch");
}

int taco_get_state(taco_env *e, uint32_t *blob, void *stream) {
    if (!e || !blob) return fail(TACO_ERR_INVALID_ARG, "taco_get_state: null argument");
    { const int rc = refresh_clock(e, stream); if (rc != TACO_OK) return rc; }
    const int n = e->cfg.num_envs;
    hipLaunchKernelGGL(export_state_kernel, dim3((n + 255) / 256, TACO_BLOB_ROWS), dim3(256), 0, (hipStream_t)stream, e->S, e->hist, e->ring, blob, n,
                       e->npad, e->head, e->hh);
    hipError_t he = hipGetLastError();
    return he == hipSuccess ? TACO_OK : hip_fail(he, "export_state_kernel launch");
}

int taco_set_state(taco_env *e, const uint32_t *blob, void *stream) {
    if (!e || !blob) return fail(TACO_ERR_INVALID_ARG, "taco_set_state: null argument");
    { const int rc = refresh_clock(e, stream); if (rc != TACO_OK) return rc; }
    const int n = e->cfg.num_envs;
    hipLaunchKernelGGL(import_state_kernel, dim3((n + 255) / 256, TACO_BLOB_ROWS), dim3(256), 0, (hipStream_t)stream, e->S, e->hist, e->ring, blob, n,
                       e->npad, e->head, e->hh);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "import_state_kernel launch");
    if (!e->params_imported) {  // the blob may carry per-env rotor / aero parameters and any queue depth: stop assuming otherwise
        e->params_imported = 1;
        derive(e);
    }
    return TACO_OK;
}

int taco_launch_geometry(const taco_env *e, int *grid, int *block) {
    if (!e || !grid || !block) return fail(TACO_ERR_INVALID_ARG, "taco_launch_geometry: null argument");
    grid_of(e, grid, block);
    return TACO_OK;
}

// ---- policy forward (second half)
static int policy_cfg_ok(const taco_policy_cfg *c) {
    auto p16 = [](int x) { return (x + 15) / 16 * 16; };
    if (!c) return fail(TACO_ERR_INVALID_ARG, "policy cfg is null");
    if (c->obs_len < 1 || c->obs_dim < 1 || p16(c->obs_len * c->obs_dim) > taco::POL_MAXW) return fail(TACO_ERR_INVALID_ARG, "policy: obs_len * obs_dim must be 1..256");
    if (c->act_dim < 1 || c->act_dim > 16) return fail(TACO_ERR_INVALID_ARG, "policy: act_dim must be 1..16");
    if (c->n_actor_hidden < 0 || c->n_actor_hidden > 4 || c->n_critic_hidden < 0 || c->n_critic_hidden > 4) return fail(TACO_ERR_INVALID_ARG, "policy: at most four hidden layers per MLP");
    for (int l = 0; l < c->n_actor_hidden; ++l) if (c->actor_hidden[l] < 1 || c->actor_hidden[l] > taco::POL_MAXW) return fail(TACO_ERR_INVALID_ARG, "policy: hidden widths must be 1..256");
    for (int l = 0; l < c->n_critic_hidden; ++l) if (c->critic_hidden[l] < 1 || c->critic_hidden[l] > taco::POL_MAXW) return fail(TACO_ERR_INVALID_ARG, "policy: hidden widths must be 1..256");
    if (c->states_len < 1 || c->states_dim < 1) return fail(TACO_ERR_INVALID_ARG, "policy: states_len / states_dim must be >= 1");
    if (c->lstm_hidden < 0 || c->lstm_hidden > taco::POL_MAXW) return fail(TACO_ERR_INVALID_ARG, "policy: lstm_hidden must be 0..256");
    if (c->lstm_hidden > 0 && (c->states_len > taco::POL_MAXT || c->states_dim > 32)) return fail(TACO_ERR_INVALID_ARG, "policy: the LSTM critic takes at most 8 frames of at most 32 features");
    if (c->lstm_hidden == 0 && p16(c->states_len * c->states_dim) > taco::POL_MAXW) return fail(TACO_ERR_INVALID_ARG, "policy: states_len * states_dim must be <= 256 without an encoder");
    if (c->flags & ~(int32_t)(TACO_P_EXACT_CELL | TACO_P_SPLIT_F16 | TACO_P_SPLIT_BF16)) return fail(TACO_ERR_INVALID_ARG, "policy: unknown bits in cfg.flags");
    if ((c->flags & TACO_P_EXACT_CELL) && (c->flags & (TACO_P_SPLIT_F16 | TACO_P_SPLIT_BF16)))
        return fail(TACO_ERR_INVALID_ARG, "policy: TACO_P_EXACT_CELL (bit-exact critic) and TACO_P_SPLIT_* (16-bit matrix pipe) exclude each other");
    if ((c->flags & TACO_P_SPLIT_F16) && (c->flags & TACO_P_SPLIT_BF16)) return fail(TACO_ERR_INVALID_ARG, "policy: choose ONE of TACO_P_SPLIT_F16 / TACO_P_SPLIT_BF16");
    return TACO_OK;
}

size_t taco_policy_blob_floats(const taco_policy_cfg *c) {
    if (policy_cfg_ok(c) != TACO_OK) return 0;
    auto p16 = [](int x) { return (x + 15) / 16 * 16; };
    size_t n = 0;
    int in = p16(c->obs_len * c->obs_dim);
    for (int l = 0; l <= c->n_actor_hidden; ++l) {
        const int out = p16(l < c->n_actor_hidden ? c->actor_hidden[l] : c->act_dim);
        n += (size_t)out * in + out;
        in = out;
    }
    n += 16;
    if (c->lstm_hidden > 0) {
        const int hp = p16(c->lstm_hidden), ip = p16(c->states_dim);
        n += (size_t)4 * hp * ip + (size_t)4 * hp * hp + (size_t)4 * hp;
        in = hp;
    } else {
        in = p16(c->states_len * c->states_dim);
    }
    for (int l = 0; l <= c->n_critic_hidden; ++l) {
        const int out = p16(l < c->n_critic_hidden ? c->critic_hidden[l] : 1);
        n += (size_t)out * in + out;
        in = out;
    }
    return n;
}

static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else return 256;
    }
    return cus;
}

static int launch_policy(const taco_policy_cfg *c, const float *blob, int n, const float *obs, const float *states, uint64_t seed, uint32_t call,
                         int deterministic, int action_only, float *action, float *logp, float *value, float *mu, float *sigma,
                         float *action_env, float act_lo, float act_hi, void *stream, uint64_t *stamps = nullptr, const uint32_t *clock = nullptr,
                         uint32_t call_delta = 0) {
    taco::PolicyParams P{};
    P.clock = clock; P.call_delta = call_delta;
    P.obs_len = c->obs_len; P.obs_dim = c->obs_dim; P.states_len = c->states_len; P.states_dim = c->states_dim; P.act_dim = c->act_dim;
    P.n_actor_hidden = c->n_actor_hidden; P.lstm_hidden = c->lstm_hidden; P.n_critic_hidden = c->n_critic_hidden;
    for (int l = 0; l < 4; ++l) { P.actor_hidden[l] = c->actor_hidden[l]; P.critic_hidden[l] = c->critic_hidden[l]; }
    P.blob = blob; P.obs = obs; P.states = states; P.action = action; P.logp = logp; P.value = value; P.mu = mu; P.sigma = sigma;
    P.action_env = action_env; P.act_lo = act_lo; P.act_hi = act_hi;
    P.stamps = (unsigned long long *)stamps;
    P.n = n; P.deterministic = deterministic ? 1 : 0;
    P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32); P.call = call;
    P.role0 = 0; P.value_tail = nullptr; P.value_split = INT_MAX;
    // the actor alone above 8 192 rows, documented architecture: the throughput form (64 rows per pass, two workgroups per CU); otherwise the
    // 16-rows-per-workgroup form, which is the faster one while its workgroups fit the chip in two rounds
    auto p16 = [](int x) { return (x + 15) / 16 * 16; };
    const bool actor_throughput = action_only && n > 8192 && !stamps && c->act_dim == 4 && p16(c->obs_len * c->obs_dim) == 32 && c->n_actor_hidden == 3 &&
                                  p16(c->actor_hidden[0]) == 128 && p16(c->actor_hidden[1]) == 128 && p16(c->actor_hidden[2]) == 128;
    if (actor_throughput) {
        const int npass = (n + taco::ACT_ROWS - 1) / taco::ACT_ROWS, cap = 2 * device_cus();
        hipLaunchKernelGGL(taco::taco_actor_kernel, dim3((unsigned)(npass < cap ? npass : cap)), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
    } else {
        hipLaunchKernelGGL(taco::taco_policy_kernel, dim3((n + taco::POL_ROWS - 1) / taco::POL_ROWS, action_only ? 1 : 2), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "policy kernel launch");
    return TACO_OK;
}

// the critic alone over `rows` state stacks: value[row] for row < split, value_tail[row - split] beyond.  The batched form
// (taco_critic_kernel) covers the documented architecture; any other geometry runs the critic role of taco_policy_kernel, same results.
static bool critic_batched_form(const taco_policy_cfg *c) {
    auto p16 = [](int x) { return (x + 15) / 16 * 16; };
    return c->lstm_hidden > 0 && p16(c->lstm_hidden) == 128 && p16(c->states_dim) == 32 && c->states_len <= taco::POL_MAXT && c->n_critic_hidden == 2 &&
           p16(c->critic_hidden[0]) == 128 && p16(c->critic_hidden[1]) == 128;
}
// cu_limit > 0: the persistent LSTM kernel uses at most that many workgroups (a stream that owns fewer CUs than the device has)
// ring_n > 0: `states` is a frame ring [slots + states_len - 1][ring_n][states_dim] and row = slot * ring_n + env (taco_policy.hpp, PolicyParams.ring_n)
static_assert(taco::kStatusCriticClamped == taco::kStatusCriticClampedBit, "the policy header's copy of the status bit");
// the batched critic's workspace: h_T [rows][128] f32, then a 256-byte tail whose first word counts the split-f16 form's clamp events of the LAST call
static uint32_t *critic_ws_tail(float *workspace, size_t rows) { return (uint32_t *)(workspace + rows * 128); }
static int launch_critic(const taco_policy_cfg *c, const float *blob, size_t rows, const float *states, float *value, float *value_tail, size_t split,
                         float *workspace, int cu_limit, void *stream, uint64_t *stamps = nullptr, int ring_n = 0, uint32_t *env_status = nullptr) {
    if (rows > (size_t)INT_MAX - 64) return fail(TACO_ERR_INVALID_ARG, "critic: too many rows for one launch");
    taco::PolicyParams P{};
    P.obs_len = c->obs_len; P.obs_dim = c->obs_dim; P.states_len = c->states_len; P.states_dim = c->states_dim; P.act_dim = c->act_dim;
    P.n_actor_hidden = c->n_actor_hidden; P.lstm_hidden = c->lstm_hidden; P.n_critic_hidden = c->n_critic_hidden;
    for (int l = 0; l < 4; ++l) { P.actor_hidden[l] = c->actor_hidden[l]; P.critic_hidden[l] = c->critic_hidden[l]; }
    P.blob = blob; P.states = states; P.value = value; P.value_tail = value_tail;
    P.value_split = split > rows ? INT_MAX : (int)split;
    P.n = (int)rows; P.role0 = 1; P.hT = workspace; P.stamps = (unsigned long long *)stamps; P.ring_n = ring_n;
    if (critic_batched_form(c)) {
        if (!workspace || ((uintptr_t)workspace & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "critic: the batched form needs a 16-byte aligned workspace (taco_critic_workspace_bytes)");
        if (((uintptr_t)states & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "critic: the batched form needs 16-byte aligned state stacks");
        const int cus = cu_limit > 0 ? cu_limit : device_cus();
        const bool exact = (c->flags & TACO_P_EXACT_CELL) != 0;
        int split = 0;   // 1 / 2: the split-f16 / split-bf16 kernels (TACO_P_SPLIT_*: the pair form of the ring LSTM with states_dim < 32, and the MLP behind it)
        // a persistent workgroup per CU walking blocks b, b + grid, ...: shrink the grid to the smallest that keeps the longest walk as short
        const size_t nblocks = (rows + taco::CR_ROWS - 1) / taco::CR_ROWS;
        const size_t walk = (nblocks + cus - 1) / cus;
        const unsigned grid = (unsigned)((nblocks + walk - 1) / walk);
        if (ring_n > 0 && c->states_len >= 2 && rows % (size_t)ring_n == 0 && rows / (size_t)ring_n >= 2) {
            // frame ring with at least two slots: blocks of 16 envs x two consecutive slots share the input projection (taco_critic_lstm_pair_kernel)
            const size_t pb = (((size_t)ring_n + 15) / 16) * ((rows / (size_t)ring_n + 1) / 2);
            const size_t pw = (pb + cus - 1) / cus;
            const dim3 pg((unsigned)((pb + pw - 1) / pw));
            split = (exact || c->states_dim >= 32) ? 0 : ((c->flags & TACO_P_SPLIT_F16) ? 1 : ((c->flags & TACO_P_SPLIT_BF16) ? 2 : 0));
            if (split == 1) {   // the f16 form counts what it saturates: the tail word starts every call at zero (a memset node when captured)
                P.clamped = critic_ws_tail(workspace, rows); P.env_status = env_status;
                const hipError_t me = hipMemsetAsync(P.clamped, 0, sizeof(uint32_t), (hipStream_t)stream);
                if (me != hipSuccess) return hip_fail(me, "critic: zeroing the clamp counter");
            }
            if (exact) hipLaunchKernelGGL(taco::taco_critic_lstm_pair_kernel<true>, pg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
            else if (split == 1) hipLaunchKernelGGL(taco::taco_critic_lstm_pair_split_kernel<1>, pg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
            else if (split == 2) hipLaunchKernelGGL(taco::taco_critic_lstm_pair_split_kernel<2>, pg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
            else hipLaunchKernelGGL(taco::taco_critic_lstm_pair_kernel<false>, pg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
        } else if (ring_n > 0) {
            if (exact) hipLaunchKernelGGL((taco::taco_critic_lstm_kernel<true, true>), dim3(grid), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
            else hipLaunchKernelGGL((taco::taco_critic_lstm_kernel<true, false>), dim3(grid), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
        } else {
            if (exact) hipLaunchKernelGGL((taco::taco_critic_lstm_kernel<false, true>), dim3(grid), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
            else hipLaunchKernelGGL((taco::taco_critic_lstm_kernel<false, false>), dim3(grid), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
        }
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) return hip_fail(he, "taco_critic_lstm_kernel launch");
        const size_t nchunks = (rows + taco::CR_MLP_ROWS - 1) / taco::CR_MLP_ROWS;
        const size_t cap = (size_t)2 * cus;
        const dim3 mg((unsigned)(nchunks < cap ? nchunks : cap));
        if (split == 1) hipLaunchKernelGGL(taco::taco_critic_mlp_split_kernel<1>, mg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);   // (the forms the split LSTM ran in)
        else if (split == 2) hipLaunchKernelGGL(taco::taco_critic_mlp_split_kernel<2>, mg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
        else hipLaunchKernelGGL(taco::taco_critic_mlp_kernel, mg, dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
    } else {
        hipLaunchKernelGGL(taco::taco_policy_kernel, dim3((unsigned)((rows + taco::POL_ROWS - 1) / taco::POL_ROWS), 1), dim3(64 * taco::POL_NW), 0, (hipStream_t)stream, P);
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "critic kernel launch");
    return TACO_OK;
}

size_t taco_critic_workspace_bytes(const taco_policy_cfg *c, int64_t rows) {
    if (policy_cfg_ok(c) != TACO_OK || rows < 1) return 0;
    return critic_batched_form(c) ? (size_t)rows * 128 * sizeof(float) + 256 : 16;
}

int taco_critic_clamped_words(const taco_policy_cfg *c, int64_t rows, const void *workspace, uint32_t *count, void *stream) {
    const int rc = policy_cfg_ok(c);
    if (rc != TACO_OK) return rc;
    if (!workspace || !count || rows < 1) return fail(TACO_ERR_INVALID_ARG, "taco_critic_clamped_words: bad argument");
    *count = 0;
    if (!critic_batched_form(c) || !(c->flags & TACO_P_SPLIT_F16) || (c->flags & TACO_P_EXACT_CELL)) return TACO_OK;   // (only the split-f16 form saturates anything)
    hipError_t he = hipMemcpyAsync(count, critic_ws_tail((float *)const_cast<void *>(workspace), (size_t)rows), sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "taco_critic_clamped_words");
    return TACO_OK;
}

int taco_critic_values(const taco_policy_cfg *c, const float *blob, int64_t rows, const float *states, float *value, void *workspace, uint64_t *stamps,
                       void *stream) {
    const int rc = policy_cfg_ok(c);
    if (rc != TACO_OK) return rc;
    if (!blob || !states || !value) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values: null buffer pointer");
    if (rows < 1) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values: rows must be >= 1");
    if (((uintptr_t)blob & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values: the weight blob must be 16-byte aligned");
    if (stamps && ((uintptr_t)stamps & 7u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values: stamps must be 8-byte aligned");
    return launch_critic(c, blob, (size_t)rows, states, value, nullptr, (size_t)rows + 1, (float *)workspace, 0, stream, stamps);
}

#ifdef TACO_LSTM_STAMPS   // A/B probe builds only (never defined in the product build): where taco_critic_values_ring's split LSTM kernel leaves its segment stamps
static uint64_t *g_lstm_stamps = nullptr;
extern "C" void taco_debug_set_lstm_stamps(uint64_t *p) { g_lstm_stamps = p; }
#endif
int taco_critic_values_ring(const taco_policy_cfg *c, const float *blob, int64_t slots, int num_envs, const float *frames, float *value, void *workspace,
                            void *stream) {
    const int rc = policy_cfg_ok(c);
    if (rc != TACO_OK) return rc;
    if (!blob || !frames || !value) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values_ring: null buffer pointer");
    if (slots < 1 || num_envs < 1) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values_ring: slots and num_envs must be >= 1");
    if (((uintptr_t)blob & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_critic_values_ring: the weight blob must be 16-byte aligned");
    const size_t rows = (size_t)slots * (size_t)num_envs;
#ifdef TACO_LSTM_STAMPS
    return launch_critic(c, blob, rows, frames, value, nullptr, rows + 1, (float *)workspace, 0, stream, g_lstm_stamps, num_envs);
#else
    return launch_critic(c, blob, rows, frames, value, nullptr, rows + 1, (float *)workspace, 0, stream, nullptr, num_envs);
#endif
}

int taco_policy_act_stamped(const taco_policy_cfg *c, const float *blob, int n, const float *obs, const float *states, uint64_t seed, uint32_t call,
                            int deterministic, int action_only, float *action, float *logp, float *value, float *mu, float *sigma, uint64_t *stamps,
                            void *stream) {
    const int rc = policy_cfg_ok(c);
    if (rc != TACO_OK) return rc;
    if (!blob || !obs || !action || !logp || !mu || !sigma) return fail(TACO_ERR_INVALID_ARG, "taco_policy_act: null buffer pointer");
    if (!action_only && (!states || !value)) return fail(TACO_ERR_INVALID_ARG, "taco_policy_act: states / value are needed unless action_only");
    if (n < 1) return fail(TACO_ERR_INVALID_ARG, "taco_policy_act: n must be >= 1");
    if (((uintptr_t)blob & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_policy_act: the weight blob must be 16-byte aligned");
    if (stamps && ((uintptr_t)stamps & 7u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_policy_act: stamps must be 8-byte aligned");
    return launch_policy(c, blob, n, obs, states, seed, call, deterministic, action_only, action, logp, value, mu, sigma, nullptr, 0.0f, 0.0f, stream, stamps);
}
int taco_policy_act(const taco_policy_cfg *c, const float *blob, int n, const float *obs, const float *states, uint64_t seed, uint32_t call,
                    int deterministic, int action_only, float *action, float *logp, float *value, float *mu, float *sigma, void *stream) {
    return taco_policy_act_stamped(c, blob, n, obs, states, seed, call, deterministic, action_only, action, logp, value, mu, sigma, nullptr, stream);
}

// The persistent actor + step kernel (taco_fused.hpp) covers the documented rollout: one observation frame per step (len_obs = 1), the actor MLP
// 26-128-128-128-4 (widths padded to 32 / 128 / 16), launches of at most kFusedMaxEnvs envs (16 envs per workgroup, one workgroup per CU at a
// time: 512 workgroups are two rounds of the chip).  Anything else runs the launch-per-step path, same results.
// Round 6 built the same kernel with ONE LANE PER ENV (taco_rollout_kernel<1>: 64 envs per workgroup, the actor over four row tiles; 32 768 envs = 512
// workgroups = two rounds of the chip instead of the quad form's eight) for the range above kFusedMaxEnvs: bit-identical (tests/test_rollout_gpu.py runs
// every case in both forms), and SLOWER than the launch-per-step path -- 2.39-2.40 vs 2.04-2.05 ms per rollout at 32 768 x 16, 9.2 vs 6.8 at 131 072 x 16
// (profiles/r06_b_rollout_ab_lane.txt): its step takes 71 k clocks = 29.7 us because the actor's f32 MFMAs of 64 rows (77.8 k MFMA cycles per step) have the
// TWO SIMDs the step and battery wavefronts leave them, i.e. >= 16.2 us even at full rate, where the stand-alone actor kernel has all four of every CU
// (28.8 us for 32 768 rows).  The per-step chain above 8 192 envs is bound by the f32 matrix pipe, not by launches: taco_set_rollout_fusion(env, 3) keeps the
// form reachable for the A/B record, the host's choice stays launch-per-step there.
constexpr int kFusedMaxEnvs = 8192;          // the quad form (16 envs per workgroup): the host's choice up to here
static bool fused_rollout_ok(const taco_env *e, const taco_policy_cfg *c) {
    auto p16 = [](int x) { return (x + 15) / 16 * 16; };
    return !e->fusion_off && !e->form_pinned && e->gather == nullptr && (e->cfg.num_envs <= kFusedMaxEnvs || e->fusion_force) && e->cfg.len_obs == 1 && c->obs_len == 1 &&
           c->obs_dim == 26 && c->act_dim == 4 && c->n_actor_hidden == 3 && p16(c->actor_hidden[0]) == 128 && p16(c->actor_hidden[1]) == 128 &&
           p16(c->actor_hidden[2]) == 128;
}

// ---- a whole PPO rollout (ppo_asymmetry.py:308-342) enqueued from C: 2 H + 2 launches, no host work between the steps
int taco_rollout_run(taco_env *e, const taco_policy_cfg *c, const float *blob, const taco_rollout_bufs *b, int horizon, uint64_t seed, uint32_t call0,
                     double gamma, double act_lo, double act_hi, int64_t *reset_buf, void *stream) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    int rc = policy_cfg_ok(c);
    if (rc != TACO_OK) return rc;
    if (!b || !blob || !reset_buf) return fail(TACO_ERR_INVALID_ARG, "taco_rollout_run: null argument");
    if (!b->obs_store || !b->states_store || !b->act_buf || !b->act_env || !b->rew_buf || !b->done_buf || !b->value_buf || !b->logp_buf || !b->mu_buf ||
        !b->sigma_buf || !b->timeout_buf || !b->last_value || !b->critic_ws)
        return fail(TACO_ERR_INVALID_ARG, "taco_rollout_run: null buffer pointer");
    if (horizon < 1) return fail(TACO_ERR_INVALID_ARG, "taco_rollout_run: horizon must be >= 1");
    if (((uintptr_t)blob & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_rollout_run: the weight blob must be 16-byte aligned");
    const taco_cfg &ec = e->cfg;
    if (c->obs_len != ec.len_obs || c->states_len != ec.len_states || c->obs_dim != 26 || c->states_dim != 26 || c->act_dim != 4)
        return fail(TACO_ERR_INVALID_ARG, "taco_rollout_run: policy and env geometry differ");
    const size_t n = (size_t)ec.num_envs;
    const size_t obs_slot = n * ec.len_obs * 26, frame = n * 26;   // states_store is a frame ring: [horizon + len_states][n][26]
    // On a capturing stream the env's clock lives on the device (launch_step); the actor then takes its noise counter from it too:
    // counter of step t = call0 + t = (device step word) + (call0 - step count at the start of the rollout), whatever replay this is.
    if (is_capturing(stream)) note_capture(e, stream);
    else if ((rc = refresh_clock(e, stream)) != TACO_OK) return rc;   // eager, after replays: the host's clock is made current first (blocking, once)
    const uint32_t *clock = e->clock_on_device ? e->ctl + taco::kCtlStep : nullptr;
    const uint32_t call_delta = call0 - (uint32_t)e->step_count;
    const bool fused = fused_rollout_ok(e, c);
    if (fused) {
        // the whole per-step chain as ONE persistent launch (taco_fused.hpp): a workgroup owns 16 envs for all `horizon` steps
        taco::RolloutParams R{};
        R.S = e->P;
        R.S.S = e->S; R.S.ring = e->ring; R.S.hist = e->hist; R.S.reset = (long long *)reset_buf; R.S.ctl = e->ctl; R.S.wclk = e->wclk; R.S.use_ctl = e->clock_on_device; R.S.stamps_on = 0;
        R.S.st_period = 0; R.S.st_phase = e->st_phase; R.S.st_front = 0;
        R.S.step = (uint32_t)e->step_count; R.S.head = e->head; R.S.hh = e->hh; R.S.gather_row = (uint32_t)taco_gather_row_floats(1); R.S.gather_bytes = 0;
        R.blob = blob; R.pseed_lo = (uint32_t)seed; R.pseed_hi = (uint32_t)(seed >> 32); R.call0 = call0; R.call_delta = call_delta;
        R.act_lo = (float)act_lo; R.act_hi = (float)act_hi;
        R.obs_store = b->obs_store; R.frames = b->states_store; R.act_buf = b->act_buf; R.mu_buf = b->mu_buf; R.sigma_buf = b->sigma_buf;
        R.logp_buf = b->logp_buf; R.rew_buf = b->rew_buf; R.done_buf = b->done_buf; R.timeout_buf = b->timeout_buf;
        R.horizon = horizon; R.len_states = ec.len_states; R.stamps = e->rollout_stamps;
        // fusion_force: 2 = the quad form whatever the env count (round 5's A/B), 3 = the one-lane form whatever the env count
        const bool lane_form = e->fusion_force == 3;
        if (lane_form) hipLaunchKernelGGL(taco::taco_rollout_kernel<1>, dim3((unsigned)((n + 63) / 64)), dim3(taco::FU_THREADS), 0, (hipStream_t)stream, R);
        else hipLaunchKernelGGL(taco::taco_rollout_kernel<4>, dim3((unsigned)((n + taco::POL_ROWS - 1) / taco::POL_ROWS)), dim3(taco::FU_THREADS), 0, (hipStream_t)stream, R);
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) return hip_fail(he, "taco_rollout_kernel launch");
        e->step_count += horizon;   // (the kernel left the clock after `horizon` steps on the device itself)
        e->head = (e->head + 10 * horizon) % TACO_RING_SLOTS;
        e->hh = (e->hh + horizon) % taco::HIST_ROWS;
    }
    for (int t = 0; !fused && t < horizon; ++t) {
        float *act_t = b->act_buf + (size_t)t * n * 4;
        rc = launch_policy(c, blob, (int)n, b->obs_store + (size_t)t * obs_slot, nullptr, seed, call0 + (uint32_t)t, 0, 1,
                           act_t, b->logp_buf + (size_t)t * n, nullptr, b->mu_buf + (size_t)t * n * 4, b->sigma_buf + (size_t)t * n * 4,
                           b->act_env, (float)act_lo, (float)act_hi, stream, nullptr, clock, call_delta);
        if (rc != TACO_OK) return rc;
        taco_rollout_io io{};
        io.actions = b->act_env;
        io.obs_prev = b->obs_store + (size_t)t * obs_slot; io.obs_next = b->obs_store + (size_t)(t + 1) * obs_slot;
        io.states_prev = nullptr; io.states_next = b->states_store + (size_t)(t + ec.len_states) * frame;   // the newest frame of slot t + 1
        io.states_newest_only = 1;
        io.rew = b->rew_buf + (size_t)t * n; io.reset_buf = reset_buf; io.timeout_buf = b->timeout_buf + (size_t)t * n;
        io.done_f32 = b->done_buf + (size_t)t * n;
        rc = launch_step(e, &io, stream);
        if (rc != TACO_OK) return rc;
    }
    // the values of all H + 1 state stacks (:311 value of slot t, :341 of the final state) in one launch: slots 0..H-1 -> value_buf, H -> last_value
    const size_t count = (size_t)horizon * n;
    rc = launch_critic(c, blob, count + n, b->states_store, b->value_buf, b->last_value, count, b->critic_ws, 0, stream, nullptr, ec.len_states > 1 ? (int)n : 0,
                       e->ctl + taco::kCtlStatus);
    if (rc != TACO_OK) return rc;
    hipLaunchKernelGGL(taco::timeout_bootstrap_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b->rew_buf, b->value_buf, b->done_buf,
                       b->timeout_buf, count, (float)gamma);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "timeout_bootstrap_kernel launch");
    return TACO_OK;
}

int taco_bind_states_ring(taco_env *e, float *ring, int rows) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    const int front = e->cfg.len_states - 1;
    if (ring) {
        if (front < 1) return fail(TACO_ERR_INVALID_ARG, "taco_bind_states_ring: len_states is 1, there is no stack to keep");
        if (rows < 2 * front + 1) return fail(TACO_ERR_INVALID_ARG, "taco_bind_states_ring: rows must be >= 2 * (len_states - 1) + 1");
        if (((uintptr_t)ring & 15u) != 0) return fail(TACO_ERR_INVALID_ARG, "taco_bind_states_ring: ring must be 16-byte aligned");
    }
    const int rc = refresh_clock(e);
    if (rc != TACO_OK) return rc;
    hipError_t he = hipDeviceSynchronize();   // (a step in flight would write ITS successor phase behind this)
    if (he == hipSuccess) {
        const int groups = e->npad / 16;
        hipLaunchKernelGGL(set_clock_kernel, dim3((groups + 255) / 256), dim3(256), 0, (hipStream_t) nullptr, e->wclk, e->ctl, groups, 0, 0u, 1);
        he = hipGetLastError();
        if (he == hipSuccess) he = hipDeviceSynchronize();
    }
    if (he != hipSuccess) return hip_fail(he, "taco_bind_states_ring: resetting the ring phase");
    e->st_ring = ring; e->st_rows = ring ? rows : 0; e->st_period = ring ? rows - front : 0; e->st_phase = 0;
    return TACO_OK;
}
int taco_states_ring_row(taco_env *e) {
    if (!e || !e->st_ring) return fail(TACO_ERR_INVALID_ARG, "taco_states_ring_row: no ring bound"), -1;
    if (refresh_clock(e) != TACO_OK) return -1;
    return (e->st_phase + e->st_period - 1) % e->st_period;   // the phase of the LAST step: it wrote row phase + len_states - 1, the newest frame of the window
}

int taco_step_ring(taco_env *e, const taco_rollout_io *io, void *stream, int32_t *phase) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (!io || !phase) return fail(TACO_ERR_INVALID_ARG, "taco_step_ring: null argument");
    if (!e->st_ring || io->states_next) return fail(TACO_ERR_INVALID_ARG, "taco_step_ring: bind a ring (taco_bind_states_ring) and pass states_next = NULL");
    if (is_capturing(stream))
        return fail(TACO_ERR_STATE, "taco_step_ring cannot be captured into a HIP graph: every replay fills another window of the frame ring, which the "
                                    "caller of this entry point wants to be told (capture taco_step_rollout with states_next = NULL instead)");
    const int rc0 = refresh_clock(e, stream);   // (after graph replays: the host's phase is made current first -- blocking, once)
    if (rc0 != TACO_OK) return rc0;
    *phase = e->st_phase;
    return launch_step(e, io, stream);
}

int taco_set_kernel_form(taco_env *e, int form) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (form < TACO_FORM_AUTO || form > TACO_FORM_QUAD_SERVED) return fail(TACO_ERR_INVALID_ARG, "taco_set_kernel_form: unknown form");
    e->form = form == TACO_FORM_AUTO ? choose_form(e->cfg) : form;
    e->form_pinned = form != TACO_FORM_AUTO;
    return TACO_OK;
}
int taco_get_kernel_form(const taco_env *e) { return e ? e->form : TACO_ERR_INVALID_ARG; }

#ifdef TACO_TEST_HOOKS
// test-hooks build only (taco_amd/build.py --test-hooks): make the battery-server wavefront late on purpose, so that the wait path of
// the LDS mailbox is exercised
int taco_test_slow_battery_server(taco_env *e, int on) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    e->slow_server = on ? 1 : 0;
    derive(e);
    return TACO_OK;
}
#endif

int taco_set_rollout_fusion(taco_env *e, int on) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    e->fusion_off = on ? 0 : 1;
    e->fusion_force = (on == 2 || on == 3) ? on : 0;   // 2: the quad form whatever the env count, 3: the one-lane form whatever the env count
    return TACO_OK;
}
int taco_bind_rollout_stamps(taco_env *e, uint64_t *stamps) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (stamps && ((uintptr_t)stamps & 7u) != 0) return fail(TACO_ERR_INVALID_ARG, "stamps must be 8-byte aligned");
    e->rollout_stamps = (unsigned long long *)stamps;
    return TACO_OK;
}

int taco_bind_phase_stamps(taco_env *e, uint64_t *stamps) {
    if (!e) return fail(TACO_ERR_INVALID_ARG, "env is null");
    if (stamps && ((uintptr_t)stamps & 7u) != 0) return fail(TACO_ERR_INVALID_ARG, "stamps must be 8-byte aligned");
    e->stamps = (unsigned long long *)stamps;
    return TACO_OK;
}

int taco_occupancy(const taco_env *e, int *resident_blocks_per_cu, int *lds_bytes_per_block) {
    if (!e || !resident_blocks_per_cu || !lds_bytes_per_block) return fail(TACO_ERR_INVALID_ARG, "taco_occupancy: null argument");
    const FormInfo f = form_info(e->form, false, wide_form(e->cfg, e->form), lin_form(e->cfg, e->form));
    hipFuncAttributes at;
    hipError_t he = hipFuncGetAttributes(&at, f.fn);
    if (he != hipSuccess) return hip_fail(he, "hipFuncGetAttributes");
    *lds_bytes_per_block = (int)at.sharedSizeBytes;
    he = hipOccupancyMaxActiveBlocksPerMultiprocessor(resident_blocks_per_cu, f.fn, f.block, 0);
    if (he != hipSuccess) return hip_fail(he, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    return TACO_OK;
}

}  // extern "C"
