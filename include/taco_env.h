/*
 * taco_env.h -- C ABI of libtaco_env.so: the MI355X-native drop-in for the reference's fpv_asymmetry.step() hot path.
 *
 * What this boundary replaces in the reference (paths under yinzikang/taco):
 *   - the Isaac Gym tensor API the task drives on every step (python/isaacgym/gymtorch.py:60-104 wrap_tensor /
 *     unwrap_tensor over python/isaacgym/_bindings/src/gymtorch/gymtorch.cpp:33-158, and the gymapi calls
 *     refresh_actor_root_state_tensor / apply_rigid_body_force_tensors / simulate / set_actor_root_state_tensor_indexed
 *     at IsaacGymEnvs/isaacgymenvs/tasks/fpv_asymmetry.py:335-336, :633-635, :506-508 and
 *     tasks/base/vec_task_asymmetry.py:313), and
 *   - the ~2 400 per-step torch launches of fpv_asymmetry.py / tasks/control/ (SURVEY.md section 3.1),
 * by ONE kernel launch per VecTask.step().
 *
 * Ownership is inverted with respect to gymtorch (where the simulator owns the memory and torch borrows it): the caller
 * (PyTorch-ROCm, or any HIP program) allocates every buffer; the library borrows raw device pointers for the duration
 * of a call and never allocates device memory, never synchronises the host and never calls exit().  All work is
 * enqueued on the caller's HIP stream.  Signatures use plain pointers and sizes only (no torch types).
 *
 * Errors: every function returns TACO_OK (0) or a negative taco_status; taco_last_error() returns a thread-local
 * message (the reference prints "***" messages and calls quit(), vec_task_asymmetry.py:268-270).
 * Threading: calls on one handle are not re-entrant; different handles are independent.
 */
#ifndef TACO_ENV_H
#define TACO_ENV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TACO_ABI_VERSION 8

typedef enum taco_status {
    TACO_OK = 0,
    TACO_ERR_INVALID_ARG = -1,  /* null pointer, bad size, unsupported cfg value */
    TACO_ERR_HIP = -2,          /* a HIP runtime call failed; message holds hipGetErrorString */
    TACO_ERR_WORKSPACE = -3,    /* workspace too small / misaligned */
    TACO_ERR_STATE = -4         /* call order violated (e.g. step before init) */
} taco_status;

/* task_mode: registry keys "Fpv_pos" / "Fpv_rotate" / "Fpv_flip" / "Fpv_mix" (tasks/__init__.py:30-39) */
enum { TACO_TASK_POS = 0, TACO_TASK_ROTATE = 1, TACO_TASK_FLIP = 2, TACO_TASK_MIX = 3 };

/* One bit per boolean cfg key the hot path reads (fpv_asymmetry.py:63-112). */
enum {
    TACO_F_RANDOM_COPTER_POS = 1u << 0,
    TACO_F_RANDOM_COPTER_QUAT = 1u << 1,
    TACO_F_RANDOM_COPTER_VEL = 1u << 2,
    TACO_F_RANDOM_TARGET_POS = 1u << 3,
    TACO_F_RANDOM_TARGET_YAW = 1u << 4,
    TACO_F_BATTERY_CONSUMPTION = 1u << 5,
    TACO_F_RANDOM_VOLTAGE = 1u << 6,
    TACO_F_ROTOR_NOISE = 1u << 7,
    TACO_F_ROTOR_DELAY = 1u << 8,
    TACO_F_ROTOR_RESPONSE = 1u << 9,
    TACO_F_RANDOM_ROTORDYNAMIC_COE = 1u << 10,
    TACO_F_RANDOM_ROTOR_DELAY = 1u << 11,
    TACO_F_RANDOM_ROTOR_RESPONSE = 1u << 12,
    TACO_F_RANDOM_ROTOR_SPEED = 1u << 13,
    TACO_F_RANDOM_AERODYNAMIC_COE = 1u << 14,
    TACO_F_RANDOM_DELAY_TIME = 1u << 15,  /* cfg key "ramdom_delay_time" (sic, fpv_asymmetry.py:89) */
    TACO_F_RANDOM_DEPLOY_TIME = 1u << 16, /* cfg key "ramdom_deploy_time" (sic, fpv_asymmetry.py:90) */
    TACO_F_RANDOM_COMMAND = 1u << 17,
    TACO_F_OBSERVATION_NOISE = 1u << 18,
    /* cfg key "world_rate_roundtrip" (no counterpart key in the reference: this is the reference's own DATA FLOW, and the DEFAULT of the host
     * layer -- a C caller should set it too).  The reference keeps the angular velocity in the root state, world frame, and re-derives the body
     * rates from it at every refresh_state (quat_rotate(conj(q), w), fpv_asymmetry.py:350) -- ten times per step.  With this flag the kernel
     * does the same: after every simulate() w = R(q) b, b = R(q)^T w; the fast (PLAIN) loop forms exist for this mode.  Without it the body
     * rates are carried across the ten substeps and the world-frame value is written once: the same numbers up to one rounding per substep,
     * which is OUTSIDE 1e-5 of the reference per step (2.1e-5 on the body rates against 7.9e-6; tests/util.py teacher_forced_one_step,
     * DESIGN.md section 5) and runs the general loop forms.  Either way bit-identical to the oracle in the same mode. */
    TACO_F_WORLD_RATE_ROUNDTRIP = 1u << 19,
    /* cfg key "record_flag" (fpv_asymmetry.py:113, :386-388): copter_rpy_old / copter_rpy_continuous are kept up to date for EVERY env, as
     * the reference does (:339-347) -- they are consumed by the flip command only, so without this flag the kernel maintains them for
     * flip envs alone and the recorder's rows 20..25 of other envs would be stale. */
    TACO_F_TRACK_RPY = 1u << 20
};

typedef struct taco_cfg {
    int32_t num_envs;         /* envs on this handle (cfg["env"]["numEnvs"], or this rank's shard of it) */
    int32_t env_offset;       /* global id of local env 0: keys the random streams and the FpvMix thirds */
    int32_t num_envs_global;  /* numEnvs of the whole job (fpv_asymmetry.py:924-926) */
    int32_t task_mode;        /* TACO_TASK_* */
    int32_t len_obs;          /* cfg["env"]["lenObservations"] */
    int32_t len_states;       /* cfg["env"]["lenStates"] */
    int32_t control_freq_inv; /* cfg["env"]["controlFrequencyInv"]; must be 10 (the delay line hard-codes 10 slots/action) */
    int32_t substeps;         /* cfg["sim"]["substeps"], 1..8 */
    int32_t max_episode_length;
    int32_t delay_time;       /* ms, 0..90 (delay_time_max is 100, fpv_asymmetry.py:87,329) */
    uint32_t flags;           /* TACO_F_* */
    uint32_t _pad;
    uint64_t seed;
    double dt;                /* cfg["sim"]["dt"] */
    double rotor_response_time;
    double difficulty;        /* writable at run time through taco_set_difficulty (ppo_asymmetry.py:173-175) */
    double clip_actions, clip_obs, clip_states; /* +inf = none (vec_task_asymmetry.py:98-100) */
    double mass;              /* composite mass of the 9 welded bodies [kg] */
    double inertia[3];        /* composite principal inertia about the chassis origin [kg m^2] */
    double arm_x, arm_y;      /* rotor offsets (+-arm_x, +-arm_y) [m] (assets/xml/fpv_without_duct.xml:8-33) */
    double gravity_z;         /* -9.81 (fpv_asymmetry.py:214-217) */
} taco_cfg;

/* ---- State blob (taco_get_state / taco_set_state; parity tests + checkpoint/restore).
 * [TACO_BLOB_ROWS][num_envs] 32-bit words, field-major.  Rows 65 and 66 are int32, all others float.
 * Rows 67.. are the pending-action delay line (actions_remained_buffer, fpv_asymmetry.py:189) in LOGICAL order,
 * row = 67 + slot*4 + channel, slot 0 = the action applied in the next 1-ms substep. */
enum {
    TACO_S_POS = 0,        /* 3  copter position (root state 0:3)            */
    TACO_S_QUAT = 3,       /* 4  copter attitude xyzw (3:7)                   */
    TACO_S_LINVEL = 7,     /* 3  world-frame linear velocity (7:10)           */
    TACO_S_ANGVEL = 10,    /* 3  world-frame angular velocity (10:13)         */
    TACO_S_TGT_POS = 13,   /* 3                                               */
    TACO_S_TGT_QUAT = 16,  /* 4                                               */
    TACO_S_RPY_OLD = 20,   /* 3  copter_rpy_old                               */
    TACO_S_RPY_CONT = 23,  /* 3  copter_rpy_continuous                        */
    TACO_S_PID_PREV = 26,  /* 3  angvel_control.previous_error                */
    TACO_S_PID_INT = 29,   /* 3  angvel_control.integral                      */
    TACO_S_BAT_E = 32, TACO_S_BAT_U1 = 33, TACO_S_BAT_T = 34, TACO_S_BAT_V = 35,
    TACO_S_OMEGA = 36,     /* 4  rotor_speed (noised value that is fed back)  */
    TACO_S_ACT = 40,       /* 4  actions                                      */
    TACO_S_ACT_OLD = 44,   /* 4  actions_old                                  */
    TACO_S_CMD = 48,       /* 2  command                                      */
    TACO_S_FLIP_RADIAN = 50,
    TACO_S_TAU = 51,       /* 4  rotor response_time                          */
    TACO_S_OPARA = 55,     /* 5  rotor omega_para                             */
    TACO_S_CF = 60, TACO_S_CT = 61, TACO_S_DX = 62, TACO_S_DY = 63, TACO_S_KT = 64, /* aero parameters */
    TACO_S_PROGRESS = 65,  /* int32 progress_buf                              */
    TACO_S_DELAY_LEN = 66, /* int32 actions_remained_length                   */
    TACO_NUM_FIELDS = 67,
    TACO_RING_SLOTS = 100,
    TACO_BLOB_ROWS = 67 + 400
};

typedef struct taco_env taco_env;

/* ABI version of the loaded library (== TACO_ABI_VERSION of the header it was built from). */
int taco_abi_version(void);
/* 16 hex digits identifying the sources (csrc/, this header, compiler flags) the binary was built from; the Python binding rebuilds
 * a library whose hash differs from the sources on disk. */
const char *taco_source_hash(void);
/* Thread-local text of the last error on this thread ("" if none). */
const char *taco_last_error(void);

/* Bytes of device workspace taco_create needs for `cfg` (16 float4 state chunks + 16 float4 action-history rows +
 * 100 float4 ring slots per env, laid out in tiles of 64 envs, env count padded to whole tiles; + 8 bytes per 16 envs for the
 * device-resident step clock + a 256-byte control block; the pointer must be 256-byte aligned). */
size_t taco_workspace_bytes(const taco_cfg *cfg);

/* Replaces FpvBase.__init__ -> VecTask.__init__ -> create_sim/prepare_sim (fpv_asymmetry.py:54-200,
 * vec_task_asymmetry.py:146-200) for the state the hot path owns.  `workspace` is caller-allocated device memory of
 * at least taco_workspace_bytes(cfg); it is initialised asynchronously on `stream` (hipStream_t, may be NULL). */
int taco_create(const taco_cfg *cfg, int device, void *workspace, size_t workspace_bytes, void *stream, taco_env **out);
void taco_destroy(taco_env *env);

/* Replaces VecTask.step (vec_task_asymmetry.py:290-334): reset dispatch + domain randomisation, delay line,
 * 10 x (state refresh, rate PID, allocator, battery, rotor lag, aero, rigid-body integrate), observation/state pack,
 * task reward, done / time-out flags.  All pointers are DEVICE pointers.
 *   actions     [num_envs][4]            f32 in   (clamped to +-clip_actions inside)
 *   obs_buf     [num_envs][len_obs][26]  f32 in/out  persistent frame stack, newest frame last, unclamped
 *   states_buf  [num_envs][len_states][26] f32 in/out
 *   rew_buf     [num_envs]               f32 out
 *   reset_buf   [num_envs]               i64 in/out  non-zero on entry = reset this env first (ones before step 1)
 *   timeout_buf [num_envs]               u8  out     extras["time_outs"]
 * One kernel launch, asynchronous on `stream`; no host synchronisation. */
/* HIP graphs: the step index, ring head and history row advance with every call.  On the eager path they are kernel arguments (and every
 * launch leaves the next values in the workspace); a call on a CAPTURING stream enqueues the step kernel reading that device-resident
 * clock and advancing it itself -- the workspace holds one copy of the clock per 16 envs, read and rewritten by the one wavefront that
 * steps those envs, so no synchronisation between workgroups exists: ONE graph node per step, no slower than an eager step, and the
 * captured graph can be replayed any number of times, continuing exactly where the eager steps (or earlier replays) stopped.
 * GRAPH MODE: from the first capture on the clock lives on the device (the graph may be replayed at any time, and the host cannot see a
 * replay that is enqueued but has not run): every later launch of the handle takes the device path, and taco_get_step_count /
 * taco_set_step_count / taco_get_state / taco_set_state / taco_get_field / taco_states_ring_row / taco_step_ring / an eager taco_rollout_run
 * re-read it first -- they BLOCK (hipDeviceSynchronize + a 16-byte copy), on EVERY call, for as long as the handle is in graph mode.
 * taco_release_graphs ends it: the caller's word that no graph holding launches of this handle will be replayed any more (one last blocking
 * re-read; afterwards eager launches take the clock from their arguments again and nothing synchronises).  A later capture re-enters graph
 * mode.  A replay AFTER the release leaves the host's copy behind the device's; taco_check reports that (TACO_ERR_STATE) and KEEPS reporting it
 * whatever eager steps follow (launches that take their clock from the device count themselves in a workspace word eager launches never
 * write; the release latches the count) -- those steps reused random-stream counters and a stale delay-ring head.
 * taco_graph_mode: 1 while in graph mode, 0 otherwise (never blocks). */
int taco_step(taco_env *env, const float *actions, float *obs_buf, float *states_buf, float *rew_buf, int64_t *reset_buf,
              uint8_t *timeout_buf, void *stream);
int taco_graph_mode(const taco_env *env);
int taco_release_graphs(taco_env *env);

/* ---- the replay buffer directly behind step() ------------------------------------------------
 * The reference's rollout loop (ppo_asymmetry.py:96-131) copies every step's obs / states / reward / done into
 * PPOReplayBuffer slots with nine torch copies (buffer_asymmetry.py:70-91 `store`).  taco_step_rollout lets the step kernel
 * write those slots itself: the frame stacks are read from the PREVIOUS slot and written to the NEXT one, and the done
 * flags are also emitted as the f32 image done_buf keeps.  NULL obs_prev / states_prev mean "in place" (= taco_step);
 * done_f32 may be NULL.  The prev / next stacks of one call may be the same buffer but must not partially overlap. */
typedef struct taco_rollout_io {
    const float *actions;      /* [num_envs][4]                                                                   */
    const float *obs_prev;     /* [num_envs][len_obs][26]     slot t   (read only when len_obs > 1), or NULL       */
    float *obs_next;           /* [num_envs][len_obs][26]     slot t+1                                            */
    const float *states_prev;  /* [num_envs][len_states][26]  slot t, or NULL                                     */
    float *states_next;        /* [num_envs][len_states][26]  slot t+1; NULL = the bound frame ring (taco_bind_states_ring) */
    float *rew;                /* [num_envs]  rew_buf[t]                                                          */
    int64_t *reset_buf;        /* [num_envs]  in/out, as taco_step                                                */
    uint8_t *timeout_buf;      /* [num_envs]                                                                      */
    float *done_f32;           /* [num_envs]  done_buf[t] (buffer_asymmetry.py:67, 82), or NULL                   */
    float *obs_out;            /* [num_envs][len_obs][26]     clamp(obs_next, +-cfg.clip_obs): the copy VecTask.step returns      */
    float *states_out;         /* [num_envs][len_states][26]  clamp(states_next, +-cfg.clip_states) (vec_task_asymmetry.py:331-332);
                                  either may be NULL; 16-byte aligned; written by the same launch                                 */
    int32_t states_newest_only; /* != 0: states_next is ONE frame per env, [num_envs][26] -- a row of a FRAME RING (taco_rollout_bufs): the stack
                                  is not materialised, nothing is shifted; states_prev / states_out must be NULL.  The reference never clears
                                  its frame stacks (fpv_asymmetry.py:392, :413 shift by one per step, resets included), so the stack of slot t
                                  is exactly frames t .. t + len_states - 1 of such a ring                                           */
} taco_rollout_io;
int taco_step_rollout(taco_env *env, const taco_rollout_io *io, void *stream);

/* VecTask.reset_done (vec_task_asymmetry.py:363-375): reset_idx on the envs whose reset_buf is set, NOW, outside a step -- fresh state and
 * command, reset_buf / progress_buf cleared (fpv_asymmetry.py:475-517); like the reference's, it also re-draws the command of envs at
 * progress 500 (:500-503).  The step clock does not advance.  The next taco_step then steps those envs WITH thrust (it sees no reset flag),
 * exactly as the reference does after reset_done(); observations are not recomputed (the reference returns the buffers as they are). */
int taco_reset_done(taco_env *env, int64_t *reset_buf, void *stream);

/* PPOReplayBuffer.compute_returns_and_advantage (buffer_asymmetry.py:93-132) on DEVICE arrays laid out [horizon][num_envs]
 * (the reference's [H, N, 1]):  adv = GAE(gamma, lam), ret = adv + value, then (normalize != 0)
 * adv = (adv - mean(adv)) / (std(adv) + 1e-8) with the unbiased std.  ret and the un-normalised adv are bit-identical
 * to the reference's fp32 arithmetic; the two reductions are done in double (deterministic, fixed tree).
 * `workspace` = taco_gae_workspace_bytes() bytes of device memory (only used when normalize != 0). */
size_t taco_gae_workspace_bytes(void);
int taco_gae(const float *rew, const float *done, const float *value, const float *last_value, int horizon, int num_envs,
             double gamma, double lam, float *adv, float *ret, int normalize, void *workspace, void *stream);

/* PPO_ActorCritic.act (algorithms/nets_asymmetry.py:326-355) for the documented configuration (README.md:60-66): actor = MLP on
 * the observation stack (ReLU hidden layers, tanh head), critic = 1-layer unidirectional LSTM over the state stack (lstm_hidden > 0;
 * 0 = no encoder, the MLP takes the flattened stack) + MLP (ReLU, linear head).  One launch; f32 MFMA, f32 throughout.
 *   blob    taco_policy_blob_floats(cfg) floats, 16-byte aligned: every matrix zero-padded to multiples of 16 ([OUTp][INp] row-major, bias
 *           [OUTp]) in the order actor layers | log_std[16] | LSTM W_ih[4][Hp][Ip], W_hh[4][Hp][Hp], (b_ih + b_hh)[4][Hp] (gates i f g o) |
 *           critic layers                                                     (taco_amd/policy.py::pack_state_dict builds it from a state_dict)
 *   obs [n][obs_len][obs_dim], states [n][states_len][states_dim]             f32 in
 *   action, mu, sigma [n][act_dim]; logp [n]; value [n]                       f32 out.  sigma = log_std repeated, as the reference returns it;
 *           the distribution is MultivariateNormal(mean, scale_tril = diag(exp(log_std)^2)) (:334-335).
 *   (seed, call) key the Philox action noise together with the row index; deterministic != 0: action = mean.
 *   action_only != 0: only the actor runs (states / value may be NULL). */
typedef struct taco_policy_cfg {
    int32_t obs_len, obs_dim, states_len, states_dim, act_dim;
    int32_t n_actor_hidden, actor_hidden[4];
    int32_t lstm_hidden;
    int32_t n_critic_hidden, critic_hidden[4];
    int32_t flags;        /* TACO_P_* (0 = defaults) */
} taco_policy_cfg;
/* The batched critic (taco_critic_values, taco_critic_values_ring, taco_rollout_run) evaluates the LSTM cell's five activations on the
 * hardware's transcendental pipe by default (v_exp_f32 / v_rcp_f32, 1 ulp each): its SIMD time is MFMA time + VALU time and the op-for-op
 * cell is 142 VALU instructions, 37 % of the kernel (profiles/r03_q_*).  Values then agree with taco_policy_act's `value` and with the CPU
 * oracle to <= 2e-6 (tests/test_policy_gpu.py) instead of bit for bit -- inside the 1e-5 bar to the reference's own vectors.
 * TACO_P_EXACT_CELL keeps the oracle's operations in those kernels (bit-identical to taco_policy_act and oracle/taco_policy_oracle.c; +23 %
 * critic time).  taco_policy_act itself always uses the exact cell. */
#define TACO_P_EXACT_CELL 1
/* OPT-IN at this boundary (the Python host layer, taco_amd/policy.py, sets TACO_P_SPLIT_F16 by default), round 5: the batched critic's LSTM over a frame ring (taco_critic_values_ring, taco_rollout_run with a state stack) on the 16-bit
 * matrix pipe with SPLIT operands -- every f32 operand as two 16-bit halves, three MFMA chains per product chain, f32 accumulation
 * (taco_policy.hpp taco_critic_lstm_pair_split_kernel, and the MLP's two hidden layers behind it: taco_critic_mlp_split_kernel; critic at 557 056 rows 3.22 -> 1.06 ms).  TACO_P_SPLIT_F16: f16 halves (11 + 11
 * significant bits): values within ~1e-6 of the exact f32 critic's on O(1) frames (1.25e-6 measured on values up to 1.5, the fast cell alone 7.2e-7; asserted
 * <= 2e-6 x max(1, |value|), the fast cell's own bar).  Frame words outside f16's range are NOT silent (ABI 8): +-inf becomes NaN and NaN stays NaN (the
 * value of a diverged env is NaN, as in the f32 kernels); a FINITE word beyond +-65 504 saturates there -- the gates it feeds are saturated either
 * way -- and is counted: taco_critic_clamped_words reads the count of the last call, and inside taco_rollout_run the env's sticky status word is
 * set as well, so taco_check returns TACO_ERR_STATE.  TACO_P_SPLIT_BF16: bf16 halves (8 + 8
 * bits): ~1.7e-5, outside that bar -- kept for the A/B record only.  Other layouts (materialised stacks, one slot, states_dim = 32)
 * ignore the bits.  Excludes TACO_P_EXACT_CELL. */
#define TACO_P_SPLIT_F16 2
#define TACO_P_SPLIT_BF16 4
size_t taco_policy_blob_floats(const taco_policy_cfg *cfg);   /* 0 (and taco_last_error) for an unsupported configuration */
int taco_policy_act(const taco_policy_cfg *cfg, const float *blob, int n, const float *obs, const float *states, uint64_t seed, uint32_t call,
                    int deterministic, int action_only, float *action, float *logp, float *value, float *mu, float *sigma, void *stream);

/* Profiling aid: taco_policy_act with `stamps` = DEVICE array of 16 uint64; workgroup x = 0 of this launch records the shader clock at
 * [0..3] actor: entry, inputs staged, MLP done, end; [8..12] critic: entry, inputs staged, LSTM done, MLP done, end. */
int taco_policy_act_stamped(const taco_policy_cfg *cfg, const float *blob, int n, const float *obs, const float *states, uint64_t seed, uint32_t call,
                            int deterministic, int action_only, float *action, float *logp, float *value, float *mu, float *sigma, uint64_t *stamps,
                            void *stream);

/* The critic alone (nets_asymmetry.py:348-352, 128-136) over `rows` state stacks [rows][states_len][states_dim] -> value[rows], ONE launch.
 * The `value` output of taco_policy_act on the same stacks: bit-identical with TACO_P_EXACT_CELL, within 2e-6 with the default cell
 * (above; other geometries: always bit-identical).  The documented architecture (LSTM hidden 128 over <= 8 frames
 * of <= 32 features, two MLP layers of 128) runs in the batched form, two launches: a persistent LSTM kernel (one workgroup per CU walking
 * 32-row blocks, W_ih / W_hh register-resident, the SIMD's two wavefronts one phase apart so one's MFMAs run beside the other's gate
 * activations) writing h_T to `workspace`, then the MLP with resident weights; other geometries run the critic role of the policy kernel
 * (workspace unused).  This is what takes the critic off the rollout's serial chain (taco_rollout_run). */
size_t taco_critic_workspace_bytes(const taco_policy_cfg *cfg, int64_t rows);   /* the LSTM's h_T between the two kernels: rows x 128 floats, + a 256-byte tail (ABI 8) */
/* ABI 8.  BLOCKS on `stream`: *count = how many times the LAST batched-critic call on (`workspace`, `rows`) saturated a finite frame word beyond +-65 504
 * (TACO_P_SPLIT_F16 only; 0 for every other form).  Clamp EVENTS: a frame of a ring is staged once per block of two slots whose stacks hold it. */
int taco_critic_clamped_words(const taco_policy_cfg *cfg, int64_t rows, const void *workspace, uint32_t *count, void *stream);
int taco_critic_values(const taco_policy_cfg *cfg, const float *blob, int64_t rows, const float *states, float *value, void *workspace,
                       uint64_t *stamps, void *stream);   /* stamps: NULL, or a DEVICE array of 259 uint64 (profiling; taco_policy.hpp) */

/* The critic over a frame ring (the layout of taco_rollout_bufs.states_store): frames [slots + states_len - 1][num_envs][states_dim], value
 * [slots][num_envs]; row (slot, env) is the stack frames[slot .. slot + states_len - 1][env].  Same kernels and the same bits as
 * taco_critic_values on the materialised stacks. */
int taco_critic_values_ring(const taco_policy_cfg *cfg, const float *blob, int64_t slots, int num_envs, const float *frames, float *value,
                            void *workspace, void *stream);

/* One PPO rollout (ppo_asymmetry.py:308-342) enqueued from C on `stream`: for t < horizon { act on replay slot t -> action / log-prob /
 * value / mean / log-std of step t; the action clipped to [act_lo, act_hi] (:310) drives taco_step_rollout, which writes slot t + 1,
 * rew_buf[t], done_buf[t], timeout_buf[t] }, then the value of the final state (:341) and the time-out bootstrap
 * rew[t] += gamma * value[t] where an env was truncated (:314-324; the value of the pre-step state IS value_buf[t]).
 * Per step only the ACTOR runs (nothing before GAE consumes `value`): the critic is evaluated afterwards for all horizon + 1 slots in one
 * batched launch (taco_critic_values' kernel; same bits as a per-step act() with TACO_P_EXACT_CELL, see there).  2 * horizon + 3 launches (4 in the persistent form, see
 * taco_set_rollout_fusion below), no host work between the steps;
 * the policy's noise counter runs call0 .. call0 + horizon - 1 (the critic draws nothing).  Capturable: on a capturing stream the env's
 * clock and the actor's noise counter are read from the device-resident step word, so every replay of the graph is the NEXT rollout
 * (fresh noise, continuing episodes), bit-identical to the same number of eager calls.
 * All arrays are DEVICE memory in the replay-buffer layout (taco_amd/rollout.py::RolloutBuffer.run). */
typedef struct taco_rollout_bufs {
    float *obs_store;     /* [horizon + 1][num_envs][len_obs][26]    slot 0 = the stacks to start from */
    float *states_store;  /* FRAME RING [horizon + len_states][num_envs][26]: the state stack of slot t is rows t .. t + len_states - 1 (row
                             t + len_states - 1 its newest frame); rows 0 .. len_states - 1 = the stack to start from; step t writes row
                             t + len_states.  One frame per step is stored instead of a shifted stack per slot (SURVEY 8d "history kept as
                             a ring"): 104 B instead of len_states x 104 B written per env-step, and the critic reads every frame once per
                             block instead of len_states times. */
    float *act_buf;       /* [horizon][num_envs][4]   sampled (un-clipped) action, what store() keeps (:326) */
    float *act_env;       /* [num_envs][4]            scratch: the clipped action of the current step */
    float *rew_buf;       /* [horizon][num_envs] */
    float *done_buf;      /* [horizon][num_envs]      f32 */
    float *value_buf;     /* [horizon][num_envs] */
    float *logp_buf;      /* [horizon][num_envs] */
    float *mu_buf;        /* [horizon][num_envs][4] */
    float *sigma_buf;     /* [horizon][num_envs][4] */
    uint8_t *timeout_buf; /* [horizon][num_envs] */
    float *last_value;    /* [num_envs] */
    float *critic_ws;     /* taco_critic_workspace_bytes(cfg, (horizon + 1) * num_envs) bytes, 16-byte aligned */
} taco_rollout_bufs;
int taco_rollout_run(taco_env *env, const taco_policy_cfg *cfg, const float *blob, const taco_rollout_bufs *bufs, int horizon, uint64_t seed,
                     uint32_t call0, double gamma, double act_lo, double act_hi, int64_t *reset_buf, void *stream);
/* For the documented rollout (len_obs = 1, actor MLP 26-128-128-128-4, at most 8 192 envs on the handle, no gather block bound, no kernel form
 * pinned) taco_rollout_run's per-step chain is ONE persistent launch: a workgroup owns 16 envs for all `horizon` steps, four of its eight
 * wavefronts run the actor on MFMA (weights register-resident) WHILE the step wavefront integrates the ten substeps -- the action of a step
 * enters the dynamics only delay_time later -- so a rollout is 4 launches (+ 1 when captured) instead of 2 * horizon + 3; same bits.
 * taco_set_rollout_fusion(env, 0) keeps the launch-per-step path (A/B measurements, tests); 1 (default) restores the choice above; 2 takes
 * the persistent kernel whatever the env count (A/B: a workgroup holds a whole CU for 16 envs, so above 8 192 envs its workgroups queue --
 * 32 768 envs are eight rounds of the chip -- and the launch-per-step path with 64 envs per workgroup wins: bench.py records both).
 * taco_bind_rollout_stamps: profiling aid, DEVICE array of 136 + ceil(num_envs / 16) uint64 filled by the persistent kernel: workgroup 0: [w] = the SIMD
 * wavefront w runs on, [8 + 2 t] / [9 + 2 t] = shader clock of the step wavefront at the start / end of step t (t < 64); [136 + b] = clocks the step
 * wavefront of workgroup b spent in its loop.  NULL unbinds. */
int taco_set_rollout_fusion(taco_env *env, int on);
int taco_bind_rollout_stamps(taco_env *env, uint64_t *stamps);

/* Multi-GPU (no counterpart in the reference, which is single-process): bind a DEVICE block
 * [num_envs][taco_gather_row_floats(len_obs)] f32, 128-byte aligned, that every following taco_step also fills with
 * (obs stack [len_obs*26] | reward | done | time-out | zero padding) per env, so the host layer can publish a rank's
 * results with ONE RCCL all-gather.  Rows are padded to a multiple of 32 floats (one or more whole 128-byte lines per
 * env).  NULL unbinds. */
int taco_gather_row_floats(int len_obs);
int taco_bind_gather_block(taco_env *env, float *block);

/* A FRAME RING behind VecTask.step()'s state stacks (no counterpart in the reference, whose states_buf is shifted by one frame per step,
 * fpv_asymmetry.py:413: len_states - 1 frames read and len_states written per env-step).  Bind a DEVICE ring [rows][num_envs][26] f32 (zeroed,
 * 16-byte aligned, rows >= 2 * (len_states - 1) + 1; period = rows - (len_states - 1)); a taco_step_rollout call with states_next == NULL then
 * writes ONE states frame per env: the step with ring phase ph (0, 1, ... period - 1, 0, ...; part of the step clock, so captured steps advance it
 * on the device) writes row ph + len_states - 1 and, while ph >= period - (len_states - 1), a twin at row ph - period + len_states - 1, so that
 * rows [ph, ph + len_states) ALWAYS hold the stack that step produced, oldest frame first: the reference's [num_envs][len_states][26] tensor is the
 * strided view ring[ph : ph + len_states] with the first two axes swapped (taco_amd/vec_env.py hands exactly that to the PPO loop).  104 B written
 * per env-step (+ (len_states - 1) / period of that for the twins) instead of (2 len_states - 1) x 104 B moved.  taco_states_ring_row = the
 * phase of the LAST step (the first row of its window; blocks in graph mode, -1 on error).  Binding resets the phase to 0 (blocks);
 * NULL unbinds.  Calls with states_next != NULL are unaffected.  Finite clip_states: pass io->states_out = a SECOND ring of the same geometry;
 * it receives clamp(frame, +-clip_states) at the same rows -- the clamped copy VecTask.step returns (vec_task_asymmetry.py:332) is the same
 * view of that ring, since clamping a stack is clamping its frames. */
int taco_bind_states_ring(taco_env *env, float *ring, int rows);
int taco_states_ring_row(taco_env *env);
/* taco_step_rollout on the bound ring (io->states_next == NULL) that also reports the ring phase the launch uses: *phase = the first row of
 * the window this step fills -- what a host layer needs to hand out the right strided view without a second call.  Refuses a capturing
 * stream (TACO_ERR_STATE, nothing is enqueued: a replay's window cannot be told in advance; capture taco_step_rollout instead); in graph mode
 * (see taco_step) it first re-reads the device clock: it blocks on every call until taco_release_graphs. */
int taco_step_ring(taco_env *env, const taco_rollout_io *io, void *stream, int32_t *phase);

/* env.difficulty = x (ppo_asymmetry.py:173-175, :376); takes effect at the next taco_step. */
int taco_set_difficulty(taco_env *env, double difficulty);

/* Number of steps taken so far (keys the counter-based random streams); settable for checkpoint/restore.  After graph replays both
 * calls first re-read the device-resident clock (hipDeviceSynchronize); while a stream is still capturing launches of this handle they fail
 * with TACO_ERR_STATE (-1 from taco_get_step_count) instead of invalidating the capture, and so do taco_get_state / taco_set_state /
 * taco_get_field (which otherwise read the clock in the order of THEIR stream and wait for that stream only). */
int64_t taco_get_step_count(taco_env *env);
int taco_set_step_count(taco_env *env, int64_t n);   /* also writes the device-resident clock (blocks): a capture may be the next launch */
/* The HOST's copy of the step count, without synchronising: exact until a captured graph has been replayed, stale afterwards (until one of
 * the two calls above, taco_get_state / taco_set_state / taco_get_field or an eager taco_rollout_run has re-read the device clock).  Legal
 * while a stream is capturing -- what a host layer needs to derive the rollout's noise-counter offset at capture time. */
int64_t taco_peek_step_count(const taco_env *env);

/* Health check (BLOCKS on `stream`): TACO_ERR_STATE if any step kernel since taco_create recorded a sticky error in the workspace's status
 * word -- a bounded wait of the four-role form's battery mailbox gave up (the voltage of the affected envs was set to NaN instead of a stale
 * value; never observed), or the split-f16 critic of a taco_rollout_run saturated finite frame words beyond +-65 504 (ABI 8) -- or if a captured
 * graph was replayed after taco_release_graphs (see taco_step).  The steps themselves return TACO_OK because they never synchronise. */
int taco_check(taco_env *env, void *stream);

/* One row of the state blob (TACO_S_* except actions_old) for every env -> DEVICE array out[num_envs]; e.g. TACO_S_PROGRESS = progress_buf. */
int taco_get_field(taco_env *env, int field, uint32_t *out, void *stream);

/* Copy the per-env state to / from a DEVICE blob of TACO_BLOB_ROWS * num_envs words (layout above). */
int taco_get_state(taco_env *env, uint32_t *blob, void *stream);
int taco_set_state(taco_env *env, const uint32_t *blob, void *stream);

/* Diagnostics for bench.py: name of the step kernel (as rocprofv3 reports it) and its launch geometry. */
const char *taco_step_kernel_name(void);
int taco_launch_geometry(const taco_env *env, int *grid, int *block);
/* The step kernel exists in six instantiations of one template -- all the same function bit for bit -- and taco_create picks one from
 * the env count.  taco_set_kernel_form overrides the choice for this handle (tuning, A/B measurements, and the
 * tests that run every instantiation at small sizes); TACO_FORM_AUTO restores the default.  No environment variable is read anywhere. */
enum {
    TACO_FORM_AUTO = 0,
    TACO_FORM_QUAD_ROLES = 1,      /* 4 lanes per env, 256-thread workgroups: one step wavefront + three post-phase role wavefronts */
    TACO_FORM_LANE_ROLES = 2,      /* 1 lane per env, the same four roles (frame stacks, 16 385 ... 65 536 envs)                    */
    TACO_FORM_QUAD = 3,            /* 4 lanes per env, one wavefront per workgroup                                                  */
    TACO_FORM_LANE = 4,            /* 1 lane per env, one wavefront per workgroup, registers unconstrained                          */
    TACO_FORM_LANE_THROUGHPUT = 5, /* 1 lane per env, one wavefront per workgroup, <= 128 VGPRs: four wavefronts per SIMD            */
    TACO_FORM_QUAD_SERVED = 6      /* 4 lanes per env, 128-thread workgroups: a step wavefront + one wavefront that serves it (rotor noise, euler) */
};
int taco_set_kernel_form(taco_env *env, int form);
int taco_get_kernel_form(const taco_env *env);  /* the TACO_FORM_* in use (never AUTO), or TACO_ERR_INVALID_ARG */
/* Introspection (profiling aid): bind a DEVICE array of 16 uint64; every following taco_step makes lane 0 of workgroup 0 record
 * the shader clock (s_memtime) at: 0 kernel entry, 1 up-front loads landed, [9] reset_idx of the workgroup's resetting envs done (no env
 * resets: right behind 1), 2 pre-phase done (delay line / slot table),
 * 3 ten substeps done, 4 state stores + frame stacks done, 5 kernel end (in the four-role instantiations these are
 * wavefront 0's: 4 and 5 coincide after its state stores; [6..8] = the ends of role wavefronts 1..3: reward, obs stack, states stack).  NULL unbinds.  Costs a scalar branch per phase. */
int taco_bind_phase_stamps(taco_env *env, uint64_t *stamps);

/* Introspection: workgroups of the step kernel the runtime can keep resident per CU, and the kernel's LDS bytes per
 * workgroup (what bounds it). */
int taco_occupancy(const taco_env *env, int *resident_blocks_per_cu, int *lds_bytes_per_block);

#ifdef __cplusplus
}
#endif
#endif /* TACO_ENV_H */
