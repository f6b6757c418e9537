/*
 * taco_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's fpv_asymmetry.step() hot path (yinzikang/taco,
 * IsaacGymEnvs/isaacgymenvs/tasks/fpv_asymmetry.py + tasks/control/{*}.py + the quaternion helpers
 * of python/isaacgym/torch_utils.py and isaacgymenvs/utils/torch_jit_utils.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (taco_amd/, libtaco_env.so) never includes, links or calls anything in oracle/.
 *
 * PARITY STATUS
 *   rows B..H'', J, O, P, R of SURVEY.md section 8(a): pinned by tests/golden/{*}.npz, which were produced
 *     by running the reference's own importable torch modules on CPU (tests/golden/make_golden.py).
 *   row I (rigid-body integrate, PhysX inside the absent libPhysXGpu_64.so of Isaac Gym 1.0.preview4):
 *     "parity unpinned" -- the arithmetic is not in /root/reference; this file restates the published
 *     PhysX scheme (semi-implicit Euler + closed-form quaternion update) and is validated against
 *     analytic cases only.
 *   random draws: the reference uses torch's generators; here a counter-based Philox4x32-10 keyed by
 *     (seed, global env id, step index, stream) -- distributional parity only (SURVEY.md section 7 "RNG").
 */
#ifndef TACO_ORACLE_H
#define TACO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* task modes (tasks/__init__.py:30-39 keys Fpv_pos / Fpv_rotate / Fpv_flip / Fpv_mix) */
enum { ORC_TASK_POS = 0, ORC_TASK_ROTATE = 1, ORC_TASK_FLIP = 2, ORC_TASK_MIX = 3 };

/* cfg flags: one bit per boolean the hot path reads (fpv_asymmetry.py:63-112) */
enum {
    ORC_F_RANDOM_COPTER_POS = 1u << 0,
    ORC_F_RANDOM_COPTER_QUAT = 1u << 1,
    ORC_F_RANDOM_COPTER_VEL = 1u << 2,
    ORC_F_RANDOM_TARGET_POS = 1u << 3,
    ORC_F_RANDOM_TARGET_YAW = 1u << 4,
    ORC_F_BATTERY_CONSUMPTION = 1u << 5,
    ORC_F_RANDOM_VOLTAGE = 1u << 6,
    ORC_F_ROTOR_NOISE = 1u << 7,
    ORC_F_ROTOR_DELAY = 1u << 8,
    ORC_F_ROTOR_RESPONSE = 1u << 9,
    ORC_F_RANDOM_ROTORDYNAMIC_COE = 1u << 10,
    ORC_F_RANDOM_ROTOR_DELAY = 1u << 11,
    ORC_F_RANDOM_ROTOR_RESPONSE = 1u << 12,
    ORC_F_RANDOM_ROTOR_SPEED = 1u << 13,
    ORC_F_RANDOM_AERODYNAMIC_COE = 1u << 14,
    ORC_F_RANDOM_DELAY_TIME = 1u << 15,  /* cfg key "ramdom_delay_time" (sic) */
    ORC_F_RANDOM_DEPLOY_TIME = 1u << 16, /* cfg key "ramdom_deploy_time" (sic) */
    ORC_F_RANDOM_COMMAND = 1u << 17,
    ORC_F_OBSERVATION_NOISE = 1u << 18,
    ORC_F_WORLD_RATE_ROUNDTRIP = 1u << 19 /* same as orc_set_world_rate_roundtrip(e, 1) */
};

/* Same field order and types as struct taco_cfg in include/taco_env.h (declared independently). */
typedef struct orc_cfg {
    int32_t num_envs;        /* envs held by this instance */
    int32_t env_offset;      /* global id of local env 0 (multi-GPU sharding) */
    int32_t num_envs_global; /* N of the whole job: drives the FpvMix thirds (fpv_asymmetry.py:924-926) */
    int32_t task_mode;
    int32_t len_obs, len_states;
    int32_t control_freq_inv; /* must be 10: the delay line hard-codes 10 one-ms slots per action */
    int32_t substeps;
    int32_t max_episode_length;
    int32_t delay_time;
    uint32_t flags;
    uint32_t _pad;
    uint64_t seed;
    double dt;
    double rotor_response_time;
    double difficulty;
    double clip_actions, clip_obs, clip_states; /* +inf = no clipping (vec_task_asymmetry.py:98-100) */
    double mass;                                /* composite mass [kg] */
    double inertia[3];                          /* composite principal inertia [kg m^2] */
    double arm_x, arm_y;                        /* rotor offsets: (+-arm_x, +-arm_y) (fpv_without_duct.xml) */
    double gravity_z;                           /* -9.81 (fpv_asymmetry.py:214-217) */
} orc_cfg;

/* ---- state blob layout: [ORC_NUM_FIELDS + 400][num_envs] 32-bit words, field-major.  Same layout spec as
 * include/taco_env.h "State blob"; words 65,66 are int32, the rest float; the last 400 rows are the delay
 * line in LOGICAL order, row = slot*4 + channel. */
enum {
    ORC_PX = 0, ORC_QX = 3, ORC_VX = 7, ORC_WX = 10, ORC_TPX = 13, ORC_TQX = 16, ORC_RPY_OLD = 20, ORC_RPY_CONT = 23,
    ORC_PID_PREV = 26, ORC_PID_INT = 29, ORC_BAT_E = 32, ORC_BAT_U1 = 33, ORC_BAT_T = 34, ORC_BAT_V = 35,
    ORC_OMEGA = 36, ORC_ACT = 40, ORC_ACT_OLD = 44, ORC_CMD = 48, ORC_FLIP_RADIAN = 50, ORC_TAU = 51, ORC_OPARA = 55,
    ORC_CF = 60, ORC_CT = 61, ORC_DX = 62, ORC_DY = 63, ORC_KT = 64, ORC_PROGRESS = 65, ORC_DELAY_LEN = 66,
    ORC_NUM_FIELDS = 67, ORC_RING_SLOTS = 100, ORC_BLOB_ROWS = 67 + 400
};

typedef struct orc_env orc_env;

/* ---- environment */
int orc_create(const orc_cfg *cfg, orc_env **out);
void orc_destroy(orc_env *e);
void orc_set_difficulty(orc_env *e, double difficulty);
void orc_set_threads(orc_env *e, int nthreads);
int orc_num_envs(const orc_env *e);
void orc_set_world_rate_roundtrip(orc_env *e, int on); /* 1: body rates go through the root state every substep, as in the reference */
/* One VecTask.step() (vec_task_asymmetry.py:290-334).  obs/states are the PERSISTENT frame stacks
 * [num_envs][len][26] (shifted in place, newest frame last, unclamped -- the caller clamps like the reference's
 * torch.clamp at :331-332); reset_buf is int64 in/out (ones before the first step, :246-247). */
int orc_reset_done(orc_env *e, int64_t *reset_buf);   /* VecTask.reset_done, VT:363-375 */
int orc_step(orc_env *e, const float *actions, float *obs_buf, float *states_buf, float *rew_buf, int64_t *reset_buf,
             uint8_t *timeout_buf);
void orc_get_state(const orc_env *e, uint32_t *blob);
void orc_set_state(orc_env *e, const uint32_t *blob);
int64_t orc_step_count(const orc_env *e);
void orc_set_step_count(orc_env *e, int64_t n);

/* ---- sub-model entry points for the golden-vector tests (n rows each; arrays are row-major [n][k]) */
void orc_quat_mul(int n, const float *a, const float *b, float *out);
void orc_quat_rotate_inv(int n, const float *q, const float *v, float *out); /* quat_rotate(quat_conjugate(q), v) */
void orc_euler_xyz_v1(int n, const float *q, float *rpy);
void orc_quat_from_euler_xyz(int n, const float *rpy, float *q);
void orc_quat_diff_rad(int n, const float *a, const float *b, float *out);
void orc_quat_to_matrix(int n, const float *q, float *m9);
void orc_pid_step(int n, float dt, const float *des, const float *cur, float *prev, float *integ, float *out);
void orc_allocator(int n, float *u /* in/out: u3 clipped in place */, float *thr);
void orc_real2sim(int n, const float *f, const float *t, float *fs, float *ts);
void orc_battery_step(int n, int enabled, float dt, const float *Pm, float *E, float *u1, float *t, float *V);
void orc_power(int n, const float *omega, float *Pm);
void orc_rotor_step(int n, const float *V, const float *thr, const float *tau, const float *para, float *omega);
void orc_aero(int n, const float *cf_ct, const float *d, const float *kt, const float *vb, const float *om, float *rf,
              float *rt, float *bf);
void orc_reward_pos(int n, const float *rel_pos_b, const float *pos, const float *q, const float *qt, const int64_t *prog,
                    float max_len, float *rew, int64_t *reset);
void orc_reward_rotate(int n, const float *rel_pos, const float *rel_v, const float *pos, const float *q, const float *cmd,
                       const int64_t *prog, float max_len, float *rew, int64_t *reset);
void orc_reward_flip(int n, const float *rel_pos_b, const float *relq, const float *pos, const float *cmd,
                     const int64_t *prog, float max_len, float *rew, int64_t *reset);
/* noise-free 26-D frame from raw state (rows C + O); task: 0 pos, 1 rotate, 2 flip */
void orc_obs_frame(int n, int task, const float *p, const float *q, const float *v, const float *w, const float *pt,
                   const float *qt, const float *V, const float *act, const float *cmd, const float *flip_radian,
                   const float *roll_cont, float *frame26, float *flip_cmd_out);
void orc_rand_float(int n, double lower, double upper, const float *u, float *out);
/* replay buffer: GAE + advantage normalisation, arrays [H][N] */
void orc_gae(int H, int N, double gamma, double lam, const float *rew, const float *done, const float *value, const float *last_value,
             float *adv_raw, float *ret);
void orc_normalize_advantage(size_t n, float *adv);

/* policy forward (second half; oracle/taco_policy_oracle.c) */
#define ORC_POLICY_MAXW 256
typedef struct orc_policy_cfg {
    int32_t obs_len, obs_dim, states_len, states_dim, act_dim;
    int32_t n_actor_hidden, actor_hidden[4];
    int32_t lstm_hidden;            /* 0: no critic encoder (the critic MLP takes the flattened state stack) */
    int32_t n_critic_hidden, critic_hidden[4];
} orc_policy_cfg;
size_t orc_policy_blob_floats(const orc_policy_cfg *c);
int orc_policy_act(const orc_policy_cfg *c, const float *blob, int n, const float *obs, const float *states, const float *eps, int deterministic,
                   float *action, float *logp, float *value, float *mu, float *sigma);
/* one PPO rollout (ppo_asymmetry.py:308-342 + buffer_asymmetry.py:49-68 store) on the oracle env: see taco_policy_oracle.c */
int orc_rollout(orc_env *e, const orc_policy_cfg *c, const float *blob, int H, uint64_t pseed, uint32_t call0, double gamma, double act_lo,
                double act_hi, int64_t *reset_buf, float *obs_store, float *states_store, float *act_buf, float *rew_buf, float *done_buf,
                float *value_buf, float *logp_buf, float *mu_buf, float *sigma_buf, uint8_t *timeout_buf, float *last_value);
void orc_policy_noise(uint64_t seed, uint32_t call, int n, int act_dim, float *eps);
float orc_expf(float x);
float orc_tanhf(float x);
float orc_sigmoidf(float x);
/* row I alone: advance root state [n][13] by one simulate(dt) under a body-frame wrench [n][6] (F then tau) */
void orc_integrate(const orc_cfg *cfg, int n, float *root13, const float *wrench6);

/* ---- own math + RNG (bit-exact twins live in the HIP kernel; tested against libm here) */
float orc_sinf(float x);
float orc_cosf(float x);
float orc_atan2f(float y, float x);
float orc_asinf(float x);
float orc_logf(float x);
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
void orc_obs_normals(uint64_t seed, uint32_t gid, uint32_t step, float nrm[12]); /* STREAM_OBS Box-Muller normals of one env-step */
float orc_uniform(uint32_t bits);
/* bulk draws for tests/test_rng_distributions.py (see taco_oracle.c) */
void orc_uniform_block(uint64_t seed, uint32_t gid0, int n_env, uint32_t step0, int n_step, uint32_t stream, int n_idx, float *out);
void orc_rounded_normal_vec(int n, const float *u, int lim, int32_t *out);
void orc_obs_normals_block(uint64_t seed, uint32_t gid0, int n_env, uint32_t step0, int n_step, float *out);

#ifdef __cplusplus
}
#endif
#endif
