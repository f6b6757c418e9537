// Torch-free consumer of the C ABI: plain HIP runtime + include/taco_env.h.
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_api_demo.cpp -L taco_amd -ltaco_env -Wl,-rpath,$PWD/taco_amd -o examples/c_api_demo
// Steps 4 096 pos-task envs with a constant hover-ish action, prints the step rate and a few sanity statistics, and checks
// the error path.  Exit code 0 = all checks passed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "taco_env.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main() {
    taco_cfg cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.num_envs = 4096; cfg.num_envs_global = 4096; cfg.task_mode = TACO_TASK_POS; cfg.len_obs = 1; cfg.len_states = 1;
    cfg.control_freq_inv = 10; cfg.substeps = 2; cfg.max_episode_length = 1000; cfg.delay_time = 20;
    cfg.flags = TACO_F_RANDOM_COPTER_POS | TACO_F_RANDOM_COPTER_QUAT | TACO_F_RANDOM_COPTER_VEL | TACO_F_RANDOM_TARGET_POS | TACO_F_RANDOM_TARGET_YAW |
                TACO_F_BATTERY_CONSUMPTION | TACO_F_RANDOM_VOLTAGE | TACO_F_ROTOR_DELAY | TACO_F_ROTOR_RESPONSE | TACO_F_RANDOM_ROTOR_SPEED | TACO_F_RANDOM_COMMAND |
                TACO_F_WORLD_RATE_ROUNDTRIP;   // (the reference's own data flow of the angular rate: the default arithmetic)
    cfg.seed = 1; cfg.dt = 0.001; cfg.rotor_response_time = 0.017; cfg.difficulty = 1.0;
    cfg.clip_actions = cfg.clip_obs = cfg.clip_states = INFINITY;
    cfg.mass = 0.4600008; cfg.inertia[0] = 5.008029448e-4; cfg.inertia[1] = 7.008019272e-4; cfg.inertia[2] = 8.00804552e-4;
    cfg.arm_x = 0.047; cfg.arm_y = 0.059; cfg.gravity_z = -9.81;
    const int n = cfg.num_envs;

    if (taco_abi_version() != TACO_ABI_VERSION) { std::printf("ABI mismatch\n"); return 1; }
    taco_env *env = nullptr;
    taco_cfg bad = cfg; bad.control_freq_inv = 3;
    if (taco_create(&bad, 0, nullptr, 0, nullptr, &env) != TACO_ERR_INVALID_ARG || !std::strstr(taco_last_error(), "control_freq_inv")) { std::printf("error path broken\n"); return 1; }

    void *ws = nullptr; float *act, *obs, *st, *rew; int64_t *reset; uint8_t *tmo;
    const size_t wsb = taco_workspace_bytes(&cfg);
    HIP_OK(hipMalloc(&ws, wsb)); HIP_OK(hipMalloc(&act, n * 4 * sizeof(float))); HIP_OK(hipMalloc(&obs, n * 26 * sizeof(float)));
    HIP_OK(hipMalloc(&st, n * 26 * sizeof(float))); HIP_OK(hipMalloc(&rew, n * sizeof(float))); HIP_OK(hipMalloc(&reset, n * sizeof(int64_t)));
    HIP_OK(hipMalloc(&tmo, n));
    std::vector<float> h_act(n * 4); for (int i = 0; i < n; ++i) { h_act[4 * i] = -0.45f; h_act[4 * i + 1] = h_act[4 * i + 2] = h_act[4 * i + 3] = 0.0f; }
    std::vector<int64_t> ones(n, 1);
    HIP_OK(hipMemcpy(act, h_act.data(), h_act.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(reset, ones.data(), n * sizeof(int64_t), hipMemcpyHostToDevice));   // vec_task_asymmetry.py:246-247
    HIP_OK(hipMemset(obs, 0, n * 26 * sizeof(float))); HIP_OK(hipMemset(st, 0, n * 26 * sizeof(float)));
    hipStream_t s; HIP_OK(hipStreamCreate(&s));
    if (taco_create(&cfg, 0, ws, wsb, s, &env) != TACO_OK) { std::printf("taco_create: %s\n", taco_last_error()); return 1; }

    const int steps = 2000;
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (int t = 0; t < 100; ++t) if (taco_step(env, act, obs, st, rew, reset, tmo, s) != TACO_OK) { std::printf("taco_step: %s\n", taco_last_error()); return 1; }
    HIP_OK(hipEventRecord(e0, s));
    for (int t = 0; t < steps; ++t) taco_step(env, act, obs, st, rew, reset, tmo, s);
    HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h_obs(n * 26), h_rew(n); std::vector<int64_t> h_done(n);
    HIP_OK(hipMemcpy(h_obs.data(), obs, h_obs.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_rew.data(), rew, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_done.data(), reset, n * sizeof(int64_t), hipMemcpyDeviceToHost));
    double rsum = 0; long dones = 0; int bad_rows = 0;
    for (int i = 0; i < n; ++i) {
        rsum += h_rew[i]; dones += h_done[i];
        const float *o = &h_obs[26 * i];
        // rows 3..11 of the frame are a rotation matrix: first row must have unit norm
        const float nr = o[3] * o[3] + o[4] * o[4] + o[5] * o[5];
        if (!(std::fabs(nr - 1.0f) < 1e-4f) || !(o[24] == 0.0f)) ++bad_rows;
    }
    std::printf("C ABI demo: %d envs x %d steps in %.2f ms = %.1f M env-steps/s (%.2f us/step); mean reward %.5f, done now %ld, bad rows %d, step count %lld\n",
                n, steps, ms, (double)n * steps / ms / 1e3, ms * 1e3 / steps, rsum / n, dones, bad_rows, (long long)taco_get_step_count(env));
    taco_destroy(env);
    return (bad_rows == 0 && taco_get_step_count(nullptr) == -1) ? 0 : 1;
}
