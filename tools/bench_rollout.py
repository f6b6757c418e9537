#!/usr/bin/env python3
"""Row N1 measurement (not the headline metric): one rollout of H steps at N envs, then GAE.
  a) reference-shaped loop: env.step_raw + the five frame/reward/done copies of PPOReplayBuffer.store + obs/states carry
  b) RolloutBuffer.collect (the step kernel writes the slots)
  c) GAE: the reference's torch loop (buffer_asymmetry.py:93-132, on the GPU) vs taco_gae
Prints one JSON line.   python3 tools/bench_rollout.py [--envs 4096] [--horizon 32] [--len-states 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from taco_amd import config  # noqa: E402
from taco_amd.rollout import RolloutBuffer  # noqa: E402
from taco_amd.vec_env import FpvBase  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--horizon", type=int, default=32)
ap.add_argument("--len-states", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
N, H = a.envs, a.horizon
cfg = config.baseline_config(1, num_envs=N)
cfg["env"]["lenStates"] = a.len_states
env = FpvBase(cfg, copy_outputs=False)
buf = RolloutBuffer(N, 26, 1, 26, a.len_states, 4, H, 4, 0.99, 0.95, "cuda:0")
g = torch.Generator().manual_seed(0)
acts = (0.3 * torch.randn((H, N, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
obs, states = torch.zeros_like(env.obs_buf), torch.zeros_like(env.states_buf)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps


ref_states = torch.zeros(H, N, buf.states_len, 26, device="cuda")


def loop_copy():
    buf.step = 0
    for t in range(H):
        env.step_raw(acts[t])
        buf.obs_buf[t].copy_(obs); ref_states[t].copy_(states)   # (the reference's [H,N,T,26] store: the product keeps a frame ring instead)
        buf.rew_buf[t].copy_(env.rew_buf.view(-1, 1)); buf.done_buf[t].copy_(env.reset_buf.view(-1, 1))
        obs.copy_(env.obs_buf); states.copy_(env.states_buf)


def loop_fused():
    buf.step = 0
    for t in range(H):
        buf.collect(env, acts[t])
    buf._obs_store[0].copy_(buf._obs_store[H]); buf._frames[:buf.states_len].copy_(buf._frames[H:H + buf.states_len])


def gae_torch():
    last = 0
    for t in reversed(range(H)):
        nv = lastv if t == H - 1 else buf.value_buf[t + 1]
        nnt = 1.0 - buf.done_buf[t].float()
        delta = buf.rew_buf[t] + nnt * 0.99 * nv - buf.value_buf[t]
        last = delta + nnt * 0.99 * 0.95 * last
        adv[t] = last
    ret = adv + buf.value_buf
    return (adv - adv.mean()) / (adv.std() + 1e-8), ret


buf.value_buf.normal_(); lastv = torch.randn(N, 1, device="cuda"); adv = torch.zeros(H, N, 1, device="cuda")
t_copy, t_fused = timed(loop_copy), timed(loop_fused)
buf.step = H
t_gt, t_gh = timed(gae_torch), timed(lambda: buf.compute_returns_and_advantage(lastv))
# d) the whole rollout incl. policy inference: Python loop (act + collect) vs one taco_rollout_run call
from taco_amd import policy as P  # noqa: E402
import numpy as np  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
rng = np.random.default_rng(0)
Hd = 128
sd = {"log_std": np.zeros(4, np.float32)}
dims = [26, Hd, Hd, Hd, 4]
for i in range(4):
    sd[f"actor_mlp.layers.{2 * i}.weight"] = (rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)
    sd[f"actor_mlp.layers.{2 * i}.bias"] = np.zeros(dims[i + 1], np.float32)
sd["critic_encoder.layers.weight_ih_l0"] = (rng.standard_normal((4 * Hd, 26)) * 0.2).astype(np.float32)
sd["critic_encoder.layers.weight_hh_l0"] = (rng.standard_normal((4 * Hd, Hd)) / np.sqrt(Hd)).astype(np.float32)
sd["critic_encoder.layers.bias_ih_l0"] = np.zeros(4 * Hd, np.float32)
sd["critic_encoder.layers.bias_hh_l0"] = np.zeros(4 * Hd, np.float32)
dims = [Hd, Hd, Hd, 1]
for i in range(3):
    sd[f"critic_mlp.layers.{2 * i}.weight"] = (rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)
    sd[f"critic_mlp.layers.{2 * i}.bias"] = np.zeros(dims[i + 1], np.float32)
pol = P.ActorCritic(sd, 1, a.len_states)


def loop_policy():
    buf.reset()
    for t in range(H):
        action, logp, value, mu, sigma = pol.act(buf.next_obs, buf.next_states)
        buf.collect(env, torch.clip(action, -1, 1), logp, value, mu, sigma, act=action)
    pol.act(buf.next_obs, buf.next_states, deterministic=True)


def run_policy():
    buf.reset()
    buf.run(env, pol)


t_loop, t_run = timed(loop_policy), timed(run_policy)
print(json.dumps({"envs": N, "horizon": H, "len_states": a.len_states,
                  "rollout_with_policy_python_loop_ms": t_loop * 1e3, "rollout_with_policy_one_call_ms": t_run * 1e3,
                  "rollout_with_policy_env_steps_per_s": N * H / t_run,
                  "rollout_copy_ms": t_copy * 1e3, "rollout_fused_ms": t_fused * 1e3,
                  "rollout_copy_env_steps_per_s": N * H / t_copy, "rollout_fused_env_steps_per_s": N * H / t_fused,
                  "gae_torch_ms": t_gt * 1e3, "gae_hip_ms": t_gh * 1e3}))
