#!/usr/bin/env python3
"""Row N1 (policy forward) measurement, not the headline metric: PPO_ActorCritic.act at N envs.
  a) taco_policy_act (one launch, f32 MFMA)
  b) the same network in stock PyTorch-ROCm (nn.Linear / nn.LSTM / MultivariateNormal, as nets_asymmetry.py builds it), eager, no_grad
Prints one JSON line:  python3 tools/bench_policy.py [--envs 4096] [--hidden 128]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from torch.distributions import MultivariateNormal  # noqa: E402
from taco_amd import policy as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--len-states", type=int, default=5)
a = ap.parse_args()
N, H = a.envs, a.hidden
dev = "cuda:0"
torch.manual_seed(0)
actor = nn.Sequential(nn.Linear(26, H), nn.ReLU(), nn.Linear(H, H), nn.ReLU(), nn.Linear(H, H), nn.ReLU(), nn.Linear(H, 4), nn.Tanh()).to(dev)
lstm = nn.LSTM(26, H, 1, batch_first=True).to(dev)
critic = nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.Linear(H, H), nn.ReLU(), nn.Linear(H, 1)).to(dev)
log_std = torch.zeros(4, device=dev)
sd = {"log_std": log_std}
sd.update({f"actor_mlp.layers.{k}": v for k, v in actor.state_dict().items()})
sd.update({f"critic_encoder.layers.{k}": v for k, v in lstm.state_dict().items()})
sd.update({f"critic_mlp.layers.{k}": v for k, v in critic.state_dict().items()})
T = a.len_states
pol = P.ActorCritic(sd, 1, T)
obs, st = torch.randn(N, 1, 26, device=dev), torch.randn(N, T, 26, device=dev)


def torch_act():
    with torch.no_grad():
        mean = actor(obs.view(N, -1))
        dist = MultivariateNormal(mean, scale_tril=torch.diag(log_std.exp() * log_std.exp()))
        action = dist.sample()
        logp = dist.log_prob(action)
        x, _ = lstm(st)
        value = critic(x[:, -1, :])
        return action, logp, value, mean, log_std.repeat(N, 1)


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps


t_hip = timed(lambda: pol.act(obs, st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    pol.act(obs, st)
e1.record(); torch.cuda.synchronize()
t_hip_gpu = e0.elapsed_time(e1) * 1e-3 / a.reps
t_torch = timed(torch_act)
mu_h, v_h = pol.act(obs, st, deterministic=True)[3], pol.act(obs, st, deterministic=True)[2]
_, _, v_t, mu_t, _ = torch_act()
flops = 2 * N * (26 * H + 2 * H * H + H * 4 + T * 4 * H * (26 + H) + 2 * H * H + H)
t_actor = timed(lambda: pol.act(obs, None, action_only=True))
st_ = torch.zeros(16, dtype=torch.int64, device=dev)
pol.stamps = st_
acc_ = torch.zeros(16, dtype=torch.float64)
for _ in range(20):
    pol.act(obs, st)
    acc_ += st_.cpu().double()
pol.stamps = None
acc_ /= 20
phases = {"actor_stage": acc_[1] - acc_[0], "actor_mlp": acc_[2] - acc_[1], "actor_tail": acc_[3] - acc_[2],
          "critic_stage": acc_[9] - acc_[8], "critic_lstm": acc_[10] - acc_[9], "critic_mlp": acc_[11] - acc_[10], "critic_tail": acc_[12] - acc_[11]}
phases = {k: round(float(v)) for k, v in phases.items()}
print(json.dumps({"envs": N, "phase_ticks": phases, "hidden": H, "hip_act_us": t_hip * 1e6, "hip_act_gpu_us": t_hip_gpu * 1e6, "torch_act_us": t_torch * 1e6,
                  "len_states": T, "actor_only_us": t_actor * 1e6, "flops": flops, "hip_tflops": flops / t_hip_gpu / 1e12,
                  "max_abs_mu_diff_vs_torch": float((mu_h - mu_t).abs().max()), "max_abs_value_diff_vs_torch": float((v_h - v_t).abs().max())}))
