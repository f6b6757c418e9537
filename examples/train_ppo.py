#!/usr/bin/env python3
"""End-to-end PPO on the HIP path: what switching from the reference looks like for a user.

The reference's run() (ppo_asymmetry.py:286-371) alternates `collect a rollout` and `update the networks`.  Here
  * the rollout is ONE call: buf.run(env, pol) = taco_rollout_run (policy forward + env step for horizon_len steps, final value, time-out
    bootstrap), then buf.compute_returns_and_advantage() = taco_gae;
  * the update is stock PyTorch-ROCm autograd on a torch module whose parameter names equal the reference's PPO_ActorCritic
    (actor_mlp.layers.N / critic_encoder.layers.* / critic_mlp.layers.N / log_std), so checkpoints are interchangeable;
  * the per-epoch schedules, the spectral-norm clamp and the TorchScript export come from taco_amd.train_utils (row N3).
This file is caller-side glue (the PPO update is out of the hot path's scope); it exists to show the wiring.

    python examples/train_ppo.py --task_mode=pos --num_envs=4096 --epochs=30 --horizon_len=32
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.distributions import MultivariateNormal  # noqa: E402
from taco_amd import config  # noqa: E402
from taco_amd.policy import ActorCritic  # noqa: E402
from taco_amd.rollout import RolloutBuffer  # noqa: E402
from taco_amd.train_utils import Schedules, save_actor_as_pt, spectral_normalize_actor_  # noqa: E402
from taco_amd.vec_env import isaacgym_task_map  # noqa: E402


def mlp(sizes, out_act):
    layers = []
    for j in range(len(sizes) - 1):
        layers += [nn.Linear(sizes[j], sizes[j + 1]), nn.ReLU() if j < len(sizes) - 2 else out_act()]
    return nn.Sequential(*layers)


class Seq(nn.Module):  # gives the parameters the reference's names: <name>.layers.<idx>.weight
    def __init__(self, net):
        super().__init__()
        self.layers = net

    def forward(self, x):
        return self.layers(x)


class TrainNet(nn.Module):
    """Training-side twin of PPO_ActorCritic.evaluate (nets_asymmetry.py:357-378) for the documented configuration."""

    def __init__(self, len_obs, hidden):
        super().__init__()
        self.actor_mlp = Seq(mlp([26 * len_obs, hidden, hidden, hidden, 4], nn.Tanh))
        self.critic_encoder = Seq(nn.LSTM(26, hidden, 1, batch_first=True))
        self.critic_mlp = Seq(mlp([hidden, hidden, hidden, 1], nn.Identity))
        self.log_std = nn.Parameter(torch.zeros(4))
        for m in list(self.actor_mlp.layers) + list(self.critic_mlp.layers):
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, 2 ** 0.5)
                nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.actor_mlp.layers[-2].weight, 0.01)

    def evaluate(self, obs, states, actions):
        mean = self.actor_mlp(obs.reshape(obs.size(0), -1))
        dist = MultivariateNormal(mean, scale_tril=torch.diag(self.log_std.exp() * self.log_std.exp()))
        x, _ = self.critic_encoder.layers(states)
        return dist.log_prob(actions), dist.entropy(), self.critic_mlp(x[:, -1, :])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task_mode", default="pos")
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--horizon_len", type=int, default=32)
    ap.add_argument("--train_iters", type=int, default=4)
    ap.add_argument("--mini_batch_num", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--lipschitz", type=float, default=0.0, help="> 0 enables the spectral-norm clamp (the reference's --lipschitz_para)")
    ap.add_argument("--export", default="", help="write the TorchScript actor here at the end")
    a = ap.parse_args()
    dev = "cuda:0"
    N, H, len_states = a.num_envs, a.horizon_len, 5
    cfg = config.default_cfg(a.task_mode, N, env_lenStates=len_states, env_clipActions=1.0)
    env = isaacgym_task_map[cfg["name"]](cfg, dev, dev, -1, True, False, False)
    net = TrainNet(1, a.hidden).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, eps=1e-5)
    pol = ActorCritic(net.state_dict(), 1, len_states, device=dev, seed=1)
    buf = RolloutBuffer(N, 26, 1, 26, len_states, 4, H, a.mini_batch_num, 0.99, 0.95, dev)
    sched = Schedules(epochs=a.epochs, lipschitz_para=a.lipschitz or 5)
    clip, vf_coef, max_grad, target_kl = 0.2, 0.5, 0.5, 0.03
    for epoch in range(a.epochs):
        lr, lip, env.difficulty = sched.at(epoch)
        opt.param_groups[0]["lr"] = lr
        t0 = time.perf_counter()
        buf.reset()
        last_value = buf.run(env, pol)                      # the whole rollout: 2 H + 2 launches
        mean_rew, dones = float(buf.rew_buf.mean()), int(buf.done_buf.sum())
        buf.compute_returns_and_advantage(last_value)       # GAE + returns + advantage normalisation: 3 launches
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        flat = lambda t: t.reshape(-1, *t.shape[2:])  # noqa: E731
        obs, st, act = flat(buf.obs_buf), flat(buf.states_buf), flat(buf.act_buf)
        ret, adv, logp_old = flat(buf.ret_buf), flat(buf.adv_buf).squeeze(-1), flat(buf.logp_buf).squeeze(-1)
        stop = False
        for _ in range(a.train_iters):
            for idx in buf.batch_idx_generator():
                idx = torch.as_tensor(idx, device=dev)
                logp, ent, value = net.evaluate(obs[idx], st[idx], act[idx])
                ratio = torch.exp(logp - logp_old[idx])
                pg = -torch.min(adv[idx] * ratio, adv[idx] * ratio.clamp(1 - clip, 1 + clip)).mean()
                loss = pg + vf_coef * F.mse_loss(ret[idx], value)
                with torch.no_grad():
                    lr_ = logp - logp_old[idx]
                    kl = float(((lr_.exp() - 1) - lr_).mean())
                if kl > 1.5 * target_kl:
                    stop = True
                    break
                opt.zero_grad()
                loss.backward()
                nn.utils.clip_grad_norm_(net.parameters(), max_grad)
                opt.step()
                if a.lipschitz > 0:
                    spectral_normalize_actor_(net.actor_mlp, lip)
            if stop:
                break
        pol.load_state_dict(net.state_dict())               # refresh the packed weights of the HIP policy
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"epoch {epoch:3d}  difficulty {env.difficulty:.2f}  lr {lr:.2e}  reward/step {mean_rew:.5f}  episodes done {dones:6d}  "
              f"rollout {1e3 * (t1 - t0):7.2f} ms ({N * H / (t1 - t0) / 1e6:6.1f} M env-steps/s)  update {1e3 * (t2 - t1):7.1f} ms", flush=True)
    if a.export:
        save_actor_as_pt(net.state_dict(), a.export, 1)
        print("TorchScript actor written to", a.export)


if __name__ == "__main__":
    main()
