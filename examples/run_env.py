#!/usr/bin/env python3
"""Thin runner with the reference launcher's call sequence (train_fpv_asymmetry_ppo.py:363-371 + ppo_asymmetry.py:297-311):
build cfg -> isaacgym_task_map[name](cfg, ...) -> env.reset() -> env.step(clipped_act) in a loop, with a stand-in policy
(a fixed random linear map of the observation).  Prints env-steps/s and the running reward / done statistics.

    python examples/run_env.py --task_mode=mix --num_envs=16384 --delay_time=20 --observation_noise=True
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from taco_amd import cfg_io  # noqa: E402
from taco_amd.vec_env import isaacgym_task_map  # noqa: E402


def main():
    task, args = cfg_io.parse()
    env = isaacgym_task_map[task["name"]](task, "cuda:0", "cuda:0", -1, True, False, False)
    g = torch.Generator(device="cuda").manual_seed(0)
    W = 0.3 * torch.randn(env.len_obs * env.num_obs, env.num_acts, device="cuda", generator=g)
    b = torch.tensor([-0.3, 0.0, 0.0, 0.0], device="cuda")
    obs = env.reset()["obs"]
    steps, ret, dones = 300, 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        act = torch.tanh(obs.reshape(env.num_envs, -1).nan_to_num() @ W + b)
        o, rew, done, info = env.step(torch.clip(act, -1, 1))
        obs = o["obs"]
        ret += float(rew.mean())
        dones += int(done.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{task['name']}: {env.num_envs} envs x {steps} steps in {dt:.3f} s = {env.num_envs * steps / dt / 1e6:.1f} M env-steps/s; "
          f"mean reward/step {ret / steps:.5f}; episodes finished {dones}")


if __name__ == "__main__":
    main()
