#!/usr/bin/env python3
"""Thin runner with the reference launcher's call sequence (train_fpv_asymmetry_ppo.py:363-371 + ppo_asymmetry.py:297-311):
build cfg -> isaacgym_task_map[name](cfg, ...) -> env.reset() -> env.step(clipped_act) in a loop, with a stand-in policy
(a fixed random linear map of the observation).  Prints env-steps/s and the running reward / done statistics.

    python examples/run_env.py --task_mode=mix --num_envs=16384 --delay_time=20 --observation_noise=True
    python examples/run_env.py --cfg my_task.yaml --steps 50 --dump out.npz     (a YAML in the reference's layout; --dump keeps the last
                                                                                  step's tensors and the env's raw state for comparison)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from taco_amd import cfg_io  # noqa: E402
from taco_amd.vec_env import isaacgym_task_map  # noqa: E402


def main():
    own = argparse.ArgumentParser(add_help=False)
    own.add_argument("--cfg", type=str, default=None)
    own.add_argument("--steps", type=int, default=300)
    own.add_argument("--dump", type=str, default=None)
    o, rest = own.parse_known_args()
    task, args = cfg_io.parse(rest, yaml_path=o.cfg)
    env = isaacgym_task_map[task["name"]](task, "cuda:0", "cuda:0", -1, True, False, False)
    g = torch.Generator(device="cuda").manual_seed(0)
    W = 0.3 * torch.randn(env.len_obs * env.num_obs, env.num_acts, device="cuda", generator=g)
    b = torch.tensor([-0.3, 0.0, 0.0, 0.0], device="cuda")
    obs = env.reset()["obs"]
    steps, ret, dones = o.steps, 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        act = torch.tanh(obs.reshape(env.num_envs, -1).nan_to_num() @ W + b)
        od, rew, done, info = env.step(torch.clip(act, -1, 1))
        obs = od["obs"]
        ret += float(rew.mean())
        dones += int(done.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if o.dump:
        import numpy as np
        np.savez(o.dump, obs=obs.cpu().numpy(), states=od["states"].cpu().numpy(), rew=rew.cpu().numpy(), done=done.cpu().numpy(),
                 state=env.get_state().cpu().numpy(), mean_reward=ret / steps, episodes=dones)
    print(f"{task['name']}: {env.num_envs} envs x {steps} steps in {dt:.3f} s = {env.num_envs * steps / dt / 1e6:.1f} M env-steps/s; "
          f"mean reward/step {ret / steps:.5f}; episodes finished {dones}")


if __name__ == "__main__":
    main()
