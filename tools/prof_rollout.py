"""Minimal driver to put under rocprofv3: `reps` taco_rollout_run rollouts (+ reset + GAE) of config 5's flags.  --unfused: launch-per-step path.
--envs=N --horizon=H: another shape (default 4 096 x 32; above 8 192 envs the rollout is launch-per-step: actor kernel + step kernel per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from taco_amd import config, policy as P
from taco_amd.rollout import RolloutBuffer
from taco_amd.vec_env import FpvBase
import test_policy_gpu as TP

_o = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
n, H, reps = int(_o.get("--envs", 4096)), int(_o.get("--horizon", 32)), int(_o.get("--reps", 20))
cfg = config.baseline_config(4, num_envs=n)
env = FpvBase(cfg, copy_outputs=False)
env.set_rollout_fusion("--unfused" not in sys.argv)
T = env.len_states
pol = P.ActorCritic(TP._random_policy(np.random.default_rng(0), 1, T, [128, 128, 128], 128, [128, 128]), 1, T)
buf = RolloutBuffer(n, 26, 1, 26, T, 4, H, 4, 0.99, 0.95, "cuda:0")
for _ in range(reps):
    buf.reset()
    last = buf.run(env, pol)
    buf.compute_returns_and_advantage(last)
torch.cuda.synchronize()
print("done", env.step_count, flush=True)
