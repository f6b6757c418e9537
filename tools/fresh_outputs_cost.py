"""What FpvBase(..., fresh_outputs=True) costs: us per VecTask.step() with the default views of the env's buffers vs newly
allocated obs / states tensors (the reference's ownership, vec_task_asymmetry.py:331-332), at several sizes / stack lengths.
    python tools/fresh_outputs_cost.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from taco_amd import config  # noqa: E402
from taco_amd.vec_env import FpvBase  # noqa: E402

for n, ls in ((4096, 1), (4096, 5), (65536, 1), (262144, 5)):
    acts = bench.make_actions(n, 8, 0, "cuda")
    row = []
    for fresh in (False, True):
        cfg = config.baseline_config(1, num_envs=n)
        cfg["env"]["lenStates"] = ls
        env = FpvBase(cfg, fresh_outputs=fresh)
        med, ws, _ = bench.steady_windows(env.step, acts, torch, 0.2, 5, 2000 if n <= 16384 else 300)
        row.append(med)
        del env
        torch.cuda.empty_cache()
    print(f"{n:7d} envs, {ls} state frame(s): step() {row[0]:7.2f} us with views of the env's buffers, {row[1]:7.2f} us with fresh_outputs=True (+ {row[1] - row[0]:.2f} us)")
