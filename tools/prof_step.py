#!/usr/bin/env python3
"""Minimal driver for profiling the step kernel under rocprofv3:  python3 tools/prof_step.py --envs N --steps K [--config I]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from taco_amd import config  # noqa: E402
from taco_amd.vec_env import FpvBase  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--form", default="auto", help="kernel form to pin (taco_amd._lib.FORMS)")
ap.add_argument("--api", action="store_true", help="call VecTask.step() (clamped copies, dict outputs) instead of step_raw()")
a = ap.parse_args()
env = FpvBase(config.baseline_config(a.config, num_envs=a.envs), copy_outputs=a.api, kernel_form=a.form)
g = torch.Generator().manual_seed(0)
acts = (0.3 * torch.randn((8, a.envs, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
for t in range(a.steps):
    if a.api:
        env.step(acts[t % 8])
    else:
        env.step_raw(acts[t % 8])
torch.cuda.synchronize()
print("done", a.envs, a.steps, env.launch_geometry())
