#!/usr/bin/env python3
"""Which instantiation should a launch take whose role wavefronts have no stack history to move (the states FRAME RING: one frame written; or
no stack at all)?  One BASELINE config's flags at several env counts, VecTask.step() with each form pinned (`auto` = the library's choice).
    python tools/ring_form_ab.py [--config=2..5] [envs ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taco_amd import config  # noqa: E402
from taco_amd.vec_env import FpvBase  # noqa: E402

number = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--config=")), 5)
sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [8192, 16384, 32768, 65536, 131072]
dev = torch.device("cuda:0")
for n in sizes:
    acts = bench.make_actions(n, 8, 11, dev)
    forms = ["auto", "lane", "lane_roles", "lane_throughput"] + (["quad", "quad_roles", "quad_served"] if n <= 16384 else [])
    row = []
    for form in forms:
        env = FpvBase(config.baseline_config(number - 1, num_envs=n), kernel_form=form)
        med, _, _ = bench.steady_windows(env.step, acts, torch, 0.1, 5, 300)
        row.append(f"{form} {med:.2f}")
        env_desc = f"{env.len_states} state frames on the ring" if env._ring_on else "no stack"
        del env
    print(f"config {number}'s flags, {n} envs ({env_desc}): " + " | ".join(row), flush=True)
