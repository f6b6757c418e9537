#!/usr/bin/env python3
"""Kernel time vs env count / substeps (HIP events around back-to-back launches)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase

def run(n, substeps=2, idx=1, steps=300):
    cfg = config.baseline_config(idx, num_envs=n); cfg["sim"]["substeps"] = substeps
    env = FpvBase(cfg, copy_outputs=False)
    g = torch.Generator().manual_seed(0)
    acts = (0.3 * torch.randn((8, n, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
    for t in range(50): env.step_raw(acts[t % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step_raw(acts[t % 8])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    return us

for n in (64, 1024, 4096, 16384, 65535, 65536, 262144, 1048576):
    print(f"N={n:8d} substeps=2: {run(n):8.1f} us/step  -> {n/run(n):8.1f} M env-steps/s", flush=True)
for ss in (1, 4, 8):
    print(f"N=4096 substeps={ss}: {run(4096, ss):8.1f} us/step", flush=True)
