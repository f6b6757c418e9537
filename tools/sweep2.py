#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
def run(n, substeps=2, idx=1, steps=100):
    cfg = config.baseline_config(idx, num_envs=n); cfg["sim"]["substeps"] = substeps
    env = FpvBase(cfg, copy_outputs=False)
    g = torch.Generator().manual_seed(0)
    acts = (0.3 * torch.randn((4, n, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
    for t in range(20): env.step_raw(acts[t % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step_raw(acts[t % 4])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
for n in (131072, 262144, 524288):
    for ss in (1, 2, 4, 8):
        print(f"N={n} substeps={ss}: {run(n, ss):8.1f} us", flush=True)
for idx in (2, 3, 4):
    print(f"config {idx} N=262144: {run(262144, 2, idx):8.1f} us", flush=True)
