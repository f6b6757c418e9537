#!/usr/bin/env python3
"""Row N1 measurement (not the headline metric): the batched critic (taco_critic_values: persistent LSTM kernel + MLP kernel) over
H x N + N state stacks of the documented architecture (LSTM 26 -> 128 over T frames, MLP 128-128-128-1).
Prints one JSON line:  python3 tools/bench_critic.py [--rows 135168] [--len-states 5] [--once]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from taco_amd import policy as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=33 * 4096)
ap.add_argument("--len-states", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--once", action="store_true", help="a few launches only (for rocprofv3 --pmc runs)")
a = ap.parse_args()
import test_policy_gpu as TP  # noqa: E402
rng = np.random.default_rng(0)
T, H = a.len_states, 128
sd = TP._random_policy(rng, 1, T, [128, 128, 128], H, [H, H])
pol = P.ActorCritic(sd, 1, T)
st = torch.randn(a.rows, T, 26, device="cuda")
if a.once:
    for _ in range(3):
        v = pol.values(st)
    torch.cuda.synchronize()
    print("done", float(v.sum()))
    sys.exit(0)
t_end = time.perf_counter() + 0.05
while time.perf_counter() < t_end:
    pol.values(st)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        pol.values(st)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e-3 / a.reps)
ts.sort()
# MFMA flops of the padded problem (what the matrix pipe executes): LSTM 4 gates x 128 x (32 + 128) per timestep (the first timestep's
# h chain is skipped: 4 x 128 x 32), MLP 128 x 128 x 2 + 128 x 16
flops_exec = 2 * a.rows * (4 * H * 32 + (T - 1) * 4 * H * (32 + H) + 2 * H * H + H * 16)
flops_model = 2 * a.rows * (T * 4 * H * (26 + H) + 2 * H * H + H)   # the network's own count (unpadded, full first timestep)
stamps = torch.zeros(259, dtype=torch.int64, device="cuda")
pol.values(st, stamps=stamps)
torch.cuda.synchronize()
sp = stamps.cpu().numpy()
ph = sp[:126].reshape(2, 21, 3)
wall_us = (sp[257] - sp[256]) / 100.0
phases = {"wave0_mfma_cycles": (ph[0, :, 1] - ph[0, :, 0]).tolist(), "wave0_cells_cycles": (ph[0, :, 2] - ph[0, :, 1]).tolist(),
          "wave4_mfma_cycles": (ph[1, :, 1] - ph[1, :, 0]).tolist(), "wave4_cells_cycles": (ph[1, :, 2] - ph[1, :, 1]).tolist(),
          "timestep_cycles": np.diff(ph[0, :, 0]).tolist(),
          "shader_clock_GHz": float((sp[258] - ph[0, 0, 0]) / (wall_us * 1e3)) if wall_us > 0 else None, "workgroup0_us": float(wall_us)}
print(json.dumps({"phases_of_workgroup_0": phases, "rows": a.rows, "len_states": T, "critic_ms": ts[2] * 1e3, "windows_ms": [t * 1e3 for t in ts],
                  "tflops_model": flops_model / ts[2] / 1e12, "tflops_executed": flops_exec / ts[2] / 1e12,
                  "frac_of_157_model": flops_model / ts[2] / 157.3e12, "rows_per_s": a.rows / ts[2]}))
