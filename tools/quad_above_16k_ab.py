"""Do the four-lanes-per-env (quad) forms pay ABOVE 16 384 envs?  choose_form stops them there (kQuadMaxEnvs) and tools/ring_form_ab.py never pinned
them beyond; this sweep does: BASELINE configs 2 ... 5's flags at 24 576 ... 65 536 envs through VecTask.step() (state stack on the frame ring), every
form pinned, us per step (bench.py's steady-state protocol).    python tools/quad_above_16k_ab.py [--sizes=24576,32768,49152,65536]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from taco_amd import config  # noqa: E402
from taco_amd.vec_env import FpvBase  # noqa: E402

sizes = [int(x) for x in next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--sizes=")), "24576,32768,49152,65536").split(",")]
forms = ["auto", "lane", "lane_roles", "lane_throughput", "quad", "quad_served", "quad_roles"]
for idx in (1, 2, 3, 4):
    for n in sizes:
        acts = bench.make_actions(n, 8, 0, "cuda")
        row = {}
        for form in forms:
            env = FpvBase(config.baseline_config(idx, num_envs=n), kernel_form=form)
            med, ws, _ = bench.steady_windows(env.step, acts, torch, 0.15, 5, 600)
            row[form if form != "auto" else f"auto={env.kernel_form}"] = med
            del env
            torch.cuda.empty_cache()
        best = min(row, key=row.get)
        print(f"config {idx + 1} {n:6d} envs: " + "  ".join(f"{k} {v:6.2f}" for k, v in row.items()) + f"   <- {best}", flush=True)
