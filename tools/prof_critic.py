"""Minimal driver to put under rocprofv3: `reps` values_ring() calls of the batched critic (the host layer's default form: split f16) over a frame
ring of `slots` slots x `envs` envs (default 17 x 32 768 = 557 056 rows, the critic pass of config 5's rollout at its per-rank shape).
--f32: the f32 MFMA kernel instead (ActorCritic(critic_split=None))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from taco_amd import policy as P
import test_policy_gpu as TP

_o = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
slots, n, reps, T = int(_o.get("--slots", 17)), int(_o.get("--envs", 32768)), int(_o.get("--reps", 30)), int(_o.get("--len-states", 5))
pol = P.ActorCritic(TP._random_policy(np.random.default_rng(0), 1, T, [128, 128, 128], 128, [128, 128]), 1, T, critic_split=None if "--f32" in sys.argv else "auto")
fr = torch.randn(slots + T - 1, n, 26, device="cuda")
for _ in range(reps):
    v = pol.values_ring(fr)
torch.cuda.synchronize()
print("done", slots * n, float(v.sum()), flush=True)
