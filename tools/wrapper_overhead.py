"""VecTask.step() (reference semantics: fresh clamped copies) vs step_raw() (one launch, buffers in place), 4 096 envs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvPos
def bench(fn, n=2000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for copy in (True, False):
    env = FpvPos(config.baseline_config(1), copy_outputs=copy)
    a = (0.3 * torch.randn(4096, 4, device="cuda")).clamp(-1, 1)
    print(f"copy_outputs={copy}: step() {bench(lambda: env.step(a)):.1f} us   step_raw() {bench(lambda: env.step_raw(a)):.1f} us")
