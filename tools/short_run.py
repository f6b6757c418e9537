"""Why a K = 20 timed region reads slower than the steady state: per-call host time and the region's wall time for bench.py's main leg
(VecTask.step(), 4 096 envs) cold (W = 5 warm-up steps only, the driver's command) and after a pre-warm of the host path and the GPU clocks.
`python tools/short_run.py`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taco_amd import config
from taco_amd.vec_env import FpvBase

n, K, W = 4096, 20, 5
cfg = config.baseline_config(1, num_envs=n)
acts = bench.make_actions(n, 64, 1000, "cuda")


def region(env, label):
    for t in range(W):
        env.step(acts[t % 64])
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for t in range(K):
        a = time.perf_counter()
        env.step(acts[t % 64])
        per.append((time.perf_counter() - a) * 1e6)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label}: {(t2 - t0) / K * 1e6:.2f} us per step ({n * K / (t2 - t0) / 1e6:.1f} M env-steps/s); host loop {(t1 - t0) * 1e6:.0f} us, final sync "
          f"{(t2 - t1) * 1e6:.0f} us; host us per call: {' '.join(f'{p:.0f}' for p in per)}", flush=True)


env = FpvBase(cfg)
region(env, "cold (first 25 calls of the process)")
region(env, "again (host path warm, GPU idle between)")
twin = FpvBase(cfg)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.25:
    for t in range(64):
        twin.step(acts[t])
torch.cuda.synchronize()
region(env, "after 0.25 s of load on a twin env")
time.sleep(0.5)
region(env, "after 0.5 s of idling")
