import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from taco_amd import config
from taco_amd.vec_env import FpvBase
from taco_amd.dist import ShardedEnv
n = 4096
cfg = config.baseline_config(1, num_envs=n)
acts = bench.make_actions(n, 64, 1000, "cuda")
env = FpvBase(cfg)
sh = ShardedEnv(cfg, rank=0, world_size=1, device=torch.device("cuda:0"), gather=False)
for name, fn in (("VecTask.step", env.step), ("ShardedEnv.step_gathered", sh.step_gathered), ("step_raw", env.step_raw)):
    for t in range(200):
        fn(acts[t % 64])
    torch.cuda.synchronize()
    # host-only cost: enqueue 300 calls, measure host loop time (GPU queue absorbs) vs total
    t0 = time.perf_counter()
    for t in range(300):
        fn(acts[t % 64])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {(t1 - t0) / 300 * 1e6:.2f} us per call, wall {(t2 - t0) / 300 * 1e6:.2f} us per step", flush=True)
