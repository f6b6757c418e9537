"""Tail-effect probe: step time vs N around the residency limit (workgroups resident per CU x 256 CUs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
def run(n, steps=60):
    env = FpvBase(config.baseline_config(1, num_envs=n), copy_outputs=False)
    g = torch.Generator().manual_seed(0)
    acts = (0.3 * torch.randn((4, n, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
    for t in range(20): env.step_raw(acts[t % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step_raw(acts[t % 4])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps, env.occupancy(), env.launch_geometry()
for n in [int(x) for x in sys.argv[1:]] or (65536, 131072, 196608, 212992, 229376, 262144, 393216, 524288, 786432, 1048576):
    t, occ, geo = run(n)
    print(f"N={n}: {t:.1f} us  {n / t / 1e3:.2f} G env-steps/s  occupancy(blocks/CU, LDS B)={occ} grid/block={geo}", flush=True)
