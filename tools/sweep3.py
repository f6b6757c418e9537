import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
def run(n, steps=100):
    env = FpvBase(config.baseline_config(1, num_envs=n), copy_outputs=False)
    g = torch.Generator().manual_seed(0)
    acts = (0.3 * torch.randn((4, n, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
    for t in range(20): env.step_raw(acts[t % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step_raw(acts[t % 4])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
print(os.environ.get("TACO_ENV_LIB"), " ".join(f"N={n}: {run(n):.1f}us" for n in (4096, 65536, 262144, 1048576)), flush=True)
