"""Step time with frame stacks (lenObservations / lenStates > 1): python3 tools/sweep5.py N len_obs len_states ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
def run(n, lo, ls, steps=60):
    cfg = config.baseline_config(1, num_envs=n)
    cfg["env"]["lenObservations"], cfg["env"]["lenStates"] = lo, ls
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
if __name__ == "__main__":
  for n in (4096, 262144):
      print(f"N={n}: " + "  ".join(f"obs{lo}/st{ls}: {run(n, lo, ls):.1f} us" for lo, ls in ((1, 1), (1, 5), (5, 5), (1, 10))), flush=True)
