"""What ONE all_gather_into_tensor per step costs on the host side of torch.distributed's RCCL backend, measured with the one rank a one-GPU box allows
(the collective degenerates to a device copy, the enqueue path -- ProcessGroupNCCL, event record / wait between the streams -- is the real one):
4 096 envs, 2 000 steps each of  step_local (no collective) / the overlapped gathered loop of bench.py / the serial gathered step.
    python tools/gather_host_cost.py [envs] [--torch]      (--torch: the process group's own collective instead of the direct call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1)
from taco_amd import config
from taco_amd.dist import ShardedEnv

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
direct = "--torch" not in sys.argv
env = ShardedEnv(config.baseline_config(1, num_envs=n), rank=0, world_size=1, device=dev, gather=True, collective_when_alone=True, direct=direct)
print("all-gather path:", "ncclAllGather called directly (taco_amd/rccl.py)" if env.direct_path else f"torch.distributed all_gather_into_tensor ({env.direct_reason})", flush=True)
g = torch.Generator().manual_seed(0)
acts = (0.3 * torch.randn((64, n, 4), generator=g) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).to(dev)
pend = []


def wait_previous():
    while pend:
        pend.pop(0).wait()


def overlapped(a):
    pend.append(env.step_async(a, before_gather=wait_previous))


def run(fn, steps=2000, finish=None):
    for t in range(200):
        fn(acts[t % 64])
    if finish:
        finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        fn(acts[t % 64])
    if finish:
        finish()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return host / steps * 1e6, (time.perf_counter() - t0) / steps * 1e6


for name, fn, fin in (("step_local (no collective)", env.step_local, None), ("overlapped gathered loop", overlapped, lambda: (wait_previous(), env.drain())),
                      ("serial gathered step", env.step_gathered, None), ("step_local (no collective)", env.step_local, None)):
    h, w = run(fn, finish=fin)
    print(f"{name:32s} host {h:7.2f} us per step to enqueue, {w:7.2f} us per step wall ({n} envs, RCCL backend, 1 rank)", flush=True)
dist.destroy_process_group()
