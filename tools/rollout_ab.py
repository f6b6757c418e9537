"""taco_rollout_run with the persistent actor + step kernel vs the launch-per-step path: ms per rollout (median of 5 x 4), the critic alone,
and the persistent kernel's in-kernel step clocks.  `python tools/rollout_ab.py [n] [horizon] [config index] [forms]`; forms: comma-separated
from quad, lane (the persistent kernel's two forms, forced), auto (the host's choice), off (launch per step); default auto,off,auto,off"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from taco_amd import config, policy as P
from taco_amd.rollout import RolloutBuffer
from taco_amd.vec_env import FpvBase
import test_policy_gpu as TP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ci = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(0)
forms = [{"auto": True, "off": False}.get(f, f) for f in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["auto", "off", "auto", "off"])]
for fused in forms:
    cfg = config.baseline_config(ci, num_envs=n)
    env = FpvBase(cfg, copy_outputs=False)
    env.set_rollout_fusion(fused)
    T = env.len_states
    sd = TP._random_policy(rng, 1, T, [128, 128, 128], 128, [128, 128])
    pol = P.ActorCritic(sd, 1, T)
    buf = RolloutBuffer(n, 26, 1, 26, T, 4, H, 4, 0.99, 0.95, "cuda:0")
    stamps = torch.zeros(136 + (n + 15) // 16, dtype=torch.int64, device="cuda")
    env.bind_rollout_stamps(stamps)
    nwg = (n + 63) // 64 if (fused == "lane" or (fused is True and n > 8192)) else (n + 15) // 16

    def run():
        buf.reset()
        last = buf.run(env, pol)
        buf.compute_returns_and_advantage(last)

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 4)
    ts.sort()
    # the per-step chain alone: events around run() minus the critic (timed alone)
    cs = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            pol.values_ring(buf._frames)
        e1.record()
        torch.cuda.synchronize()
        cs.append(e0.elapsed_time(e1) / 4)
    cs.sort()
    line = f"n={n} H={H} config {ci + 1} fused={fused}: {ts[2] * 1e3:.3f} ms per rollout ({n * H / ts[2] / 1e6:.1f} M env-steps/s), critic alone {cs[2]:.3f} ms"
    if fused:
        st = stamps.cpu().numpy()
        per = (st[9:9 + 2 * min(H, 64):2] - st[8:8 + 2 * min(H, 64):2])
        gap = (st[10:8 + 2 * min(H, 64):2] - st[9:7 + 2 * min(H, 64):2])
        line += f"; in-kernel step ticks (step wavefront, workgroup 0): median {int(np.median(per))}, max {int(per.max())}, total {int(st[7 + 2 * min(H, 64)] - st[8])}; simd {st[:8].tolist()}; per-workgroup loop ticks ({nwg} workgroups): min {int(st[136:136 + nwg].min())} median {int(np.median(st[136:136 + nwg]))} max {int(st[136:136 + nwg].max())}"
    print(line, flush=True)
