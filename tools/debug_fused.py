"""Debugging aid: the persistent rollout kernel vs the launch-per-step path, array by array, for growing horizons (prints the first mismatch of each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from taco_amd import config, policy as P
from taco_amd.rollout import RolloutBuffer
from taco_amd.vec_env import FpvBase
import test_policy_gpu as TP

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 5
rng = np.random.default_rng(7)
sd = TP._random_policy(rng, 1, T, [128, 128, 128], 128, [128, 128])
keys = ("mu_buf", "act_buf", "logp_buf", "sigma_buf", "obs_buf", "states_buf", "rew_buf", "done_buf", "value_buf")
for H in (1, 2, 3, 8):
    outs = []
    for fused in (True, False):
        cfg = config.default_cfg("pos", n, env_lenStates=T, env_maxEpisodeLength=30, seed=4)
        env = FpvBase(cfg, copy_outputs=False)
        env.set_rollout_fusion(fused)
        pol = P.ActorCritic(sd, 1, T, seed=21)
        buf = RolloutBuffer(n, 26, 1, 26, T, 4, H, 1, 0.99, 0.95, "cuda:0")
        snaps = []
        for epoch in range(2):
            buf.reset()
            last = buf.run(env, pol)
            snaps.append({k: getattr(buf, k).clone().cpu().numpy() for k in keys} | {"obs_store": buf._obs_store.clone().cpu().numpy(), "frames": buf._frames.clone().cpu().numpy(),
                          "state": env.get_state().cpu().numpy(), "reset": env.reset_buf.cpu().numpy()})
        torch.cuda.synchronize()
        outs.append(snaps)
    for epoch in range(2):
        for k in outs[0][epoch]:
            a, b = outs[0][epoch][k], outs[1][epoch][k]
            bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            if bad.any():
                idx = np.argwhere(bad)
                print(f"H={H} epoch {epoch} {k}: {bad.sum()}/{bad.size} differ; first {idx[0]}: fused {a[tuple(idx[0])]} vs {b[tuple(idx[0])]}; last {idx[-1]}")
            else:
                print(f"H={H} epoch {epoch} {k}: equal")
