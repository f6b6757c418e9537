"""Which workgroups of the persistent rollout kernel are slow?  Per-workgroup loop clocks (taco_bind_rollout_stamps) against the number of
steps in which the workgroup held a resetting env, by task third (config 5's flags, 4 096 envs x 32 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from taco_amd import config, policy as P
from taco_amd.rollout import RolloutBuffer
from taco_amd.vec_env import FpvBase
import test_policy_gpu as TP

n, H = 4096, 32
env = FpvBase(config.baseline_config(4, num_envs=n), copy_outputs=False)
T = env.len_states
pol = P.ActorCritic(TP._random_policy(np.random.default_rng(0), 1, T, [128, 128, 128], 128, [128, 128]), 1, T)
buf = RolloutBuffer(n, 26, 1, 26, T, 4, H, 4, 0.99, 0.95, "cuda:0")
stamps = torch.zeros(136 + n // 16, dtype=torch.int64, device="cuda")
env.bind_rollout_stamps(stamps)
for rep in range(6):
    buf.reset()
    prev_done = env.reset_buf.clone()
    buf.run(env, pol)
    torch.cuda.synchronize()
done = torch.cat([prev_done.view(1, n).float(), buf.done_buf.view(H, n)[:-1]], 0)   # reset flag each step STARTS with
wg_reset_steps = (done.view(H, n // 16, 16).sum(2) > 0).sum(0).cpu().numpy()
ticks = stamps[136:].cpu().numpy()
third = np.minimum(np.arange(n // 16) * 16 // (n // 3 + 1), 2)
for g, name in enumerate(("pos", "rotate", "flip")):
    m = third == g
    print(f"{name:6s} workgroups {m.sum():3d}: loop clocks min {ticks[m].min()} median {int(np.median(ticks[m]))} max {ticks[m].max()}; reset-steps per workgroup median {np.median(wg_reset_steps[m])} max {wg_reset_steps[m].max()}")
A = np.stack([np.ones_like(wg_reset_steps, dtype=float), wg_reset_steps, (third == 1).astype(float), (third == 2).astype(float)], 1)
coef, *_ = np.linalg.lstsq(A, ticks.astype(float), rcond=None)
print(f"least squares: loop clocks = {coef[0]:.0f} + {coef[1]:.0f} per reset-step + {coef[2]:.0f} if rotate + {coef[3]:.0f} if flip   (per step: base {coef[0] / H:.0f}, flip +{coef[3] / H:.0f})")
print("resets per env-step:", float(done.mean()))
blob = env.get_state()[:40].cpu().numpy()
bad_env = ~np.isfinite(blob).all(0)
big_env = (np.abs(np.nan_to_num(blob[10:13], nan=0.0)) > 100).any(0)
bad_wg = bad_env.reshape(-1, 16).any(1)
print("envs with a non-finite state word:", int(bad_env.sum()), " workgroups holding one:", int(bad_wg.sum()), " envs with |omega| > 100 rad/s:", int(big_env.sum()))
if bad_wg.any():
    print(f"loop clocks: workgroups with a non-finite env median {int(np.median(ticks[bad_wg]))}, without {int(np.median(ticks[~bad_wg]))}")
order = np.argsort(ticks)
print("slowest 8 workgroups:", [(int(b), int(ticks[b]), int(wg_reset_steps[b]), bool(bad_wg[b])) for b in order[-8:]])
print("fastest 8 workgroups:", [(int(b), int(ticks[b]), int(wg_reset_steps[b]), bool(bad_wg[b])) for b in order[:8]])
full = env.get_state().cpu().numpy()
L = full[66].view(np.int32).reshape(-1, 16)
prog = full[65].view(np.int32).reshape(-1, 16)
print("workgroups above 1.0 M clocks:", int((ticks > 1.0e6).sum()), "of", len(ticks))
for b in order[-6:]:
    print(f"  wg {int(b)} clocks {int(ticks[b])}: pending slots L of its envs {L[b].tolist()} progress {prog[b].tolist()}")
for b in order[:3]:
    print(f"  wg {int(b)} clocks {int(ticks[b])}: pending slots L of its envs {L[b].tolist()}")
