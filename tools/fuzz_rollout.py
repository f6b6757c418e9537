"""Rollout fuzz: taco_rollout_run with the persistent actor + step kernel vs the launch-per-step path of the same call, bit for bit, over random
configurations (tools/fuzz.py's generator restricted to what the persistent form takes: one obs frame, at most 8 192 envs, unclamped stacks).

    python tools/fuzz_rollout.py [--cases 200] [--seed 0] [--oracle] [--form auto|quad|lane]

--oracle: both paths run with the exact LSTM cell, and every case of at most 20 000 env-steps per rollout is ALSO held against the CPU oracle's
orc_rollout (oracle/taco_policy_oracle.c) directly: every replay-buffer array, the final value, the env's state, bit for bit.

Every case draws a task, 1..6 000 envs (one in eight 6 000..8 192), all flags, delay_time (0 and 1 often: the actor on the step's critical
path), horizon 1..24, 1..5 state frames, a short episode length (resets and time-out bootstraps inside the rollout), a fresh random policy of the
documented architecture; runs two rollouts on each path and compares every replay-buffer array, reset_buf and the env's whole state bit for bit
(NaNs of a diverged env compare as NaNs: their sign and payload depend on operand order, which the two builds of the reward code do not share).  A hang
would show as the harness's timeout: every case prints a line."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=-1, help="run only this case of the sequence (the draws of the earlier ones are replayed)")
    ap.add_argument("--oracle", action="store_true", help="exact LSTM cell; small cases are also compared with the CPU oracle's orc_rollout")
    ap.add_argument("--form", default="auto", choices=["auto", "quad", "lane"], help="the persistent kernel's form: auto (the host's choice: quad at these sizes), "
                    "quad, or lane (round 6's one-lane form, 64 envs per workgroup, forced)")
    args = ap.parse_args()
    import torch
    import fuzz
    import test_policy_gpu as TP
    from taco_amd import policy as P
    from taco_amd.rollout import RolloutBuffer
    from taco_amd.vec_env import FpvBase
    rng = np.random.default_rng(args.seed)
    keys = ("obs_buf", "states_buf", "act_buf", "rew_buf", "done_buf", "value_buf", "logp_buf", "mu_buf", "sigma_buf", "ret_buf", "adv_buf")
    t0 = time.time()
    resets = bootstraps = vs_oracle = 0
    nan_cases = set()
    if args.oracle:
        from oracle import oracle as O
        from taco_amd import config
    for i in range(args.cases):
        cfg, _, _ = fuzz.draw_case(rng)
        env_c = cfg["env"]
        env_c["numEnvs"] = int(rng.integers(6000, 8193)) if rng.random() < 0.125 else int(rng.integers(1, 6000) if rng.random() < 0.5 else rng.integers(1, 400))
        env_c["lenObservations"] = 1
        env_c["lenStates"] = int(rng.choice([1, 2, 3, 5]))
        env_c["clipObservations"] = env_c["clipStates"] = float("inf")
        env_c["maxEpisodeLength"] = int(rng.choice([5, 11, 25, 1000]))
        cfg["delay_time"] = int(rng.choice([0, 0, 1, 1, 3, 9, 10, 20, 35, 85]))
        n, T, H = env_c["numEnvs"], env_c["lenStates"], int(rng.integers(1, 25))
        if args.only >= 0 and i != args.only:
            continue
        sd = TP._random_policy(np.random.default_rng(1000 + i), 1, T, [128, 128, 128], 128, [128, 128])
        outs = []
        for fused in (True if args.form == "auto" else args.form, False):
            env = FpvBase(cfg, copy_outputs=False)
            env.set_rollout_fusion(fused)
            stamps = torch.zeros(136 + (n + 15) // 16, dtype=torch.int64, device="cuda")
            env.bind_rollout_stamps(stamps)
            pol = P.ActorCritic(sd, 1, T, seed=i, exact_critic=args.oracle)
            buf = RolloutBuffer(n, 26, 1, 26, T, 4, H, 1, 0.99, 0.95, "cuda:0")
            snaps = []
            check_oracle = args.oracle and fused and n * H <= 20000
            if check_oracle:
                orc = O.OracleEnv(config.flat_cfg(cfg), threads=8)
                oc, blob = O.policy_cfg(1, T, [128, 128, 128], 128, [128, 128]), P.pack_state_dict(pol.cfg, sd)
            for epoch in range(2):
                buf.reset()
                last = buf.run(env, pol)
                if check_oracle:
                    exp = O.rollout(orc, oc, blob, H, i, epoch * H, 0.99, orc.obs_buf.copy(), orc.states_buf.copy(), act_lo=-1.0, act_hi=1.0)
                    got = {"obs": buf._obs_store.cpu().numpy().reshape(H + 1, n, 1, 26), "states": torch.stack([buf._stack_view(t, 1)[0] for t in range(H + 1)]).cpu().numpy(),
                           "act": buf.act_buf.cpu().numpy(), "rew": buf.rew_buf.cpu().numpy()[:, :, 0], "done": buf.done_buf.cpu().numpy()[:, :, 0],
                           "value": buf.value_buf.cpu().numpy()[:, :, 0], "logp": buf.logp_buf.cpu().numpy()[:, :, 0], "mu": buf.mu_buf.cpu().numpy(),
                           "sigma": buf.sigma_buf.cpu().numpy(), "timeout": buf.time_outs.cpu().numpy(), "last_value": last.cpu().numpy()[:, 0]}
                    got["state"], exp["state"] = env.get_state().cpu().numpy().view(np.uint32).copy(), orc.get_state().copy()
                    untracked = np.array([not env.tracks_rpy(e) for e in range(n)])   # (copter_rpy_old / _continuous are kept for flip envs only)
                    got["state"][20:26, untracked] = exp["state"][20:26, untracked] = 0
                    for k in exp:
                        a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(exp[k])
                        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
                        if a.dtype == np.float32:
                            same |= np.isnan(a) & np.isnan(b)
                        if not same.all():
                            raise SystemExit(f"case {i} FAILED vs the ORACLE: rollout {epoch} {k}: {int((~same).sum())} words differ, first at {np.argwhere(~same)[0].tolist()}: "
                                             f"H={H} {fuzz.describe(cfg)}")
                    vs_oracle += 1 if epoch == 1 else 0
                buf.compute_returns_and_advantage(last, normalize=n * H >= 2)
                snaps.append({k: getattr(buf, k).clone() for k in keys} | {"last": last.clone(), "tmo": buf.time_outs.clone(), "reset": env.reset_buf.clone(),
                                                                             "state": env.get_state().view(torch.int32).clone()})
            torch.cuda.synchronize()
            ran = bool(stamps[8:8 + 2 * min(H, 64)].min() > 0)
            assert ran == bool(fused), f"case {i}: persistent kernel ran = {ran} with fusion = {fused}: {fuzz.describe(cfg)}"
            outs.append(snaps)
        failed = []
        for epoch in range(2):
            for k in outs[0][epoch]:
                a, b = outs[0][epoch][k], outs[1][epoch][k]
                ai = a.view(torch.int32) if a.dtype == torch.float32 else a
                bi = b.view(torch.int32) if b.dtype == torch.float32 else b
                if not torch.equal(ai, bi):
                    ne = ai != bi
                    if a.dtype == torch.float32:
                        ne &= ~(torch.isnan(a) & torch.isnan(b))     # a NaN is a NaN: sign and payload of a diverged env's NaN are not part of the contract
                    if not bool(ne.any()):
                        nan_cases.add(i)
                        continue
                    bad = ne.nonzero()
                    failed.append(f"rollout {epoch} {k}: {bad.shape[0]} words differ, first at {bad[0].tolist()}: persistent {a[tuple(bad[0].tolist())].item()!r} "
                                  f"vs launch-per-step {b[tuple(bad[0].tolist())].item()!r}; envs {sorted(set(bad[:, 1 if bad.shape[1] > 1 and a.shape[0] != n else 0].tolist()))[:12]}")
        if failed:
            raise SystemExit(f"case {i} FAILED: H={H} {fuzz.describe(cfg)}\n  " + "\n  ".join(failed))
        resets += int(sum(float(s["done_buf"].sum()) for s in outs[0]))
        bootstraps += int(sum(int(s["tmo"].sum()) for s in outs[0]))
        print(f"case {i} ok ({time.time() - t0:.0f} s): H={H} {fuzz.describe(cfg)}", flush=True)
    if args.oracle:
        print(f"{vs_oracle} of the cases were also held against the CPU oracle's orc_rollout, bit for bit")
    print(f"rollout fuzz ok: {args.cases} cases, {resets} episode ends and {bootstraps} time-out bootstraps inside the rollouts; {len(nan_cases)} cases held a "
          f"diverged env whose NaNs differ in sign / payload between the two paths (same positions)")


if __name__ == "__main__":
    main()
