"""SURVEY.md 8f row N1 on the GPU: taco_step_rollout (the step kernel writing replay-buffer slots) and taco_gae
(PPOReplayBuffer.compute_returns_and_advantage, buffer_asymmetry.py:93-132) against the CPU oracle and the golden
vectors generated from the reference's own PPOReplayBuffer (tests/golden/gae.npz)."""
import os

import numpy as np
import pytest
import torch

from taco_amd import config
from util import assert_bits_equal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _buffer(n, H, len_obs=1, len_states=1, gamma=0.99, lam=0.95):
    from taco_amd.rollout import RolloutBuffer
    return RolloutBuffer(n, 26, len_obs, 26, len_states, 4, H, 4, gamma, lam, "cuda:0")


def test_gae_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "gae.npz"))
    H, N = g["rew"].shape
    buf = _buffer(N, H, gamma=float(g["gamma"]), lam=float(g["lam"]))
    buf.rew_buf.copy_(torch.from_numpy(g["rew"]).view(H, N, 1))
    buf.done_buf.copy_(torch.from_numpy(g["done_f32"]).view(H, N, 1))
    buf.value_buf.copy_(torch.from_numpy(g["value"]).view(H, N, 1))
    buf.compute_returns_and_advantage(torch.from_numpy(g["last_value"]).cuda().view(N, 1))
    assert_bits_equal(buf.ret_buf.cpu().numpy()[:, :, 0], g["ret"], "ret_buf vs the reference")
    # mean / std reductions: torch's summation order is not specified -> tolerance 1e-6 absolute (values are O(1))
    np.testing.assert_allclose(buf.adv_buf.cpu().numpy()[:, :, 0], g["adv_norm"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("H,N", [(1, 2), (7, 63), (32, 4096), (16, 70001)])
def test_gae_matches_oracle_bitwise(H, N):
    from oracle import oracle as O
    rng = np.random.default_rng(H * 1000 + N)
    rew = (rng.random((H, N)) * 0.02).astype(np.float32)
    done = (rng.random((H, N)) < 0.05).astype(np.float32)
    val = (rng.standard_normal((H, N)) * 0.3).astype(np.float32)
    last = (rng.standard_normal(N) * 0.3).astype(np.float32)
    adv_o, ret_o = O.gae(rew, done, val, last, 0.98, 0.9)
    buf = _buffer(N, H, gamma=0.98, lam=0.9)
    buf.rew_buf.copy_(torch.from_numpy(rew).view(H, N, 1))
    buf.done_buf.copy_(torch.from_numpy(done).view(H, N, 1))
    buf.value_buf.copy_(torch.from_numpy(val).view(H, N, 1))
    lv = torch.from_numpy(last).cuda()
    buf.compute_returns_and_advantage(lv, normalize=False)
    assert_bits_equal(buf.adv_buf.cpu().numpy()[:, :, 0], adv_o, "raw advantage")
    assert_bits_equal(buf.ret_buf.cpu().numpy()[:, :, 0], ret_o, "returns")
    buf.compute_returns_and_advantage(lv, normalize=True)
    a1 = buf.adv_buf.cpu().numpy()[:, :, 0].copy()
    np.testing.assert_allclose(a1, O.normalize_advantage(adv_o), rtol=0, atol=2e-6)
    assert abs(float(a1.astype(np.float64).mean())) < 1e-5 and abs(float(a1.astype(np.float64).std(ddof=1)) - 1) < 1e-5
    buf.compute_returns_and_advantage(lv, normalize=True)   # fixed reduction tree: bit-reproducible
    assert_bits_equal(buf.adv_buf.cpu().numpy()[:, :, 0], a1, "normalised advantage, second run")


@pytest.mark.parametrize("n,len_obs,len_states,task", [(63, 5, 3, "mix"), (256, 1, 1, "pos"), (130, 1, 4, "flip"), (10001, 2, 3, "mix"), (17003, 1, 5, "mix")])
def test_collect_equals_step_plus_store_and_oracle(n, len_obs, len_states, task):
    """The fused path (kernel writes slot t+1 / rew_buf[t] / done_buf[t]) == env.step_raw + PPOReplayBuffer.store copies
    == the oracle, bit for bit, across two rollouts (reset() carries the last stacks into slot 0).  n = 63 makes every
    slot base only 8-byte aligned."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    H = 6
    cfg = config.default_cfg(task, n, env_lenObservations=len_obs, env_lenStates=len_states, env_maxEpisodeLength=9, seed=5)
    fused, plain = FpvBase(cfg, copy_outputs=False), FpvBase(cfg, copy_outputs=False)
    orc = O.OracleEnv(config.flat_cfg(cfg), threads=4)
    bf, bp = _buffer(n, H, len_obs, len_states), _buffer(n, H, len_obs, len_states)
    rng = np.random.default_rng(3)
    for epoch in range(2):
        bf.reset(); bp.reset()
        for t in range(H):
            a = torch.from_numpy(np.clip(0.3 * rng.standard_normal((n, 4)), -1, 1).astype(np.float32)).cuda()
            obs_t, st_t = plain.obs_buf.clone(), plain.states_buf.clone()   # what the reference calls obs / states
            assert torch.equal(bf.next_obs, obs_t) and torch.equal(bf.next_states, st_t)
            rew, done, tmo = bf.collect(fused, a)
            plain.step_raw(a)
            bp.store(obs_t, st_t, a, plain.rew_buf, torch.zeros(n, device="cuda"), plain.reset_buf, torch.zeros(n, 1, device="cuda"), a, a)
            orc.step(a.cpu().numpy())
            assert_bits_equal(rew.cpu().numpy(), orc.rew_buf, f"epoch {epoch} step {t} reward")
            assert_bits_equal(done.cpu().numpy(), orc.reset_buf, "done")
            assert_bits_equal(tmo.cpu().numpy().astype(np.uint8), orc.timeout_buf, "time-outs")
            assert_bits_equal(bf.next_obs.cpu().numpy(), orc.obs_buf, "slot t+1 obs vs oracle")
            assert_bits_equal(bf.next_states.cpu().numpy(), orc.states_buf, "slot t+1 states vs oracle")
        for name in ("obs_buf", "states_buf", "act_buf", "rew_buf", "done_buf"):
            assert torch.equal(getattr(bf, name), getattr(bp, name)), name
        assert bf.done_buf.sum() > 0 or epoch == 0
        assert_bits_equal(fused.get_state().cpu().numpy().view(np.uint32), plain.get_state().cpu().numpy().view(np.uint32), "env state")
    with pytest.raises(AssertionError):
        bf.collect(fused, a)       # buffer_asymmetry.py:50-51 "Rollout buffer overflow"


def test_collect_refuses_what_it_cannot_do_exactly():
    from taco_amd._lib import TacoError
    from taco_amd.vec_env import FpvPos
    env = FpvPos(config.default_cfg("pos", 64, env_clipObservations=5.0))
    buf = _buffer(64, 4)
    with pytest.raises(TacoError):
        buf.collect(env, env.zero_actions().cuda())
    env2 = FpvPos(config.default_cfg("pos", 32))
    with pytest.raises(ValueError):
        buf.collect(env2, env2.zero_actions().cuda())
    assert len(buf.batch_idx_generator()) == 4 and sorted(sum(buf.batch_idx_generator(), [])) == list(range(64 * 4))


@pytest.mark.parametrize("n,task,len_states,arch,exact", [(300, "pos", 5, ([48, 32], 32, [40]), False), (96, "mix", 3, ([48, 32], 32, [40]), False),
                                                           (333, "mix", 5, ([128, 128, 128], 128, [128, 128]), True),   # the batched critic's form
                                                           (333, "mix", 5, ([128, 128, 128], 128, [128, 128]), False)])
def test_rollout_run_equals_the_python_loop(n, task, len_states, arch, exact):
    """taco_rollout_run (one C call: the actor + the env step per step, the critic batched over all H + 1 slots afterwards) == the act() /
    collect() / bootstrap loop of ppo_asymmetry.py:308-342, bit for bit -- for the documented architecture with the exact LSTM cell
    (TACO_P_EXACT_CELL); with the default cell the critic's values, and what GAE derives from them, agree to 2e-6 / 1e-4 and everything
    else (trajectories, actions, log-probs, dones, time-outs, the env's state) stays bit-identical."""
    from taco_amd import policy as P
    from taco_amd.vec_env import FpvBase
    import test_policy_gpu as TP
    H = 12
    rng = np.random.default_rng(5)
    sd = TP._random_policy(rng, 1, len_states, *arch)
    cfg = config.default_cfg(task, n, env_lenStates=len_states, env_maxEpisodeLength=7, seed=3)
    outs = []
    for mode in ("run", "loop"):
        env = FpvBase(cfg, copy_outputs=False)
        pol = P.ActorCritic(sd, 1, len_states, seed=21, exact_critic=exact)
        buf = _buffer(n, H, 1, len_states)
        for epoch in range(2):
            buf.reset()
            if mode == "run":
                last = buf.run(env, pol)
                tmo = buf.time_outs.clone()
            else:
                tmo = torch.zeros(H, n, dtype=torch.uint8, device="cuda")
                for t in range(H):
                    action, logp, value, mu, sigma = pol.act(buf.next_obs, buf.next_states)
                    rew, dones, time_outs = buf.collect(env, torch.clip(action, -1, 1), logp, value, mu, sigma, act=action)
                    tmo[t] = time_outs.to(torch.uint8)
                    ids = (time_outs & (dones != 0)).nonzero().squeeze(-1)
                    if ids.numel():
                        buf.add_timeout_bootstrap(t, ids, value[ids])
                last = pol.act(buf.next_obs, buf.next_states, deterministic=True)[2]
            buf.compute_returns_and_advantage(last)
        assert tmo.sum() > 0 and buf.done_buf.sum() > tmo.sum() * 0   # short episodes: time-outs happened, the bootstrap path ran
        outs.append({k: getattr(buf, k).clone() for k in ("obs_buf", "states_buf", "act_buf", "rew_buf", "done_buf", "value_buf", "logp_buf",
                                                            "mu_buf", "sigma_buf", "ret_buf", "adv_buf")} | {"last": last.clone(), "tmo": tmo,
                                                                                                             "state": env.get_state().view(torch.int32)})
    batched_fast = arch[1] == 128 and not exact          # only the documented architecture runs the batched critic kernels
    loose = {"value_buf": 2e-6, "last": 2e-6, "rew_buf": 2e-6, "ret_buf": 1e-4, "adv_buf": 1e-4} if batched_fast else {}
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        if k in loose:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=loose[k], err_msg=k)
        elif a.dtype == torch.float32:
            assert_bits_equal(a.cpu().numpy(), b.cpu().numpy(), k)
        else:
            assert torch.equal(a, b), k


def test_rollout_run_is_graph_capturable_and_replays_are_the_next_rollouts():
    """taco_rollout_run captured into ONE HIP graph (2 H + 3 kernels + the clock advances): the env's clock and the actor's noise counter
    come from the device-resident step word, so replay k is rollout k -- fresh action noise, episodes continuing -- bit-identical to k eager
    calls (every replay-buffer array, the env's whole state)."""
    from taco_amd import policy as P
    from taco_amd.vec_env import FpvBase
    import test_policy_gpu as TP
    n, H, T = 333, 9, 5
    rng = np.random.default_rng(11)
    sd = TP._random_policy(rng, 1, T, [128, 128, 128], 128, [128, 128])
    cfg = config.default_cfg("mix", n, env_lenStates=T, env_maxEpisodeLength=23, seed=8, observation_noise=True, rotor_noise=True)
    keys = ("obs_buf", "states_buf", "act_buf", "rew_buf", "done_buf", "value_buf", "logp_buf", "mu_buf", "sigma_buf")

    def snapshot(buf, env, last):
        return {k: getattr(buf, k).clone() for k in keys} | {"last": last.clone(), "tmo": buf.time_outs.clone()}

    # eager: four rollouts (the buffer carries the final stacks of one into slot 0 of the next, as PPO.run does)
    env, pol, buf = FpvBase(cfg, copy_outputs=False), P.ActorCritic(sd, 1, T, seed=5), _buffer(n, H, 1, T)
    eager = []
    for k in range(4):
        buf.reset()
        eager.append(snapshot(buf, env, buf.run(env, pol)))
        if k == 2:
            eager_state = env.get_state().view(torch.int32).clone()

    # captured: rollout 0 eagerly on a side stream (allocations, warm-up), then rollout 1 captured and replayed twice = rollouts 1 and 2
    env, pol, buf = FpvBase(cfg, copy_outputs=False), P.ActorCritic(sd, 1, T, seed=5), _buffer(n, H, 1, T)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        buf.reset()
        got0 = snapshot(buf, env, buf.run(env, pol))
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            buf.reset()
            last = buf.run(env, pol)
    for k in (1, 2):
        g.replay()
        torch.cuda.synchronize()
        got = snapshot(buf, env, last)
        for name in got:
            a, b = got[name], eager[k][name]
            if a.dtype == torch.float32:
                assert_bits_equal(a.cpu().numpy(), b.cpu().numpy(), f"replay {k} {name}")
            else:
                assert torch.equal(a, b), f"replay {k} {name}"
    for name in got0:
        assert torch.equal(got0[name].view(torch.int32) if got0[name].dtype == torch.float32 else got0[name],
                           eager[0][name].view(torch.int32) if eager[0][name].dtype == torch.float32 else eager[0][name]), name
    assert torch.equal(env.get_state().view(torch.int32), eager_state)
    assert env.step_count == 3 * H
    # ... and back to EAGER sampling after the replays: the Python-side noise counter (policy.calls) never saw the 2 H counters the replays
    # consumed; run() re-derives it from the env's clock, so rollout 3 draws fresh noise = the eager twin's rollout 3
    assert pol.calls == 2 * H
    buf.reset()
    got3 = snapshot(buf, env, buf.run(env, pol))
    assert pol.calls == 4 * H
    for name in got3:
        a, b = got3[name], eager[3][name]
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b), f"eager rollout after replays: {name}"
    env.check()


@pytest.mark.parametrize("n,task,len_states,kw", [
    (4096, "mix", 5, dict(observation_noise=True, rotor_noise=True, ramdom_delay_time=True, ramdom_deploy_time=True, random_rotordynamic_coe=True)),  # config 5's flags
    (333, "pos", 1, dict()),                                  # ragged last workgroup, one frame
    (1000, "flip", 3, dict(delay_time=5)),                    # 2..9 pending slots: the action is patched into the slot table mid-loop
    (700, "rotate", 1, dict(delay_time=1, ramdom_delay_time=True)),   # 0..1 pending slots (and up to 4 by the random delay): waits for the actor up front / patches at substep 1
    (500, "rotate", 2, dict(delay_time=85, ramdom_deploy_time=True)),   # the delay line's overflow regime: DENSE envs take the early path
    (600, "mix", 5, dict(battery_consumption=False)),         # no battery server
])
@pytest.mark.parametrize("fused_form", ["quad", "lane"])
def test_persistent_rollout_kernel_equals_the_launch_per_step_path(n, task, len_states, kw, fused_form):
    """taco_rollout_run with the persistent actor + step kernel (one workgroup owns 16 envs for the whole horizon, the actor's MFMAs run
    under the substeps) against the same call with fusion switched off (2 H launches): every replay-buffer array, the env's whole state and
    reset_buf bit for bit, over three rollouts with episode ends; and the hardware really places the eight wavefronts two per SIMD in
    launch order, which is what keeps the actor off the step wavefront's SIMD.  fused_form: "quad" = 16 envs per workgroup (what the host picks up to
    8 192 envs), "lane" = round 6's one-lane form, 64 envs per workgroup, the actor over four row tiles (what it picks above) -- both forced here."""
    from taco_amd import policy as P
    from taco_amd.vec_env import FpvBase
    import test_policy_gpu as TP
    H = 16
    rng = np.random.default_rng(7)
    sd = TP._random_policy(rng, 1, len_states, [128, 128, 128], 128, [128, 128])
    cfg = config.default_cfg(task, n, env_lenStates=len_states, env_maxEpisodeLength=21, seed=4, **kw)
    keys = ("obs_buf", "states_buf", "act_buf", "rew_buf", "done_buf", "value_buf", "logp_buf", "mu_buf", "sigma_buf", "ret_buf", "adv_buf")
    outs = []
    for fused in (True, False):
        env = FpvBase(cfg, copy_outputs=False)
        env.set_rollout_fusion(fused_form if fused else False)
        stamps = torch.zeros(136 + (n + 15) // 16, dtype=torch.int64, device="cuda")
        env.bind_rollout_stamps(stamps)
        pol = P.ActorCritic(sd, 1, len_states, seed=21)
        buf = _buffer(n, H, 1, len_states)
        snaps = []
        for epoch in range(3):
            buf.reset()
            last = buf.run(env, pol)
            buf.compute_returns_and_advantage(last)
            snaps.append({k: getattr(buf, k).clone() for k in keys} | {"last": last.clone(), "tmo": buf.time_outs.clone(), "reset": env.reset_buf.clone(),
                                                                         "state": env.get_state().view(torch.int32).clone()})
        env.check()
        assert env.step_count == 3 * H and pol.calls == 3 * H
        if fused:
            simd = stamps[:8].cpu().tolist()
            assert simd[0] == simd[4] and simd[1] == simd[5] and simd[2] == simd[6] and simd[3] == simd[7] and len(set(simd[:4])) == 4, f"wavefront -> SIMD placement {simd}"
            assert stamps[8:8 + 2 * H].min() > 0, "the persistent kernel did not run"
        else:
            assert stamps.abs().sum() == 0, "fusion was switched off, yet the persistent kernel ran"
        outs.append(snaps)
    assert sum(float(s["done_buf"].sum()) for s in outs[0]) > 0 and sum(int(s["tmo"].sum()) for s in outs[0]) > 0
    for epoch in range(3):
        for k in outs[0][epoch]:
            a, b = outs[0][epoch][k], outs[1][epoch][k]
            if a.dtype == torch.float32:
                assert_bits_equal(a.cpu().numpy(), b.cpu().numpy(), f"rollout {epoch} {k}")
            else:
                assert torch.equal(a, b), f"rollout {epoch} {k}"


@pytest.mark.parametrize("n,task,len_states,fused,kw", [
    # config 5's flags (taco_amd/config.py baseline_config(4)), ragged last workgroup: the PERSISTENT actor + step kernel
    (1000, "mix", 5, True, dict(random_rotordynamic_coe=True, random_rotor_response=True, random_aerodynamic_coe=True, observation_noise=True,
                                rotor_noise=True, ramdom_delay_time=True, ramdom_deploy_time=True)),
    (333, "flip", 3, True, dict(delay_time=5)),        # the late-patch path of the persistent kernel (2..9 pending slots)
    (700, "mix", 5, False, dict(observation_noise=True, rotor_noise=True)),   # the launch-per-step chain
    # round 6: the persistent kernel's ONE-LANE form (64 envs per workgroup), forced at small env counts: config 5's flags, the late patch, no battery
    (1000, "mix", 5, "lane", dict(random_rotordynamic_coe=True, random_rotor_response=True, random_aerodynamic_coe=True, observation_noise=True,
                                  rotor_noise=True, ramdom_delay_time=True, ramdom_deploy_time=True)),
    (333, "flip", 3, "lane", dict(delay_time=5)),
    (650, "rotate", 2, "lane", dict(delay_time=1, ramdom_delay_time=True, battery_consumption=False)),
])
def test_rollout_run_equals_the_oracle_rollout(n, task, len_states, fused, kw):
    """taco_rollout_run DIRECTLY against the CPU oracle's orc_rollout (actor forward + clip + env step + store + time-out bootstrap per step,
    the final value, then GAE; oracle/taco_policy_oracle.c, pinned piece by piece by the reference's own vectors): every replay-buffer array,
    the returns / raw advantages, the env's whole state and reset_buf bit for bit, over two rollouts with episode ends and time-outs.  (The
    exact LSTM cell -- TACO_P_EXACT_CELL -- is the one that equals the oracle op for op; the default cell is held to 2e-6 in test_policy_gpu.)"""
    from oracle import oracle as O
    from taco_amd import policy as P
    from taco_amd.vec_env import FpvBase
    import test_policy_gpu as TP
    H, gamma, lam = 16, 0.99, 0.95
    rng = np.random.default_rng(17)
    arch = ([128, 128, 128], 128, [128, 128])
    sd = TP._random_policy(rng, 1, len_states, *arch)
    cfg = config.default_cfg(task, n, env_lenStates=len_states, env_maxEpisodeLength=19, seed=6, **kw)
    env = FpvBase(cfg, copy_outputs=False)
    env.set_rollout_fusion(fused)
    stamps = torch.zeros(136 + (n + 15) // 16, dtype=torch.int64, device="cuda")
    env.bind_rollout_stamps(stamps)
    pol = P.ActorCritic(sd, 1, len_states, seed=31, exact_critic=True)
    buf = _buffer(n, H, 1, len_states, gamma, lam)
    orc = O.OracleEnv(config.flat_cfg(cfg), threads=8)
    oc = O.policy_cfg(1, len_states, *arch)
    blob = P.pack_state_dict(pol.cfg, sd)
    n_tmo = n_done = 0
    for ro in range(2):
        buf.reset()
        last = buf.run(env, pol)
        buf.compute_returns_and_advantage(last, normalize=False)
        exp = O.rollout(orc, oc, blob, H, 31, ro * H, gamma, orc.obs_buf.copy(), orc.states_buf.copy())
        adv_o, ret_o = O.gae(exp["rew"], exp["done"], exp["value"], exp["last_value"], gamma, lam)
        got = {"obs": buf._obs_store.cpu().numpy().reshape(H + 1, n, 1, 26),
               "states": torch.stack([buf._stack_view(t, 1)[0] for t in range(H + 1)]).cpu().numpy(),
               "act": buf.act_buf.cpu().numpy(), "rew": buf.rew_buf.cpu().numpy()[:, :, 0], "done": buf.done_buf.cpu().numpy()[:, :, 0],
               "value": buf.value_buf.cpu().numpy()[:, :, 0], "logp": buf.logp_buf.cpu().numpy()[:, :, 0], "mu": buf.mu_buf.cpu().numpy(),
               "sigma": buf.sigma_buf.cpu().numpy(), "timeout": buf.time_outs.cpu().numpy(), "last_value": last.cpu().numpy()[:, 0]}
        for k in exp:
            assert_bits_equal(got[k], exp[k], f"rollout {ro} {k}")
        assert_bits_equal(buf.adv_buf.cpu().numpy()[:, :, 0], adv_o, f"rollout {ro} raw advantage")
        assert_bits_equal(buf.ret_buf.cpu().numpy()[:, :, 0], ret_o, f"rollout {ro} returns")
        gs, es = env.get_state().cpu().numpy().view(np.uint32).copy(), orc.get_state().copy()
        untracked = np.array([not env.tracks_rpy(i) for i in range(n)])   # (copter_rpy_old / _continuous: kept for flip envs only, nothing else reads them)
        gs[20:26, untracked] = es[20:26, untracked] = 0
        assert_bits_equal(gs, es, f"rollout {ro} env state")
        assert np.array_equal(env.reset_buf.cpu().numpy(), orc.reset_buf)
        n_tmo += int(exp["timeout"].sum()); n_done += int(exp["done"].sum())
    assert n_tmo > 0 and n_done >= n_tmo and (task != "mix" or n_done > n_tmo)
    assert (stamps[8:8 + 2 * H].min() > 0) == bool(fused), "the persistent kernel ran / did not run as asked"
    env.check()
