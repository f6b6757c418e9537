"""The VecTask surface of taco_amd.vec_env on a real GPU (what ppo_asymmetry.py / train_fpv_asymmetry_ppo.py touch),
and full-size parity: at BASELINE.json's sizes the oracle re-simulates random SLICES of global env ids (envs are
independent and keyed by global id) and must agree bit-for-bit with the same rows of the full-size GPU run."""
import math
import os

import numpy as np
import pytest
import torch

from taco_amd import config
from util import assert_bits_equal

pytestmark = pytest.mark.gpu


def test_vectask_surface_matches_the_reference_contract():
    from taco_amd.vec_env import isaacgym_task_map
    cfg = config.default_cfg("pos", 128, env_lenStates=5, env_clipObservations=0.75, env_maxEpisodeLength=50)
    env = isaacgym_task_map["Fpv_pos"](cfg, "cuda:0", "cuda:0", -1, True, False, False)
    assert (env.num_envs, env.num_obs, env.len_obs, env.num_states, env.len_states, env.num_acts) == (128, 26, 1, 26, 5, 4)
    assert env.observation_space.shape == (1, 26) and env.state_space.shape == (5, 26) and env.action_space.shape == (4,)
    assert (env.action_space.low == -1).all() and env.max_episode_length == 50 and cfg["env"]["numObservations"] == 26
    d = env.reset()                                   # VT:352-361: zero buffers, nothing is reset yet
    assert set(d) == {"obs", "states"} and not d["obs"].any() and d["obs"].shape == (128, 1, 26) and d["states"].shape == (128, 5, 26)
    assert env.reset_buf.dtype == torch.long and env.reset_buf.all()
    a = env.zero_actions()
    obs, rew, done, info = env.step(a)
    assert obs["obs"].dtype == torch.float32 and rew.shape == (128,) and done.dtype == torch.long and info["time_outs"].dtype == torch.bool
    assert obs["obs"].abs().max() <= 0.75 and env.obs_buf.abs().max() > 0.75      # returned copy is clamped, obs_buf is not
    assert obs["obs"].data_ptr() != env.obs_buf.data_ptr()
    assert (env.progress_buf == 1).all() and not done.any()
    # frame stack: newest frame last, older frames shift towards index 0
    s1 = env.states_buf.clone()
    env.step(a)
    assert torch.equal(env.states_buf[:, 3], s1[:, 4]) and not torch.equal(env.states_buf[:, 4], s1[:, 4])
    # done -> re-initialised at the START of the next step; time_outs only together with done
    saw_timeout = False
    for t in range(60):
        obs, rew, done, info = env.step(a)
        assert not (info["time_outs"] & (done == 0)).any()
        saw_timeout |= bool(info["time_outs"].any())
        prog = env.progress_buf
        assert (prog[done != 0] >= 1).all() and (prog <= 50).all()
    assert saw_timeout
    # writing reset_buf marks envs for the next step (or for reset_done()); reset_idx(ids) re-initialises NOW, like the reference's (FA:475-517)
    env.reset_buf[[3, 5]] = 1
    obs, ids = env.reset_done()
    assert {3, 5} <= set(ids.tolist())
    env.step(a)
    assert (env.progress_buf[[3, 5]] == 1).all()
    before, n0 = env.get_state()[:13, 7].clone(), env.step_count
    env.reset_buf[9] = 1
    env.reset_idx(torch.tensor([7, 9], device="cuda"))
    assert (env.progress_buf[[7, 9]] == 0).all() and not env.reset_buf[[7, 9]].any() and not torch.equal(env.get_state()[:13, 7], before)
    assert env.step_count == n0 == 63 and torch.equal(env.randomize_buf, torch.full((128,), 63, device="cuda"))   # (VT:252, FA:376: += 1 per step)
    env.step(a)
    assert (env.progress_buf[[7, 9]] == 1).all()
    # difficulty is writable and read back (ppo_asymmetry.py:173-175)
    env.difficulty = 0.25
    assert env.difficulty == 0.25 and cfg["difficulty"] == 0.25
    with pytest.raises(ValueError):
        env.step(torch.zeros(7, 4, device="cuda"))


def test_step_is_stream_ordered_and_async():
    """taco_step only enqueues: steps issued on a side stream interleave correctly with torch ops on that stream."""
    from taco_amd.vec_env import FpvPos
    cfg = config.baseline_config(1, num_envs=512)
    ref = FpvPos(cfg, copy_outputs=False)
    env = FpvPos(config.baseline_config(1, num_envs=512), copy_outputs=False)
    acts = (0.3 * torch.randn(30, 512, 4, device="cuda")).clamp(-1, 1)
    for t in range(30):
        ref.step_raw(acts[t])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for t in range(30):
            a = acts[t] * 1.0            # a torch kernel on the same stream produces the action
            env.step_raw(a)
    s.synchronize()
    assert torch.equal(ref.get_state().view(torch.int32), env.get_state().view(torch.int32))


@pytest.mark.parametrize("idx,steps", [(2, 150), (3, 150), (4, 120)])
def test_full_size_parity_on_random_slices(idx, steps):
    """BASELINE configs[2..4] at FULL size (16 384 / 65 536 / 262 144 envs): four random slices of 192 global env ids are
    re-simulated by the oracle (env_offset / num_envs_global) and compared bit-for-bit every 10 steps."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    cfg = config.baseline_config(idx)
    flat = config.flat_cfg(cfg)
    n = flat["num_envs"]
    env = FpvBase(cfg, copy_outputs=False)
    rng = np.random.default_rng(100 + idx)
    width = 192
    n1, n2 = int(n / 3 * 1), int(n / 3 * 2)
    starts = sorted({0, n - width, n1 - width // 2, n2 - width // 2} if idx == 4 else set(int(x) for x in rng.integers(0, n - width, 4)))
    orcs = [O.OracleEnv(dict(flat, num_envs=width, env_offset=lo, num_envs_global=n), threads=8) for lo in starts]
    g = torch.Generator().manual_seed(idx)
    for t in range(steps):
        a = (0.3 * torch.randn((n, 4), generator=g)).clamp(-1, 1)
        a[:, 0] = (a[:, 0] - 0.2).clamp(-1, 1)
        ad = a.cuda()
        env.step_raw(ad)
        an = a.numpy()
        for lo, orc in zip(starts, orcs):
            orc.step(an[lo:lo + width])
        if t % 10 == 9 or t == steps - 1:
            obs = env.obs_buf.cpu().numpy(); st = env.states_buf.cpu().numpy()
            rew = env.rew_buf.cpu().numpy(); done = env.reset_buf.cpu().numpy(); tmo = env.timeout_buf.cpu().numpy().astype(np.uint8)
            for lo, orc in zip(starts, orcs):
                what = f"config {idx} step {t} slice {lo}"
                assert_bits_equal(obs[lo:lo + width], orc.obs_buf, what + " obs")
                assert_bits_equal(st[lo:lo + width], orc.states_buf, what + " states")
                assert_bits_equal(rew[lo:lo + width], orc.rew_buf, what + " rew")
                assert_bits_equal(done[lo:lo + width], orc.reset_buf, what + " done")
                assert_bits_equal(tmo[lo:lo + width], orc.timeout_buf, what + " time_outs")
    # size-independent invariants on the whole batch
    blob = env.get_state()
    q = blob[3:7]
    fin = torch.isfinite(q).all(0)
    assert fin.float().mean() > 0.999
    assert ((q[:, fin] ** 2).sum(0) - 1).abs().max() < 1e-5
    prog = blob[65].view(torch.int32)
    assert (prog >= 1).all() and (prog <= flat["max_episode_length"]).all()


# (baseline_config index, envs per rank, steps, the kernel form the product picks there): BASELINE configs 2 ... 5 at the per-rank shapes of
# N = 1, 2, 4, 8 ranks (DESIGN section 8 lists them): pos 4 096 / N; rotate 16 384 / N; flip 65 536 / N; mix 262 144 / N with every flag and 5 frames
PER_RANK_SHAPES = [(2, 16384, 60, "quad_served"), (3, 16384, 60, "quad_served"), (4, 32768, 40, "lane_roles"), (1, 65536, 25, "lane_throughput"),
                   (1, 512, 60, "quad_roles"), (2, 8192, 40, "quad_roles"), (2, 2048, 60, "quad_roles"), (3, 32768, 25, "lane_roles"),
                   (3, 8192, 40, "quad_roles"), (4, 65536, 16, "lane_roles"), (4, 131072, 10, "lane_throughput")]


@pytest.mark.parametrize("idx,n,steps,form", PER_RANK_SHAPES)
def test_whole_batch_parity_at_the_per_rank_shapes(idx, n, steps, form):
    """SURVEY 8(d)'s configs 3 / 4 / 5 at their per-GPU shapes (rotate 16 384, flip 16 384, mix 32 768 with every flag and 5 state frames),
    the first throughput-form size, and the shapes the same totals give on 2 / 4 / 8 ranks, in the kernel form the product picks there:
    EVERY env against the oracle (all host threads), outputs each few steps and the whole state at the end, bit for bit."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    cfg = config.baseline_config(idx, num_envs=n)
    flat = config.flat_cfg(cfg)
    env = FpvBase(cfg, copy_outputs=False)
    assert env.kernel_form == form
    orc = O.OracleEnv(flat, threads=min(32, os.cpu_count() or 8))
    g = torch.Generator().manual_seed(40 + idx)
    for t in range(steps):
        a = (0.3 * torch.randn((n, 4), generator=g)).clamp(-1, 1)
        a[:, 0] = (a[:, 0] - 0.3).clamp(-1, 1)
        env.step_raw(a.cuda())
        orc.step(a.numpy())
        if t % 8 == 7 or t == steps - 1:
            what = f"config {idx + 1} step {t}"
            assert_bits_equal(env.obs_buf.cpu().numpy(), orc.obs_buf, what + " obs")
            assert_bits_equal(env.states_buf.cpu().numpy(), orc.states_buf, what + " states")
            assert_bits_equal(env.rew_buf.cpu().numpy(), orc.rew_buf, what + " rew")
            assert_bits_equal(env.reset_buf.cpu().numpy(), orc.reset_buf, what + " done")
    gb, ob = env.get_state().cpu().numpy(), orc.get_state().view(np.float32)
    flip = (np.arange(n) >= int(n / 3 * 2)) if flat["task_mode"] == "mix" else np.full(n, flat["task_mode"] == "flip")
    assert_bits_equal(gb[:20], ob[:20], "state fields 0..19")
    assert_bits_equal(gb[26:], ob[26:], "state fields 26.. and the delay line")
    assert_bits_equal(gb[20:26][:, flip], ob[20:26][:, flip], "rpy of the flip envs")


def test_maximum_size_two_million_envs_with_stacked_states():
    """The largest launch taco_create accepts (2 000 000 envs), with the documented 5-frame state stack: slices at the first, a middle
    and the last envs are re-simulated by the oracle and compared bit for bit; the stack of every env shifts by one frame."""
    from oracle import oracle as O
    from taco_amd._lib import TacoError
    from taco_amd.vec_env import FpvBase
    n, width, steps = 2_000_000, 64, 7
    cfg = config.default_cfg("mix", n, env_lenStates=5, seed=2)
    flat = config.flat_cfg(cfg)
    env = FpvBase(cfg, copy_outputs=False)
    starts = [0, 1_333_300, n - width]
    orcs = [O.OracleEnv(dict(flat, num_envs=width, env_offset=lo, num_envs_global=n), threads=4) for lo in starts]
    g = torch.Generator().manual_seed(9)
    prev = None
    for t in range(steps):
        a = (0.3 * torch.randn((n, 4), generator=g)).clamp(-1, 1)
        env.step_raw(a.cuda())
        for lo, orc in zip(starts, orcs):
            orc.step(a[lo:lo + width].numpy())
            assert_bits_equal(env.states_buf[lo:lo + width].cpu().numpy(), orc.states_buf, f"step {t} slice {lo} states")
            assert_bits_equal(env.obs_buf[lo:lo + width].cpu().numpy(), orc.obs_buf, f"step {t} slice {lo} obs")
            assert_bits_equal(env.rew_buf[lo:lo + width].cpu().numpy(), orc.rew_buf, f"step {t} slice {lo} rew")
        if prev is not None:   # size-independent property over ALL envs: frames 1..4 of the old stack are frames 0..3 of the new one
            assert torch.equal(env.states_buf[:, :4].view(torch.int32), prev[:, 1:].view(torch.int32))
        prev = env.states_buf.clone()
    del env, prev
    torch.cuda.empty_cache()
    with pytest.raises(TacoError):
        FpvBase(config.default_cfg("pos", 2_000_001), copy_outputs=False)


def test_checkpoint_file_roundtrip_and_recorder(tmp_path):
    """N2: save -> load into a fresh env -> both continue bit-identically; the recorder pulls env 0 with the reference's keys"""
    from taco_amd import checkpoint
    from taco_amd.recorder import EpisodeRecorder
    from taco_amd.vec_env import FpvFlip
    mk = lambda: FpvFlip(config.default_cfg("flip", 300, observation_noise=True, rotor_noise=True, env_lenStates=3, env_maxEpisodeLength=40), copy_outputs=False)
    a = mk()
    rec = EpisodeRecorder(str(tmp_path / "rec"))
    acts = (0.3 * torch.randn(70, 300, 4, device="cuda")).clamp(-1, 1)
    for t in range(33):
        a.step_raw(acts[t]); rec.record(a)
    checkpoint.save(a, str(tmp_path / "ck.pt"))
    b = mk()
    checkpoint.load(b, str(tmp_path / "ck.pt"))
    for t in range(33, 70):
        a.step_raw(acts[t]); b.step_raw(acts[t]); rec.record(a)
        assert torch.equal(a.obs_buf.view(torch.int32), b.obs_buf.view(torch.int32)) and torch.equal(a.reset_buf, b.reset_buf)
    assert torch.equal(a.get_state().view(torch.int32), b.get_state().view(torch.int32))
    with pytest.raises(ValueError):
        checkpoint.load(FpvFlip(config.default_cfg("flip", 64), copy_outputs=False), str(tmp_path / "ck.pt"))
    assert rec.dump_index >= 1 and {"copter_pos0.npy", "observations0.csv", "actions_old0.npy"} <= set(__import__("os").listdir(tmp_path / "rec"))
    pos = np.load(tmp_path / "rec" / "copter_pos0.npy")
    assert pos.ndim == 2 and pos.shape[1] == 3


def test_recorder_logs_the_live_rpy_continuous_of_a_pos_env(tmp_path):
    """FA:346, :662: the reference logs the LIVE copter_rpy_continuous of env 0 whatever the task.  The kernel maintains it for flip envs
    only unless cfg['record_flag'] is set: the recorder refuses an env that does not keep it, and with the flag its rows equal the oracle's."""
    from oracle import oracle as O
    from taco_amd._lib import TacoError
    from taco_amd.recorder import EpisodeRecorder
    from taco_amd.vec_env import FpvPos
    with pytest.raises(TacoError, match="record_flag"):
        EpisodeRecorder(str(tmp_path / "x")).record(FpvPos(config.default_cfg("pos", 64), copy_outputs=False))
    cfg = config.default_cfg("pos", 64, record_flag=True, record_path=str(tmp_path / "rec"), env_maxEpisodeLength=60)
    env, orc = FpvPos(cfg, copy_outputs=False), O.OracleEnv(config.flat_cfg(cfg))
    rec = EpisodeRecorder(cfg["record_path"])
    g = torch.Generator().manual_seed(1)
    exp = []
    for t in range(100):
        a = (0.5 * torch.randn(64, 4, generator=g)).clamp(-1, 1)
        a[:, 0] -= 0.2
        env.step_raw(a.cuda()); orc.step(a.numpy()); rec.record(env)
        exp.append(orc.get_state().view(np.float32)[23:26, 0].copy())
    rec.dump_buffer()
    got = np.concatenate([np.load(tmp_path / "rec" / f"copter_rpy_continuous{k}.npy") for k in range(rec.dump_index)])
    assert np.array_equal(got, np.array(exp)) and np.ptp(got, axis=0).max() > 0.5, "rows 23..25 of a pos env must be live, not the reset-time value"


def test_checkpoint_restores_across_shardings():
    """A checkpoint of one 600-env run restores two shards of it (250 + 350 envs, as two ranks would hold them), and the two shards'
    checkpoints merge()d restore a 600-env run again -- everything continues bit-identically (every per-env word is a blob column and
    every random stream is keyed by the global env id)."""
    from taco_amd import checkpoint
    from taco_amd.vec_env import FpvBase
    cfg = config.default_cfg("mix", 600, observation_noise=True, rotor_noise=True, ramdom_deploy_time=True, env_lenStates=3, env_maxEpisodeLength=45, seed=5)
    whole = FpvBase(cfg, copy_outputs=False)
    acts = (0.3 * torch.randn(90, 600, 4, device="cuda") + torch.tensor([-0.45, 0, 0, 0], device="cuda")).clamp(-1, 1)
    for t in range(30):
        whole.step_raw(acts[t])
    sd = checkpoint.state_dict(whole)
    parts = [FpvBase(cfg, copy_outputs=False, env_offset=lo, num_envs_local=n) for lo, n in ((0, 250), (250, 350))]
    for p in parts:
        checkpoint.load_state_dict(p, sd)
    for t in range(30, 60):
        whole.step_raw(acts[t])
        for p, (lo, n) in zip(parts, ((0, 250), (250, 350))):
            p.step_raw(acts[t, lo:lo + n].contiguous())
            assert torch.equal(p.obs_buf.view(torch.int32), whole.obs_buf[lo:lo + n].view(torch.int32)), (t, lo)
            assert torch.equal(p.rew_buf.view(torch.int32), whole.rew_buf[lo:lo + n].view(torch.int32)) and torch.equal(p.reset_buf, whole.reset_buf[lo:lo + n])
    merged = checkpoint.merge([checkpoint.state_dict(parts[1]), checkpoint.state_dict(parts[0])])
    again = FpvBase(cfg, copy_outputs=False)
    checkpoint.load_state_dict(again, merged)
    for t in range(60, 90):
        whole.step_raw(acts[t]); again.step_raw(acts[t])
    assert torch.equal(again.get_state().view(torch.int32), whole.get_state().view(torch.int32))
    assert torch.equal(again.obs_buf.view(torch.int32), whole.obs_buf.view(torch.int32)) and whole.reset_buf.sum() >= 0
    with pytest.raises(ValueError, match="merge"):
        checkpoint.load_state_dict(again, checkpoint.state_dict(parts[0]))     # one shard does not cover the whole run
    with pytest.raises(ValueError, match="adjoin"):
        checkpoint.merge([checkpoint.state_dict(parts[1])] * 2)
    with pytest.raises(ValueError, match="does not fit"):
        again.set_state(torch.zeros(3, 600))


def test_c_abi_consumer_without_torch():
    """examples/c_api_demo.cpp (plain HIP runtime + include/taco_env.h, built by __graft_entry__.build()) steps the env
    through the C ABI alone and checks the error path; it exits 0 when its own checks pass."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "c_api_demo")
    if not os.path.exists(exe):
        pytest.skip("examples/c_api_demo not built (run __graft_entry__.build())")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad rows 0" in out.stdout


def test_training_example_runs_and_learns_something(tmp_path):
    """examples/train_ppo.py: rollout on the HIP path + torch update, a few epochs on the hover task; the mean reward must not
    collapse and the exported TorchScript actor must load."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_ppo.py"), "--num_envs=1024", "--epochs=6", "--horizon_len=16",
                          "--hidden=64", f"--export={tmp_path / 'actor.pt'}"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rewards = [float(l.split("reward/step")[1].split()[0]) for l in out.stdout.splitlines() if "reward/step" in l]
    assert len(rewards) == 6 and all(math.isfinite(r) and r > 0 for r in rewards)
    assert torch.jit.load(str(tmp_path / "actor.pt"))(torch.zeros(1, 1, 26)).shape == (1, 4)


def test_step_is_graph_capturable_and_replays_continue_the_run():
    """taco_step on a capturing stream enqueues the step kernel reading the DEVICE-resident clock (step index, ring head, history row) plus
    the one-thread kernel that advances it, so a captured step can be replayed: eager steps, then 100 replays, then eager steps again are
    bit-identical to 100 + k eager steps of a twin env -- every output and every state word, resets and command re-draws included."""
    import torch
    from taco_amd import config
    from taco_amd.vec_env import FpvBase
    n = 700
    kw = dict(env_maxEpisodeLength=40, env_lenStates=3, seed=5, ramdom_deploy_time=True, observation_noise=True, rotor_noise=True)
    a_env = FpvBase(config.default_cfg("mix", n, **kw), copy_outputs=False)
    b_env = FpvBase(config.default_cfg("mix", n, **kw), copy_outputs=False)
    g = torch.Generator().manual_seed(3)
    acts = (0.3 * torch.randn((8, n, 4), generator=g)).clamp(-1, 1).cuda()
    act = torch.zeros((n, 4), device="cuda")      # the graph reads its action from this fixed buffer
    for t in range(7):                            # eager prefix on both
        a_env.step_raw(acts[t % 8]); b_env.step_raw(acts[t % 8])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        graph.capture_begin()
        a_env.step_raw(act)                       # captured, not executed
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    for t in range(7, 107):
        act.copy_(acts[t % 8])
        graph.replay()
        b_env.step_raw(acts[t % 8])
        if t % 10 == 0 or t == 106:
            for name in ("obs_buf", "states_buf", "rew_buf", "reset_buf", "timeout_buf"):
                assert_bits_equal(getattr(a_env, name).cpu().numpy(), getattr(b_env, name).cpu().numpy(), f"replay {t} {name}")
    assert a_env.step_count == b_env.step_count == 107        # re-reads the device clock the replays advanced
    assert torch.equal(a_env.get_state().view(torch.int32), b_env.get_state().view(torch.int32))
    for t in range(107, 130):                     # and eagerly on
        a_env.step_raw(acts[t % 8]); b_env.step_raw(acts[t % 8])
    assert torch.equal(a_env.get_state().view(torch.int32), b_env.get_state().view(torch.int32))
    assert_bits_equal(a_env.obs_buf.cpu().numpy(), b_env.obs_buf.cpu().numpy(), "eager steps after the replays")
    a_env.check(); b_env.check()


def test_capture_as_the_first_launch_after_a_checkpoint_restore():
    """A restored env whose FIRST launch is a capture: taco_set_step_count (checkpoint.load_state_dict) must have put the step count into
    the device-resident clock too, or every replay would draw from the Philox counters of step 0.  Replays vs the eager twin, bit for bit.
    Also: reading the clock back while the capture is in progress is refused (TACO_ERR_STATE) instead of invalidating the capture."""
    from taco_amd import checkpoint
    from taco_amd._lib import TacoError
    from taco_amd.vec_env import FpvBase
    n = 500
    kw = dict(env_maxEpisodeLength=30, env_lenStates=2, seed=9, ramdom_deploy_time=True, observation_noise=True, rotor_noise=True)
    twin = FpvBase(config.default_cfg("mix", n, **kw), copy_outputs=False)
    g = torch.Generator().manual_seed(4)
    acts = (0.3 * torch.randn((8, n, 4), generator=g)).clamp(-1, 1).cuda()
    for t in range(21):
        twin.step_raw(acts[t % 8])
    sd = checkpoint.state_dict(twin)
    env = FpvBase(config.default_cfg("mix", n, **kw), copy_outputs=False)      # fresh handle: its device clock reads step 0
    checkpoint.load_state_dict(env, sd)
    act = torch.zeros((n, 4), device="cuda")
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        graph.capture_begin()
        env.step_raw(act)
        with pytest.raises(TacoError, match="capturing"):
            env.get_state()
        with pytest.raises(TacoError, match="capturing"):
            env.step_count
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    for t in range(21, 70):
        act.copy_(acts[t % 8])
        graph.replay()
        twin.step_raw(acts[t % 8])
        if t % 8 == 0 or t == 69:
            for name in ("obs_buf", "states_buf", "rew_buf", "reset_buf", "timeout_buf"):
                assert_bits_equal(getattr(env, name).cpu().numpy(), getattr(twin, name).cpu().numpy(), f"replay {t} {name}")
    assert env.step_count == twin.step_count == 70
    assert torch.equal(env.get_state().view(torch.int32), twin.get_state().view(torch.int32))
    # the one-row export agrees with the blob on the `actions` row of an env that has not stepped since its reset (FA:572: zeros)
    env.reset_idx(torch.tensor([3], device="cuda"))
    out = torch.empty(n, device="cuda")
    from taco_amd import _lib
    _lib.check(env.lib.taco_get_field(env._h, 40, out.data_ptr(), None), env.lib)
    assert out[3] == 0 and torch.equal(out, env.get_state()[40])


@pytest.mark.parametrize("n,len_states,form", [(300, 1, "auto"), (300, 5, "auto"), (70000, 1, "auto"), (20000, 5, "auto"), (300, 3, "lane"), (300, 3, "quad")])
def test_step_returns_kernel_written_clamped_copies(n, len_states, form):
    """VecTask.step returns clamp(obs_buf, +-clipObservations) / clamp(states_buf, +-clipStates) (vec_task_asymmetry.py:331-332).  The step
    kernel writes those copies itself -- step() is one launch -- in every store path (single frame, obs stacks moved by the role wavefronts or
    by the step wavefront, the state stack's frame ring with its clamped twin ring); they equal torch.clamp of the buffers bit for bit, NaN
    frames included, and a returned tensor survives the next step."""
    from taco_amd.vec_env import FpvBase
    cfg = config.default_cfg("mix", n, env_lenStates=len_states, env_lenObservations=2 if len_states == 3 else 1, env_clipObservations=0.6,
                             env_clipStates=0.9, env_maxEpisodeLength=30, observation_noise=True)
    env = FpvBase(cfg, kernel_form=form)
    g = torch.Generator().manual_seed(1)
    acts = (0.3 * torch.randn((6, n, 4), generator=g)).clamp(-1, 1).cuda()
    acts[3, :5, 1] = float("nan")                 # a NaN action reaches the frames `delay_time` ms later: clamp must pass it through
    prev = None
    # (a state stack lives in the frame ring: step() returns the view of a SECOND ring that holds the clamped frames; several wraps of it)
    for t in range(3 * env._st_period + 5 if env._ring_on else 40):
        d, rew, done, info = env.step(acts[t % 6])
        assert torch.equal(_bits(d["obs"]), _bits(torch.clamp(env.obs_buf, -0.6, 0.6))), t
        assert torch.equal(_bits(d["states"]), _bits(torch.clamp(env.states_buf, -0.9, 0.9))), t
        assert d["obs"].data_ptr() != env.obs_buf.data_ptr() and d["states"].data_ptr() != env.states_buf.data_ptr()
        if prev is not None:
            assert torch.equal(_bits(prev[0]), _bits(prev[1])), "the previous step's tensors must survive this step"
        prev = (d["states"], d["states"].clone())
    assert torch.isnan(env.states_buf).any() or n > 1000
    prog = env.progress_buf
    assert torch.equal(prog, env.get_state()[65].view(torch.int32).to(torch.long))
    env.check()


def _bits(t):
    return t.contiguous().view(torch.int32)


@pytest.mark.parametrize("n,len_obs,len_states,form", [(300, 1, 1, "auto"), (300, 2, 5, "auto"), (20000, 1, 5, "auto"), (70000, 1, 1, "auto"), (300, 3, 3, "lane"), (300, 1, 4, "quad"),
                                                       (5000, 1, 8, "auto"), (333, 1, 2, "auto")])
def test_step_returns_kernel_written_buffers_without_a_copy(n, len_obs, len_states, form):
    """Default clip (+inf): step() is ONE launch and nothing else.  The obs stack is read from the current buffer and written to the other one
    of an alternating pair; a state stack (len_states > 1) lives in a FRAME RING (taco_bind_states_ring): the launch writes one frame per env
    and step() returns the strided [n, len, 26] view of the window it filled.  Bit-identical to an env stepped in place on contiguous stacks,
    through every store path, across several wraps of the ring (its twin rows); the tensors of step t survive the following steps (at least
    one for the obs pair, period - len_states for the ring); env.obs_buf / states_buf always name what the last step produced."""
    from taco_amd.vec_env import FpvBase
    kw = dict(env_lenStates=len_states, env_lenObservations=len_obs, env_maxEpisodeLength=30, observation_noise=True)
    env = FpvBase(config.default_cfg("mix", n, **kw), kernel_form=form)
    ref = FpvBase(config.default_cfg("mix", n, **kw), kernel_form=form, copy_outputs=False)
    ring = len_states > 1
    assert env._ring_on == ring
    steps = 3 * env._st_period + 7 if ring else 45
    g = torch.Generator().manual_seed(1)
    acts = (0.3 * torch.randn((6, n, 4), generator=g)).clamp(-1, 1).cuda()
    held = []
    for t in range(steps):
        d, rew, done, info = env.step(acts[t % 6])
        ref.step_raw(acts[t % 6])
        assert d["obs"].data_ptr() == env.obs_buf.data_ptr() and d["states"].data_ptr() == env.states_buf.data_ptr()
        assert tuple(d["states"].shape) == (n, len_states, 26)
        assert torch.equal(_bits(d["obs"]), _bits(ref.obs_buf)) and torch.equal(_bits(d["states"]), _bits(ref.states_buf)), t
        assert torch.equal(rew.view(torch.int32), ref.rew_buf.view(torch.int32)) and torch.equal(done, ref.reset_buf)
        for back, (old, copy) in enumerate(reversed(held[-(env._st_period - len_states if ring else 1):])):
            assert torch.equal(_bits(old), _bits(copy)), f"the tensors of step {t - 1 - back} must survive step {t}"
        if held:
            assert held[-1][0].data_ptr() != d["states"].data_ptr()
        if len(held) > 1 and not ring:
            assert held[-2][0].data_ptr() == d["states"].data_ptr()          # two pairs alternate
        held.append((d["states"], d["states"].clone()))
    assert torch.equal(env.get_state().view(torch.int32), ref.get_state().view(torch.int32))
    env.step_raw(acts[0]); ref.step_raw(acts[0])                              # in place on the current obs buffer / the ring
    assert torch.equal(_bits(env.states_buf), _bits(ref.states_buf)) and torch.equal(_bits(env.obs_buf), _bits(ref.obs_buf))
    env.check()


@pytest.mark.parametrize("len_states,clip", [(1, math.inf), (5, math.inf), (5, 0.8)])
def test_fresh_outputs_returns_newly_allocated_obs_and_states_like_the_reference(len_states, clip):
    """FpvBase(..., fresh_outputs=True): step() returns NEW obs / states tensors every call (vec_task_asymmetry.py:331-332: torch.clamp allocates),
    so a caller may keep them for any number of steps; same values as the default (aliasing) env, which hands out views of its buffers."""
    from taco_amd.vec_env import FpvBase
    n = 700
    kw = dict(env_lenStates=len_states, env_maxEpisodeLength=30, observation_noise=True, env_clipObservations=clip, env_clipStates=clip)
    env = FpvBase(config.default_cfg("mix", n, **kw), fresh_outputs=True)
    ref = FpvBase(config.default_cfg("mix", n, **kw))
    g = torch.Generator().manual_seed(2)
    acts = (0.3 * torch.randn((6, n, 4), generator=g)).clamp(-1, 1).cuda()
    kept = []
    for t in range(2 * (ref._st_period if len_states > 1 else 4) + 3):
        d, rew, done, info = env.step(acts[t % 6])
        r, *_ = ref.step(acts[t % 6])
        assert d["states"].is_contiguous() and tuple(d["states"].shape) == (n, len_states, 26)
        assert torch.equal(_bits(d["obs"]), _bits(r["obs"])) and torch.equal(_bits(d["states"]), _bits(r["states"]))
        assert all(d["obs"].data_ptr() != k[0].data_ptr() and d["states"].data_ptr() != k[1].data_ptr() for k in kept)
        kept.append((d["obs"], d["states"], d["obs"].clone(), d["states"].clone()))
    for o, s_, oc, sc in kept:   # every tensor ever returned still holds what it held when it was returned
        assert torch.equal(_bits(o), _bits(oc)) and torch.equal(_bits(s_), _bits(sc))
    env.check()


def test_ring_backed_state_stack_through_capture_checkpoint_and_the_c_abi(tmp_path):
    """The frame ring behind the state stack (include/taco_env.h taco_bind_states_ring): (1) VecTask.step() refuses to be captured with it
    (every replay fills another window); (2) the C entry point itself IS capturable -- the ring phase is part of the device-resident clock, a
    captured step is ONE kernel node that advances it -- and replays continue the run bit for bit, the window named by taco_states_ring_row;
    (3) a checkpoint taken at an arbitrary ring phase restores into a fresh env (phase 0, twins rebuilt) and both continue identically."""
    import ctypes as C
    from taco_amd import _lib, checkpoint
    from taco_amd.vec_env import FpvBase
    n, T = 900, 5
    kw = dict(env_maxEpisodeLength=35, env_lenStates=T, seed=4, ramdom_deploy_time=True, observation_noise=True, rotor_noise=True)
    env = FpvBase(config.default_cfg("mix", n, **kw))
    ref = FpvBase(config.default_cfg("mix", n, **kw), copy_outputs=False)
    g = torch.Generator().manual_seed(8)
    acts = (0.3 * torch.randn((8, n, 4), generator=g)).clamp(-1, 1).cuda()
    for t in range(11):
        env.step(acts[t % 8]); ref.step_raw(acts[t % 8])
    # (3) checkpoint at phase 11 -> fresh env
    checkpoint.save(env, str(tmp_path / "ring.pt"))
    twin = FpvBase(config.default_cfg("mix", n, **kw))
    checkpoint.load(twin, str(tmp_path / "ring.pt"))
    assert torch.equal(_bits(twin.states_buf), _bits(ref.states_buf))
    # (1) step() under capture is refused
    act = torch.zeros((n, 4), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    io = _lib.RolloutIO(act.data_ptr(), None, env.obs_buf.data_ptr(), None, None, env.rew_buf.data_ptr(), env.reset_buf.data_ptr(), env.timeout_buf.data_ptr(), None)
    with torch.cuda.stream(s):
        graph.capture_begin()
        with pytest.raises(_lib.TacoError, match="cannot be captured"):
            env.step(act)                         # refused, nothing enqueued
        # (2) ... the C entry point without the window report is capturable: two steps on the action in `act`
        for _ in range(2):
            _lib.check(env.lib.taco_step_rollout(env._h, C.byref(io), C.c_void_p(s.cuda_stream)), env.lib)
        with pytest.raises(_lib.TacoError, match="capturing"):
            env.release_graphs()                  # (graph mode cannot be left while the capture is still open: that would synchronise into it)
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    t = 11
    for rep in range(40):                         # 80 steps: more than a period of the ring
        act.copy_(acts[t % 8])
        for k in range(2):
            ref.step_raw(acts[t % 8]); twin.step(acts[t % 8])
        t += 2
        graph.replay()
        assert torch.equal(_bits(env.states_buf), _bits(ref.states_buf)) and torch.equal(_bits(twin.states_buf), _bits(ref.states_buf)), rep
        assert torch.equal(_bits(env.obs_buf), _bits(ref.obs_buf))
    # ... and an OBS stack on the alternating buffer pair refuses capture too (a replay would shift the never-updated buffer of the frozen pair)
    env2 = FpvBase(config.default_cfg("pos", 64, env_lenObservations=2))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g2.capture_begin()
        with pytest.raises(_lib.TacoError, match="cannot be captured"):
            env2.step(act[:64])
        env2.step_raw(act[:64])                    # (in place on the current buffers: capturable)
        g2.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    assert env.step_count == ref.step_count == 91
    for t2 in range(t, t + 9):                    # eagerly on, through step(): the host picks the ring phase up again
        d, *_ = env.step(acts[t2 % 8]); ref.step_raw(acts[t2 % 8]); twin.step(acts[t2 % 8])
        assert torch.equal(_bits(d["states"]), _bits(ref.states_buf))
    assert torch.equal(env.get_state().view(torch.int32), ref.get_state().view(torch.int32))
    assert torch.equal(twin.get_state().view(torch.int32), ref.get_state().view(torch.int32))
    assert torch.equal(_bits(twin.states_buf), _bits(ref.states_buf))
    env.check(); twin.check()
    # GRAPH MODE and its end (include/taco_env.h taco_graph_mode / taco_release_graphs): since the capture every clock-reading call of `env`
    # synchronises the device; release_graphs() puts the handle back on the eager path (host-side lookups, no sync), bit for bit the same run
    assert env.lib.taco_graph_mode(env._h) == 1 and twin.lib.taco_graph_mode(twin._h) == 0
    env.release_graphs()
    assert env.lib.taco_graph_mode(env._h) == 0
    for t3 in range(t2 + 1, t2 + 70):             # (more than a period of the ring)
        d, *_ = env.step(acts[t3 % 8]); ref.step_raw(acts[t3 % 8])
        assert torch.equal(_bits(d["states"]), _bits(ref.states_buf)) and torch.equal(_bits(env.states_buf), _bits(ref.states_buf))
    assert torch.equal(env.get_state().view(torch.int32), ref.get_state().view(torch.int32)) and env.step_count == ref.step_count
    env.check()
    # ... and a replay AFTER the release is not silent: the device clock runs ahead of the host's copy, check() says so
    graph.replay()
    with pytest.raises(_lib.TacoError, match="replayed after taco_release_graphs"):
        env.check()
    # ... and STAYS reported once eager steps have overwritten the device clock's step word (round 5's advisor: the comparison of the two clocks
    # alone went quiet after one env.step()): launches that read the device clock count themselves, the release latched the count
    for t4 in range(3):
        env.step(acts[t4 % 8])
    with pytest.raises(_lib.TacoError, match="replayed after taco_release_graphs"):
        env.check()


def _run_bench(extra, env_extra=None, timeout=420):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads (metric/value/unit/n_gpus/steps/warmup/ms_per_step/
    higher_is_better/scaling/vs_baseline/dtype/data/config) plus roofline, cpu_baseline, steady_state, step_api and the configs array."""
    d = _run_bench(["--steps", "100", "--warmup", "20", "--no-large-n", "--cpu-seconds", "1.5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "steady_state", "step_api", "configs"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 100 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread", "configs"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and [x["config"] for x in c["configs"]] == [1, 3]
    assert d["parity"]["all_words_bit_equal"] and d["parity"]["done_flags_equal"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # SURVEY 8(d) configs 3-5 at their per-GPU shapes
    cfgs = d["configs"]
    assert [(x["config"], x.get("task_mode"), x["envs"]) for x in cfgs[:3]] == [(3, "rotate", 16384), (4, "flip", 16384), (5, "mix", 32768)]
    # (config 5's state stack lives in the frame ring behind VecTask.step(): one frame written, the algorithmic 820 B; the materialised stack beside it at SURVEY's 1 236 B)
    assert cfgs[2]["len_states"] == 5 and cfgs[2]["algorithmic_bytes_per_env_step"] == 820.0 and "observation_noise" in cfgs[2]["flags_on"]
    assert cfgs[2]["materialised_stack"]["algorithmic_bytes_per_env_step"] == 1236.0 and cfgs[2]["materialised_stack"]["kernel_us"] > cfgs[2]["kernel_us"]
    assert cfgs[3]["config"] == 5 and cfgs[3]["env_steps_per_s"] > 0, cfgs[3]
    # the headline IS VecTask.step(): ping-pong buffer pairs, the launch step_raw() makes (the host-side dict is all it adds)
    assert "VecTask.step()" in d["metric"] and d["steady_state"]["api"] == "VecTask.step()"
    assert d["steady_state"]["us_per_step"] < 1.12 * d["step_raw"]["us_per_step"], "step() is one launch: it must cost about what step_raw costs"
    assert d["step_api_finite_clip"]["us_per_step"] < 1.3 * d["step_raw"]["us_per_step"]
    r = d["roofline"]
    assert "traffic_frac" in r and (r["traffic"] is None or abs(r["traffic_frac"] - r["traffic"] / (r["kernel_avg_us"] * 1e-6) / 1e9 / 8000.0) < 1e-9)
    lf = d["latency_floor"]
    assert 0.4 < lf["frac_of_kernel_avg"] < 1.0 and lf["critical_path_clocks"] >= lf["step_wavefront_clocks"] > 10000, lf
    ro = cfgs[4]
    assert ro["config"] == "5 at 4096 envs x 32 steps" and "persistent" in ro["per_step_chain"] and ro["ms_per_rollout"] < ro["ms_per_rollout_launch_per_step"]
    # BASELINE configs 3-5 at their TOTAL env counts (here on one rank), the rollout of config 5 included
    bc = d["baseline_configs"]
    assert [(x["config"], x["envs_total"]) for x in bc] == [(3, 16384), (4, 65536), (5, 262144), (5, 262144)] and all("error" not in x for x in bc), bc
    assert bc[2]["len_states"] == 5 and bc[3]["horizon"] == 16 and bc[3]["value"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun starts the two ranks itself (as children, before touching the GPU), relays rank 0's
    line and exits 0.  Rehearsed here on ONE GPU over gloo (TACO_BENCH_BACKEND / TACO_BENCH_ONE_DEVICE); the driver runs it over RCCL."""
    d = _run_bench(["--gpus", "2", "--steps", "60", "--warmup", "10"], {"TACO_BENCH_BACKEND": "gloo", "TACO_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["config"]["envs_total"] == 8192 and d["scaling"] == "weak"
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the headline of N > 1 carries north_star's per-step all-gather (overlapped); the gather-free and the serial figures sit beside it
    assert d["config"]["collective"] != "none" and "all_gather" in d["config"]["collective"] and "step_async" in d["config"]["api"]
    assert "allgather_legs_error" not in d and d["without_allgather"]["value"] > 0 and d["with_allgather_serial"]["value"] > 0
    assert d["without_allgather"]["value"] >= 0.8 * d["value"], "the gather-free leg cannot be much slower than the gathered one"
    assert d["allgather"]["bytes_per_rank"] == 4096 * 32 * 4 and d["allgather"]["backend"] == "gloo"
    assert len(d["kernel_us_per_rank"]) == 2 and d["envs_per_rank"] == [4096.0, 4096.0] and d["n1_on_rank0"]["value"] > 0
    bc = d["baseline_configs"]      # BASELINE configs 3-5 at their real totals, sharded over the two ranks; config 5 with its rollout
    assert [(x["config"], x["envs_total"], x["envs_per_rank"]) for x in bc] == [(3, 16384, 8192), (4, 65536, 32768), (5, 262144, 131072), (5, 262144, 131072)], bc
    assert all("error" not in x and x["value"] > 0 for x in bc) and all("error" not in x["with_allgather"] for x in bc[:3])
    s = d["strong_scaling"]
    assert "error" not in s and s["envs_total"] == 4096 and s["envs_per_rank"] == 2048 and s["value"] > 0


def test_bench_main_leg_on_another_baseline_config_and_backend_fallback():
    """`--config 4`: BASELINE configs[3] (flip, 65 536 envs in total) as the MAIN timed leg, sharded over two ranks; the process group is asked
    for over a backend that cannot initialise here ("ucc"): the ranks fall back to gloo, the line is printed and says what happened."""
    d = _run_bench(["--gpus", "2", "--steps", "40", "--warmup", "10", "--config", "4", "--no-configs"], {"TACO_BENCH_BACKEND": "ucc", "TACO_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["envs_total"] == 65536 and d["config"]["envs_per_gpu"] == 32768
    assert "configs[3]" in d["metric"] and abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["backend"] == "gloo" and "ucc" in d["collective_init_error"] and d["ranks_seen"] == [0, 1]
    assert "all_gather" in d["config"]["collective"] and d["without_allgather"]["value"] > 0


def test_bench_no_gather_flag_keeps_the_gather_free_headline():
    """`--no-gather`: the main timed leg of N > 1 is the gather-free ShardedEnv.step_local; the gathered forms are still timed beside it"""
    d = _run_bench(["--gpus", "2", "--steps", "40", "--warmup", "10", "--no-gather", "--no-configs"], {"TACO_BENCH_BACKEND": "gloo", "TACO_BENCH_ONE_DEVICE": "1"})
    assert d["config"]["collective"].startswith("none in the timed region") and "step_local" in d["config"]["api"]
    assert d["with_allgather"]["value"] > 0 and d["with_allgather_serial"]["value"] > 0 and "without_allgather" not in d


def test_yaml_launcher_equals_the_python_built_config(tmp_path):
    """SURVEY 8f row N4 end to end: a YAML in the reference's layout + CLI overrides -> examples/run_env.py (the launcher's call sequence,
    train_fpv_asymmetry_ppo.py:363-371) -> 50 steps, against the same configuration built in Python and stepped in this process:
    the last step's tensors and the env's whole raw state, bit for bit."""
    import subprocess
    import sys
    from taco_amd import cfg_io, config
    from taco_amd.vec_env import isaacgym_task_map
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    yml = cfg_io.write_default_yaml(str(tmp_path / "Fpv_asymmetry_PPO_mix.yaml"), "mix", delay_time=30)
    dump = str(tmp_path / "out.npz")
    cli = ["--task_mode=mix", "--num_envs=777", "--lenStates=3", "--rotor_noise=True", "--observation_noise=True", "--ramdom_deploy_time=True",
           "--rotor_response_time=0.018", "--seed=11"]
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "run_env.py"), "--cfg", yml, "--steps", "50", "--dump", dump] + cli,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(dump)
    # the same configuration, built in Python
    cfg = config.default_cfg("mix", 777, delay_time=30, rotor_noise=True, observation_noise=True, ramdom_deploy_time=True, rotor_response_time=0.018, seed=11,
                             env_lenStates=3)
    ap = cfg_io.add_env_arguments(__import__("argparse").ArgumentParser())
    task = cfg_io.apply_overrides({"Task": cfg}, ap.parse_args(["--task_mode=mix"]))
    env = isaacgym_task_map[task["name"]](task, "cuda:0", "cuda:0", -1, True, False, False)
    g = torch.Generator(device="cuda").manual_seed(0)
    W = 0.3 * torch.randn(env.len_obs * env.num_obs, env.num_acts, device="cuda", generator=g)
    b = torch.tensor([-0.3, 0.0, 0.0, 0.0], device="cuda")
    obs = env.reset()["obs"]
    for _ in range(50):
        act = torch.tanh(obs.reshape(env.num_envs, -1).nan_to_num() @ W + b)
        od, rew, done, info = env.step(torch.clip(act, -1, 1))
        obs = od["obs"]
    assert_bits_equal(obs.cpu().numpy(), got["obs"], "obs")
    assert_bits_equal(od["states"].cpu().numpy(), got["states"], "states")
    assert_bits_equal(rew.cpu().numpy(), got["rew"], "rew")
    assert np.array_equal(done.cpu().numpy(), got["done"])
    assert np.array_equal(env.get_state().view(torch.int32).cpu().numpy(), got["state"].view(np.int32))


def test_sharded_env_step_local_is_the_single_gpu_launch_and_rebinding_works():
    """ShardedEnv.step_local(): this rank's slice through VecTask.step() with no gather block bound (what bench.py's ungathered main leg
    times for N > 1).  Same results as the block-filling step of a twin shard, and a gathered step after it fills its block again."""
    from taco_amd.dist import ShardedEnv, unpack_block
    cfg = config.default_cfg("mix", 333, seed=5, env_maxEpisodeLength=6)
    dev = torch.device("cuda:0")
    a = ShardedEnv(cfg, rank=1, world_size=2, device=dev, gather=False)     # envs 167 .. 332 of the job
    b = ShardedEnv(cfg, rank=1, world_size=2, device=dev, gather=False)
    n = a.hi - a.lo
    g = torch.Generator(device="cpu").manual_seed(1)
    for t in range(14):
        act = (torch.rand(n, 4, generator=g) * 2 - 1).cuda()
        blk = a.step_gathered(act)
        obs_a, rew_a, done_a, tmo_a = unpack_block(blk, a.len_obs)
        if t % 3 == 2:      # now and then the gathered path on b too: the block must be bound again and filled
            blk_b = b.step_gathered(act)
            assert torch.equal(blk_b, blk)
            continue
        o, rew_b, done_b, ex = b.step_local(act)
        assert b._bound is None
        assert torch.equal(o["obs"].reshape(n, -1), obs_a.reshape(n, -1)) and torch.equal(rew_b, rew_a)
        assert torch.equal(done_b.to(done_a.dtype), done_a) and torch.equal(ex["time_outs"].to(tmo_a.dtype), tmo_a)
    assert torch.equal(a.env.get_state(), b.env.get_state())


@pytest.mark.parametrize("direct", [True, False])
def test_rccl_backend_runs_the_gather_path_on_one_rank(direct):
    """torch.distributed's "nccl" backend IS RCCL on ROCm.  The test box has ONE GPU and RCCL refuses two ranks on one device, so the N > 1 tests
    run over gloo -- here the RCCL branch itself executes, with a 1-rank group: communicator init, ShardedEnv.step_async / step_gathered issuing
    all_gather_into_tensor on the kernel-filled device block (async handle, the two alternating blocks, the stream-side wait), results equal to
    the env's own buffers.  direct=True (the default when the backend is RCCL, round 6): the gather is ONE ncclAllGather called directly on a comm
    stream (taco_amd/rccl.py: unique id through the process group, self-check, event-ordered); False: the process group's collective.
    In a child process (a process group is process-wide state)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29583", rank=0, world_size=1)
from taco_amd import config
from taco_amd.dist import ShardedEnv, unpack_block
from taco_amd.vec_env import FpvBase
dev = torch.device("cuda:0")
n = 3000
cfg = config.default_cfg("mix", n, env_lenObservations=2, env_lenStates=3, env_maxEpisodeLength=20, observation_noise=True)
env = ShardedEnv(cfg, rank=0, world_size=1, device=dev, gather=True, collective_when_alone=True, direct=DIRECT)
assert env.direct_path == DIRECT, env.direct_reason     # round 6: ONE ncclAllGather called directly on a comm stream (taco_amd/rccl.py) / the process group's collective
ref = FpvBase(config.default_cfg("mix", n, env_lenObservations=2, env_lenStates=3, env_maxEpisodeLength=20, observation_noise=True), copy_outputs=False)
g = torch.Generator().manual_seed(2)
acts = (0.3 * torch.randn((6, n, 4), generator=g)).clamp(-1, 1).to(dev)
pend = []
for t in range(40):
    ref.step_raw(acts[t % 6])
    if t % 2:                                     # serial form
        rows = env.step_gathered(acts[t % 6])
    else:                                         # overlapped form: waited for one step later
        pend.append(env.step_async(acts[t % 6]))
        if len(pend) > 1:
            pend.pop(0).wait()
        continue
    while pend:
        pend.pop(0).wait()
    obs, rew, done, tmo = unpack_block(rows, env.len_obs)
    torch.cuda.synchronize()
    assert torch.equal(obs, ref.obs_buf) and torch.equal(rew, ref.rew_buf) and torch.equal(done, ref.reset_buf) and torch.equal(tmo, ref.timeout_buf), t
env.drain()
# bench.py's overlapped loop (round 6): the previous step's gather is waited for between this step's kernel launch and this step's collective;
# what each gather delivered is checked one step late, against a copy of the reference's buffers of THAT step
pend, held = [], []
def wait_previous():
    while pend:
        g_, want = pend.pop(0), held.pop(0)
        rows_ = g_.wait().global_rows()
        o_, r_, d_, t_ = unpack_block(rows_, env.len_obs)
        torch.cuda.synchronize()
        assert torch.equal(o_, want[0]) and torch.equal(r_, want[1]) and torch.equal(d_, want[2]) and torch.equal(t_, want[3]), "overlapped gather"
for t in range(40, 70):
    ref.step_raw(acts[t % 6])
    want = (ref.obs_buf.clone(), ref.rew_buf.clone(), ref.reset_buf.clone(), ref.timeout_buf.clone())
    g_ = env.step_async(acts[t % 6], before_gather=wait_previous)
    pend.append(g_); held.append(want)
wait_previous()
env.drain()
assert dist.get_backend() == "nccl"
# ... and the collectives bench.py brackets its timed region with (barrier, MAX-reduction and all_gather of fp64 scalars on the device)
t = torch.tensor([3.5], device=dev, dtype=torch.float64)
dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX)
outs = [torch.zeros_like(t)]
dist.all_gather(outs, t)
assert float(t.item()) == 3.5 and float(outs[0].item()) == 3.5
print("RCCL-GATHER-OK")
dist.destroy_process_group()
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code.replace("DIRECT", repr(bool(direct)))], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL-GATHER-OK" in out.stdout, (out.stdout + out.stderr)[-3000:]
