"""GPU parity: the HIP step kernel (through the C ABI / taco_amd.vec_env) against the CPU oracle on identical seeded
inputs.  Bar: every state word, observation, reward and flag BIT-EXACT after every step (the kernel and the oracle
implement the same IEEE-754 operation sequence; see DESIGN.md "arithmetic contract").  The north-star tolerance
(trajectory L-inf <= 1e-5 over 1 000 steps, integer flags exact) is asserted as well, as the weaker documented bar.
"""
import os

import numpy as np
import pytest
import torch

from taco_amd import config
from util import GLUE_CASES, GLUE_CASES_IEEE, GLUE_CASES_LONG, assert_bits_equal

pytestmark = pytest.mark.gpu


def action_stream(n, steps, seed):
    """a_t = clamp(0.3*N(0,1) + (-0.45,0,0,0), -1, 1)"""
    rng = np.random.default_rng(seed)
    a = 0.3 * rng.standard_normal((steps, n, 4)).astype(np.float32)
    a[:, :, 0] -= 0.45
    return np.clip(a, -1, 1).astype(np.float32)


def run_pair(cfg, steps, seed=0, check_every=1, hover_bias=False, form="auto", lib=None, slow_server=False, api=False):
    """api: a second env is stepped through VecTask.step() (the alternating obs buffers, the state stack on the frame ring or the kernel-written
    clamped copies) and what it RETURNS is held against the oracle's clamped buffers as well"""
    from oracle import oracle as O
    from taco_amd import _lib
    from taco_amd.vec_env import FpvBase
    flat = config.flat_cfg(cfg)
    n = flat["num_envs"]
    env = FpvBase(cfg, copy_outputs=False, kernel_form=form, lib=lib)
    api_env = FpvBase(cfg, kernel_form=form, lib=lib) if api else None
    if slow_server:
        _lib.check(lib.taco_test_slow_battery_server(env._h, 1), lib)
    orc = O.OracleEnv(flat, threads=8)
    gids = np.arange(n)
    if flat["task_mode"] == "mix":
        flip_envs = gids >= int(n / 3 * 2)
    else:
        flip_envs = np.full(n, flat["task_mode"] == "flip")
    if flat.get("record_flag", False):
        flip_envs = np.full(n, True)   # record_flag: the kernel keeps copter_rpy_continuous current for every env (TACO_F_TRACK_RPY)
    acts = action_stream(n, steps, seed)
    if hover_bias:
        acts[:, :, 0] = np.clip(acts[:, :, 0] + 0.25, -1, 1)
    acts_d = torch.from_numpy(acts).cuda()
    max_err = 0.0
    n_done = 0
    for t in range(steps):
        env.step_raw(acts_d[t])
        orc.step(acts[t])
        if api_env is not None:
            d, a_rew, a_done, a_info = api_env.step(acts_d[t])
        if t % check_every == 0 or t == steps - 1:
            what = f"step {t}"
            if api_env is not None:   # VecTask.step()'s return values = clamp(buffers) (vec_task_asymmetry.py:331-334)
                co, cs = np.float32(flat["clip_obs"]), np.float32(flat["clip_states"])
                assert_bits_equal(d["obs"].cpu().numpy(), np.clip(orc.obs_buf, -co, co), what + " step() obs")
                assert_bits_equal(d["states"].cpu().numpy(), np.clip(orc.states_buf, -cs, cs), what + " step() states")
                assert_bits_equal(a_rew.cpu().numpy(), orc.rew_buf, what + " step() reward")
                assert_bits_equal(a_done.cpu().numpy(), orc.reset_buf, what + " step() done")
                assert_bits_equal(a_info["time_outs"].cpu().numpy().astype(np.uint8), orc.timeout_buf, what + " step() time_outs")
            assert_bits_equal(env.reset_buf.cpu().numpy(), orc.reset_buf, what + " reset_buf")
            assert_bits_equal(env.timeout_buf.cpu().numpy().astype(np.uint8), orc.timeout_buf, what + " timeout_buf")
            assert_bits_equal(env.rew_buf.cpu().numpy(), orc.rew_buf, what + " rew_buf")
            assert_bits_equal(env.obs_buf.cpu().numpy(), orc.obs_buf, what + " obs_buf")
            assert_bits_equal(env.states_buf.cpu().numpy(), orc.states_buf, what + " states_buf")
            gb = env.get_state().cpu().numpy()
            ob = orc.get_state().view(np.float32)
            assert_bits_equal(gb.view(np.uint32)[65:67], ob.view(np.uint32)[65:67], what + " progress / delay length")
            # copter_rpy_old / copter_rpy_continuous (rows 20..25) are maintained by the kernel only where they are consumed:
            # envs of the flip task (fpv_asymmetry.py:831, :930).  Elsewhere they are a logging-only quantity of the
            # reference and the kernel keeps the reset-time value.
            assert_bits_equal(gb[:20], ob[:20], what + " state fields 0..19")
            assert_bits_equal(gb[26:65], ob[26:65], what + " state fields 26..64")
            assert_bits_equal(gb[20:26][:, flip_envs], ob[20:26][:, flip_envs], what + " rpy_old / rpy_continuous of the flip envs")
            assert_bits_equal(gb[67:], ob[67:], what + " delay line")
            fin = np.isfinite(ob[:13]) & np.isfinite(gb[:13])
            if fin.any():
                max_err = max(max_err, float(np.abs(gb[:13] - ob[:13])[fin].max()))
        n_done += int(orc.reset_buf.sum())
    assert max_err <= 1e-5, "north-star bar: root-state trajectory L-inf <= 1e-5"
    return n_done


def test_pos_plumbing_config0():
    """BASELINE configs[0]: pos, 64 envs, all randomisation off, battery off."""
    run_pair(config.baseline_config(0), steps=300)


def test_pos_config1_4096_envs_1000_steps():
    """BASELINE configs[1] at full size, 1 000 steps (north-star horizon); episodes end and re-randomise on the way."""
    done = run_pair(config.baseline_config(1), steps=1000, check_every=50)
    assert done > 1000, "the run must exercise resets"


def test_rotate_config2():
    run_pair(config.baseline_config(2, num_envs=1024), steps=600, check_every=25, hover_bias=True)


def test_flip_config3():
    run_pair(config.baseline_config(3, num_envs=1024), steps=600, check_every=25, hover_bias=True)


def test_mix_all_randomisation_config4():
    """BASELINE configs[4] flags: every random_* on, observation/rotor noise, random delay + deploy time, lenStates 5;
    1 000 envs so the thirds are uneven (333/666)."""
    run_pair(config.baseline_config(4, num_envs=1000), steps=700, check_every=25, hover_bias=True)


def test_ragged_sizes_and_large_block():
    for n in (1, 63, 65, 200):
        run_pair(config.baseline_config(1, num_envs=n), steps=40)
    run_pair(config.baseline_config(4, num_envs=65536 + 77), steps=12, check_every=4)  # 256-thread workgroups, ragged tail


def test_clipping_and_substeps():
    cfg = config.baseline_config(1, num_envs=256)
    cfg["env"]["clipActions"] = 0.5
    cfg["sim"]["substeps"] = 4
    cfg["env"]["maxEpisodeLength"] = 120
    cfg["env"]["lenObservations"] = 3
    cfg["env"]["lenStates"] = 2
    run_pair(cfg, steps=260)


def test_delay_line_overflow_regime():
    """delay_time = 85 with random deploy time: L + T crosses 100, the mask write is truncated and the stale tail of the
    dense buffer is read (SURVEY section 7 'delay-line overflow').  The ring must reproduce the dense buffer exactly."""
    cfg = config.baseline_config(1, num_envs=512)
    cfg["delay_time"] = 85
    cfg["ramdom_deploy_time"] = True
    cfg["ramdom_delay_time"] = True
    run_pair(cfg, steps=400, check_every=10, hover_bias=True)
    # start just BELOW the tail (L + T = 84..86 <= 90): envs begin with the implied-zero encoding and random-walk into the
    # dense regime mid-episode (the kernel's one-off materialisation of the implied zeros is on this path)
    cfg = config.baseline_config(1, num_envs=512)
    cfg["delay_time"] = 75
    cfg["ramdom_deploy_time"] = True
    cfg["env"]["maxEpisodeLength"] = 300
    run_pair(cfg, steps=500, check_every=5, hover_bias=True)


@pytest.mark.parametrize("form", ["auto", "quad", "lane", "lane_throughput", "lane_roles", "quad_served"])
@pytest.mark.parametrize("len_obs", [1, 3])
def test_gather_block_matches_outputs(len_obs, form):
    """the packed per-rank block the kernel fills for the all-gather == pack_block(obs, rew, done, time_outs), in every instantiation
    (role wavefronts: the obs wavefront writes the obs part of a row, the reward wavefront its three tail words)"""
    import ctypes as C
    from taco_amd import dist, _lib
    from taco_amd.vec_env import FpvBase
    n = 300
    cfg = config.baseline_config(4, num_envs=n)
    cfg["env"]["lenObservations"] = len_obs
    cfg["env"]["maxEpisodeLength"] = 30
    env = FpvBase(cfg, copy_outputs=False, kernel_form=form)
    assert env.lib.taco_gather_row_floats(len_obs) == dist.block_row(len_obs)
    block = torch.full((n, dist.block_row(len_obs)), -7.0, device="cuda")
    _lib.check(env.lib.taco_bind_gather_block(env._h, C.c_void_p(block.data_ptr())))
    acts = torch.from_numpy(action_stream(n, 80, 3)).cuda()
    for t in range(80):
        env.step_raw(acts[t])
        want = dist.pack_block(env.obs_buf, env.rew_buf, env.reset_buf, env.timeout_buf)
        row = len_obs * 26 + 3
        assert torch.equal(block[:, :row].view(torch.int32), want[:, :row].view(torch.int32)), f"step {t}"
        o, r, d, tm = dist.unpack_block(block, len_obs)
        assert torch.equal(d, env.reset_buf) and torch.equal(tm, env.timeout_buf)
    assert env.timeout_buf.any() or True


def test_state_blob_roundtrip_and_restore():
    """taco_get_state / taco_set_state (+ step count) restore a run exactly: a second handle loaded from a mid-run blob
    continues bit-identically (covers the logical<->physical ring mapping and the implied-zero delay-line encoding)."""
    from taco_amd.vec_env import FpvBase
    n = 500
    cfg = config.baseline_config(4, num_envs=n)
    cfg["delay_time"] = 60
    a = FpvBase(cfg, copy_outputs=False)
    acts = torch.from_numpy(action_stream(n, 90, 11)).cuda()
    for t in range(37):
        a.step_raw(acts[t])
    blob = a.get_state()
    b = FpvBase(config.baseline_config(4, num_envs=n) | {"delay_time": 60}, copy_outputs=False)
    b.set_state(blob)
    b.step_count = a.step_count
    b.obs_buf.copy_(a.obs_buf); b.states_buf.copy_(a.states_buf); b.reset_buf.copy_(a.reset_buf)
    assert torch.equal(b.get_state().view(torch.int32), blob.view(torch.int32))
    for t in range(37, 90):
        a.step_raw(acts[t]); b.step_raw(acts[t])
        assert torch.equal(a.obs_buf.view(torch.int32), b.obs_buf.view(torch.int32)), t
        assert torch.equal(a.states_buf.view(torch.int32), b.states_buf.view(torch.int32)), t
        assert torch.equal(a.reset_buf, b.reset_buf) and torch.equal(a.rew_buf.view(torch.int32), b.rew_buf.view(torch.int32)), t
    assert torch.equal(a.get_state().view(torch.int32), b.get_state().view(torch.int32))


@pytest.mark.parametrize("task,n", [("pos", 1000), ("mix", 333), ("flip", 257), ("rotate", 4096), ("mix", 8232), ("rotate", 12500), ("posnoise", 777)])
def test_quad_layout_equals_one_lane_per_env(task, n):
    """The six instantiations of the step kernel (4 lanes per env with and without the role wavefronts or with one serving wavefront / 1 lane per
    env with 64- and 256-thread workgroups and with role wavefronts) are the same function: every output and every state word bit-identical, all randomisation on, stacked frames.
    (Up to 8 192 envs the four-role quad form is the build whose obs role wavefront runs the linear chain, above it the build that keeps it in the
    step wavefront -- taco_step.hpp LIN, taco_capi.hip lin_form: the 8 232-env case is the second.  The served pair pinned on a launch WITHOUT flip
    envs runs the linear chain in its serving wavefront: the pos / rotate / posnoise cases.)"""
    from taco_amd.vec_env import FpvBase
    kw = dict(env_lenObservations=2, env_lenStates=3, env_maxEpisodeLength=40, seed=11)
    if task == "mix":
        kw.update(rotor_noise=True, observation_noise=True, ramdom_deploy_time=True, ramdom_delay_time=True)
    if task == "posnoise":   # (pos / rotate in the served pair: its serving wavefront runs the LINEAR CHAIN, round 5 -- here behind the rotor-noise table)
        task = "pos"
        kw.update(rotor_noise=True, ramdom_deploy_time=True)
    if n == 12500:           # (the size the product picks the pair for: no stacks)
        kw.update(env_lenObservations=1, env_lenStates=1)
    envs = []
    for form, lpe, block in (("quad", 4, 64), ("lane", 1, 64), ("lane_throughput", 1, 64), ("quad_roles", 4, 256), ("lane_roles", 1, 256), ("quad_served", 4, 128)):
        e = FpvBase(config.default_cfg(task, n, **kw), copy_outputs=False, kernel_form=form)
        assert e.kernel_form == form
        assert e.launch_geometry() == ((n * lpe + 63) // 64, block)  # one step wavefront per 64 / lpe envs in every form
        envs.append(e)
    acts = torch.from_numpy(action_stream(n, 90, 4)).cuda()
    for t in range(90):
        for e in envs:
            e.step_raw(acts[t])
        ref = envs[1]
        for e in (envs[0], envs[2], envs[3], envs[4], envs[5]):
            for name in ("obs_buf", "states_buf", "rew_buf", "reset_buf", "timeout_buf"):
                assert_bits_equal(getattr(e, name).cpu().numpy(), getattr(ref, name).cpu().numpy(), f"step {t} {name}")
        if t % 10 == 9:
            b = [e.get_state().view(torch.int32) for e in envs]
            assert all(torch.equal(x, b[1]) for x in b), f"step {t} state blob"
    assert int(envs[1].reset_buf.sum()) >= 0


@pytest.mark.parametrize("form", ["lane", "lane_throughput", "lane_roles"])
def test_shared_reset_draws_match_the_oracle_for_every_count_of_resetting_lanes(form):
    """The one-lane forms draw a resetting env's Philox blocks with sixteen lanes at once when at most four lanes of the wavefront reset, and lane
    by lane otherwise (taco_step.hpp "shared reset draws"): planted reset flags -- 1, 2, 3, 4 (shared pass), 5, 17, 64 (own draws) resetting lanes in
    a wavefront, none in another, and the ragged last wavefront's last lanes (whose tail lanes shadow env n - 1) -- against the CPU oracle, every
    randomisation on (all ten blocks drawn), state blob and outputs bit for bit."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    n = 64 * 8 + 37
    cfg = config.default_cfg("mix", n, env_lenObservations=1, env_lenStates=3, env_maxEpisodeLength=400, seed=23, rotor_noise=True, observation_noise=True,
                             ramdom_deploy_time=True, ramdom_delay_time=True, random_rotordynamic_coe=True, random_rotor_response=True,
                             random_aerodynamic_coe=True)
    flat = config.flat_cfg(cfg)
    env = FpvBase(cfg, copy_outputs=False, kernel_form=form)
    assert env.kernel_form == form
    orc = O.OracleEnv(flat, threads=8)
    acts = action_stream(n, 12, 6)
    acts_d = torch.from_numpy(acts).cuda()
    rng = np.random.default_rng(2)
    counts = {0: 1, 1: 2, 2: 3, 3: 4, 4: 5, 5: 17, 6: 64, 7: 0}
    for t in range(12):
        if t in (3, 6, 9):   # (every step 0 resets all envs: the own-draws path with 64 lanes)
            plant = np.zeros(n, np.int64)
            for wv, c in counts.items():
                plant[64 * wv + rng.choice(64, c, replace=False)] = 1
            plant[n - 2:] = 1   # the last wavefront: 37 live lanes, lanes 35 and 36 reset, 27 tail lanes shadow env n - 1
            keep = orc.reset_buf.astype(bool)
            orc.reset_buf[:] = np.where(keep, 1, plant)
            env.reset_buf.copy_(torch.from_numpy(orc.reset_buf.copy()))
        env.step_raw(acts_d[t])
        orc.step(acts[t])
        for name in ("obs_buf", "states_buf", "rew_buf", "reset_buf"):
            assert_bits_equal(getattr(env, name).cpu().numpy(), getattr(orc, name), f"{form} step {t} {name}")
        gb, ob = env.get_state().cpu().numpy(), orc.get_state().view(np.float32)
        assert_bits_equal(gb[:20], ob[:20], f"{form} step {t} state fields 0..19")
        assert_bits_equal(gb[26:], ob[26:], f"{form} step {t} state fields 26..")


def test_battery_mailbox_wait_path():
    """The role-wavefront battery server normally posts each voltage long before wavefront 0 needs it; with the server delayed on purpose
    (taco_test_slow_battery_server of the -DTACO_TEST_HOOKS build; the product library has no such hook) wavefront 0 has to take the
    mailbox's wait path every substep -- the results must not change."""
    from taco_amd import _lib
    hooks = _lib.load_test_hooks()
    assert not hasattr(_lib.load(), "taco_test_slow_battery_server"), "the product library must not carry test hooks"
    run_pair(config.baseline_config(1, num_envs=512), steps=40, lib=hooks, slow_server=True)


def test_random_configurations():
    """a bounded slice of tools/fuzz.py: random task / size / flag / stack-length / clip / substep combinations, every word bit-exact
    (the full generator ran 1 500 cases clean on the GPU box; see profiles/README.md)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz
    rng = np.random.default_rng(2024)
    for i in range(40):
        cfg, steps, form = fuzz.draw_case(rng)   # (a quarter of the small cases pin a one-lane kernel form)
        try:
            run_pair(cfg, steps, seed=i, check_every=1 if steps > 8 else 2, hover_bias=bool(i & 1), api=cfg["env"]["numEnvs"] <= 20000, form=form)   # (+ what VecTask.step() returns)
        except AssertionError as e:
            raise AssertionError(f"case {i}: form={form} {fuzz.describe(cfg)}: {e}") from e


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("TACO_FUZZ_CASES"), reason="opt-in: TACO_FUZZ_CASES=1500 python -m pytest tests -m gpu -k full_fuzz (about 25 min)")
def test_full_fuzz_run():
    """The whole generator of tools/fuzz.py (the run recorded in profiles/README.md), as a test: TACO_FUZZ_CASES cases from TACO_FUZZ_SEED (0)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz
    cases, seed = int(os.environ["TACO_FUZZ_CASES"]), int(os.environ.get("TACO_FUZZ_SEED", "0"))
    rng = np.random.default_rng(seed)
    for i in range(cases):
        cfg, steps, form = fuzz.draw_case(rng)
        try:
            run_pair(cfg, steps, seed=i, check_every=1 if steps > 8 else 2, hover_bias=bool(i & 1), form=form)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} case {i}: form={form} {fuzz.describe(cfg)}: {e}") from e
        if i % 50 == 49:
            print(f"fuzz: {i + 1} / {cases} cases clean", flush=True)


@pytest.mark.parametrize("form", ["auto", "quad", "lane", "lane_throughput", "lane_roles", "quad_served"])
def test_loop_forms_hand_over_on_rare_forms_and_nonfinite_inputs(form):
    """The substep loop runs a FIN form (no NaN-carrying selects) and a PLAIN form (compile-time flags, unrolled integrator) while a
    wave-uniform predicate says they are the same function as the exact form.  Two stress cases against the oracle, in every
    instantiation: (1) an airframe with 1/3000 of the inertia -- body rates shoot past the integrator's common form within a few
    substeps, at different substeps in different envs (hand-over in mid-step), and run on to inf / NaN; (2) NaN, inf and huge
    actions (the predicate is false from the first substep for the wavefronts that consume them, `delay_time` ms later)."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    n = 300
    # (1) tiny inertia
    for scale in (1 / 300.0, 1 / 3000.0, 1 / 100000.0):
        cfg = config.baseline_config(1, num_envs=n)
        _, J = config.composite_body()
        cfg["inertia"] = tuple(j * scale for j in J)
        cfg["env"]["maxEpisodeLength"] = 40
        run_pair(cfg, steps=60, seed=3, hover_bias=True, form=form)
    # (2) non-finite / huge actions in some envs, some steps
    cfg = config.baseline_config(1, num_envs=n)
    cfg["env"]["maxEpisodeLength"] = 50
    flat = config.flat_cfg(cfg)
    env = FpvBase(cfg, copy_outputs=False, kernel_form=form)
    orc = O.OracleEnv(flat, threads=8)
    acts = action_stream(n, 70, 5)
    rng = np.random.default_rng(9)
    bad = np.array([np.nan, np.inf, -np.inf, 1e35, -3e37, 1e30, 5e29], np.float32)
    for t in range(5, 70, 3):
        rows = rng.integers(0, n, 12)
        acts[t, rows, rng.integers(0, 4, 12)] = bad[rng.integers(0, len(bad), 12)]
    acts_d = torch.from_numpy(acts).cuda()
    for t in range(70):
        env.step_raw(acts_d[t])
        orc.step(acts[t])
        what = f"non-finite actions, step {t}"
        assert_bits_equal(env.obs_buf.cpu().numpy(), orc.obs_buf, what + " obs")
        assert_bits_equal(env.rew_buf.cpu().numpy(), orc.rew_buf, what + " rew")
        assert_bits_equal(env.reset_buf.cpu().numpy(), orc.reset_buf, what + " reset")
        gb, ob = env.get_state().cpu().numpy(), orc.get_state().view(np.float32)
        assert_bits_equal(gb[:20], ob[:20], what + " state 0..19")
        assert_bits_equal(gb[26:65], ob[26:65], what + " state 26..64")


def test_reset_done_resets_now_like_the_reference():
    """VecTask.reset_done (VT:363-375) = reset_idx on the flagged envs outside a step: fresh state and command, reset_buf / progress_buf
    cleared, the step clock unchanged -- and the step that follows applies thrust to those envs (it sees no reset flag).  HIP vs oracle,
    every state word, through several episodes' worth of interleaved step() / reset_done() calls; the progress == 500 command re-draw
    of reset_command_condition (FA:500-503) is exercised by planting progress 500 through the state blob."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    cfg = config.baseline_config(4, num_envs=777)            # mix, every randomisation on, 5 state frames
    cfg["env"]["maxEpisodeLength"] = 37
    flat = config.flat_cfg(cfg)
    n = flat["num_envs"]
    env, orc = FpvBase(cfg, copy_outputs=False), O.OracleEnv(flat, threads=8)
    acts = action_stream(n, 160, 3)
    acts_d = torch.from_numpy(acts).cuda()

    def same(what):
        gb, ob = env.get_state().cpu().numpy(), orc.get_state().view(np.float32)
        flip_envs = np.arange(n) >= int(n / 3 * 2)
        assert_bits_equal(gb[:20], ob[:20], what + " state fields 0..19")
        assert_bits_equal(gb[26:67], ob[26:67], what + " state fields 26..66")
        assert_bits_equal(gb[20:26][:, flip_envs], ob[20:26][:, flip_envs], what + " rpy of the flip envs")
        assert_bits_equal(gb[67:], ob[67:], what + " delay line")
        assert_bits_equal(env.reset_buf.cpu().numpy(), orc.reset_buf, what + " reset_buf")

    resets = 0
    for t in range(160):
        env.step_raw(acts_d[t]); orc.step(acts[t])
        if t % 7 == 3:                                       # every few steps: reset the done envs NOW instead of inside the next step
            if t == 80:                                      # plant progress 500 in some envs that are not resetting
                for side in (env, orc):
                    b = side.get_state()
                    b = b.clone() if torch.is_tensor(b) else b.copy()
                    v = b.view(torch.int32) if torch.is_tensor(b) else b.view(np.int32)
                    v[65, 600:700] = 500
                    side.set_state(b)
            pending = int(orc.reset_buf.sum())
            _, ids = env.reset_done()
            oids = orc.reset_done()
            assert np.array_equal(ids.cpu().numpy(), oids) and len(oids) == pending
            assert int(env.reset_buf.sum()) == 0 and int(orc.reset_buf.sum()) == 0
            assert env.step_count == t + 1                   # the clock did not move
            same(f"after reset_done at step {t}")
            prog = env.progress_buf.cpu().numpy()
            assert (prog[oids] == 0).all()
            resets += pending
        if t % 5 == 0:
            same(f"step {t}")
            assert_bits_equal(env.obs_buf.cpu().numpy(), orc.obs_buf, f"step {t} obs")
            assert_bits_equal(env.rew_buf.cpu().numpy(), orc.rew_buf, f"step {t} rew")
    assert resets > 50


# ---------------------------------------------------------------------------------------------------------------------------------------
# The chain HIP kernel -> the reference's own VecTask.step(), closed directly: the glue fixtures (tests/golden/glue_*.npz, traces of
# fpv_asymmetry.py + vec_task_asymmetry.py running on CPU, tests/golden/make_glue_golden.py) replayed through the HIP env -- same cfg,
# seed and actions -- and compared with what the REFERENCE held after every recorded step, with no oracle in between.  EXACT: flags,
# progress, delay length, the whole 400-word pending-action line, actions, parameters, pos / rotate commands; continuous state within the
# tolerances of tests/util.py::GLUE_TOL.  Both arithmetic modes of the angular rate (see tests/test_oracle_golden.py): the product's
# rounds 1-3 default ("carried") and the reference's literal data flow ("roundtrip", cfg["world_rate_roundtrip"] = True: the default since round 4), the latter also bit for
# bit against the oracle in the same mode.  Measured maxima, HIP vs the reference's run (identical to the oracle's, as they must be):
#   carried  : root state 1.6e-3, obs / states 5.2e-4, reward 1.4e-6        roundtrip: 3.1e-4, 1.0e-4, 3.4e-7 (round 5; both fixture families alike:
#   the drift is seeded by the attitudes a reset draws through libm's sin / cos on the reference's side, not by the battery's sqrt)
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,mode", [(c, m) for c in GLUE_CASES for m in ("carried", "roundtrip")] + [(c, "roundtrip") for c in GLUE_CASES_IEEE])
def test_hip_kernel_equals_the_references_own_step(case, mode, golden):
    from taco_amd.vec_env import FpvBase
    from util import check_against_glue_fixture, check_glue_trace_content, glue_case
    g = golden("glue_" + case)
    cfg, acts, rec_steps = glue_case(g)
    cfg["world_rate_roundtrip"] = mode == "roundtrip"
    cfg["record_flag"] = True   # rows 20..25 (copter_rpy_old / _continuous) of every env are compared, as the reference maintains them
    env = FpvBase(cfg, copy_outputs=False)
    acts_d = torch.from_numpy(acts).cuda()
    seen, maxima = {}, {}
    for t in range(acts.shape[0]):
        env.step_raw(acts_d[t])
        if t in rec_steps:
            check_against_glue_fixture(g, rec_steps[t], f"HIP glue_{case} [{mode}] step {t}", env.get_state().cpu().numpy(), env.obs_buf.cpu().numpy(),
                                       env.states_buf.cpu().numpy(), env.rew_buf.cpu().numpy(), env.reset_buf.cpu().numpy(),
                                       env.timeout_buf.cpu().numpy(), maxima, seen)
    check_glue_trace_content(case, seen)
    env.check()
    print(f"HIP glue_{case} [{mode}] maxima: " + ", ".join(f"{k} {v:.2e}" for k, v in maxima.items()))


# Teacher-forced ONE-STEP parity of the HIP kernel with the reference's own step() (tests/util.py::teacher_forced_one_step): north_star's
# "within 1e-5 of the reference", asserted for the default arithmetic mode on every consecutive pair of recorded steps of all eleven traces
# (incl. the flag sets of BASELINE configs 2-5), from the REFERENCE's state: blob + frame stacks + reset flags + step counter loaded through
# the C ABI (taco_set_state / taco_set_step_count), one taco_step, compared with what the reference held next.  What a reset draws (E_c, the
# initial rotor speeds) is compared bit for bit through taco_reset_done.
class _HipOneStep:
    def __init__(self, cfg, mode):
        from taco_amd.vec_env import FpvBase
        cfg["world_rate_roundtrip"] = mode == "roundtrip"
        cfg["record_flag"] = True
        self.env = FpvBase(cfg, copy_outputs=False)

    def load(self, blob, obs, states, reset, t):
        e = self.env
        e.set_state(torch.from_numpy(np.ascontiguousarray(blob).view(np.float32)))
        e.obs_buf.copy_(torch.from_numpy(obs))
        e.states_buf.copy_(torch.from_numpy(states))
        e.reset_buf.copy_(torch.from_numpy(reset))
        e.step_count = t

    def _blob(self):
        return self.env.get_state().cpu().numpy()

    def step(self, a):
        e = self.env
        e.step_raw(torch.from_numpy(a).cuda())
        return self._blob(), e.obs_buf.cpu().numpy(), e.states_buf.cpu().numpy(), e.rew_buf.cpu().numpy(), e.reset_buf.cpu().numpy(), e.timeout_buf.cpu().numpy()

    def reset_now(self):
        self.env.reset_done()
        return self._blob()


@pytest.mark.parametrize("mode", ["roundtrip", "carried"])
@pytest.mark.parametrize("case", GLUE_CASES)
def test_hip_one_step_from_the_references_own_state(case, mode, golden):
    from util import ONE_STEP_TOL, ONE_STEP_TOL_CARRIED, exact_table, glue_case, teacher_forced_one_step
    g = golden("glue_" + case)
    cfg, _, _ = glue_case(g)
    impl = _HipOneStep(cfg, mode)
    exact = {}
    maxima, cnt = teacher_forced_one_step(g, impl, f"HIP glue_{case} [{mode}]", ONE_STEP_TOL if mode == "roundtrip" else ONE_STEP_TOL_CARRIED, exact=exact)
    impl.env.check()
    assert cnt["pairs"] >= 60
    print(f"HIP one step, glue_{case} [{mode}] {cnt}: " + ", ".join(f"{k} {v:.1e}" for k, v in maxima.items()))
    print(exact_table(f"HIP glue_{case} [{mode}] (ABSOLUTE deviations in the last column)", exact))


# The same against the IEEE-sqrt family (tests/util.py EXACT_GROUPS; glue_*_ieee.npz: the reference with a correctly rounded square root in its
# battery model, i.e. its real platform's sqrtf): ONE step of the HIP kernel from the reference's own state is BIT-EQUAL to the reference's own
# step in root state, PID memory, battery, rotor speeds and every libm-free word of the obs / states stacks, for every env that did not reset
# in the step; what sits behind a libm call of the reference is reported in ulp.  The table (incl. ABSOLUTE deviations) is printed with -s.
@pytest.mark.parametrize("case", GLUE_CASES_IEEE)
def test_hip_one_step_is_bit_exact_against_the_ieee_sqrt_reference(case, golden):
    from util import ONE_STEP_TOL, exact_table, glue_case, teacher_forced_one_step
    g = golden("glue_" + case)
    cfg, _, _ = glue_case(g)
    impl = _HipOneStep(cfg, "roundtrip")
    exact = {}
    maxima, cnt = teacher_forced_one_step(g, impl, f"HIP glue_{case}", ONE_STEP_TOL, exact=exact, assert_exact=True)
    impl.env.check()
    assert cnt["pairs"] >= 60
    for name in ("root state", "PID memory", "battery state / voltage", "rotor speeds (~300 rev/s)", "obs stack, libm-free words", "states stack, libm-free words"):
        assert exact[name]["differ (no reset)"] == 0 and exact[name]["words"] > 0
    print(f"HIP one step, glue_{case} {cnt}: " + ", ".join(f"{k} {v:.1e}" for k, v in maxima.items()))
    print(exact_table(f"HIP glue_{case}", exact))


@pytest.mark.parametrize("form", ["quad_roles", "quad", "lane", "lane_roles", "lane_throughput", "quad_served"])
def test_world_rate_roundtrip_mode_equals_the_oracle_bitwise(form):
    """the reference's literal data flow of the angular rate (TACO_F_WORLD_RATE_ROUNDTRIP): every instantiation against the oracle in the same mode"""
    cfg = config.baseline_config(4, num_envs=600)   # mix, every randomisation on, 5 state frames
    cfg["world_rate_roundtrip"] = True
    run_pair(cfg, steps=120, check_every=10, hover_bias=True, form=form)
    cfg = config.baseline_config(1, num_envs=300)
    cfg["world_rate_roundtrip"] = True
    run_pair(cfg, steps=150, check_every=10, form=form)


@pytest.mark.parametrize("task", ["pos", "rotate", "mix"])
def test_record_flag_tracks_rpy_continuous_for_every_env(task):
    """record_flag (fpv_asymmetry.py:113): rows 20..25 are live for pos / rotate envs too, as in the reference (FA:339-347) -- what the
    episode recorder logs (FA:662).  run_pair compares them for every env when the flag is set."""
    cfg = config.default_cfg(task, 500, record_flag=True, env_maxEpisodeLength=200)
    for form in ("auto", "lane"):
        run_pair(cfg, steps=260, check_every=20, hover_bias=True, form=form)


def test_reset_idx_resets_the_given_envs_now():
    """FpvBase.reset_idx(env_ids) (FA:475-517) called from outside a step: those envs get their fresh state at once (taco_reset_done on an id
    mask), reset_buf / progress_buf of them cleared, nothing else touched; HIP vs oracle word for word, then 30 more steps."""
    from oracle import oracle as O
    from taco_amd.vec_env import FpvBase
    cfg = config.baseline_config(4, num_envs=300)
    flat = config.flat_cfg(cfg)
    env, orc = FpvBase(cfg, copy_outputs=False), O.OracleEnv(flat, threads=4)
    acts = action_stream(300, 60, 5)
    acts_d = torch.from_numpy(acts).cuda()
    flip_envs = np.arange(300) >= 200
    for t in range(60):
        env.step_raw(acts_d[t]); orc.step(acts[t])
        if t in (10, 11, 29):
            ids = np.array([0, 17, 150, 299] if t != 11 else [17, 201])
            env.reset_idx(torch.from_numpy(ids).cuda()); orc.reset_idx(ids)
            assert env.step_count == t + 1
        gb, ob = env.get_state().cpu().numpy(), orc.get_state().view(np.float32)
        assert_bits_equal(gb[:20], ob[:20], f"step {t} state fields 0..19")
        assert_bits_equal(gb[26:], ob[26:], f"step {t} state fields 26.. + delay line")
        assert_bits_equal(gb[20:26][:, flip_envs], ob[20:26][:, flip_envs], f"step {t} rpy of the flip envs")
        assert_bits_equal(env.reset_buf.cpu().numpy(), orc.reset_buf, f"step {t} reset_buf")
        assert_bits_equal(env.obs_buf.cpu().numpy(), orc.obs_buf, f"step {t} obs")


def test_ducted_airframe_preset():
    """assets/xml/fpv_with_duct.xml:6 (0.570 kg, J = (1.09e-3, 1.47e-3, 1.226e-3)) through config.airframe(): the general (non-literal) constants"""
    cfg = config.default_cfg("pos", 500, **config.airframe("with_duct"))
    assert abs(config.flat_cfg(cfg)["mass"] - 0.5700008) < 1e-9
    for form in ("auto", "lane", "lane_throughput"):
        run_pair(cfg, steps=150, check_every=10, form=form)


# north_star's parity sentence -- "state trajectories within 1e-5 fp32 over 1 000 steps" -- against the reference's OWN 1 060-step runs
# (glue_long_*_ieee.npz, tests/golden/make_glue_golden.py cases 12-15: pos, rotate, mix with 5 state frames, mix with every flag of BASELINE config 5):
# the reference's task state at an anchor step loaded through the C ABI (taco_set_state / taco_set_step_count), the stored actions replayed open loop
# through taco_step, and after EVERY step root state, PID memory, battery, rotor speeds, bookkeeping, flags and every libm-free word of the newest
# obs / states frame BIT-EQUAL to what the reference held, for every env that has not reset since the anchor; euler-angle rows, the flip command,
# noised obs words and the reward (behind libm calls of the reference) within tests/util.py::LONG_TOL; the envs that did reset tabulated apart
# (reset step: what libm's sin / cos costs; afterwards: the closed loop).  Printed with -s.
@pytest.mark.parametrize("case,anchor", [(c, a) for c in GLUE_CASES_LONG for a in (0, 530)])
def test_hip_thousand_steps_from_the_references_own_state(case, anchor, golden):
    from util import glue_case, long_replay, long_table
    g = golden("glue_" + case)
    cfg, _, _ = glue_case(g)
    impl = _HipOneStep(cfg, "roundtrip")
    st, tab, summary = long_replay(g, impl, f"HIP glue_{case} from step {anchor}", anchor_step=anchor)
    impl.env.check()
    assert summary["steps"] >= (1000 if anchor == 0 else 500)
    print(long_table(f"HIP glue_{case} from step {anchor}", st, tab, summary))
