"""The CPU oracle (oracle/taco_oracle.c) against the golden vectors captured from the reference's own torch
sub-models (tests/golden/make_golden.py).  Tolerances, all stated here:

  * 0 ulp (bit-exact) wherever the reference's arithmetic is +,-,*,/ only;
  * <= 1 ulp where the reference's CPU build uses torch.sqrt (its vectorised sqrt is not correctly rounded;
    the oracle's is);
  * <= 5e-7 absolute on angles that go through atan2/asin/sin/cos (the oracle uses its own polynomial
    implementations so that CPU and GPU agree bit-for-bit; they are within 2 ulp of torch's);
  * integer done flags: exact.
"""
import numpy as np
import pytest

from oracle import oracle as O
from util import GLUE_CASES, GLUE_CASES_IEEE, GLUE_CASES_LONG, assert_bits_equal, assert_ulp

TWO_PI = np.float32(2 * np.pi)


def test_quaternion_helpers(golden):
    q = golden("quat")
    assert_bits_equal(O.quat_mul(q["q"], q["p"]), q["mul"], "quat_mul TU:19-40")
    assert_bits_equal(O.quat_rotate_inv(q["q"], q["v"]), q["rot_inv"], "quat_rotate(conj) TU:58-68")
    conj = q["q"].copy()
    conj[:, :3] *= -1
    assert_bits_equal(conj, q["conj"], "quat_conjugate TU:84-88")
    assert_bits_equal(O.quat_mul(conj, q["p"]), q["rel"], "relative_quat_body FA:356")
    assert_bits_equal(O.quat_to_matrix(q["q"]), q["mat"], "quaternion_to_matrix TJ:389-416")
    np.testing.assert_allclose(O.euler_xyz_v1(q["q"]), q["rpy"], rtol=0, atol=5e-7, err_msg="get_euler_xyz_v1 TU:175-196")
    np.testing.assert_allclose(O.quat_from_euler_xyz(q["e"]), q["from_euler"], rtol=0, atol=2e-7, err_msg="quat_from_euler_xyz TU:199-213")
    np.testing.assert_allclose(O.quat_diff_rad(q["q"], q["p"]), q["diff_rad"], rtol=0, atol=5e-7, err_msg="quat_diff_rad TJ:145-164")


def test_rate_pid(golden):
    p = golden("pid")
    T, n, _ = p["des"].shape
    prev = np.zeros((n, 3), np.float32)
    integ = np.zeros((n, 3), np.float32)
    for t in range(T):
        if t == int(p["reset_step"]):
            prev[p["reset_ids"]] = 0
            integ[p["reset_ids"]] = 0
        out = O.pid_step(float(p["dt"]), p["des"][t], p["cur"][t], prev, integ)
        assert_bits_equal(out, p["out"][t], f"pid out t={t}")
        assert_bits_equal(prev, p["prev"][t], f"pid previous_error t={t}")
        assert_bits_equal(integ, p["integ"][t], f"pid integral t={t}")
    # SURVEY Appendix A smoke values
    prev = np.zeros((1, 3), np.float32); integ = np.zeros((1, 3), np.float32)
    o1 = O.pid_step(0.001, [[1, -2, 0.5]], [[0.2, 0.1, -0.3]], prev, integ)
    np.testing.assert_allclose(o1, [[8.8, -42.0, 64.0]], rtol=1e-6)
    o2 = O.pid_step(0.001, [[1, -2, 0.5]], [[0.25, 0.0, -0.3]], prev, integ)
    np.testing.assert_allclose(o2, [[-1.75, -20.0, 64.0]], rtol=1e-4)


def test_allocator_and_real2sim(golden):
    a = golden("alloc")
    thr, u_after = O.allocator(a["u"])
    assert_bits_equal(thr, a["thr"], "control_allocator CTRL/fpv_dynamics.py:35-46")
    assert_bits_equal(u_after, a["u_after"], "u[:,3] in-place clip")
    fs, ts = O.real2sim(a["f_real"], a["t_real"])
    assert_bits_equal(fs, a["f_sim"], "real2sim forces")
    assert_bits_equal(ts, a["t_sim"], "real2sim torques")
    thr, _ = O.allocator([[500., 10, -20, 400], [900, 300, 300, 300], [50, 0, 0, 0], [1000, 0, 0, 0]])
    np.testing.assert_array_equal(thr, [[220, 760, 280, 740], [100, 100, 100, 1000], [100] * 4, [1000] * 4])


def test_battery(golden):
    b = golden("battery")
    E = b["E0"].copy(); u1 = np.zeros_like(E); t = np.zeros_like(E)
    Vs = []
    for k in range(b["Pm"].shape[0]):
        Vs.append(O.battery_step(True, float(b["dt"]), b["Pm"][k], E, u1, t))
    Vs = np.stack(Vs)[b["V_keep_idx"]]
    assert np.isnan(b["V"]).any(), "fixture must contain the sqrt(<0) NaN case"
    assert_ulp(Vs, b["V"], 1, "battery voltage (torch.sqrt is not correctly rounded)")
    assert_bits_equal(E, b["E_end"], "E_c after 2000 steps")
    assert_bits_equal(u1, b["u1_end"], "u_1 after 2000 steps")
    assert_bits_equal(t, b["t_end"], "time after 2000 steps")
    z = np.zeros(4, np.float32)
    assert_bits_equal(O.battery_step(False, 0.001, np.ones(4, np.float32), z.copy(), z.copy(), z.copy()), b["V_off"], "battery off = 26.1 V")


def test_rotor_first_order_lag(golden):
    r = golden("rotor")
    om = r["om0"].copy()
    oms = []
    for k in range(r["thr"].shape[0]):
        O.rotor_step(r["V0"] + np.float32(0.001 * k), r["thr"][k], r["tau"], r["para"], om)
        oms.append(om.copy())
    assert_bits_equal(np.stack(oms)[r["keep_idx"]], r["omega"], "RotorDynamics.sim_process CTRL/thrust_dynamics.py:98-104")


def test_aero(golden):
    a = golden("aero")
    rf, rt, bf = O.aero(a["cf_ct"], a["d"], a["kt"], a["vb"], a["om"])
    assert_bits_equal(rf, a["rf"], "rotor force")
    assert_bits_equal(rt, a["rt"], "rotor torque")
    assert_bits_equal(bf, a["bf"], "body drag force")
    assert not a["bt"].any()


def test_rewards_and_done_flags(golden):
    w = golden("reward")
    ml = float(w["max_len"])
    r, d = O.reward_pos(w["rel_pos_b"], w["pos"], w["q"], w["qt"], w["prog"], ml)
    np.testing.assert_allclose(r, w["r_pos"], rtol=1e-6, atol=0)
    np.testing.assert_array_equal(d, w["d_pos"])
    r, d = O.reward_rotate(w["rel_pos"], w["rel_v"], w["pos"], w["q"], w["cmd"], w["prog"], ml)
    assert_ulp(r, w["r_rot"], 2, "rotating reward")
    np.testing.assert_array_equal(d, w["d_rot"])
    r, d = O.reward_flip(w["rel_pos_b"], w["relq"], w["pos"], w["cmd"], w["prog"], ml)
    assert_ulp(r, w["r_flip"], 1, "flip reward")
    np.testing.assert_array_equal(d, w["d_flip"])
    assert w["d_pos"].any() and not w["d_pos"].all(), "fixture must exercise both flag values"


def test_observation_frame(golden):
    o = golden("obs")
    for task, key in ((0, "f_pos"), (1, "f_rot"), (2, "f_flip")):
        fr, fc = O.obs_frame(task, o["p"], o["q"], o["v"], o["w"], o["pt"], o["qt"], o["V"], o["act"], o["cmd"], o["flip_radian"], o["roll_cont"])
        assert_bits_equal(fr, o[key], f"26-D frame {key} FA:415-421")
    assert_bits_equal(fc, o["flip_cmd"], "flip command FA:831-832")


def test_rand_float_and_reset_quats(golden):
    import math
    r = golden("reset")
    for col, (name, lo, hi) in enumerate((("pm2", -2.0, 2.0), ("pmpi", -math.pi, math.pi), ("u0_400", 0.0, 400.0), ("u0_2p2", 0.0, 2.2),
                                          ("pm6", -6.0, 6.0), ("dr095", 1 - 0.05 * 0.7, 1 + 0.05 * 0.7), ("tau", 0.017 - 0.001, 0.017 + 0.001),
                                          ("noise", 1 - 10 / 700, 1 + 10 / 700))):
        assert_bits_equal(O.rand_float(lo, hi, r["u"][:, col]), r[name], f"torch_rand_float {name} TU:216-219")
    z = np.zeros(len(r["e"]), np.float32)
    np.testing.assert_allclose(O.quat_from_euler_xyz(r["e"]), r["quat_xyz"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(O.quat_from_euler_xyz(np.stack([r["e"][:, 0], z, z], 1)), r["quat_roll_only"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(O.quat_from_euler_xyz(np.stack([z, z, r["e"][:, 2]], 1)), r["quat_yaw_only"], rtol=0, atol=2e-7)


def test_composed_substep_chain(golden):
    """rows C -> H'' chained exactly as mid_physics_step does, over a prescribed rigid-body trajectory."""
    c = golden("chain")
    T, n = c["q"].shape[:2]
    dt = float(c["dt"])
    rpy_old = O.euler_xyz_v1(c["q"][0]); rpy_cont = rpy_old.copy()
    prev = np.zeros((n, 3), np.float32); integ = np.zeros((n, 3), np.float32)
    E = c["E0"].copy(); u1 = np.zeros(n, np.float32); tt = np.zeros(n, np.float32); om = c["om0"].copy()
    for t in range(T):
        q, v, w, a = c["q"][t], c["v"][t], c["w"][t], c["act"][t]
        rpy = O.euler_xyz_v1(q)
        d = rpy - rpy_old
        d = np.where(d > 1, d - TWO_PI, d); d = np.where(d < -1, d + TWO_PI, d)
        rpy_cont = rpy_cont + d; rpy_old = rpy
        vb = O.quat_rotate_inv(q, v); wb = O.quat_rotate_inv(q, w)
        u = np.zeros((n, 4), np.float32)
        u[:, 0] = (a[:, 0] + np.float32(1)) / np.float32(2) * np.float32(1000)
        u[:, 1:] = O.pid_step(dt, a[:, 1:] * np.float32(20), wb, prev, integ)
        thr, u = O.allocator(u)
        V = O.battery_step(True, dt, O.power(om), E, u1, tt)
        O.rotor_step(V, thr, c["tau"], c["para"], om)
        rf, rt, bf = O.aero(c["cf_ct"], c["d"], c["kt"], vb, om)
        fs, ts = O.real2sim(rf, rt)
        np.testing.assert_allclose(rpy, c["rpy"][t], rtol=0, atol=5e-7)
        np.testing.assert_allclose(rpy_cont, c["rpy_cont"][t], rtol=0, atol=1e-5)
        assert_bits_equal(vb, c["vb"][t], f"v_body t={t}"); assert_bits_equal(wb, c["wb"][t], f"w_body t={t}")
        assert_bits_equal(u, c["u"][t], f"u t={t}"); assert_bits_equal(thr, c["thr"][t], f"throttle t={t}")
        assert_ulp(V, c["V"][t], 1, f"V t={t}")
        np.testing.assert_allclose(om, c["om"][t], rtol=3e-7); np.testing.assert_allclose(fs, c["f_sim"][t], rtol=6e-7)
        np.testing.assert_allclose(ts, c["t_sim"][t], rtol=6e-7); assert_bits_equal(bf, c["body_f"][t], f"body force t={t}")
    assert np.abs(c["rpy_cont"][-1] - c["rpy"][-1]).max() > 3, "fixture must cross the +-pi euler wrap"
    np.testing.assert_allclose(E, c["E_end"], rtol=1e-7)


def test_gae_returns_and_advantage_vs_reference_buffer(golden):
    """orc_gae / orc_normalize_advantage vs PPOReplayBuffer.compute_returns_and_advantage run on CPU (gae.npz).
    ret is bit-exact; the golden `adv_raw` is ret - value (one extra rounding), the normalised advantage depends on
    torch's reduction order -> 1e-6 absolute on O(1) values."""
    g = golden("gae")
    assert np.array_equal(g["done_f32"], g["done"].astype(np.float32)) and np.array_equal(g["rew_buf"], g["rew"])
    adv, ret = O.gae(g["rew"], g["done_f32"], g["value"], g["last_value"], float(g["gamma"]), float(g["lam"]))
    assert_bits_equal(ret, g["ret"], "ret_buf")
    np.testing.assert_allclose(adv, g["adv_raw"], rtol=0, atol=1.5e-7)
    np.testing.assert_allclose(O.normalize_advantage(adv), g["adv_norm"], rtol=0, atol=1e-6)
    # pure-numpy restatement of the recurrence (fp32, same association) as a second witness
    H, N = g["rew"].shape
    gm, lm = np.float32(g["gamma"]), np.float32(g["lam"])
    last = np.zeros(N, np.float32)
    for t in reversed(range(H)):
        nv = g["last_value"] if t == H - 1 else g["value"][t + 1]
        nnt = np.float32(1) - g["done_f32"][t]
        delta = (g["rew"][t] + nnt * gm * nv) - g["value"][t]
        last = delta + nnt * gm * lm * last
        assert_bits_equal(last, adv[t], f"adv step {t}")


def test_policy_forward_vs_reference_module(golden):
    """orc_policy_act vs PPO_ActorCritic.act / evaluate run on CPU (policy.npz: widths 26-64-40-4, LSTM 26->24 over 5 frames,
    24-48-1; weights from the reference's own initialisation).  Tolerance 1e-5: the reference goes through the platform BLAS and
    vectorised tanh/sigmoid, the oracle through the kernel's fma chain order and its own exp/tanh."""
    from taco_amd import policy as P
    g = golden("policy")
    sd = {k[3:]: g[k] for k in g if k.startswith("sd.")}
    cfg = P.cfg_from_state_dict(sd, 1, 5)
    assert (list(cfg.actor_hidden)[:cfg.n_actor_hidden], cfg.lstm_hidden, list(cfg.critic_hidden)[:cfg.n_critic_hidden]) == ([64, 40], 24, [48])
    blob = P.pack_state_dict(cfg, sd)
    oc = O.policy_cfg(1, 5, [64, 40], 24, [48])
    assert blob.size == O.policy_blob_floats(oc)
    a, lp, v, mu, sg = O.policy_act(oc, blob, g["obs"], g["states"], deterministic=True)
    np.testing.assert_allclose(mu, g["mu"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(v, g["value"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(lp, g["logp_det"], rtol=0, atol=1e-5)
    assert_bits_equal(a, mu, "deterministic action == mean") and None
    assert_bits_equal(sg, g["sigma"], "sigma = log_std repeated")
    a, lp, v, mu, sg = O.policy_act(oc, blob, g["obs"], g["states"], eps=g["eps"])
    np.testing.assert_allclose(a, g["action_s"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(lp, g["logp_s"], rtol=1e-6, atol=1e-5)


def test_policy_forward_vs_reference_module_documented_widths(golden):
    """The same at the documented widths (policy_documented.npz: actor 26-128-128-128-4, critic LSTM 26->128 over 5 frames + 128-128-128-1),
    the architecture the batched critic kernels are specialised for; incl. an all-zero stack and a stack with zero leading frames."""
    from taco_amd import policy as P
    g = golden("policy_documented")
    sd = {k[3:]: g[k] for k in g if k.startswith("sd.")}
    cfg = P.cfg_from_state_dict(sd, 1, 5)
    assert (list(cfg.actor_hidden)[:cfg.n_actor_hidden], cfg.lstm_hidden, list(cfg.critic_hidden)[:cfg.n_critic_hidden]) == ([128, 128, 128], 128, [128, 128])
    oc = O.policy_cfg(1, 5, [128, 128, 128], 128, [128, 128])
    a, lp, v, mu, sg = O.policy_act(oc, P.pack_state_dict(cfg, sd), g["obs"], g["states"], deterministic=True)
    np.testing.assert_allclose(mu, g["mu"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(v, g["value"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(lp, g["logp_det"], rtol=0, atol=1e-5)
    assert np.abs(g["value"]).max() > 0.05 and np.abs(g["mu"]).max() > 0.3     # the fixture is not degenerate


def test_policy_transcendentals_and_noise():
    x = np.linspace(-30, 30, 6001).astype(np.float32)
    xd = x.astype(np.float64)
    assert np.max(np.abs(O.expf(x).astype(np.float64) - np.exp(xd)) / np.exp(xd)) < 2e-7
    assert np.max(np.abs(O.tanhf(x).astype(np.float64) - np.tanh(xd))) < 1.5e-7
    assert np.max(np.abs(O.sigmoidf(x).astype(np.float64) - 1 / (1 + np.exp(-xd)))) < 1.5e-7
    e = O.expf([-200.0, 200.0, np.nan])         # the argument is clamped: straight-line code, saturating instead of 0 / inf
    assert 0 < e[0] < 1.3e-38 and e[1] > 3.3e38 and np.isnan(e[2]) and np.isnan(O.tanhf([np.nan])[0]) and np.isnan(O.sigmoidf([np.nan])[0])
    assert np.array_equal(O.tanhf([100.0, -100.0, 0.0]), np.array([1, -1, 0], np.float32)) and np.array_equal(O.sigmoidf([200.0]), np.array([1], np.float32))
    assert O.sigmoidf([-200.0])[0] < 1e-38 and O.sigmoidf([-80.0])[0] > 0
    e = O.policy_noise(3, 1, 100000)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01 and abs(np.corrcoef(e[:, 0], e[:, 1])[0, 1]) < 0.01
    assert not np.array_equal(e, O.policy_noise(3, 2, 100000)) and np.array_equal(e, O.policy_noise(3, 1, 100000))


# ---------------------------------------------------------------------------------------------------------------------------------------
# The task GLUE, pinned to the reference's own fpv_asymmetry.py / vec_task_asymmetry.py: tests/golden/make_glue_golden.py runs the
# reference's VecTask.step() on CPU (stub modules for the Isaac Gym imports, row I plugged in as gym.simulate, the build's uniforms fed
# through the reference's own random-number expressions) and stores what the task holds after every step.  The oracle replays the same
# actions; tests/util.py::check_against_glue_fixture states what must be EXACT (control flow, bookkeeping, copies: reset dispatch and
# ordering, what a reset zeroes, the mask write and its truncation at slot 100, the 10-slot shift with the stale tail, the delayed-action
# index, the command re-draw at progress 500, FpvMix's thirds) and the tolerances of the continuous state.
# Both arithmetic modes of the angular rate are held against the fixtures: "roundtrip" = the reference's literal data flow (cfg key
# world_rate_roundtrip / TACO_F_WORLD_RATE_ROUNDTRIP: the body rates pass through the root state after every simulate(), FA:350) and
# "carried" = the product's DEFAULT (body rates carried across the ten substeps: one rounding per substep apart, amplified by the closed
# loop).  Same exact set, same tolerances.  Measured maxima over the six traces (this container, printed with -s):
#                       root state  rpy      PID      battery  rotor    cmd      obs/states  reward
#   roundtrip           3.8e-4      1.3e-4   6.1e-5   5.7e-6   1.2e-3   1.5e-5   1.3e-4      6.5e-7
#   carried (default)   1.6e-3      2.5e-4   2.0e-4   1.1e-5   4.7e-3   5.0e-5   5.2e-4      1.4e-6
# The same comparison runs on the GPU with the HIP kernel in the oracle's place (tests/test_parity_gpu.py).
# ---------------------------------------------------------------------------------------------------------------------------------------
GLUE_MAXIMA = {}


@pytest.mark.parametrize("case,mode", [(c, m) for c in GLUE_CASES for m in ("roundtrip", "carried")] + [(c, "roundtrip") for c in GLUE_CASES_IEEE])
def test_task_glue_equals_the_references_own_step(case, mode, golden):
    from oracle import oracle as O
    from taco_amd import config
    from util import check_against_glue_fixture, check_glue_trace_content, glue_case
    g = golden("glue_" + case)
    cfg, acts, rec_steps = glue_case(g)
    cfg["world_rate_roundtrip"] = mode == "roundtrip"
    orc = O.OracleEnv(config.flat_cfg(cfg), threads=4)
    seen, maxima = {}, GLUE_MAXIMA.setdefault(mode, {})
    for t in range(acts.shape[0]):
        obs, states, rew, done, tmo = orc.step(acts[t])
        if t in rec_steps:
            check_against_glue_fixture(g, rec_steps[t], f"glue_{case} [{mode}] step {t}", orc.get_state(), obs, states, rew, done, tmo, maxima, seen)
    check_glue_trace_content(case, seen)
    print(f"glue_{case} [{mode}] running maxima: " + ", ".join(f"{k} {v:.2e}" for k, v in maxima.items()))


# ---------------------------------------------------------------------------------------------------------------------------------------
# Teacher-forced ONE-STEP parity with the reference's own step() (tests/util.py::teacher_forced_one_step): north_star's 1e-5, asserted for the
# default arithmetic mode (world-rate round trip) on every consecutive pair of recorded steps of all eleven traces; the "carried" mode is held
# to its own, looser, bound.  Also pins what a reset draws (E_c, initial rotor speeds) bit for bit.
# ---------------------------------------------------------------------------------------------------------------------------------------
class _OracleOneStep:
    def __init__(self, cfg, mode):
        from taco_amd import config
        cfg["world_rate_roundtrip"] = mode == "roundtrip"
        self.e = O.OracleEnv(config.flat_cfg(cfg), threads=1)

    def load(self, blob, obs, states, reset, t):
        e = self.e
        e.set_state(blob)
        e.obs_buf[:], e.states_buf[:], e.reset_buf[:] = obs, states, reset
        e.step_count = t

    def step(self, a):
        o, s, r, d, tm = self.e.step(a)
        return self.e.get_state(), o, s, r, d, tm

    def reset_now(self):
        self.e.reset_done()
        return self.e.get_state()


@pytest.mark.parametrize("mode", ["roundtrip", "carried"])
@pytest.mark.parametrize("case", GLUE_CASES)
def test_one_step_from_the_references_own_state(case, mode, golden):
    from util import ONE_STEP_TOL, ONE_STEP_TOL_CARRIED, exact_table, glue_case, teacher_forced_one_step
    g = golden("glue_" + case)
    cfg, _, _ = glue_case(g)
    exact = {}
    maxima, cnt = teacher_forced_one_step(g, _OracleOneStep(cfg, mode), f"glue_{case} [{mode}]",
                                          ONE_STEP_TOL if mode == "roundtrip" else ONE_STEP_TOL_CARRIED, exact=exact)
    assert cnt["pairs"] >= 60
    print(f"one step, glue_{case} [{mode}] {cnt}: " + ", ".join(f"{k} {v:.1e}" for k, v in maxima.items()))
    print(exact_table(f"glue_{case} [{mode}] (ABSOLUTE deviations in the last column)", exact))


# The same against the IEEE-sqrt family (tests/util.py EXACT_GROUPS): with a correctly rounded square root in the reference's battery model
# (battery_dynamics.py:68 -- its real platform's sqrtf) one step of the oracle from the reference's own state is BIT-EQUAL to the reference's
# own step in root state, PID memory, battery, rotor speeds and every libm-free word of the obs / states stacks, for every env that did not
# reset in the step; what sits behind a libm call of the reference (reset attitudes, euler angles, the flip command, the noised obs words,
# the reward's asin) is reported in ulp.  Round-trip mode only: "carried" is not the reference's data flow.
@pytest.mark.parametrize("case", GLUE_CASES_IEEE)
def test_one_step_is_bit_exact_against_the_ieee_sqrt_reference(case, golden):
    from util import ONE_STEP_TOL, exact_table, glue_case, teacher_forced_one_step
    g = golden("glue_" + case)
    cfg, _, _ = glue_case(g)
    exact = {}
    maxima, cnt = teacher_forced_one_step(g, _OracleOneStep(cfg, "roundtrip"), f"glue_{case}", ONE_STEP_TOL, exact=exact, assert_exact=True)
    assert cnt["pairs"] >= 60
    for name in ("root state", "PID memory", "battery state / voltage", "rotor speeds (~300 rev/s)", "obs stack, libm-free words", "states stack, libm-free words"):
        assert exact[name]["differ (no reset)"] == 0 and exact[name]["words"] > 0
    print(f"one step, glue_{case} {cnt}: " + ", ".join(f"{k} {v:.1e}" for k, v in maxima.items()))
    print(exact_table(f"glue_{case}", exact))


# ---------------------------------------------------------------------------------------------------------------------------------------
# north_star's parity sentence ("state trajectories within 1e-5 fp32 over 1 000 steps") against the reference's OWN 1 060-step runs
# (glue_long_*_ieee.npz; tests/util.py::long_replay states what is bit-equal and what is held to 1e-5).  Here: the oracle; the HIP kernel
# takes the same test in tests/test_parity_gpu.py.
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,anchor", [(c, a) for c in GLUE_CASES_LONG for a in (0, 530)])
def test_thousand_steps_from_the_references_own_state(case, anchor, golden):
    from util import glue_case, long_replay, long_table
    g = golden("glue_" + case)
    cfg, acts, _ = glue_case(g)
    st, tab, summary = long_replay(g, _OracleOneStep(cfg, "roundtrip"), f"glue_{case} from step {anchor}", anchor_step=anchor)
    assert summary["steps"] >= (1000 if anchor == 0 else 500)
    print(long_table(f"glue_{case} from step {anchor}", st, tab, summary))
