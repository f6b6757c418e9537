"""SURVEY.md 8f row N1, second half, on the GPU: taco_policy_act (one MFMA launch for PPO_ActorCritic.act) against the CPU oracle
(bit-exact: the f32 MFMA is a k-ordered fma chain and the oracle follows the kernel's fragment order) and against the golden
vectors generated from the reference module (tolerance 1e-5, see tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

from util import assert_bits_equal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _golden_policy():
    from taco_amd import policy as P
    g = np.load(os.path.join(GOLD, "policy.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    return g, sd, P.ActorCritic(sd, 1, 5, seed=9)


def test_act_matches_reference_golden():
    g, sd, pol = _golden_policy()
    obs, st = torch.from_numpy(g["obs"]).cuda(), torch.from_numpy(g["states"]).cuda()
    action, logp, value, mu, sigma = pol.act(obs, st, deterministic=True)
    assert action.shape == (96, 4) and logp.shape == (96,) and value.shape == (96, 1) and sigma.shape == (96, 4)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(value.cpu().numpy()[:, 0], g["value"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp_det"], rtol=0, atol=1e-5)
    assert torch.equal(action, mu) and np.array_equal(sigma.cpu().numpy(), g["sigma"])
    assert torch.equal(pol.forward(obs), mu) and torch.equal(pol.act(obs, None, deterministic=True, action_only=True), mu)


@pytest.mark.parametrize("kind,bar", [("f16", 2e-6), ("bf16", 1e-4)])
@pytest.mark.parametrize("slots,n,states_len,scale", [(8, 256, 5, 1.0), (33, 1000, 5, 1.0), (7, 90, 5, 3.0), (5, 70, 3, 0.2), (9, 333, 8, 1.0), (6, 64, 2, 20.0)])
def test_split_precision_critic_stays_within_its_bar_of_the_f32_kernel(kind, bar, slots, n, states_len, scale):
    """ActorCritic(critic_split=...) -- the ring-form LSTM on the 16-bit matrix pipe with split operands (TACO_P_SPLIT_F16 / _BF16; "f16" is the host layer's default for the ring form):
    against the exact f32 critic on the same frame ring.  f16 halves: <= 2e-6 (the fast cell's own bar; measured ~2e-7); bf16 halves with
    three terms: measured 1e-5 ... 3e-5 (9e-5 on 20-sigma inputs) -- OUTSIDE 2e-6, which is why it is kept for the A/B record only; asserted
    against a loose 1e-4 so that the record stays reproducible.  (The bars scale with max(1, |value|): 20-sigma inputs give |value| ~ 2.)  Even / odd slot counts, ragged env groups, short and long stacks, small and 20-sigma inputs."""
    from taco_amd import policy as P
    rng = np.random.default_rng(slots * 100 + n)
    sd = _random_policy(rng, 1, states_len, [32], 128, [128, 128])
    frames = torch.from_numpy((rng.standard_normal((slots + states_len - 1, n, 26)) * scale).astype(np.float32)).cuda()
    frames[1, n // 2] = 0.0
    ref = P.ActorCritic(sd, 1, states_len, seed=3, exact_critic=True).values_ring(frames).cpu().numpy()
    got = P.ActorCritic(sd, 1, states_len, seed=3, critic_split=kind).values_ring(frames).cpu().numpy()
    assert got.shape == ref.shape == (slots, n, 1) and np.isfinite(got).all()
    d = np.abs(got.astype(np.float64) - ref).max()
    print(f"critic_split={kind} slots {slots} n {n} T {states_len} scale {scale}: values in [{ref.min():.3f}, {ref.max():.3f}], max |difference| {d:.2e}")
    if scale > 3.0:
        bar *= 5    # 20-sigma frames (pre-activations of ~100, |value| ~ 2): 3e-6 ... 5e-6 measured with f16 halves -- a stress case, not what an env produces
    assert d <= bar * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError):
        P.ActorCritic(sd, 1, states_len, exact_critic=True, critic_split=kind)


@pytest.mark.parametrize("states_dim", [17, 19, 30, 31, 32])
def test_split_precision_critic_other_state_widths(states_dim):
    """The split kernel stages the frames as they lie in the ring (LDS-DMA where 16-byte pieces allow, plain loads otherwise), splits them into its
    LDS planes and puts the bias into the weights' column `states_dim`: odd and even widths, 31 (the last free column) -- and 32, which has no free
    column: the host sends it to the f32 kernel (bit-identical to critic_split=None)."""
    from taco_amd import policy as P
    rng = np.random.default_rng(states_dim)
    sd = _random_policy(rng, 1, 4, [32], 128, [128, 128], states_dim=states_dim)
    frames = torch.from_numpy(rng.standard_normal((9, 123, states_dim)).astype(np.float32)).cuda()
    ref = P.ActorCritic(sd, 1, 4, seed=3, states_dim=states_dim, exact_critic=True).values_ring(frames).cpu().numpy()
    got = P.ActorCritic(sd, 1, 4, seed=3, states_dim=states_dim, critic_split="f16").values_ring(frames).cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    if states_dim == 32:
        f32 = P.ActorCritic(sd, 1, 4, seed=3, states_dim=states_dim, critic_split=None).values_ring(frames).cpu().numpy()
        assert_bits_equal(got, f32, "states_dim = 32: the split flag falls back to the f32 kernel")


def test_split_precision_critic_does_not_hide_out_of_range_frames():
    """include/taco_env.h TACO_P_SPLIT_F16 (ABI 8), round 5's verdict / advisor: the split-f16 critic (the host layer's default) must not turn a diverged
    env into a plausible value, nor clamp silently.  (a) a +-inf or NaN frame word -> the value of every stack holding it is NaN, exactly the rows the f32
    kernel poisons; (b) FINITE words of magnitude 1e5 (states that were never normalised) saturate at +-65 504 and are COUNTED
    (ActorCritic.clamped_words / .check; the gates they feed are saturated either way: the values stay within 2e-6 of the f32 kernel's);
    (c) in-range frames count nothing."""
    from taco_amd import _lib, policy as P
    rng = np.random.default_rng(77)
    T, slots, n = 5, 8, 96
    sd = _random_policy(rng, 1, T, [32], 128, [128, 128])
    base = rng.standard_normal((slots + T - 1, n, 26)).astype(np.float32)
    f32, split = P.ActorCritic(sd, 1, T, seed=3, critic_split=None), P.ActorCritic(sd, 1, T, seed=3)
    assert split.cfg.flags & P.P_SPLIT_F16
    # (c)
    fr = torch.from_numpy(base).cuda()
    split.values_ring(fr)
    assert split.clamped_words() == 0
    split.check()
    # (a) poison: frame 6 of env 10 is +inf, frame 3 of env 20 is -inf, frame 9 of env 30 NaN
    bad = base.copy()
    bad[6, 10, 4], bad[3, 20, 17], bad[9, 30, 0] = np.inf, -np.inf, np.nan
    fr = torch.from_numpy(bad).cuda()
    v_ref, v = f32.values_ring(fr).cpu().numpy()[:, :, 0], split.values_ring(fr).cpu().numpy()[:, :, 0]
    # every stack holding a poisoned frame, and no other row (the f32 kernel poisons a subset: an inf word can saturate all four gates to finite values there)
    assert np.isnan(v[2:7, 10]).all() and np.isnan(v[0:4, 20]).all() and np.isnan(v[5:8, 30]).all() and np.isnan(v).sum() == 5 + 4 + 3
    assert not (np.isnan(v_ref) & ~np.isnan(v)).any() and np.isnan(v_ref[5:8, 30]).all()
    ok = ~np.isnan(v)
    assert np.abs(v[ok] - v_ref[ok]).max() <= 2e-6 * max(1.0, np.abs(v_ref[ok]).max())
    assert split.clamped_words() == 0, "inf / NaN words poison, they are not counted as clamps"
    # (b) huge finite words
    big = base.copy()
    big[2, 5, 1], big[7, 40, 20], big[7, 41, 3] = 1.0e5, -3.0e5, 7.0e4
    fr = torch.from_numpy(big).cuda()
    v_ref, v = f32.values_ring(fr).cpu().numpy(), split.values_ring(fr).cpu().numpy()
    assert np.isfinite(v).all() and np.abs(v - v_ref).max() <= 2e-6 * max(1.0, np.abs(v_ref).max())
    cnt = split.clamped_words()
    assert 3 <= cnt <= 3 * ((T + 2) // 2 + 1), cnt          # three words; each staged by every two-slot block whose stacks hold its frame
    with pytest.raises(_lib.TacoError, match="saturated"):
        split.check()
    assert f32.clamped_words() == 0
    split.values_ring(torch.from_numpy(base).cuda())
    assert split.clamped_words() == 0, "the count is per call"


def test_split_precision_critic_large_magnitude_weights():
    """round 5's advisor: the 2e-6 bar was measured on random-init weights only.  Weights scaled up (x 4 / x 8 on the LSTM: saturated gates; x 3 / x 6 on
    the MLP: hidden activations of tens, |value| up to hundreds): the error relative to max(1, |value|) grows with the weights' magnitude -- the value is a
    sum of terms larger than itself, and the split operands carry 22 bits against f32's 24.  Printed next to the f32 fast cell's own distance from the exact
    cell (same weights): the growth is the MLP's gain on the cell's error, not the split -- asserted: split <= 4 x the fast cell's distance; random-init scale (1, 1) holds the plain 2e-6.  DESIGN.md section 4.3 quotes this table."""
    from taco_amd import policy as P
    for seed, lstm_scale, mlp_scale in ((0, 1.0, 1.0), (1, 4.0, 3.0), (2, 8.0, 1.0), (3, 1.0, 6.0)):
        rng = np.random.default_rng(seed)
        sd = _random_policy(rng, 1, 5, [32], 128, [128, 128])
        for k in sd:
            if k.startswith("critic_encoder."):
                sd[k] = (sd[k] * lstm_scale).astype(np.float32)
            elif k.startswith("critic_mlp."):
                sd[k] = (sd[k] * mlp_scale).astype(np.float32)
        frames = torch.from_numpy(rng.standard_normal((12, 200, 26)).astype(np.float32)).cuda()
        ref = P.ActorCritic(sd, 1, 5, seed=3, exact_critic=True).values_ring(frames).cpu().numpy()
        fast = P.ActorCritic(sd, 1, 5, seed=3, critic_split=None).values_ring(frames).cpu().numpy()
        pol = P.ActorCritic(sd, 1, 5, seed=3, critic_split="f16")
        got = pol.values_ring(frames).cpu().numpy()
        rel = lambda x: (np.abs(x.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))).max()
        print(f"weights x ({lstm_scale}, {mlp_scale}): values in [{ref.min():.2f}, {ref.max():.2f}]; |difference| / max(1, |value|) from the exact f32 critic: "
              f"split f16 {rel(got):.2e}, f32 fast cell {rel(fast):.2e}")
        assert np.isfinite(got).all() and pol.clamped_words() == 0
        # measured (round 6, r06_a): (1, 1) split 6.9e-7 / fast cell 4.8e-7; (4, 3) 1.8e-5 / 1.3e-5; (8, 1) 2.8e-6 / 9.6e-7; (1, 6) 3.2e-5 / 3.7e-5:
        # the MLP's gain amplifies the cell's ~1e-6 on h_T; the split operands add at most 3 x on top of what the f32 fast cell already differs by
        assert rel(got) <= 4.0 * max(rel(fast), 7e-7), "the split form stays within 4 x the f32 fast cell's own distance from the exact cell"
        if lstm_scale == mlp_scale == 1.0:
            assert rel(got) <= 2e-6


def test_split_precision_critic_matches_reference_golden_and_rollout_trajectories_do_not_move():
    """policy_documented.npz through the split-f16 critic: inside the 1e-5 bar to the reference's own numbers (a ring built from the fixture's
    stacks); and a rollout with critic_split='f16' produces bit-identical observations / actions / log-probs / rewards / dones -- only
    `value` (and what GAE makes of it) differs, within the bar."""
    from taco_amd import policy as P
    g = np.load(os.path.join(GOLD, "policy_documented.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    st = g["states"]                                   # [100, 5, 26] independent stacks -> ring with ONE slot per stack is not the pair form;
    frames = np.zeros((2 + 4, 100, 26), np.float32)    # build a 2-slot ring whose slot 0 holds the fixture's stacks: frames[k] = stack frame k
    frames[:5] = st.transpose(1, 0, 2)
    frames[5] = st[:, 4]
    fr = torch.from_numpy(frames).cuda()
    v = P.ActorCritic(sd, 1, 5, seed=9, critic_split="f16").values_ring(fr).cpu().numpy()
    np.testing.assert_allclose(v[0, :, 0], g["value"], rtol=0, atol=1e-5)
    vx = P.ActorCritic(sd, 1, 5, seed=9, exact_critic=True).values_ring(fr).cpu().numpy()
    assert np.abs(v - vx).max() <= 2e-6


def test_documented_architecture_matches_reference_golden():
    """policy_documented.npz (the reference's own module at the documented widths): act() and the batched critic against the reference's
    numbers (1e-5), and against each other (bit for bit with the exact cell)."""
    from taco_amd import policy as P
    g = np.load(os.path.join(GOLD, "policy_documented.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    pol = P.ActorCritic(sd, 1, 5, seed=9)
    obs, st = torch.from_numpy(g["obs"]).cuda(), torch.from_numpy(g["states"]).cuda()
    action, logp, value, mu, sigma = pol.act(obs, st, deterministic=True)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(value.cpu().numpy()[:, 0], g["value"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp_det"], rtol=0, atol=1e-5)
    batched = pol.values(st)     # default cell: the hardware's 2^x / reciprocal
    np.testing.assert_allclose(batched.cpu().numpy()[:, 0], g["value"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(batched.cpu().numpy(), value.cpu().numpy(), rtol=0, atol=CELL_TOL)
    exact = P.ActorCritic(sd, 1, 5, seed=9, exact_critic=True).values(st)
    assert_bits_equal(exact.cpu().numpy(), value.cpu().numpy(), "batched critic (exact cell) vs act()")


def _random_policy(rng, obs_len, states_len, actor_hidden, lstm, critic_hidden, states_dim=26):
    sd = {"log_std": rng.uniform(-1, 0.5, 4).astype(np.float32)}
    dims = [obs_len * 26] + actor_hidden + [4]
    for i in range(len(dims) - 1):
        sd[f"actor_mlp.layers.{2 * i}.weight"] = (rng.standard_normal((dims[i + 1], dims[i])) * (1.5 / np.sqrt(dims[i]))).astype(np.float32)
        sd[f"actor_mlp.layers.{2 * i}.bias"] = (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)
    if lstm:
        sd["critic_encoder.layers.weight_ih_l0"] = (rng.standard_normal((4 * lstm, states_dim)) * 0.3).astype(np.float32)
        sd["critic_encoder.layers.weight_hh_l0"] = (rng.standard_normal((4 * lstm, lstm)) * (1.0 / np.sqrt(lstm))).astype(np.float32)
        sd["critic_encoder.layers.bias_ih_l0"] = (rng.standard_normal(4 * lstm) * 0.1).astype(np.float32)
        sd["critic_encoder.layers.bias_hh_l0"] = (rng.standard_normal(4 * lstm) * 0.1).astype(np.float32)
    dims = [lstm if lstm else states_len * states_dim] + critic_hidden + [1]
    for i in range(len(dims) - 1):
        sd[f"critic_mlp.layers.{2 * i}.weight"] = (rng.standard_normal((dims[i + 1], dims[i])) * (1.5 / np.sqrt(dims[i]))).astype(np.float32)
        sd[f"critic_mlp.layers.{2 * i}.bias"] = (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)
    return sd


@pytest.mark.parametrize("n,obs_len,states_len,actor_hidden,lstm,critic_hidden", [
    (96, 1, 5, [64, 40], 24, [48]),            # the golden shape (ragged widths: padding to 16)
    (4096, 1, 5, [128, 128, 128], 128, [128, 128]),   # the default production shape
    (1000, 3, 4, [256], 0, [256, 32]),         # frame-stacked actor, no critic encoder, ragged row count
    (17, 1, 8, [], 256, []),                   # no hidden layers, widest LSTM, longest stack
])
def test_act_equals_oracle_bitwise(n, obs_len, states_len, actor_hidden, lstm, critic_hidden):
    from oracle import oracle as O
    from taco_amd import policy as P
    rng = np.random.default_rng(n + lstm)
    sd = _random_policy(rng, obs_len, states_len, actor_hidden, lstm, critic_hidden)
    pol = P.ActorCritic(sd, obs_len, states_len, seed=77)
    oc = O.policy_cfg(obs_len, states_len, actor_hidden, lstm, critic_hidden)
    blob = P.pack_state_dict(pol.cfg, sd)
    obs = rng.standard_normal((n, obs_len, 26)).astype(np.float32)
    st = rng.standard_normal((n, states_len, 26)).astype(np.float32)
    obs[0, 0, :3] = [1e4, -1e4, 0.0]           # saturate a few units
    for call, det in ((0, True), (0, False), (1, False)):     # a deterministic call does not consume a noise counter
        assert pol.calls == call
        got = pol.act(torch.from_numpy(obs).cuda(), torch.from_numpy(st).cuda(), deterministic=det)
        eps = None if det else O.policy_noise(77, call, n)
        exp = O.policy_act(oc, blob, obs, st, eps=eps, deterministic=det)
        for name, gt, ex in zip(("action", "logp", "value", "mu", "sigma"), got, exp):
            assert_bits_equal(gt.cpu().numpy().reshape(ex.shape), ex, f"call {call} {name}")
    a1 = pol.act(torch.from_numpy(obs).cuda(), torch.from_numpy(st).cuda())[0]
    a2 = pol.act(torch.from_numpy(obs).cuda(), torch.from_numpy(st).cuda())[0]
    assert not torch.equal(a1, a2)             # a fresh noise draw per call


def test_policy_drives_the_env_rollout():
    """act -> collect for a short rollout: the pieces of ppo_asymmetry.py:308-342 on one stream, no host sync inside the loop."""
    from taco_amd import config
    from taco_amd.rollout import RolloutBuffer
    from taco_amd.vec_env import FpvPos
    g, sd, pol = _golden_policy()
    n, H = 256, 8
    env = FpvPos(config.default_cfg("pos", n, env_lenStates=5), copy_outputs=False)
    buf = RolloutBuffer(n, 26, 1, 26, 5, 4, H, 4, 0.99, 0.95, "cuda:0")
    for t in range(H):
        action, logp, value, mu, sigma = pol.act(buf.next_obs, buf.next_states)
        buf.collect(env, torch.clip(action, -1, 1), logp, value, mu, sigma, act=action)
    buf.compute_returns_and_advantage(pol.act(buf.next_obs, buf.next_states)[2])
    assert torch.isfinite(buf.adv_buf).all() and torch.isfinite(buf.ret_buf).all() and buf.value_buf.abs().sum() > 0
    assert torch.equal(buf.obs_buf[1], buf._obs_store[1]) and buf.act_buf.abs().max() > 0


def test_unsupported_configurations_fail_loudly():
    from taco_amd import policy as P
    from taco_amd._lib import TacoError
    rng = np.random.default_rng(0)
    sd = _random_policy(rng, 1, 5, [32], 16, [32])
    sd["actor_encoder.layers.weight_ih_l0"] = np.zeros((4, 4), np.float32)
    with pytest.raises(TacoError):
        P.ActorCritic(sd, 1, 5)
    sd = _random_policy(rng, 1, 9, [32], 16, [32])
    with pytest.raises(TacoError):
        P.ActorCritic(sd, 1, 9)                # LSTM stack longer than 8 frames
    sd = _random_policy(rng, 1, 5, [32], 16, [32])
    pol = P.ActorCritic(sd, 1, 5)
    with pytest.raises(ValueError):
        pol.act(torch.zeros(4, 2, 26).cuda(), torch.zeros(4, 5, 26).cuda())


def test_torchscript_export_agrees_with_the_hip_forward(tmp_path):
    """Row N3: the TorchScript actor written by save_actor_as_pt (the on-board deployment format, ppo_asymmetry.py:458-468) and the HIP
    policy kernel are the same function (1e-6: torch's GEMM order vs the kernel's fma chain)."""
    from taco_amd.train_utils import save_actor_as_pt
    g, sd, pol = _golden_policy()
    save_actor_as_pt(sd, str(tmp_path / "actor_1.pt"), 1)
    ts = torch.jit.load(str(tmp_path / "actor_1.pt"))
    obs = torch.from_numpy(g["obs"])
    np.testing.assert_allclose(pol.forward(obs.cuda()).cpu().numpy(), ts(obs).detach().numpy(), rtol=0, atol=1e-6)


# ---- the batched critic (taco_critic_values): what takes the critic off the rollout's per-step chain
# Its LSTM cell runs on the hardware's 2^x / reciprocal by default (include/taco_env.h TACO_P_EXACT_CELL): bit for bit against act() and the
# oracle with exact_critic=True, within CELL_TOL (values are O(1); the reference's own vectors are matched to 1e-5) without.
CELL_TOL = 2e-6


def _same_values(got, exp, exact, what):
    if exact:
        assert_bits_equal(got, exp, what)
    else:
        assert np.isfinite(got).all(), what
        np.testing.assert_allclose(got, exp, rtol=0, atol=CELL_TOL, err_msg=what)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("rows,states_len,lstm,critic_hidden", [
    (1, 5, 128, [128, 128]), (31, 1, 128, [128, 128]), (33, 2, 128, [128, 128]), (4133, 5, 128, [128, 128]), (20000, 8, 128, [128, 128]),
    (700, 5, 120, [128, 116]),      # ragged widths inside the batched form's padding
    (300, 4, 64, [128, 128]),       # narrower LSTM: runs the policy kernel's critic role
    (300, 3, 128, [64]),            # another MLP: likewise
    (300, 2, 0, [256, 32]),         # no encoder
])
def test_batched_critic_equals_act_and_oracle(rows, states_len, lstm, critic_hidden, exact):
    from oracle import oracle as O
    from taco_amd import policy as P
    rng = np.random.default_rng(rows + lstm)
    sd = _random_policy(rng, 1, states_len, [32], lstm, critic_hidden)
    pol = P.ActorCritic(sd, 1, states_len, seed=3, exact_critic=exact)
    st = rng.standard_normal((rows, states_len, 26)).astype(np.float32)
    st[rows // 2] = 0.0                                   # an all-zero stack
    st[0, 0, :3] = [1e4, -1e4, 0.0]
    std = torch.from_numpy(st).cuda()
    got = pol.values(std)
    assert got.shape == (rows, 1)
    ref = pol.act(torch.zeros(rows, 1, 26, device="cuda"), std, deterministic=True)[2]
    _same_values(got.cpu().numpy(), ref.cpu().numpy(), exact, "batched critic vs act()")
    m = min(rows, 200)                                    # the CPU oracle on a slice (it takes seconds per thousand rows)
    pick = np.unique(np.concatenate([np.arange(m // 2), rows - 1 - np.arange(m // 2), [rows // 2]]))
    oc = O.policy_cfg(1, states_len, [32], lstm, critic_hidden)
    exp = O.policy_act(oc, P.pack_state_dict(pol.cfg, sd), np.zeros((len(pick), 1, 26), np.float32), st[pick], deterministic=True)[2]
    _same_values(got.cpu().numpy()[pick].reshape(exp.shape), exp, exact, "batched critic vs oracle")
    # leading dimensions are kept: [H, N, T, 26] -> [H, N, 1]
    if rows % 3 == 0:
        assert torch.equal(pol.values(std.view(3, rows // 3, states_len, 26)), got.view(3, rows // 3, 1))


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("slots,n,states_len,lstm,critic_hidden", [
    (33, 4096, 5, 128, [128, 128]),   # the rollout's shape: H + 1 slots, the batched form, 16-byte aligned pieces (LDS-DMA)
    (7, 333, 5, 128, [128, 128]),     # odd N: pieces only 8-byte aligned (plain-load staging), blocks straddle slot boundaries
    (5, 70, 3, 128, [128, 128]),      # ragged last block
    (4, 96, 1, 128, [128, 128]),      # one frame per stack
    (6, 100, 4, 64, [128, 128]),      # narrower LSTM: the policy kernel's critic role on a ring
    (6, 100, 2, 0, [256, 32]),        # no encoder: the flattened stack gathered from the ring
])
def test_critic_on_a_frame_ring_equals_the_materialised_stacks(slots, n, states_len, lstm, critic_hidden, exact):
    """taco_critic_values_ring: frames [slots + T - 1][N][26], row (slot, env) = frames[slot : slot + T, env] -- the replay store's layout
    (one frame per step instead of a shifted stack per slot).  Bit-identical to the critic on the materialised stacks, with either cell (the
    paired-slot kernel and the stack kernel run the same operations on the same accumulators)."""
    from taco_amd import policy as P
    rng = np.random.default_rng(slots * 1000 + n)
    sd = _random_policy(rng, 1, states_len, [32], lstm, critic_hidden)
    pol = P.ActorCritic(sd, 1, states_len, seed=3, exact_critic=exact, critic_split=None)   # (the f32 kernels: the split-precision default of the ring form has its own test)
    frames = torch.from_numpy(rng.standard_normal((slots + states_len - 1, n, 26)).astype(np.float32)).cuda()
    frames[1, n // 2] = 0.0
    stacks = torch.as_strided(frames, (slots, n, states_len, 26), (n * 26, 26, n * 26, 1)).contiguous()
    got = pol.values_ring(frames)
    assert got.shape == (slots, n, 1)
    assert_bits_equal(got.cpu().numpy(), pol.values(stacks).cpu().numpy(), "critic on the ring vs on the stacks")


def test_batched_critic_first_timestep_shortcut_keeps_the_sign_of_zero():
    """The batched LSTM skips the W_hh h_{-1} chain of the first timestep (h_{-1} = +0).  fma(+0, w, acc) changes acc only when acc is a
    zero: -0 survives iff every weight of the column is negative.  Build exactly that: zero input rows, biases of -0.0, one gate column
    all-negative, another mixed -- and compare with the kernel that runs the whole chain."""
    from taco_amd import policy as P
    rng = np.random.default_rng(5)
    sd = _random_policy(rng, 1, 3, [32], 128, [128, 128])
    whh = sd["critic_encoder.layers.weight_hh_l0"]
    whh[7] = -np.abs(whh[7])                 # gate i, column 7: every weight negative
    whh[128 + 9] = -np.abs(whh[128 + 9])     # gate f, column 9
    whh[256 + 9, 5] = 0.0                    # gate g, column 9: a +0 among them (not "all negative")
    sd["critic_encoder.layers.bias_ih_l0"][:] = 0.0
    sd["critic_encoder.layers.bias_hh_l0"][:] = -0.0
    sd["critic_encoder.layers.bias_ih_l0"][:] = -0.0
    pol = P.ActorCritic(sd, 1, 3, seed=3, exact_critic=True)
    st = rng.standard_normal((96, 3, 26)).astype(np.float32)
    st[::2, 0] = 0.0                          # first frame zero: the x chain leaves the -0 bias in place
    st[1::4, 0] = -0.0
    std = torch.from_numpy(st).cuda()
    ref = pol.act(torch.zeros(96, 1, 26, device="cuda"), std, deterministic=True)[2]
    assert_bits_equal(pol.values(std).cpu().numpy(), ref.cpu().numpy(), "batched critic vs act(), zero-sign case")


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("states_dim", [17, 19, 30, 32])
def test_batched_critic_other_state_widths(states_dim, exact):
    """The batched LSTM kernel copies the state stacks into LDS as they lie in memory and addresses [row][t][states_dim] in its fragment
    reads: odd widths take the 4-byte read path, even ones the 8-byte path, 32 has no padding columns to zero."""
    from oracle import oracle as O
    from taco_amd import policy as P
    rng = np.random.default_rng(states_dim)
    sd = _random_policy(rng, 1, 4, [32], 128, [128, 128], states_dim=states_dim)
    pol = P.ActorCritic(sd, 1, 4, seed=3, states_dim=states_dim, exact_critic=exact)
    rows = 1000 + states_dim
    st = rng.standard_normal((rows, 4, states_dim)).astype(np.float32)
    std = torch.from_numpy(st).cuda()
    got = pol.values(std)
    ref = pol.act(torch.zeros(rows, 1, 26, device="cuda"), std, deterministic=True)[2]
    _same_values(got.cpu().numpy(), ref.cpu().numpy(), exact, "batched critic vs act()")
    oc = O.policy_cfg(1, 4, [32], 128, [128, 128], states_dim=states_dim)
    exp = O.policy_act(oc, P.pack_state_dict(pol.cfg, sd), np.zeros((64, 1, 26), np.float32), st[-64:], deterministic=True)[2]
    _same_values(got.cpu().numpy()[-64:].reshape(exp.shape), exp, exact, "batched critic vs oracle")


@pytest.mark.parametrize("n", [8193, 20000, 65536 + 3])
def test_actor_throughput_form_equals_the_latency_form_and_oracle(n):
    """Above 8 192 rows act(action_only=True) runs taco_actor_kernel (64 rows per pass, two workgroups per CU): same actions, log-probs, means as
    the 16-row form (which a full act() still uses) and as the oracle, bit for bit, sampled and deterministic."""
    from oracle import oracle as O
    from taco_amd import policy as P
    rng = np.random.default_rng(n)
    sd = _random_policy(rng, 1, 5, [128, 128, 128], 128, [128, 128])
    pol = P.ActorCritic(sd, 1, 5, seed=13)
    obs = rng.standard_normal((n, 1, 26)).astype(np.float32)
    obs[5, 0, :3] = [1e4, -1e4, 0.0]
    st = rng.standard_normal((n, 5, 26)).astype(np.float32)
    obs_d, st_d = torch.from_numpy(obs).cuda(), torch.from_numpy(st).cuda()
    for det in (True, False):
        call = pol.calls
        full = pol.act(obs_d, st_d, deterministic=det)              # the 16-row kernel (actor + critic roles)
        pol.calls = call
        fast = pol._run(obs_d, None, det, True)                      # the throughput actor
        for name, k in (("action", 0), ("logp", 1), ("mu", 3), ("sigma", 4)):
            assert_bits_equal(fast[k].cpu().numpy(), full[k].cpu().numpy(), f"det={det} {name}")
        pick = np.r_[0:40, n - 40:n]
        oc = O.policy_cfg(1, 5, [128, 128, 128], 128, [128, 128])
        eps = None if det else O.policy_noise(13, call, n)[pick]
        exp = O.policy_act(oc, P.pack_state_dict(pol.cfg, sd), obs[pick], st[pick], eps=eps, deterministic=det)
        assert_bits_equal(fast[0].cpu().numpy()[pick], exp[0], f"det={det} action vs oracle")
        assert_bits_equal(fast[1].cpu().numpy()[pick], exp[1], f"det={det} logp vs oracle")
