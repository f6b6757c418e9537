"""Host-side logic that needs no GPU: config authoring / flattening, sharding arithmetic, gather block packing,
and the 'fail loudly' contract of the Python layer."""
import math

import numpy as np
import pytest
import torch

from taco_amd import config, dist, _lib


def test_composite_body_from_mjcf_numbers():
    m, J = config.composite_body()
    assert abs(m - (0.46 + 8e-7)) < 1e-12
    assert abs(J[0] - (5e-4 + 8e-7 + 1e-7 * (8 * 0.059 ** 2 + 4 * 0.02 ** 2))) < 1e-15
    assert abs(J[2] - (8e-4 + 8e-7 + 8e-7 * (0.047 ** 2 + 0.059 ** 2))) < 1e-15
    m2, J2 = config.composite_body("with_duct")
    assert m2 > 0.57 and J2[1] > 0.00147


def test_flat_cfg_reads_the_reference_key_set():
    cfg = config.default_cfg("rotate", 512, env_lenStates=5, delay_time=40, sim_substeps=4)
    f = config.flat_cfg(cfg)
    assert f["task_mode"] == "rotate" and f["num_envs"] == f["num_envs_global"] == 512
    assert f["len_obs"] == 1 and f["len_states"] == 5 and f["substeps"] == 4 and f["delay_time"] == 40
    assert f["control_freq_inv"] == 10 and f["dt"] == 0.001 and math.isinf(f["clip_obs"])
    for k in _lib.FLAG_BITS:
        assert k in f, k
    c = _lib.make_cfg(f)
    assert c.flags & (1 << _lib.FLAG_BITS["random_command"]) and not c.flags & (1 << _lib.FLAG_BITS["observation_noise"])
    with pytest.raises(ValueError):
        config.flat_cfg(dict(cfg, delay_time_max=50))
    f2 = config.flat_cfg(cfg, env_offset=128, num_envs_local=64)
    assert (f2["num_envs"], f2["env_offset"], f2["num_envs_global"]) == (64, 128, 512)


def test_baseline_configs_are_the_five_of_baseline_json():
    modes = [config.baseline_config(i)["task_mode"] for i in range(5)]
    sizes = [config.baseline_config(i)["env"]["numEnvs"] for i in range(5)]
    assert modes == ["pos", "pos", "rotate", "flip", "mix"] and sizes == [64, 4096, 16384, 65536, 262144]
    c4 = config.baseline_config(4)
    assert c4["observation_noise"] and c4["rotor_noise"] and c4["ramdom_deploy_time"] and c4["env"]["lenStates"] == 5


def test_shard_bounds_cover_the_range_contiguously():
    for n, w in ((4096, 8), (10, 3), (7, 8), (262144, 8), (1000, 6)):
        b = [dist.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_gather_block_pack_unpack_roundtrip():
    g = torch.Generator().manual_seed(0)
    for len_obs in (1, 3):
        obs = torch.randn(17, len_obs, 26, generator=g)
        rew = torch.randn(17, generator=g)
        done = torch.randint(0, 2, (17,), generator=g)
        tmo = torch.randint(0, 2, (17,), generator=g).bool() & (done != 0)
        blk = dist.pack_block(obs, rew, done, tmo)
        assert blk.shape == (17, dist.block_row(len_obs))
        o, r, d, t = dist.unpack_block(blk, len_obs)
        assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(d, done) and torch.equal(t, tmo)
        assert d.dtype == torch.long and t.dtype == torch.bool


def test_python_layer_fails_loudly_without_a_hip_device():
    from taco_amd.vec_env import FpvPos, isaacgym_task_map
    assert set(isaacgym_task_map) == {"Fpv_pos", "Fpv_rotate", "Fpv_flip", "Fpv_mix"}
    with pytest.raises(_lib.TacoError):
        FpvPos(config.baseline_config(0), rl_device="cpu", sim_device="cpu")   # no CPU fallback for the step path
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            FpvPos(config.baseline_config(0), rl_device="cuda:0", sim_device="cuda:0")


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtaco_env.so")
    with pytest.raises(_lib.TacoError, match="no CPU fallback"):
        _lib.load()


def test_rollout_and_policy_refuse_to_run_without_the_gpu():
    """The N1 host classes are thin shells over HIP kernels: on a CPU device they fail loudly instead of falling back to torch."""
    import numpy as np
    from taco_amd._lib import TacoError
    from taco_amd.policy import ActorCritic
    from taco_amd.rollout import RolloutBuffer
    with pytest.raises(TacoError):
        RolloutBuffer(8, 26, 1, 26, 5, 4, 4, 2, 0.99, 0.95, "cpu")
    sd = {"log_std": np.zeros(4, np.float32), "actor_mlp.layers.0.weight": np.zeros((4, 26), np.float32), "actor_mlp.layers.0.bias": np.zeros(4, np.float32),
          "critic_mlp.layers.0.weight": np.zeros((1, 130), np.float32), "critic_mlp.layers.0.bias": np.zeros(1, np.float32)}
    with pytest.raises(TacoError):
        ActorCritic(sd, 1, 5, device="cpu")


def test_policy_packing_matches_the_layout_both_sides_read():
    """pack_state_dict: blob size == what the C ABI and the oracle expect for the same widths; fragment-major round trip of one matrix."""
    import numpy as np
    from oracle import oracle as O
    from taco_amd import policy as P
    rng = np.random.default_rng(1)
    for actor_h, lstm, critic_h, lo, ls in (([64, 40], 24, [48], 1, 5), ([], 0, [], 2, 3), ([256, 16, 16, 100], 200, [7], 1, 8)):
        sd = {"log_std": np.zeros(4, np.float32)}
        dims = [lo * 26] + actor_h + [4]
        for i in range(len(dims) - 1):
            sd[f"actor_mlp.layers.{2 * i}.weight"] = rng.standard_normal((dims[i + 1], dims[i])).astype(np.float32)
            sd[f"actor_mlp.layers.{2 * i}.bias"] = rng.standard_normal(dims[i + 1]).astype(np.float32)
        if lstm:
            sd["critic_encoder.layers.weight_ih_l0"] = rng.standard_normal((4 * lstm, 26)).astype(np.float32)
            sd["critic_encoder.layers.weight_hh_l0"] = rng.standard_normal((4 * lstm, lstm)).astype(np.float32)
            sd["critic_encoder.layers.bias_ih_l0"] = rng.standard_normal(4 * lstm).astype(np.float32)
            sd["critic_encoder.layers.bias_hh_l0"] = rng.standard_normal(4 * lstm).astype(np.float32)
        dims = [lstm if lstm else ls * 26] + critic_h + [1]
        for i in range(len(dims) - 1):
            sd[f"critic_mlp.layers.{2 * i}.weight"] = rng.standard_normal((dims[i + 1], dims[i])).astype(np.float32)
            sd[f"critic_mlp.layers.{2 * i}.bias"] = rng.standard_normal(dims[i + 1]).astype(np.float32)
        cfg = P.cfg_from_state_dict(sd, lo, ls)
        blob = P.pack_state_dict(cfg, sd)
        assert blob.size == O.policy_blob_floats(O.policy_cfg(lo, ls, actor_h, lstm, critic_h))
    W = np.arange(32 * 48, dtype=np.float32).reshape(32, 48)
    F = P._frag(W).reshape(2, 3, 64, 4)                      # [tile][k block][lane = 16 g + r][t]
    for tile, s, g, r, t in ((0, 0, 0, 0, 0), (1, 2, 3, 15, 3), (0, 1, 2, 5, 1)):
        assert F[tile, s, 16 * g + r, t] == W[16 * tile + r, 16 * s + 4 * g + t]


def test_checkpoint_merge_and_slice_by_global_env_id():
    """taco_amd/checkpoint.py on stand-in envs (no GPU): merge() joins adjoining shards in env order and refuses gaps or mixed runs;
    load_state_dict() hands an env exactly the columns / rows of its [env_offset, env_offset + num_envs) range."""
    import torch
    from taco_amd import checkpoint, _lib

    class Env:
        def __init__(self, lo, n, total):
            self._flat = {"num_envs": n, "env_offset": lo, "num_envs_global": total, "task_mode": "mix", "len_obs": 1, "len_states": 2, "seed": 3}
            self.device = torch.device("cpu")
            self.difficulty, self.step_count = 0.0, 0
            self.blob = None
            self.obs_buf, self.states_buf = torch.zeros(n, 1, 26), torch.zeros(n, 2, 26)
            self.rew_buf, self.reset_buf, self.timeout_buf = torch.zeros(n), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.bool)

        def get_state(self):
            return self.blob

        def set_state(self, b):
            self.blob = b.clone()

        def load_stacks(self, obs, states):   # (FpvBase.load_stacks: the state stack may live in a frame ring there)
            self.obs_buf.copy_(obs); self.states_buf.copy_(states)

    def filled(lo, n, total, step=17):
        e = Env(lo, n, total)
        ids = torch.arange(lo, lo + n, dtype=torch.float32)
        e.blob = ids.repeat(_lib.BLOB_ROWS, 1) + torch.arange(_lib.BLOB_ROWS, dtype=torch.float32)[:, None] * 1000
        e.obs_buf += ids[:, None, None]; e.states_buf += ids[:, None, None]; e.rew_buf += ids; e.reset_buf += ids.long()
        e.difficulty, e.step_count = 0.4, step
        return e

    parts = [filled(0, 5, 12), filled(5, 4, 12), filled(9, 3, 12)]
    merged = checkpoint.merge([checkpoint.state_dict(p) for p in (parts[2], parts[0], parts[1])])
    assert merged["flat_cfg"]["num_envs"] == 12 and merged["flat_cfg"]["env_offset"] == 0 and merged["blob"].shape == (_lib.BLOB_ROWS, 12)
    assert torch.equal(merged["blob"][0], torch.arange(12, dtype=torch.float32)) and torch.equal(merged["rew_buf"], torch.arange(12, dtype=torch.float32))
    tgt = Env(3, 6, 12)
    checkpoint.load_state_dict(tgt, merged)
    assert tgt.step_count == 17 and tgt.difficulty == 0.4
    assert torch.equal(tgt.blob[0], torch.arange(3, 9, dtype=torch.float32)) and torch.equal(tgt.obs_buf[:, 0, 0], torch.arange(3, 9, dtype=torch.float32))
    with pytest.raises(ValueError, match="adjoin"):
        checkpoint.merge([checkpoint.state_dict(parts[0]), checkpoint.state_dict(parts[2])])
    with pytest.raises(ValueError, match="step_count"):
        checkpoint.merge([checkpoint.state_dict(parts[0]), checkpoint.state_dict(filled(5, 4, 12, step=18))])
    with pytest.raises(ValueError, match="merge"):
        checkpoint.load_state_dict(Env(3, 6, 12), checkpoint.state_dict(parts[0]))
    with pytest.raises(ValueError, match="num_envs_global"):
        checkpoint.load_state_dict(Env(0, 5, 13), merged)
    # the arithmetic mode is part of the run's identity: a checkpoint written under the other data flow of the angular rate is refused, and so
    # is one from before ABI 6 (which carries no such key: "carried" was the only arithmetic then) by an env on today's default
    rt = Env(3, 6, 12)
    rt._flat["world_rate_roundtrip"] = True
    with pytest.raises(ValueError, match="world_rate_roundtrip=None|world_rate_roundtrip=False"):
        checkpoint.load_state_dict(rt, dict(merged, abi=5))
    sd = dict(merged, flat_cfg=dict(merged["flat_cfg"], world_rate_roundtrip=False))
    with pytest.raises(ValueError, match="world_rate_roundtrip=False"):
        checkpoint.load_state_dict(rt, sd)
    checkpoint.load_state_dict(rt, sd, strict=False)
    with pytest.warns(UserWarning, match="ABI"):
        checkpoint.load_state_dict(rt, dict(merged, abi=_lib.ABI_VERSION - 1, flat_cfg=dict(merged["flat_cfg"], world_rate_roundtrip=True)))
    # round 5's advisor: (a) an ABI-6 checkpoint (arithmetic revision 1: the stored world-frame rate was rebuilt once more) is refused under strict;
    good = dict(merged, flat_cfg=dict(merged["flat_cfg"], world_rate_roundtrip=True))
    with pytest.raises(ValueError, match="arithmetic revision 1"):
        checkpoint.load_state_dict(rt, {k: v for k, v in dict(good, abi=6).items() if k != "arith_rev"})
    checkpoint.load_state_dict(rt, {k: v for k, v in dict(good, abi=6).items() if k != "arith_rev"}, strict=False)
    # (b) the numeric cfg values the arithmetic reads are compared;
    rt._flat["rotor_response_time"] = 0.017
    with pytest.raises(ValueError, match="rotor_response_time=0.02"):
        checkpoint.load_state_dict(rt, dict(good, flat_cfg=dict(good["flat_cfg"], rotor_response_time=0.02)))
    rt._flat["inertia"] = (1.0, 2.0, 3.0)
    with pytest.raises(ValueError, match="inertia"):
        checkpoint.load_state_dict(rt, dict(good, flat_cfg=dict(good["flat_cfg"], rotor_response_time=0.017, inertia=[1.0, 2.0, 3.5])))
    checkpoint.load_state_dict(rt, dict(good, flat_cfg=dict(good["flat_cfg"], rotor_response_time=0.017, inertia=[1.0, 2.0, 3.0])))
    # (c) a recording env does not take over the stale rpy rows of a checkpoint taken without record_flag (the other direction is fine);
    rt._flat["record_flag"] = True
    full = dict(good, flat_cfg=dict(good["flat_cfg"], rotor_response_time=0.017, inertia=[1.0, 2.0, 3.0]))
    with pytest.raises(ValueError, match="record_flag=False"):
        checkpoint.load_state_dict(rt, full)
    checkpoint.load_state_dict(rt, dict(full, flat_cfg=dict(full["flat_cfg"], record_flag=True)))
    rt._flat["record_flag"] = False
    checkpoint.load_state_dict(rt, dict(full, flat_cfg=dict(full["flat_cfg"], record_flag=True)))
    # (d) merge() refuses shards taken under different flags / constants (it compared five identity keys only)
    a, b = checkpoint.state_dict(parts[0]), checkpoint.state_dict(parts[1])
    b["flat_cfg"] = dict(b["flat_cfg"], rotor_noise=True)
    a["flat_cfg"] = dict(a["flat_cfg"], rotor_noise=False)
    with pytest.raises(ValueError, match="rotor_noise"):
        checkpoint.merge([a, b])


def test_replay_store_frame_ring_views_are_the_state_stacks():
    """RolloutBuffer keeps the state stacks as a frame ring [H + T][N][D]; states_buf [H,N,T,D] / next_states [N,T,D] are overlapping strided
    views of it: stack (slot, env) = ring rows slot .. slot + T - 1 of that env.  The view arithmetic, on CPU tensors (no GPU needed)."""
    import torch
    from taco_amd.rollout import RolloutBuffer
    H, N, T, D = 6, 5, 3, 4
    buf = object.__new__(RolloutBuffer)
    buf.num_envs, buf.states_len, buf.states_dim, buf.horizon_len = N, T, D, H
    buf._frames = torch.arange((H + T) * N * D, dtype=torch.float32).view(H + T, N, D)
    sb = buf._stack_view(0, H)
    assert sb.shape == (H, N, T, D)
    for t in range(H):
        for n in range(N):
            for k in range(T):
                assert torch.equal(sb[t, n, k], buf._frames[t + k, n])
    buf.step = 4
    ns = buf.next_states
    assert ns.shape == (N, T, D) and torch.equal(ns[2, 1], buf._frames[5, 2]) and torch.equal(ns.contiguous()[:, -1], buf._frames[4 + T - 1])
    flat = sb.reshape(-1, T, D)                       # what a PPO update does before indexing minibatches: materialises a copy
    assert flat.shape == (H * N, T, D) and torch.equal(flat[3 * N + 2], sb[3, 2])
