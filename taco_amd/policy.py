"""Policy forward next to the env (second half): `PPO_ActorCritic.act`
(IsaacGymEnvs/algorithms/nets_asymmetry.py:270-355) for the documented training configuration (README.md:60-66: the actor is an
MLP on the observation stack, the critic a 1-layer LSTM over the state stack followed by an MLP) as ONE kernel launch
(taco_policy_act: f32 MFMA 16x16x4 tiles, 16 envs per workgroup, weights streamed from L2, activations in LDS).

`ActorCritic` mirrors the reference module's inference surface: `act(actor_input, critic_input, deterministic=False,
action_only=False)` with the same five return values, `forward(actor_input)`, `load_state_dict` with the reference's parameter
names.  Training (`evaluate`, autograd) stays with the caller's torch module; `load_state_dict(agent.state_dict())` after each
update refreshes the packed weights.  There is no torch fallback for the arithmetic.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class PolicyCfg(C.Structure):
    """struct taco_policy_cfg (include/taco_env.h)"""
    _fields_ = [("obs_len", C.c_int32), ("obs_dim", C.c_int32), ("states_len", C.c_int32), ("states_dim", C.c_int32), ("act_dim", C.c_int32),
                ("n_actor_hidden", C.c_int32), ("actor_hidden", C.c_int32 * 4), ("lstm_hidden", C.c_int32),
                ("n_critic_hidden", C.c_int32), ("critic_hidden", C.c_int32 * 4), ("flags", C.c_int32)]


P_EXACT_CELL = 1   # TACO_P_EXACT_CELL
P_SPLIT_F16, P_SPLIT_BF16 = 2, 4   # TACO_P_SPLIT_F16 / TACO_P_SPLIT_BF16


MAX_WIDTH = 256


def make_cfg(obs_len, states_len, actor_hidden, lstm_hidden, critic_hidden, obs_dim=26, states_dim=26, act_dim=4):
    if len(actor_hidden) > 4 or len(critic_hidden) > 4:
        raise ValueError("at most four hidden layers per MLP")
    c = PolicyCfg(obs_len, obs_dim, states_len, states_dim, act_dim)
    c.n_actor_hidden = len(actor_hidden)
    for i, h in enumerate(actor_hidden):
        c.actor_hidden[i] = int(h)
    c.lstm_hidden = int(lstm_hidden)
    c.n_critic_hidden = len(critic_hidden)
    for i, h in enumerate(critic_hidden):
        c.critic_hidden[i] = int(h)
    return c


def _pad16(x):
    return (int(x) + 15) // 16 * 16


def _np(v):
    return v.detach().cpu().numpy().astype(np.float32) if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)


def _linear_keys(sd, prefix):
    """weights / biases of the nn.Linear members of an nn.Sequential named `prefix`, in order"""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")})
    return [(sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"]) for i in idx]


def cfg_from_state_dict(sd, obs_len, states_len, obs_dim=26, states_dim=26):
    """Read the layer widths off a PPO_ActorCritic state_dict (actor_mlp.layers.N.*, critic_encoder.layers.*_l0, critic_mlp.layers.N.*)."""
    if any(k.startswith("actor_encoder.") for k in sd):
        raise _lib.TacoError("actor encoders are not supported by the HIP policy forward (documented configuration: use_actor_encoder=False)")
    if any(k.endswith("_l1") or k.endswith("_reverse") for k in sd):
        raise _lib.TacoError("only a 1-layer unidirectional LSTM critic encoder is supported")
    actor = _linear_keys(sd, "actor_mlp.layers.")
    critic = _linear_keys(sd, "critic_mlp.layers.")
    lstm = sd["critic_encoder.layers.weight_hh_l0"].shape[1] if "critic_encoder.layers.weight_hh_l0" in sd else 0
    if lstm == 0 and any(k.startswith("critic_encoder.") for k in sd):
        raise _lib.TacoError("only the LSTM critic encoder is supported")
    return make_cfg(obs_len, states_len, [w.shape[0] for w, _ in actor[:-1]], lstm, [w.shape[0] for w, _ in critic[:-1]],
                    obs_dim, states_dim, actor[-1][0].shape[0])


def _frag(W):
    """[OUTp][INp] -> fragment-major [OUTp/16][INp/16][lane = 16 g + r][t] with W[16 tile + r][16 s + 4 g + t]: the 16 bytes lane (r, g)
    feeds to the four MFMAs of k block s are contiguous, and a wavefront's load of one block is 1 KiB contiguous."""
    o, i = W.shape
    return np.ascontiguousarray(W.reshape(o // 16, 16, i // 16, 4, 4).transpose(0, 2, 3, 1, 4)).ravel()


def pack_state_dict(cfg, sd):
    """state_dict (torch tensors or numpy arrays) -> the flat fp32 weight blob taco_policy_act / the oracle read.  Every matrix is
    zero-padded to multiples of 16 in both dimensions and stored fragment-major (_frag), followed by its bias [OUTp]; the LSTM is
    stored per gate (i f g o): W_ih [4][Hp x Ip], W_hh [4][Hp x Hp], b_ih + b_hh [4][Hp]."""
    out = []

    def lin(w, b, inp):
        w, b = _np(w), _np(b)
        o, i = w.shape
        if _pad16(o) > MAX_WIDTH or inp > MAX_WIDTH:
            raise _lib.TacoError(f"layer widths above {MAX_WIDTH} are not supported")
        W = np.zeros((_pad16(o), inp), np.float32)
        W[:o, :i] = w
        B = np.zeros(_pad16(o), np.float32)
        B[:o] = b
        out.extend([_frag(W), B])
        return _pad16(o)

    inp = _pad16(cfg.obs_len * cfg.obs_dim)
    for w, b in _linear_keys(sd, "actor_mlp.layers."):
        inp = lin(w, b, inp)
    ls = np.zeros(16, np.float32)
    ls[:cfg.act_dim] = _np(sd["log_std"])
    out.append(ls)
    if cfg.lstm_hidden > 0:
        H, hp, ip = cfg.lstm_hidden, _pad16(cfg.lstm_hidden), _pad16(cfg.states_dim)
        wih, whh = _np(sd["critic_encoder.layers.weight_ih_l0"]), _np(sd["critic_encoder.layers.weight_hh_l0"])
        bs = _np(sd["critic_encoder.layers.bias_ih_l0"]) + _np(sd["critic_encoder.layers.bias_hh_l0"])
        Wih, Whh, B = np.zeros((4, hp, ip), np.float32), np.zeros((4, hp, hp), np.float32), np.zeros((4, hp), np.float32)
        for q in range(4):
            Wih[q, :H, :cfg.states_dim] = wih[q * H:(q + 1) * H]
            Whh[q, :H, :H] = whh[q * H:(q + 1) * H]
            B[q, :H] = bs[q * H:(q + 1) * H]
        out.extend([np.concatenate([_frag(Wih[q]) for q in range(4)]), np.concatenate([_frag(Whh[q]) for q in range(4)]), B.ravel()])
        inp = hp
    else:
        inp = _pad16(cfg.states_len * cfg.states_dim)
    for w, b in _linear_keys(sd, "critic_mlp.layers."):
        inp = lin(w, b, inp)
    return np.concatenate(out)


class ActorCritic:
    """Inference-side mirror of PPO_ActorCritic (nets_asymmetry.py:270-355) on the HIP policy kernel."""

    def __init__(self, state_dict, obs_len, states_len, device="cuda:0", seed=0, obs_dim=26, states_dim=26, exact_critic=False, critic_split="auto"):
        """exact_critic: the batched critic (values / values_ring / RolloutBuffer.run) keeps the oracle's op-for-op LSTM cell, bit-identical
        to act()'s `value`; default: the hardware's 2^x / reciprocal in the cell, 19 % faster, values within 2e-6 (TACO_P_EXACT_CELL).
        critic_split: "auto" (default) | None | "f16" | "bf16" -- the ring-form LSTM of the batched critic (values_ring, RolloutBuffer.run) on
        the 16-bit matrix pipe with split operands (include/taco_env.h TACO_P_SPLIT_F16 / TACO_P_SPLIT_BF16): 2.6 x the f32 MFMA kernel.
        "f16": values within 1e-6 of the exact f32 critic's on O(1) frames -- the fast cell's own 2e-6 bar (the error table
        this default rests on); "bf16": ~2e-5, outside it (kept for the A/B record); None: the f32 MFMA kernel.  "auto" = "f16" unless
        exact_critic (which excludes a split).  The C ABI's own default (flags = 0) is the f32 kernel: the host layer opts in."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TacoError("the policy kernel runs on an MI355X; device must be a cuda:N (HIP) device")
        self.cfg = cfg_from_state_dict(state_dict, obs_len, states_len, obs_dim, states_dim)
        if critic_split == "auto":
            critic_split = None if exact_critic else "f16"
        if critic_split not in (None, "f16", "bf16"):
            raise ValueError("critic_split must be 'auto', None, 'f16' or 'bf16'")
        if exact_critic and critic_split:
            raise ValueError("exact_critic and critic_split exclude each other")
        self.cfg.flags = (P_EXACT_CELL if exact_critic else 0) | {None: 0, "f16": P_SPLIT_F16, "bf16": P_SPLIT_BF16}[critic_split]
        self.seed = int(seed)
        self.calls = 0  # Philox counter of the action noise: (seed, env index, call number); advanced by every SAMPLING call
        self.reuse_outputs = False  # True: act() returns the same five tensors every call (no allocations on the hot loop)
        self.stamps = None          # profiling: a [16] int64 device tensor that workgroup 0 of every act() launch fills with shader-clock stamps
        self._out = None
        self._last_critic_rows = 0
        n = self.lib.taco_policy_blob_floats(C.byref(self.cfg))
        if n == 0:
            raise _lib.TacoError(f"unsupported policy configuration: {self.lib.taco_last_error().decode()}")
        self._blob = torch.empty(n, dtype=torch.float32, device=self.device)
        self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict):
        blob = pack_state_dict(self.cfg, state_dict)
        assert blob.size == self._blob.numel()
        self._blob.copy_(torch.from_numpy(blob))
        self.log_std = torch.as_tensor(_np(state_dict["log_std"]), device=self.device)

    def _run(self, actor_input, critic_input, deterministic, action_only):
        n = actor_input.shape[0]
        a = self.cfg.act_dim

        def prep(t, length, dim):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                t = t.to(device=self.device, dtype=torch.float32).contiguous()
            if t.numel() != n * length * dim:
                raise ValueError(f"input of shape {tuple(t.shape)} does not match the policy ({length} x {dim} per env)")
            return t

        obs = prep(actor_input, self.cfg.obs_len, self.cfg.obs_dim)
        st = prep(critic_input, self.cfg.states_len, self.cfg.states_dim) if critic_input is not None else None
        dev = self.device
        if self.reuse_outputs and self._out is not None and self._out[0].shape[0] == n:
            action, logp, value, mu, sigma = self._out      # the previous call's tensors are overwritten
        else:
            action, mu, sigma = (torch.empty(n, a, device=dev) for _ in range(3))
            logp, value = torch.empty(n, device=dev), torch.empty(n, 1, device=dev)
            self._out = (action, logp, value, mu, sigma)
        s = _lib.stream_ptr(dev)
        _lib.check(self.lib.taco_policy_act_stamped(C.byref(self.cfg), self._blob.data_ptr(), n, obs.data_ptr(), st.data_ptr() if st is not None else None,
                                                    C.c_uint64(self.seed), C.c_uint32(self.calls), 1 if deterministic else 0, 1 if action_only else 0,
                                                    action.data_ptr(), logp.data_ptr(), value.data_ptr(), mu.data_ptr(), sigma.data_ptr(),
                                                    self.stamps.data_ptr() if self.stamps is not None else None, s))
        if not deterministic:
            self.calls += 1     # (a deterministic call draws nothing)
        return action, logp, value, mu, sigma

    def act(self, actor_input, critic_input, deterministic=False, action_only=False):
        """nets_asymmetry.py:326-355: (action, log_p [N], value [N,1], action_mean, log_std repeated [N,act])."""
        action, logp, value, mu, sigma = self._run(actor_input, None if action_only else critic_input, deterministic, action_only)
        if action_only:
            return action
        return action, logp, value, mu, sigma

    def values(self, critic_input, stamps=None):
        """The critic alone (nets_asymmetry.py:348-352) over any number of state stacks [..., states_len, states_dim] -> [..., 1], one
        launch (taco_critic_values): act()'s `value` on the same stacks -- bit for bit with exact_critic=True, within 2e-6 otherwise."""
        lead = critic_input.shape[:-2]
        st = critic_input
        if st.dtype != torch.float32 or not st.is_contiguous() or st.device != self.device:
            st = st.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(st.shape[-2:]) != (self.cfg.states_len, self.cfg.states_dim):
            raise ValueError(f"input of shape {tuple(st.shape)} does not match the critic ({self.cfg.states_len} x {self.cfg.states_dim} per row)")
        rows = st.numel() // (self.cfg.states_len * self.cfg.states_dim)
        out = torch.empty(rows, device=self.device)
        ws = self.critic_workspace(rows)
        s = _lib.stream_ptr(self.device)
        _lib.check(self.lib.taco_critic_values(C.byref(self.cfg), self._blob.data_ptr(), rows, st.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                               stamps.data_ptr() if stamps is not None else None, s))
        return out.view(*lead, 1)

    def values_ring(self, frames):
        """The critic over a FRAME RING [slots + states_len - 1, N, states_dim] (the replay store's layout, taco_amd/rollout.py): value
        [slots, N, 1] of the stacks frames[slot : slot + states_len, env].  Same kernels, same bits as values() on the materialised stacks."""
        T = self.cfg.states_len
        if frames.dim() != 3 or frames.shape[0] < T or frames.shape[2] != self.cfg.states_dim:
            raise ValueError(f"frame ring of shape {tuple(frames.shape)} does not match the critic ([slots + {T - 1}, N, {self.cfg.states_dim}])")
        fr = frames
        if fr.dtype != torch.float32 or not fr.is_contiguous() or fr.device != self.device:
            fr = fr.to(device=self.device, dtype=torch.float32).contiguous()
        slots, n = fr.shape[0] - T + 1, fr.shape[1]
        out = torch.empty(slots * n, device=self.device)
        ws = self.critic_workspace(slots * n)
        self._last_critic_rows = slots * n
        s = _lib.stream_ptr(self.device)
        _lib.check(self.lib.taco_critic_values_ring(C.byref(self.cfg), self._blob.data_ptr(), slots, n, fr.data_ptr(), out.data_ptr(), ws.data_ptr(), s), self.lib)
        return out.view(slots, n, 1)

    def clamped_words(self, rows=None):
        """BLOCKS.  How many finite frame words beyond +-65 504 the LAST batched split-f16 critic call (values_ring / RolloutBuffer.run; `rows` = its row
        count, default: that of the last values_ring call) saturated; 0 for every other critic form.  +-inf / NaN words are not counted: they poison the
        row's value with NaN, as in the f32 kernels.  (include/taco_env.h taco_critic_clamped_words)"""
        rows = self._last_critic_rows if rows is None else int(rows)
        if not rows or getattr(self, "_critic_ws", None) is None:
            return 0
        n = C.c_uint32(0)
        _lib.check(self.lib.taco_critic_clamped_words(C.byref(self.cfg), rows, self._critic_ws.data_ptr(), C.byref(n), _lib.stream_ptr(self.device)), self.lib)
        return int(n.value)

    def check(self, rows=None):
        """raise if the last batched critic call saturated frame words (states not normalised: use critic_split=None / exact_critic=True)"""
        n = self.clamped_words(rows)
        if n:
            raise _lib.TacoError(f"the split-f16 critic saturated {n} finite frame words beyond +-65 504: normalise the states or construct ActorCritic(critic_split=None)")

    def critic_workspace(self, rows):
        """the batched critic's workspace for `rows` state stacks (kept and reused while it is large enough)"""
        need = int(self.lib.taco_critic_workspace_bytes(C.byref(self.cfg), rows))
        if getattr(self, "_critic_ws", None) is None or self._critic_ws.numel() < need:
            self._critic_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._critic_ws

    def forward(self, actor_input):
        """nets_asymmetry.py:380-387: the action mean (what the TorchScript export traces)."""
        return self._run(actor_input, None, True, True)[3]

    __call__ = forward
