"""Replay buffer directly behind step().

Mirrors `PPOReplayBuffer` (IsaacGymEnvs/algorithms/buffer_asymmetry.py:9-139): same constructor, same attribute names
and shapes (`obs_buf [H,N,L,D]`, `states_buf`, `act_buf`, `rew_buf [H,N,1]`, `done_buf`, `ret_buf`, `value_buf`,
`adv_buf`, `mu_buf`, `sigma_buf`, `logp_buf`, `step`), same `store / reset / compute_returns_and_advantage /
batch_idx_generator`, so `PPO.update()` (ppo_asymmetry.py:178-194) can index it unchanged.

Three things are done the MI355X way:
* `collect(env, actions)` lets the step kernel write slot t+1 of the obs storage, the newest states frame, `rew_buf[t]` and
  `done_buf[t]` itself (taco_step_rollout), replacing five of the nine per-step torch copies of `store` and the two
  `obs.copy_(next_obs)` of the rollout loop (ppo_asymmetry.py:326-329).  The obs storage has H+1 slots; `obs_buf` is a view of
  the first H, `next_obs` a view of slot `step`.
* The STATE STACKS are kept as a FRAME RING `[H + T][N][D]` (T = states_len): the reference's frame stacks are shifted by one frame
  per step and never cleared, not even by a reset (fpv_asymmetry.py:392, :413), so the stack of slot t is exactly ring rows
  t .. t+T-1.  One frame (104 B) is written per env-step instead of a shifted T-frame stack, and the batched critic reads every frame
  once per block instead of T times.  `states_buf [H,N,T,D]` and `next_states [N,T,D]` are OVERLAPPING STRIDED VIEWS of the ring
  (`torch.as_strided`): index / gather them freely (`PPO.update`'s minibatch indexing), `.contiguous()` materialises; never write
  through them.  `store()` (the generic copying path) therefore keeps only the newest frame of the stack it is given (all T at slot
  0) -- exact for stacks that evolve as the reference's do (`check_shift=True` verifies it).
* `compute_returns_and_advantage` is three kernel launches (taco_gae) instead of ~9 H torch launches.
PyTorch only owns the memory.  There is no torch fallback for the arithmetic.
"""
import ctypes as C
import math

import torch

from . import _lib


class RolloutBuffer:
    def __init__(self, num_envs, obs_dim, obs_len, states_dim, states_len, act_dim, horizon_len, mini_batch_num, gamma, lam, device):
        self.num_envs, self.obs_dim, self.obs_len = num_envs, obs_dim, obs_len
        self.states_dim, self.states_len, self.act_dim = states_dim, states_len, act_dim
        self.horizon_len, self.mini_batch_num = horizon_len, mini_batch_num
        self.gamma, self.lam = gamma, lam
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TacoError("RolloutBuffer lives in HBM next to the env: device must be a cuda:N (HIP) device")
        self.lib = _lib.load()
        H, N, dev = horizon_len, num_envs, self.device
        self._obs_store = torch.zeros(H + 1, N, obs_len, obs_dim, dtype=torch.float32, device=dev)
        self._frames = torch.zeros(H + states_len, N, states_dim, dtype=torch.float32, device=dev)   # the states frame ring
        self.obs_buf = self._obs_store[:H]
        self.states_buf = self._stack_view(0, H)
        self.check_shift = False   # store(): verify that the stacks handed in are shifts of each other (debugging aid; one sync per call)
        self._gae_ws = torch.empty(self.lib.taco_gae_workspace_bytes(), dtype=torch.uint8, device=dev)
        self._alloc_small()
        self.step = 0

    def _alloc_small(self):
        """the nine per-step arrays of buffer_asymmetry.py:24-47 as views of ONE allocation: reset() clears them with a single launch"""
        H, N, dev, A = self.horizon_len, self.num_envs, self.device, self.act_dim
        widths = (("act_buf", A), ("rew_buf", 1), ("done_buf", 1), ("ret_buf", 1), ("value_buf", 1), ("adv_buf", 1), ("mu_buf", A), ("sigma_buf", A), ("logp_buf", 1))
        self._small = torch.zeros(sum(w for _, w in widths) * H * N, dtype=torch.float32, device=dev)
        off = 0
        for name, w in widths:
            setattr(self, name, self._small[off:off + H * N * w].view(H, N, w))
            off += H * N * w

    def _stack_view(self, slot0, slots):
        """[slots, N, T, D] view of the ring: stack (slot, env) = ring rows slot .. slot + T - 1 of that env (overlapping, read-only)"""
        N, T, D = self.num_envs, self.states_len, self.states_dim
        return torch.as_strided(self._frames, (slots, N, T, D), (N * D, D, N * D, 1), storage_offset=self._frames.storage_offset() + slot0 * N * D)

    # ---- the reference's API -------------------------------------------------------------------------------------------
    def store(self, obs, states, act, rew, log_prob, done, value, mu, sigma):
        """buffer_asymmetry.py:49-68, the generic (copying) path; usable with any env."""
        if self.step >= self.horizon_len:
            raise AssertionError("Rollout buffer overflow")
        t, T = self.step, self.states_len
        self.obs_buf[t].copy_(obs)
        states = states.view(self.num_envs, T, self.states_dim)
        if t == 0:
            self._frames[:T].copy_(states.transpose(0, 1))
        else:
            if self.check_shift and not torch.equal(states[:, :-1], self._stack_view(t, 1)[0][:, :-1]):
                raise AssertionError("store(): the state stack is not the previous one shifted by a frame: the frame-ring replay store cannot hold it")
            self._frames[t + T - 1].copy_(states[:, -1])
        self.rew_buf[t].copy_(rew.view(-1, 1))
        self.done_buf[t].copy_(done.view(-1, 1))
        self._store_policy(t, act, log_prob, value, mu, sigma)
        self.step += 1

    def reset(self):
        """buffer_asymmetry.py:70-91.  The reference re-allocates zeroed tensors; here the frame-stack storage is kept and
        the stacks the env last wrote (slot `step`) become slot 0 of the next rollout."""
        if self.step > 0:
            self._obs_store[0].copy_(self._obs_store[self.step])
            self._frames[:self.states_len].copy_(self._frames[self.step:self.step + self.states_len].clone() if self.step < self.states_len
                                                 else self._frames[self.step:self.step + self.states_len])
        self._small.zero_()  # all nine per-step arrays, in place (one launch): the pointers stay valid for captured graphs
        self.step = 0

    def compute_returns_and_advantage(self, last_values, normalize=True):
        """buffer_asymmetry.py:93-132 in three launches."""
        lv = last_values.to(device=self.device, dtype=torch.float32).contiguous()
        if lv.numel() != self.num_envs:
            raise ValueError("last_values must hold one value per env")
        s = _lib.stream_ptr(self.device)
        _lib.check(self.lib.taco_gae(self.rew_buf.data_ptr(), self.done_buf.data_ptr(), self.value_buf.data_ptr(), lv.data_ptr(),
                                     self.horizon_len, self.num_envs, float(self.gamma), float(self.lam),
                                     self.adv_buf.data_ptr(), self.ret_buf.data_ptr(), 1 if normalize else 0,
                                     self._gae_ws.data_ptr(), s))

    def batch_idx_generator(self):
        """buffer_asymmetry.py:134-139 (host-side index lists, as the reference)."""
        idx = torch.randperm(self.num_envs * self.horizon_len)
        return idx.reshape(self.mini_batch_num, -1).tolist()

    # ---- the fused rollout path ----------------------------------------------------------------------------------------
    @property
    def next_obs(self):
        """The frame stacks the policy acts on at the current step (slot `step`): what the reference keeps in `obs`."""
        return self._obs_store[self.step]

    @property
    def next_states(self):
        """[N, T, D] strided view of the ring: the state stacks of slot `step`"""
        return self._stack_view(self.step, 1)[0]

    def collect(self, env, actions, log_prob=None, value=None, mu=None, sigma=None, act=None):
        """env.step(actions) + store(...) of ppo_asymmetry.py:311-329 with the env writing this step's slots in place.
        `actions` is what the env executes (the reference clips first, :310), `act` what goes into act_buf (the un-clipped sample,
        :326; defaults to `actions`).  Returns (rewards [N], dones [N] int64 = env.reset_buf, time_outs [N] bool = env.timeout_buf)."""
        if self.step >= self.horizon_len:
            raise AssertionError("Rollout buffer overflow")
        if math.isfinite(env.clip_obs) or math.isfinite(env.clip_states):
            raise _lib.TacoError("collect() stores the env's unclamped frame stacks; with finite clipObservations / clipStates use env.step() + store()")
        if (env.num_envs, env.len_obs, env.num_obs, env.len_states, env.num_states) != (self.num_envs, self.obs_len, self.obs_dim, self.states_len, self.states_dim):
            raise ValueError("env and buffer geometry differ")
        t = self.step
        env.step_into(actions, self._obs_store[t], self._obs_store[t + 1], None, None, self.rew_buf[t], self.done_buf[t],
                      states_newest=self._frames[t + self.states_len])
        self._store_policy(t, actions if act is None else act, log_prob, value, mu, sigma)
        self.step += 1
        return self.rew_buf[t].view(-1), env.reset_buf, env.timeout_buf

    def run(self, env, policy, act_low=-1.0, act_high=1.0):
        """The whole rollout of ppo_asymmetry.py:308-342 as ONE C call (taco_rollout_run): horizon x (policy.act on slot t, clipped
        action -> env step writing slot t + 1), then the critic over all H + 1 slots in one batched pass (nothing before GAE reads a
        value) + the time-out bootstrap: 2 H + 3 launches enqueued back to back with no host work in between.  Fills every buffer `store` fills; returns last_values [N, 1] for
        compute_returns_and_advantage.  Equivalent to the act()/collect() loop (tests/test_rollout_gpu.py): bit for bit with
        ActorCritic(exact_critic=True); with the default critic (hardware LSTM cell, split-f16 operands on the 16-bit matrix pipe) the values agree to 2e-6 and everything else stays bit-identical."""
        if self.step != 0:
            raise AssertionError("run() fills a whole rollout: call reset() first")
        if math.isfinite(env.clip_obs) or math.isfinite(env.clip_states):
            raise _lib.TacoError("run() stores the env's unclamped frame stacks; with finite clipObservations / clipStates use env.step() + store()")
        if (env.num_envs, env.len_obs, env.num_obs, env.len_states, env.num_states) != (self.num_envs, self.obs_len, self.obs_dim, self.states_len, self.states_dim):
            raise ValueError("env and buffer geometry differ")
        H, N, dev = self.horizon_len, self.num_envs, self.device
        if not hasattr(self, "_run_scratch"):
            self._act_env = torch.empty(N, self.act_dim, device=dev)
            self._timeouts = torch.zeros(H, N, dtype=torch.uint8, device=dev)
            self._last_value = torch.empty(N, 1, device=dev)
            self._run_scratch = True
        b = _lib.RolloutBufs(self._obs_store.data_ptr(), self._frames.data_ptr(), self.act_buf.data_ptr(), self._act_env.data_ptr(),
                             self.rew_buf.data_ptr(), self.done_buf.data_ptr(), self.value_buf.data_ptr(), self.logp_buf.data_ptr(),
                             self.mu_buf.data_ptr(), self.sigma_buf.data_ptr(), self._timeouts.data_ptr(), self._last_value.data_ptr(),
                             policy.critic_workspace((H + 1) * N).data_ptr())
        policy._last_critic_rows = (H + 1) * N      # (what policy.clamped_words() / policy.check() look at afterwards)
        # The actor's noise counter is (env step word) + call_delta on a capturing stream (taco_rollout_run), so replays of a captured rollout
        # consume counters the Python-side policy.calls never sees: the first EAGER run() after replays re-derives it from the env's clock.
        if torch.cuda.is_current_stream_capturing():
            self._captured = True
        elif getattr(self, "_captured", False):
            self.sync_policy_counter(env, policy)
        self._call_delta = (policy.calls - self.lib.taco_peek_step_count(env._h)) & 0xffffffff   # == the C side's call_delta for this call
        s = _lib.stream_ptr(dev)
        _lib.check(self.lib.taco_rollout_run(env._h, C.byref(policy.cfg), policy._blob.data_ptr(), C.byref(b), H, C.c_uint64(policy.seed),
                                             C.c_uint32(policy.calls), float(self.gamma), float(act_low), float(act_high),
                                             env.reset_buf.data_ptr(), s))
        policy.calls += H
        self.step = H
        return self._last_value

    def sync_policy_counter(self, env, policy):
        """After replays of a captured run(): set policy.calls to the counter the NEXT rollout step will use (env step count + the offset the
        captured rollout was recorded with), so that eager policy.act() / run() calls draw fresh noise instead of repeating the streams the
        replays consumed.  run() does this by itself; call it before going back to policy.act() by hand.  Blocks (re-reads the device clock)."""
        if getattr(self, "_call_delta", None) is not None:
            policy.calls = (env.step_count + self._call_delta) & 0xffffffff
        self._captured = False

    @property
    def time_outs(self):
        """[H, N] uint8: extras["time_outs"] of every step of the last run()"""
        return self._timeouts

    def add_timeout_bootstrap(self, t, env_ids, time_out_value):
        """rewards_augmented[truncated] += gamma * V(s_t)  (ppo_asymmetry.py:320-324) on the stored reward of step t."""
        self.rew_buf[t].view(-1)[env_ids] += self.gamma * time_out_value.view(-1)

    def seed_stacks(self, env):
        """Adopt the env's current frame stacks as slot `step` (e.g. when attaching to an env that already stepped)."""
        self._obs_store[self.step].copy_(env.obs_buf)
        self._frames[self.step:self.step + self.states_len].copy_(env.states_buf.transpose(0, 1))

    def _store_policy(self, t, act, log_prob, value, mu, sigma):
        self.act_buf[t].copy_(act)
        if value is not None:
            self.value_buf[t].copy_(value.view(-1, 1))
        if mu is not None:
            self.mu_buf[t].copy_(mu)
        if sigma is not None:
            self.sigma_buf[t].copy_(sigma)
        if log_prob is not None:
            self.logp_buf[t].copy_(log_prob.view(-1, 1))


# the reference's class name, so `from taco_amd.rollout import PPOReplayBuffer` is a drop-in (buffer_asymmetry.py:9)
PPOReplayBuffer = RolloutBuffer
