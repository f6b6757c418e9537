"""Python host layer: the reference's VecTask surface over the HIP step library.

Mirrors what the PPO loop sees of `FpvBase(VecTask)` (fpv_asymmetry.py:34-211, vec_task_asymmetry.py:48-139, :231-254,
:290-375): constructor signature, `num_envs / num_obs / len_obs / num_states / len_states / num_acts`,
`observation_space / state_space / action_space`, writable `difficulty`, `obs_buf / states_buf / rew_buf / reset_buf /
progress_buf / timeout_buf`, `reset()`, `step()`, `reset_done()`, `reset_idx()`.

PyTorch is plumbing here: it owns device memory (the library borrows `data_ptr()`s) and provides the stream.  Every
`step()` is ONE kernel launch through the C ABI; there is no torch fallback for the arithmetic.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .config import flat_cfg


class Box:
    """Three-attribute stand-in for gym.spaces.Box (gym is not a dependency of the hot path)."""

    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape

    def __repr__(self):
        return f"Box{self.shape}"


_stream_ptr = _lib.stream_ptr


class FpvBase:
    """Vectorised FPV environment; subclasses only fix `task_mode` (registry names of tasks/__init__.py:30-39)."""

    task_mode = None
    num_commands = 2

    def __init__(self, cfg, rl_device="cuda:0", sim_device="cuda:0", graphics_device_id=-1, headless=True,
                 virtual_screen_capture=False, force_render=False, env_offset=0, num_envs_local=None, copy_outputs=True, kernel_form="auto",
                 lib=None, states_ring=None, fresh_outputs=False):
        self.cfg = cfg
        # fresh_outputs=True: step() returns NEWLY ALLOCATED obs / states tensors, as the reference does (vec_task_asymmetry.py:331-332:
        # torch.clamp(...) allocates) -- for callers that keep observations across steps without cloning.  The default (False) hands out
        # views of the env's alternating buffers / frame ring (see step(): ALIASING CONTRACT), which is what the PPO loop needs
        # (ppo_asymmetry.py:326-329 copies at once) and costs nothing; the clones cost two copy kernels per step.
        # rew / done / time_outs are the env's own buffers either way -- as in the reference (`.to(rl_device)` of a tensor already there
        # returns the tensor itself, :329, :334).
        self.fresh_outputs = bool(fresh_outputs)
        if self.task_mode is not None:
            cfg["task_mode"] = self.task_mode
        self.lib = lib or _lib.load()  # raises if libtaco_env.so is missing: no fallback
        self.device = torch.device(sim_device)
        if self.device.type != "cuda":
            raise _lib.TacoError("the step kernel runs on an MI355X; sim_device must be a cuda:N (HIP) device")
        self.rl_device = torch.device(rl_device)
        self.headless = headless
        self.copy_outputs = copy_outputs
        self._flat = flat_cfg(cfg, env_offset=env_offset, num_envs_local=num_envs_local)
        self._c = _lib.make_cfg(self._flat)

        env = cfg["env"]
        self.num_envs = self._c.num_envs
        self.num_envs_global = self._c.num_envs_global
        self.env_offset = self._c.env_offset
        self.num_agents = 1
        self.num_acts = self.num_actions = env["numActions"] = 4
        self.num_obs = env["numObservations"] = 18 + 1 + self.num_acts + 1 + self.num_commands  # 26, fpv_asymmetry.py:107
        self.num_states = env["numStates"] = self.num_obs
        self.len_obs = self._c.len_obs
        self.len_states = self._c.len_states
        self.control_freq_inv = self._c.control_freq_inv
        self.max_episode_length = self._c.max_episode_length
        self.clip_obs = env.get("clipObservations", math.inf)
        self.clip_states = env.get("clipStates", math.inf)
        self.clip_actions = env.get("clipActions", math.inf)
        self.dt = self._c.dt
        self.obs_space = Box(np.full((self.len_obs, self.num_obs), -np.inf), np.full((self.len_obs, self.num_obs), np.inf))
        self.state_space = Box(np.full((self.len_states, self.num_states), -np.inf), np.full((self.len_states, self.num_states), np.inf))
        self.act_space = Box(-np.ones(self.num_acts), np.ones(self.num_acts))

        # allocate_buffers, vec_task_asymmetry.py:231-254
        dev = self.device
        # obs_buf / states_buf: with copy_outputs (the default) TWO buffers each, used alternately -- step() reads the frame stacks from the
        # current pair and writes the next one (taco_step_rollout's prev / next), so what a step() returned stays untouched by the next
        # step() without any copy: the reference returns a clamped COPY of its buffers (vec_task_asymmetry.py:331-332), and with the
        # default clip of +inf that copy is the buffer's content.  `env.obs_buf` / `env.states_buf` are always the current pair.
        self._want_pp = bool(copy_outputs)
        self._finite_clip = bool(math.isfinite(self.clip_obs) or math.isfinite(self.clip_states))
        # A state stack (len_states > 1, the documented 5) behind step() is a FRAME RING (taco_bind_states_ring): the kernel writes ONE frame per
        # env-step and `states_buf` / the tensor step() returns is a strided [num_envs, len_states, 26] view of it -- the reference's shifted
        # stack (fpv_asymmetry.py:413) moves 9 frames per env-step for the same content.  (copy_outputs=False keeps the contiguous in-place
        # stack.  Finite clipObservations / clipStates: a SECOND ring receives the clamped frames -- clamping a stack is clamping its frames --
        # and step() returns the view of that one; `states_buf` stays the unclamped stack, as in the reference.)
        # states_ring: None = with copy_outputs; True / False force it (ShardedEnv steps in place on the obs buffer but keeps the ring)
        self._ring_on = (self._want_pp if states_ring is None else bool(states_ring)) and self.len_states > 1
        npp = 2 if self._want_pp else 1
        self._obs_pp = [torch.zeros((self.num_envs, self.len_obs, self.num_obs), device=dev, dtype=torch.float32) for _ in range(npp)]
        self._states_pp = [] if self._ring_on else \
            [torch.zeros((self.num_envs, self.len_states, self.num_states), device=dev, dtype=torch.float32) for _ in range(npp)]
        self._pp = 0
        # a graph captured around step() would freeze ONE (current -> next) pair of the alternating obs buffers: with an obs stack every replay
        # would shift the never-updated `current` one (refused in step())
        self._capture_unsafe = self._want_pp and self.len_obs > 1 and not self._finite_clip
        self._ring_clamped = self._ring_on and self._finite_clip   # (step() hands out clamped copies: a second ring holds the clamped frames)
        self.rew_buf = torch.zeros(self.num_envs, device=dev, dtype=torch.float32)
        self.reset_buf = torch.ones(self.num_envs, device=dev, dtype=torch.long)
        self.timeout_buf = torch.zeros(self.num_envs, device=dev, dtype=torch.bool)
        self.extras = {}
        self.obs_dict = {}
        # finite clipObservations / clipStates: the clamped copies step() returns are written by the step kernel itself (its OUT
        # instantiation) into two alternating pairs -- no torch.clamp launch, no allocation per step
        self._out = [(torch.zeros_like(self._obs_pp[0]), None if self._ring_on else torch.zeros_like(self._states_pp[0])) for _ in range(2)] if self._finite_clip else None
        self._out_k = 0
        self._step_io = {}
        self._same_device = self.rl_device == self.device
        self._progress = torch.zeros(self.num_envs, device=dev, dtype=torch.int32)

        nbytes = self.lib.taco_workspace_bytes(C.byref(self._c))
        self._workspace = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        self._h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(self.lib.taco_create(C.byref(self._c), dev.index or 0, C.c_void_p(self._workspace.data_ptr()), nbytes,
                                            _stream_ptr(dev), C.byref(self._h)))
        self._difficulty = float(cfg["difficulty"])
        if kernel_form != "auto":
            self.set_kernel_form(kernel_form)
        if self._ring_on:
            front, row_bytes = self.len_states - 1, self.num_envs * self.num_states * 4
            # <= 1 GiB of ring where that leaves period >= len_states + 1 (what step t returned survives step t + 1); twins cost front / period of a frame per step
            self._st_period = int(min(64, max(front + 2, (1 << 30) // row_bytes - front)))
            self._st_ring = torch.zeros((self._st_period + front, self.num_envs, self.num_states), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                torch.cuda.synchronize()   # (the bind resets the ring phase on the device, behind the workspace's initialisation)
                _lib.check(self.lib.taco_bind_states_ring(self._h, self._st_ring.data_ptr(), self._st_period + front), self.lib)
            self._st_views = [self._st_ring[k:k + self.len_states].permute(1, 0, 2) for k in range(self._st_period)]
            self._st_ring_c = torch.zeros_like(self._st_ring) if self._ring_clamped else None
            self._st_views_c = [self._st_ring_c[k:k + self.len_states].permute(1, 0, 2) for k in range(self._st_period)] if self._ring_clamped else None
            self._st_last = self._st_period - 1   # the window of the last step (before the first: zeros, like every other)
            self._st_phase = C.c_int32(0)

    def set_kernel_form(self, name):
        """pin one of the six instantiations of the step kernel (`_lib.FORMS`; "auto" = the library's choice for this env count)"""
        _lib.check(self.lib.taco_set_kernel_form(self._h, _lib.FORMS[name]), self.lib)

    def set_rollout_fusion(self, on=True):
        """taco_rollout_run's persistent actor + step kernel (taco_fused.hpp) on / off for this env (on by default where it applies: the quad form -- 16
        envs per workgroup -- up to 8 192 envs, the one-lane form -- 64 envs per workgroup, round 6 -- above); on="quad" (= "force") / "lane": that form
        whatever the env count (A/B measurements: above 8 192 envs the quad form's workgroups queue)"""
        _lib.check(self.lib.taco_set_rollout_fusion(self._h, {"force": 2, "quad": 2, "lane": 3}.get(on, 1 if on else 0) if isinstance(on, str) else (1 if on else 0)), self.lib)

    def bind_rollout_stamps(self, stamps):
        """profiling: a [136 + ceil(num_envs / 16)] int64 device tensor workgroup 0 of the persistent rollout kernel fills (SIMD of its 8 wavefronts, per-step clocks); None unbinds"""
        self._rollout_stamps = stamps
        _lib.check(self.lib.taco_bind_rollout_stamps(self._h, stamps.data_ptr() if stamps is not None else None), self.lib)

    @property
    def kernel_form(self):
        f = self.lib.taco_get_kernel_form(self._h)
        return next(k for k, v in _lib.FORMS.items() if v == f)

    # ---- spaces / attributes the callers read (train_fpv_asymmetry_ppo.py:372-393, ppo_asymmetry.py:42-52)
    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def difficulty(self):
        return self._difficulty

    @difficulty.setter
    def difficulty(self, value):  # ppo_asymmetry.py:173-175 writes env.difficulty every epoch
        self._difficulty = float(value)
        self.cfg["difficulty"] = self._difficulty
        _lib.check(self.lib.taco_set_difficulty(self._h, self._difficulty))

    @property
    def obs_buf(self):
        """[num_envs, len_obs, 26] frame stacks, newest frame last, unclamped (fpv_asymmetry.py:392): the CURRENT buffer of the alternating pair"""
        return self._obs_pp[self._pp]

    @property
    def states_buf(self):
        """[num_envs, len_states, 26]; with the frame ring a STRIDED, READ-ONLY view of the window the last step filled (each frame contiguous,
        frames num_envs * 26 floats apart): reads, .cpu(), indexing work; .view() needs .reshape() / .contiguous().  Do NOT write into it: a
        write reaches rows [ph, ph + len_states) only, never the twin rows the ring keeps for the wrap-around, so it is lost as soon as the
        window wraps -- load_stacks(states=...) is the way to set the stack (it rebuilds the twins).  LIFETIME of a view handed out earlier
        (by step() or this property): its oldest frame is overwritten period - len_states + 1 >= 2 steps later (the ring's period is at
        least len_states + 1), i.e. what step t returned is intact while step t + 1 runs, as with the alternating obs buffers.
        Outside graph mode (include/taco_env.h taco_graph_mode) the library answers from the host's copy of the clock (no sync); in graph mode it blocks."""
        if self._ring_on:
            # the LIBRARY is the source of truth for the window (round 5's advisor: a cached phase goes stale when something other than step() advances the
            # ring -- a direct taco_step_rollout with states_next = NULL on the bound ring, as tools do).  Outside graph mode taco_states_ring_row is a
            # host-side lookup (it never synchronises there); in graph mode it re-reads the device-resident clock (device-wide sync).  step() itself does
            # not come here: taco_step_ring hands it the phase.
            ph = self.lib.taco_states_ring_row(self._h)
            if ph < 0:
                raise _lib.TacoError(f"libtaco_env: {self.lib.taco_last_error().decode()}")
            self._st_last = ph
            return self._st_views[self._st_last]
        return self._states_pp[self._pp]

    def release_graphs(self):
        """taco_release_graphs: the caller's word that no HIP graph holding launches of this env will be replayed any more.  From the first
        captured launch on the env is in GRAPH MODE -- every step() / states_buf / step_count / get_state re-reads the device-resident clock
        and blocks (a replay may have advanced it behind the host's back) --; this ends it (one last blocking re-read).  A replay after the
        release is reported by check()."""
        _lib.check(self.lib.taco_release_graphs(self._h), self.lib)
        if self._ring_on:
            ph = self.lib.taco_states_ring_row(self._h)
            if ph < 0:
                raise _lib.TacoError(f"libtaco_env: {self.lib.taco_last_error().decode()}")
            self._st_last = ph

    def load_stacks(self, obs=None, states=None):
        """make `obs` / `states` ([num_envs, len, 26]) the env's current frame stacks (checkpoint restore)"""
        if obs is not None:
            self.obs_buf.copy_(obs.to(self.device))
        if states is not None:
            states = states.to(self.device)
            if not self._ring_on:
                self.states_buf.copy_(states)
                return
            # ring: restart at phase 0 -- the window of the "last step" is rows [period - 1, period - 1 + len), and the frames the NEXT windows
            # share with it must also sit in the twin rows [0, len - 1)
            front = self.len_states - 1
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
                _lib.check(self.lib.taco_bind_states_ring(self._h, self._st_ring.data_ptr(), self._st_period + front), self.lib)
            self._st_last = self._st_period - 1
            self._st_views[self._st_last].copy_(states)
            self._st_ring[0:front].copy_(states[:, 1:].transpose(0, 1))
            if self._ring_clamped:
                self._st_ring_c.copy_(torch.clamp(self._st_ring, -self.clip_states, self.clip_states))

    @property
    def progress_buf(self):
        """progress_buf of the reference (fpv_asymmetry.py:375): one row of the state, exported by a one-row kernel"""
        _lib.check(self.lib.taco_get_field(self._h, _lib.NUM_FIELDS - 2, self._progress.data_ptr(), _stream_ptr(self.device).value), self.lib)
        return self._progress.to(torch.long)

    @property
    def randomize_buf(self):
        """vec_task_asymmetry.py:252 / fpv_asymmetry.py:376: incremented by one per step() for every env and never cleared on this path
        (its only consumer, apply_randomizations, is never called by the task) -- i.e. the number of steps taken"""
        return torch.full((self.num_envs,), int(self.step_count), device=self.device, dtype=torch.long)

    def tracks_rpy(self, env_index=0):
        """is copter_rpy_old / copter_rpy_continuous (blob rows 20..25) of this env kept up to date?  (flip envs always; every env with
        cfg['record_flag'], TACO_F_TRACK_RPY)"""
        if self._flat.get("record_flag", False):
            return True
        mode, gid = self._flat["task_mode"], self.env_offset + int(env_index)
        return mode == "flip" or (mode == "mix" and gid >= int(self.num_envs_global / 3 * 2))

    def check(self):
        """raise if any step kernel recorded a sticky error since the env was created (blocks on the current stream)"""
        _lib.check(self.lib.taco_check(self._h, _stream_ptr(self.device).value), self.lib)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.taco_destroy(h)
            self._h = None

    # ---- VecTask API
    def _outputs(self):
        obs = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs) if (self.copy_outputs or math.isfinite(self.clip_obs)) else self.obs_buf
        st = torch.clamp(self.states_buf, -self.clip_states, self.clip_states) if (self.copy_outputs or math.isfinite(self.clip_states)) else self.states_buf
        self.obs_dict["obs"] = obs.to(self.rl_device)
        self.obs_dict["states"] = st.to(self.rl_device)
        return self.obs_dict

    def reset(self):
        """vec_task_asymmetry.py:352-361: returns the (zero) buffers; the real reset happens inside the first step()."""
        return self._outputs()

    def step_raw(self, actions):
        """One taco_step launch on the current stream; updates obs_buf/states_buf/rew_buf/reset_buf/timeout_buf in place."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.shape != (self.num_envs, self.num_acts):
            raise ValueError(f"actions must be [{self.num_envs}, {self.num_acts}], got {tuple(actions.shape)}")
        if self._ring_on:   # (in place for obs, the bound frame ring for the state stack)
            io = _lib.RolloutIO(actions.data_ptr(), None, self.obs_buf.data_ptr(), None, None, self.rew_buf.data_ptr(), self.reset_buf.data_ptr(),
                                self.timeout_buf.data_ptr(), None, None, self._st_ring_c.data_ptr() if self._ring_clamped else None)
            self._ring_step(io)
            return
        rc = self.lib.taco_step(self._h, actions.data_ptr(), self.obs_buf.data_ptr(), self.states_buf.data_ptr(), self.rew_buf.data_ptr(),
                                self.reset_buf.data_ptr(), self.timeout_buf.data_ptr(), _stream_ptr(self.device).value)
        _lib.check(rc)

    def _ring_step(self, io):
        """one launch on the frame ring (taco_step_ring); remembers which window it fills"""
        rc = self.lib.taco_step_ring(self._h, C.byref(io), _stream_ptr(self.device), C.byref(self._st_phase))
        if rc != 0:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.TacoError("VecTask.step() with a state stack cannot be captured into a HIP graph (each replay fills another window of the "
                                     "frame ring): capture step_raw() of an env built with copy_outputs=False, RolloutBuffer.run(), or taco_step_rollout")
            _lib.check(rc, self.lib)
        self._st_last = self._st_phase.value

    def step_into(self, actions, obs_prev, obs_next, states_prev, states_next, rew, done_f32=None, states_newest=None):
        """taco_step_rollout: like step_raw, but the frame stacks are read from `*_prev` and written to `*_next` (replay-buffer
        slots, see taco_amd/rollout.py), the reward goes to `rew` and the new done flags also to `done_f32` (fp32).
        `states_newest` ([num_envs, 26], instead of states_prev / states_next): only the newest states frame is written, into that row of
        a frame ring (taco_rollout_io.states_newest_only) -- no stack is shifted.
        reset_buf / timeout_buf stay the env's own.  Tensors must be contiguous fp32 on the env's device."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.shape != (self.num_envs, self.num_acts):
            raise ValueError(f"actions must be [{self.num_envs}, {self.num_acts}], got {tuple(actions.shape)}")
        if (states_newest is None) == (states_next is None):
            raise ValueError("step_into: give either states_prev / states_next (stacks) or states_newest (one frame per env)")
        frame_shape = (self.num_envs, self.num_states)
        shapes = ((obs_prev, self.obs_buf.shape), (obs_next, self.obs_buf.shape), (states_prev, self.states_buf.shape),
                  (states_next, self.states_buf.shape), (states_newest, frame_shape), (rew, None), (done_f32, None))
        for t, shp in shapes:
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("step_into: buffers must be contiguous fp32 tensors on the env's device")
            if (shp is not None and tuple(t.shape) != tuple(shp)) or (shp is None and t.numel() != self.num_envs):
                raise ValueError(f"step_into: buffer of shape {tuple(t.shape)} does not match the env")
        if states_newest is not None:
            io = _lib.RolloutIO(actions.data_ptr(), obs_prev.data_ptr(), obs_next.data_ptr(), None, states_newest.data_ptr(),
                                rew.data_ptr(), self.reset_buf.data_ptr(), self.timeout_buf.data_ptr(),
                                done_f32.data_ptr() if done_f32 is not None else None, None, None, 1)
        else:
            io = _lib.RolloutIO(actions.data_ptr(), obs_prev.data_ptr(), obs_next.data_ptr(), states_prev.data_ptr(), states_next.data_ptr(),
                                rew.data_ptr(), self.reset_buf.data_ptr(), self.timeout_buf.data_ptr(),
                                done_f32.data_ptr() if done_f32 is not None else None)
        _lib.check(self.lib.taco_step_rollout(self._h, C.byref(io), _stream_ptr(self.device)), self.lib)

    def step(self, actions):
        """vec_task_asymmetry.py:290-334: ONE kernel launch, nothing else on the device.

        What is returned as obs / states: the reference returns clamp(obs_buf, +-clipObservations) -- a fresh tensor.  Here, with the default
        clip of +inf, step() reads the frame stacks from the current buffer pair and writes the OTHER pair, which becomes `env.obs_buf` /
        `env.states_buf` and is returned as it is (no copy exists anywhere; the launch is the one step_raw() makes); with a finite clip the
        same launch also writes the clamped copies (the kernel's OUT instantiation) into two alternating output pairs.
        ALIASING CONTRACT (differs from the reference's freshly allocated tensors): the obs / states tensors of step t are overwritten by
        step t + 2; rew / done / time_outs are the env's own buffers, overwritten by step t + 1.  The PPO loop copies them into its replay
        buffer straight away (ppo_asymmetry.py:326-329), which is the intended use; a caller that keeps observations across more than one
        step must clone() -- or build the env with fresh_outputs=True, which returns newly allocated obs / states like the reference.
        copy_outputs=False: everything in place, the returned tensors ARE the buffers."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.shape != (self.num_envs, self.num_acts):
            raise ValueError(f"actions must be [{self.num_envs}, {self.num_acts}], got {tuple(actions.shape)}")
        if self._capture_unsafe and torch.cuda.is_current_stream_capturing():
            raise _lib.TacoError("VecTask.step() with an observation stack alternates between two buffer pairs and cannot be captured into a HIP graph: "
                                 "capture step_raw() of an env built with copy_outputs=False")
        cur = self._pp
        nxt = cur ^ 1 if (self._want_pp and not self._finite_clip) else cur   # (finite clip: the clamped copies are what survives)
        k = self._out_k if self._finite_clip else 0
        io = self._step_io.get((cur, k))
        if io is None:   # the argument block of this (buffer pair, output pair) combination: built once, only the action pointer changes
            obs_out, states_out = self._out[k] if self._finite_clip else (None, None)
            ring = self._ring_on
            if ring and self._ring_clamped:
                states_out = self._st_ring_c   # (the ring of clamped frames: the kernel writes the same rows of it)
            io = _lib.RolloutIO(None, self._obs_pp[cur].data_ptr() if nxt != cur else None, self._obs_pp[nxt].data_ptr(),
                                self._states_pp[cur].data_ptr() if (nxt != cur and not ring) else None,
                                None if ring else self._states_pp[nxt].data_ptr(), self.rew_buf.data_ptr(),
                                self.reset_buf.data_ptr(), self.timeout_buf.data_ptr(), None,
                                obs_out.data_ptr() if obs_out is not None else None, states_out.data_ptr() if states_out is not None else None)
            self._step_io[(cur, k)] = io
        io.actions = actions.data_ptr()
        if self._ring_on:
            self._ring_step(io)
        else:
            rc = self.lib.taco_step_rollout(self._h, C.byref(io), _stream_ptr(self.device))
            if rc != 0:
                _lib.check(rc, self.lib)
        self._pp = nxt
        if self._finite_clip:
            self._out_k ^= 1
            obs, st = self._out[k]
            if self._ring_on:
                st = self._st_views_c[self._st_last]
        elif self._ring_on:
            obs, st = self._obs_pp[nxt], self._st_views[self._st_last]
        else:
            obs, st = self._obs_pp[nxt], self._states_pp[nxt]
        if self.fresh_outputs:
            obs, st = obs.clone(), st.clone(memory_format=torch.contiguous_format)
        if self._same_device:
            self.obs_dict["obs"], self.obs_dict["states"], self.extras["time_outs"] = obs, st, self.timeout_buf
            return self.obs_dict, self.rew_buf, self.reset_buf, self.extras
        self.obs_dict["obs"], self.obs_dict["states"] = obs.to(self.rl_device), st.to(self.rl_device)
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def reset_idx(self, env_ids):
        """fpv_asymmetry.py:475-517: re-initialise the envs `env_ids` NOW (fresh state, controller memory, delay line, target; command
        re-drawn for them and for envs at progress 500; progress_buf and reset_buf of `env_ids` cleared, :510-511) -- one launch of the
        RESET_ONLY instantiation on a scratch mask, the step clock does not advance.  Inside step() the same reset is fused into the step
        kernel, driven by reset_buf; to only MARK envs for the next step() write env.reset_buf[ids] = 1, as the reference's callers do.
        (Called with ids other than the flagged ones -- which the reference's own callers never do -- the reference re-draws the commands of
        the envs flagged in reset_buf, :500-503 / :587-604; this one those of `env_ids`.  env_ids: indices, or a bool mask.)"""
        ids = torch.as_tensor(env_ids, device=self.device)
        ids = ids.flatten().nonzero().flatten() if ids.dtype == torch.bool else ids.to(torch.long).flatten()   # (a bool mask means mask.nonzero())
        if ids.numel() == 0:
            return
        if not hasattr(self, "_reset_mask"):
            self._reset_mask = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        self._reset_mask.zero_()
        self._reset_mask[ids] = 1
        _lib.check(self.lib.taco_reset_done(self._h, self._reset_mask.data_ptr(), _stream_ptr(self.device).value), self.lib)
        self.reset_buf[ids] = 0

    def reset_done(self):
        """vec_task_asymmetry.py:363-375: reset_idx on the flagged envs NOW (taco_reset_done: fresh state and command, reset_buf and
        progress_buf cleared), then the observation dict as it is -- like the reference, nothing recomputes the frames -- and the ids."""
        done_env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        _lib.check(self.lib.taco_reset_done(self._h, self.reset_buf.data_ptr(), _stream_ptr(self.device).value))
        return self._outputs(), done_env_ids

    def zero_actions(self):
        return torch.zeros((self.num_envs, self.num_acts), dtype=torch.float32, device=self.rl_device)

    # ---- state blob (parity tests, checkpoint / restore)
    def get_state(self):
        blob = torch.empty((_lib.BLOB_ROWS, self.num_envs), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.taco_get_state(self._h, blob.data_ptr(), _stream_ptr(self.device).value))
        return blob

    def set_state(self, blob):
        blob = blob.to(device=self.device).contiguous()
        if tuple(blob.shape) != (_lib.BLOB_ROWS, self.num_envs) or blob.element_size() != 4:
            raise ValueError(f"state blob of shape {tuple(blob.shape)} ({blob.dtype}) does not fit this env: expected ({_lib.BLOB_ROWS}, {self.num_envs}) 32-bit "
                             "words -- a blob holds one fixed set of envs (num_envs, len_obs and len_states are part of its layout)")
        _lib.check(self.lib.taco_set_state(self._h, blob.data_ptr(), _stream_ptr(self.device).value))

    @property
    def step_count(self):
        n = self.lib.taco_get_step_count(self._h)
        if n < 0:
            raise _lib.TacoError(f"libtaco_env: taco_get_step_count failed: {self.lib.taco_last_error().decode()}")
        return n

    @step_count.setter
    def step_count(self, n):
        _lib.check(self.lib.taco_set_step_count(self._h, int(n)))

    def launch_geometry(self):
        g, b = C.c_int(), C.c_int()
        _lib.check(self.lib.taco_launch_geometry(self._h, C.byref(g), C.byref(b)))
        return g.value, b.value


    def phase_stamps(self, actions, steps=20, back_to_back=False):
        """Profiling aid: run `steps` steps with the phase stamps bound and return the mean shader-clock ticks between the
        phase boundaries of workgroup 0 (entry->loads, ->pre-phase, ->substeps, ->stores+frames, ->end).  back_to_back: launch the steps
        without host synchronisation in between and read the LAST launch's stamps (warm instruction cache, as in a rollout)."""
        st = torch.zeros(16, dtype=torch.int64, device=self.device)
        self._last_stamps = st
        _lib.check(self.lib.taco_bind_phase_stamps(self._h, st.data_ptr()))
        acc = torch.zeros(5, dtype=torch.float64)
        try:
            for _ in range(steps):
                self.step_raw(actions)
                if not back_to_back:
                    t = st.cpu()
                    acc += (t[1:6] - t[0:5]).double()
            if back_to_back:
                t = st.cpu()
                acc = (t[1:6] - t[0:5]).double() * steps
        finally:
            _lib.check(self.lib.taco_bind_phase_stamps(self._h, None))
        return (acc / steps).tolist()

    def occupancy(self):
        """(resident workgroups per CU, LDS bytes per workgroup) of the step kernel this env launches"""
        b, l = C.c_int(), C.c_int()
        _lib.check(self.lib.taco_occupancy(self._h, C.byref(b), C.byref(l)))
        return b.value, l.value


class FpvPos(FpvBase):
    task_mode = "pos"


class FpvRotate(FpvBase):
    task_mode = "rotate"


class FpvFlip(FpvBase):
    task_mode = "flip"


class FpvMix(FpvBase):
    task_mode = "mix"


# tasks/__init__.py:30-39
isaacgym_task_map = {"Fpv_pos": FpvPos, "Fpv_rotate": FpvRotate, "Fpv_flip": FpvFlip, "Fpv_mix": FpvMix}
