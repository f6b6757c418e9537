"""ctypes front-end of the CPU ORACLE (oracle/libtaco_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under taco_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtaco_oracle.so")

TASKS = {"pos": 0, "rotate": 1, "flip": 2, "mix": 3}
FLAG_BITS = {
    "random_copter_pos": 0, "random_copter_quat": 1, "random_copter_vel": 2, "random_target_pos": 3,
    "random_target_yaw": 4, "battery_consumption": 5, "random_voltage": 6, "rotor_noise": 7, "rotor_delay": 8,
    "rotor_response": 9, "random_rotordynamic_coe": 10, "random_rotor_delay": 11, "random_rotor_response": 12,
    "random_rotor_speed": 13, "random_aerodynamic_coe": 14, "ramdom_delay_time": 15, "ramdom_deploy_time": 16,
    "random_command": 17, "observation_noise": 18, "world_rate_roundtrip": 19,
}
NUM_FIELDS = 67
BLOB_ROWS = 67 + 400
FIELD = dict(PX=0, QX=3, VX=7, WX=10, TPX=13, TQX=16, RPY_OLD=20, RPY_CONT=23, PID_PREV=26, PID_INT=29, BAT_E=32,
             BAT_U1=33, BAT_T=34, BAT_V=35, OMEGA=36, ACT=40, ACT_OLD=44, CMD=48, FLIP_RADIAN=50, TAU=51, OPARA=55,
             CF=60, CT=61, DX=62, DY=63, KT=64, PROGRESS=65, DELAY_LEN=66)


class OrcCfg(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("env_offset", C.c_int32), ("num_envs_global", C.c_int32), ("task_mode", C.c_int32),
        ("len_obs", C.c_int32), ("len_states", C.c_int32), ("control_freq_inv", C.c_int32), ("substeps", C.c_int32),
        ("max_episode_length", C.c_int32), ("delay_time", C.c_int32), ("flags", C.c_uint32), ("_pad", C.c_uint32),
        ("seed", C.c_uint64), ("dt", C.c_double), ("rotor_response_time", C.c_double), ("difficulty", C.c_double),
        ("clip_actions", C.c_double), ("clip_obs", C.c_double), ("clip_states", C.c_double), ("mass", C.c_double),
        ("inertia", C.c_double * 3), ("arm_x", C.c_double), ("arm_y", C.c_double), ("gravity_z", C.c_double),
    ]


def build(force=False):
    """Compile oracle/libtaco_oracle.so with the committed Makefile (gcc only, no reference sources involved)."""
    srcs = [os.path.join(_HERE, f) for f in ("taco_oracle.c", "taco_policy_oracle.c", "taco_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libtaco_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_create.argtypes = [C.POINTER(OrcCfg), C.POINTER(C.c_void_p)]
        _lib.orc_create.restype = C.c_int
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_set_difficulty.argtypes = [C.c_void_p, C.c_double]
        _lib.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_set_world_rate_roundtrip.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_reset_done.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_reset_done.restype = C.c_int
        _lib.orc_step.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        _lib.orc_step.restype = C.c_int
        _lib.orc_get_state.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_set_state.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_step_count.argtypes = [C.c_void_p]
        _lib.orc_step_count.restype = C.c_int64
        _lib.orc_set_step_count.argtypes = [C.c_void_p, C.c_int64]
        for name in ("orc_sinf", "orc_cosf", "orc_asinf", "orc_logf"):
            getattr(_lib, name).argtypes = [C.c_float]
            getattr(_lib, name).restype = C.c_float
        _lib.orc_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.orc_atan2f.restype = C.c_float
        _lib.orc_uniform.argtypes = [C.c_uint32]
        _lib.orc_uniform.restype = C.c_float
        _lib.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    return _lib


def make_cfg(d):
    """dict (the keys of taco_amd.config.flat_cfg) -> OrcCfg"""
    c = OrcCfg()
    flags = 0
    for k, bit in FLAG_BITS.items():
        if d.get(k, False):
            flags |= 1 << bit
    c.num_envs = d["num_envs"]
    c.env_offset = d.get("env_offset", 0)
    c.num_envs_global = d.get("num_envs_global", d["num_envs"])
    c.task_mode = TASKS[d["task_mode"]] if isinstance(d["task_mode"], str) else d["task_mode"]
    c.len_obs = d.get("len_obs", 1)
    c.len_states = d.get("len_states", 1)
    c.control_freq_inv = d.get("control_freq_inv", 10)
    c.substeps = d.get("substeps", 2)
    c.max_episode_length = d.get("max_episode_length", 1000)
    c.delay_time = d.get("delay_time", 20)
    c.flags = flags
    c.seed = d.get("seed", 0)
    c.dt = d.get("dt", 0.001)
    c.rotor_response_time = d.get("rotor_response_time", 0.017)
    c.difficulty = d.get("difficulty", 1.0)
    c.clip_actions = d.get("clip_actions", float("inf"))
    c.clip_obs = d.get("clip_obs", float("inf"))
    c.clip_states = d.get("clip_states", float("inf"))
    c.mass = d["mass"]
    for i in range(3):
        c.inertia[i] = d["inertia"][i]
    c.arm_x = d["arm_x"]
    c.arm_y = d["arm_y"]
    c.gravity_z = d.get("gravity_z", -9.81)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleEnv:
    """The reference's VecTask.step() semantics on CPU (numpy in/out)."""

    def __init__(self, cfg_dict, threads=1, world_rate_roundtrip=False):
        self.cfg = make_cfg(cfg_dict)
        self.n = self.cfg.num_envs
        self._h = C.c_void_p()
        rc = lib().orc_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            raise ValueError("orc_create rejected the configuration")
        lib().orc_set_threads(self._h, threads)
        if world_rate_roundtrip:  # the reference's data flow for the angular rate (see orc_set_world_rate_roundtrip)
            lib().orc_set_world_rate_roundtrip(self._h, 1)
        self.obs_buf = np.zeros((self.n, self.cfg.len_obs, 26), np.float32)
        self.states_buf = np.zeros((self.n, self.cfg.len_states, 26), np.float32)
        self.rew_buf = np.zeros(self.n, np.float32)
        self.reset_buf = np.ones(self.n, np.int64)
        self.timeout_buf = np.zeros(self.n, np.uint8)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_difficulty(self, d):
        lib().orc_set_difficulty(self._h, float(d))

    def step(self, actions):
        a = _f32(actions)
        assert a.shape == (self.n, 4)
        lib().orc_step(self._h, _p(a), _p(self.obs_buf), _p(self.states_buf), _p(self.rew_buf), _p(self.reset_buf), _p(self.timeout_buf))
        return self.obs_buf, self.states_buf, self.rew_buf, self.reset_buf, self.timeout_buf

    def reset_done(self):
        """VecTask.reset_done (VT:363-375): the flagged envs are reset now; returns their ids"""
        ids = np.nonzero(self.reset_buf)[0]
        lib().orc_reset_done(self._h, _p(self.reset_buf))
        return ids

    def reset_idx(self, ids):
        """FpvBase.reset_idx(env_ids) (FA:475-517) called directly: those envs are re-initialised now, their reset_buf / progress_buf cleared"""
        mask = np.zeros(self.n, np.int64)
        mask[np.asarray(ids, np.int64)] = 1
        lib().orc_reset_done(self._h, _p(mask))
        self.reset_buf[np.asarray(ids, np.int64)] = 0

    def get_state(self):
        blob = np.zeros((BLOB_ROWS, self.n), np.uint32)
        lib().orc_get_state(self._h, _p(blob))
        return blob

    def set_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint32)
        assert blob.shape == (BLOB_ROWS, self.n)
        lib().orc_set_state(self._h, _p(blob))

    @property
    def step_count(self):
        return lib().orc_step_count(self._h)

    @step_count.setter
    def step_count(self, n):
        lib().orc_set_step_count(self._h, int(n))


# ---- sub-model wrappers (golden-vector tests)
def _call(name, n, *arrs):
    fn = getattr(lib(), name)
    fn.restype = None
    args = [C.c_int(n)]
    for a in arrs:
        if isinstance(a, np.ndarray):
            args.append(_p(a))
        else:
            args.append(a)
    fn(*args)


def quat_mul(a, b):
    a, b = _f32(a), _f32(b); out = np.empty_like(a); _call("orc_quat_mul", len(a), a, b, out); return out


def quat_rotate_inv(q, v):
    q, v = _f32(q), _f32(v); out = np.empty_like(v); _call("orc_quat_rotate_inv", len(q), q, v, out); return out


def euler_xyz_v1(q):
    q = _f32(q); out = np.empty((len(q), 3), np.float32); _call("orc_euler_xyz_v1", len(q), q, out); return out


def quat_from_euler_xyz(rpy):
    rpy = _f32(rpy); out = np.empty((len(rpy), 4), np.float32); _call("orc_quat_from_euler_xyz", len(rpy), rpy, out); return out


def quat_diff_rad(a, b):
    a, b = _f32(a), _f32(b); out = np.empty(len(a), np.float32); _call("orc_quat_diff_rad", len(a), a, b, out); return out


def quat_to_matrix(q):
    q = _f32(q); out = np.empty((len(q), 9), np.float32); _call("orc_quat_to_matrix", len(q), q, out); return out


def pid_step(dt, des, cur, prev, integ):
    des, cur = _f32(des), _f32(cur); out = np.empty_like(des)
    _call("orc_pid_step", len(des), C.c_float(dt), des, cur, prev, integ, out); return out


def allocator(u):
    u = _f32(u).copy(); thr = np.empty_like(u); _call("orc_allocator", len(u), u, thr); return thr, u


def real2sim(f, t):
    f, t = _f32(f), _f32(t); fs, ts = np.empty_like(f), np.empty_like(t); _call("orc_real2sim", len(f), f, t, fs, ts); return fs, ts


def battery_step(enabled, dt, Pm, E, u1, t):
    Pm = _f32(Pm); V = np.empty_like(Pm)
    _call("orc_battery_step", len(Pm), C.c_int(int(enabled)), C.c_float(dt), Pm, E, u1, t, V); return V


def power(omega):
    omega = _f32(omega); out = np.empty(len(omega), np.float32); _call("orc_power", len(omega), omega, out); return out


def rotor_step(V, thr, tau, para, omega):
    _call("orc_rotor_step", len(omega), _f32(V), _f32(thr), _f32(tau), _f32(para), omega); return omega


def aero(cf_ct, d, kt, vb, om):
    n = len(vb); rf = np.empty((n, 4), np.float32); rt = np.empty((n, 4), np.float32); bf = np.empty((n, 3), np.float32)
    _call("orc_aero", n, _f32(cf_ct), _f32(d), _f32(kt), _f32(vb), _f32(om), rf, rt, bf); return rf, rt, bf


def reward_pos(rel_pos_b, pos, q, qt, prog, max_len):
    n = len(pos); r = np.empty(n, np.float32); d = np.empty(n, np.int64)
    _call("orc_reward_pos", n, _f32(rel_pos_b), _f32(pos), _f32(q), _f32(qt), np.ascontiguousarray(prog, np.int64), C.c_float(max_len), r, d); return r, d


def reward_rotate(rel_pos, rel_v, pos, q, cmd, prog, max_len):
    n = len(pos); r = np.empty(n, np.float32); d = np.empty(n, np.int64)
    _call("orc_reward_rotate", n, _f32(rel_pos), _f32(rel_v), _f32(pos), _f32(q), _f32(cmd), np.ascontiguousarray(prog, np.int64), C.c_float(max_len), r, d); return r, d


def reward_flip(rel_pos_b, relq, pos, cmd, prog, max_len):
    n = len(pos); r = np.empty(n, np.float32); d = np.empty(n, np.int64)
    _call("orc_reward_flip", n, _f32(rel_pos_b), _f32(relq), _f32(pos), _f32(cmd), np.ascontiguousarray(prog, np.int64), C.c_float(max_len), r, d); return r, d


def obs_frame(task, p, q, v, w, pt, qt, V, act, cmd, flip_radian, roll_cont):
    n = len(p); fr = np.empty((n, 26), np.float32); fc = np.zeros(n, np.float32)
    _call("orc_obs_frame", n, C.c_int(task), _f32(p), _f32(q), _f32(v), _f32(w), _f32(pt), _f32(qt), _f32(V), _f32(act), _f32(cmd),
          _f32(flip_radian), _f32(roll_cont), fr, fc)
    return fr, fc


def rand_float(lower, upper, u):
    u = _f32(u); out = np.empty_like(u); _call("orc_rand_float", len(u), C.c_double(lower), C.c_double(upper), u, out); return out


def integrate(cfg_dict, root13, wrench6):
    c = make_cfg(cfg_dict); r = _f32(root13).copy()
    fn = lib().orc_integrate; fn.restype = None
    fn(C.byref(c), C.c_int(len(r)), _p(r), _p(_f32(wrench6))); return r


def philox(seed, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(seed, c0, c1, c2, c3, out)
    return list(out)


def uniform_block(seed, n_env, n_step, stream, n_idx, gid0=0, step0=0):
    """[n_env, n_step, n_idx] uniforms of the build's generator: env gid0 + e, step step0 + t, stream, index k (tests/test_rng_distributions.py)"""
    out = np.empty((n_env, n_step, n_idx), np.float32)
    fn = lib().orc_uniform_block
    fn.restype = None
    fn.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    fn(seed, gid0, n_env, step0, n_step, stream, n_idx, _p(out))
    return out


def rounded_normal(u, lim):
    """round(N(0,1)) clamped to +-lim by inverse CDF on the uniforms u (the deploy / delay length draws, FA:324, :576)"""
    u = _f32(u).ravel()
    out = np.empty(u.size, np.int32)
    fn = lib().orc_rounded_normal_vec
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    fn(u.size, _p(u), lim, _p(out))
    return out


def obs_normals_block(seed, n_env, n_step, gid0=0, step0=0):
    """[n_env, n_step, 12] observation-noise normals (Box-Muller on STREAM_OBS uniforms 4 .. 15, FA:402-410)"""
    out = np.empty((n_env, n_step, 12), np.float32)
    fn = lib().orc_obs_normals_block
    fn.restype = None
    fn.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    fn(seed, gid0, n_env, step0, n_step, _p(out))
    return out


def vec(fn_name, *xs):
    """apply a scalar own-math function elementwise"""
    f = getattr(lib(), fn_name)
    return np.array([f(*[float(v) for v in t]) for t in zip(*xs)], np.float32)


def gae(rew, done, value, last_value, gamma, lam):
    rew, done, value, last_value = _f32(rew), _f32(done), _f32(value), _f32(last_value)
    H, N = rew.shape
    adv = np.empty_like(rew); ret = np.empty_like(rew)
    fn = lib().orc_gae; fn.restype = None
    fn(C.c_int(H), C.c_int(N), C.c_double(gamma), C.c_double(lam), _p(rew), _p(done), _p(value), _p(last_value), _p(adv), _p(ret))
    return adv, ret


def normalize_advantage(adv):
    a = _f32(adv).copy()
    fn = lib().orc_normalize_advantage; fn.restype = None
    fn(C.c_size_t(a.size), _p(a))
    return a


# ---- policy forward (row N1, second half)
class OrcPolicyCfg(C.Structure):
    _fields_ = [("obs_len", C.c_int32), ("obs_dim", C.c_int32), ("states_len", C.c_int32), ("states_dim", C.c_int32), ("act_dim", C.c_int32),
                ("n_actor_hidden", C.c_int32), ("actor_hidden", C.c_int32 * 4), ("lstm_hidden", C.c_int32),
                ("n_critic_hidden", C.c_int32), ("critic_hidden", C.c_int32 * 4)]


def policy_cfg(obs_len, states_len, actor_hidden, lstm_hidden, critic_hidden, obs_dim=26, states_dim=26, act_dim=4):
    c = OrcPolicyCfg(obs_len, obs_dim, states_len, states_dim, act_dim)
    c.n_actor_hidden = len(actor_hidden)
    for i, h in enumerate(actor_hidden):
        c.actor_hidden[i] = h
    c.lstm_hidden = lstm_hidden
    c.n_critic_hidden = len(critic_hidden)
    for i, h in enumerate(critic_hidden):
        c.critic_hidden[i] = h
    return c


def policy_blob_floats(cfg):
    fn = lib().orc_policy_blob_floats; fn.restype = C.c_size_t; fn.argtypes = [C.POINTER(OrcPolicyCfg)]
    return fn(C.byref(cfg))


def policy_act(cfg, blob, obs, states, eps=None, deterministic=False):
    obs, states, blob = _f32(obs), _f32(states), _f32(blob)
    n = obs.shape[0]
    assert blob.size == policy_blob_floats(cfg)
    a = cfg.act_dim
    action, mu, sigma = np.empty((n, a), np.float32), np.empty((n, a), np.float32), np.empty((n, a), np.float32)
    logp, value = np.empty(n, np.float32), np.empty(n, np.float32)
    e = _f32(eps) if eps is not None else None
    fn = lib().orc_policy_act; fn.restype = C.c_int
    rc = fn(C.byref(cfg), _p(blob), C.c_int(n), _p(obs), _p(states), _p(e) if e is not None else None, C.c_int(1 if deterministic else 0),
            _p(action), _p(logp), _p(value), _p(mu), _p(sigma))
    if rc != 0:
        raise ValueError("orc_policy_act rejected the configuration")
    return action, logp, value, mu, sigma


def expf(x):
    fn = lib().orc_expf; fn.restype = C.c_float; fn.argtypes = [C.c_float]
    return np.array([fn(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def tanhf(x):
    fn = lib().orc_tanhf; fn.restype = C.c_float; fn.argtypes = [C.c_float]
    return np.array([fn(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def sigmoidf(x):
    fn = lib().orc_sigmoidf; fn.restype = C.c_float; fn.argtypes = [C.c_float]
    return np.array([fn(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def policy_noise(seed, call, n, act_dim=4):
    eps = np.empty((n, act_dim), np.float32)
    fn = lib().orc_policy_noise; fn.restype = None
    fn(C.c_uint64(seed), C.c_uint32(call), C.c_int(n), C.c_int(act_dim), _p(eps))
    return eps


def rollout(env, cfg, blob, horizon, seed, call0, gamma, obs0, states0, act_lo=-1.0, act_hi=1.0):
    """orc_rollout: one PPO rollout (ppo_asymmetry.py:308-342) of OracleEnv `env` under the policy (cfg, blob), starting from the stacks
    obs0 [n, obs_len, 26] / states0 [n, states_len, 26] and env.reset_buf.  -> dict of the replay-buffer arrays ([H(+1), n, ...], materialised
    stacks) + last_value; env (state, reset_buf) is left where the rollout ended."""
    n, H = env.n, int(horizon)
    blob = _f32(blob)
    assert blob.size == policy_blob_floats(cfg)
    out = {"obs": np.zeros((H + 1, n, cfg.obs_len, 26), np.float32), "states": np.zeros((H + 1, n, cfg.states_len, 26), np.float32),
           "act": np.zeros((H, n, 4), np.float32), "rew": np.zeros((H, n), np.float32), "done": np.zeros((H, n), np.float32),
           "value": np.zeros((H, n), np.float32), "logp": np.zeros((H, n), np.float32), "mu": np.zeros((H, n, 4), np.float32),
           "sigma": np.zeros((H, n, 4), np.float32), "timeout": np.zeros((H, n), np.uint8), "last_value": np.zeros(n, np.float32)}
    out["obs"][0], out["states"][0] = obs0, states0
    fn = lib().orc_rollout
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(OrcPolicyCfg), C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 12
    rc = fn(env._h, C.byref(cfg), _p(blob), H, seed, call0, float(gamma), float(act_lo), float(act_hi), _p(env.reset_buf), _p(out["obs"]), _p(out["states"]),
            _p(out["act"]), _p(out["rew"]), _p(out["done"]), _p(out["value"]), _p(out["logp"]), _p(out["mu"]), _p(out["sigma"]), _p(out["timeout"]),
            _p(out["last_value"]))
    if rc != 0:
        raise ValueError("orc_rollout rejected its arguments")
    env.obs_buf[:], env.states_buf[:] = out["obs"][H], out["states"][H]
    return out
