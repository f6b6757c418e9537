"""ctypes binding of libtaco_env.so (include/taco_env.h).  There is NO fallback: if the HIP library is missing or
fails to load, importing the environment raises -- the product path is the HIP kernel or nothing."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TACO_ENV_LIB", os.path.join(HERE, "libtaco_env.so"))  # override = A/B builds of the same ABI (taken as it is, never rebuilt)
SKIP_ABI_CHECK = os.environ.get("TACO_ENV_LIB_SKIP_ABI", "") == "1"   # explicit opt-out for A/B runs ACROSS ABI revisions (missing symbols are stubbed)

ABI_VERSION = 8
NUM_FIELDS = 67
RING_SLOTS = 100
BLOB_ROWS = NUM_FIELDS + 4 * RING_SLOTS

TASKS = {"pos": 0, "rotate": 1, "flip": 2, "mix": 3}
FORMS = {"auto": 0, "quad_roles": 1, "lane_roles": 2, "quad": 3, "lane": 4, "lane_throughput": 5, "quad_served": 6}  # TACO_FORM_* (include/taco_env.h)
FLAG_BITS = {
    "random_copter_pos": 0, "random_copter_quat": 1, "random_copter_vel": 2, "random_target_pos": 3,
    "random_target_yaw": 4, "battery_consumption": 5, "random_voltage": 6, "rotor_noise": 7, "rotor_delay": 8,
    "rotor_response": 9, "random_rotordynamic_coe": 10, "random_rotor_delay": 11, "random_rotor_response": 12,
    "random_rotor_speed": 13, "random_aerodynamic_coe": 14, "ramdom_delay_time": 15, "ramdom_deploy_time": 16,
    "random_command": 17, "observation_noise": 18, "world_rate_roundtrip": 19, "record_flag": 20,
}
# every symbol include/taco_env.h declares
EXPORTS = ["taco_abi_version", "taco_source_hash", "taco_last_error", "taco_workspace_bytes", "taco_create", "taco_destroy", "taco_step",
           "taco_gather_row_floats", "taco_bind_gather_block", "taco_set_difficulty", "taco_get_step_count", "taco_set_step_count", "taco_get_state", "taco_set_state",
           "taco_step_kernel_name", "taco_launch_geometry", "taco_set_kernel_form", "taco_get_kernel_form", "taco_check", "taco_get_field", "taco_step_rollout", "taco_reset_done", "taco_gae_workspace_bytes", "taco_gae", "taco_occupancy", "taco_bind_phase_stamps", "taco_policy_blob_floats", "taco_policy_act", "taco_rollout_run", "taco_policy_act_stamped", "taco_critic_values", "taco_critic_workspace_bytes", "taco_peek_step_count", "taco_critic_values_ring", "taco_set_rollout_fusion", "taco_bind_rollout_stamps", "taco_bind_states_ring",
           "taco_states_ring_row", "taco_step_ring", "taco_graph_mode", "taco_release_graphs", "taco_critic_clamped_words"]


class TacoCfg(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("env_offset", C.c_int32), ("num_envs_global", C.c_int32), ("task_mode", C.c_int32),
        ("len_obs", C.c_int32), ("len_states", C.c_int32), ("control_freq_inv", C.c_int32), ("substeps", C.c_int32),
        ("max_episode_length", C.c_int32), ("delay_time", C.c_int32), ("flags", C.c_uint32), ("_pad", C.c_uint32),
        ("seed", C.c_uint64), ("dt", C.c_double), ("rotor_response_time", C.c_double), ("difficulty", C.c_double),
        ("clip_actions", C.c_double), ("clip_obs", C.c_double), ("clip_states", C.c_double), ("mass", C.c_double),
        ("inertia", C.c_double * 3), ("arm_x", C.c_double), ("arm_y", C.c_double), ("gravity_z", C.c_double),
    ]


class RolloutIO(C.Structure):
    """struct taco_rollout_io (include/taco_env.h)"""
    _fields_ = [("actions", C.c_void_p), ("obs_prev", C.c_void_p), ("obs_next", C.c_void_p), ("states_prev", C.c_void_p),
                ("states_next", C.c_void_p), ("rew", C.c_void_p), ("reset_buf", C.c_void_p), ("timeout_buf", C.c_void_p),
                ("done_f32", C.c_void_p), ("obs_out", C.c_void_p), ("states_out", C.c_void_p), ("states_newest_only", C.c_int32)]


class RolloutBufs(C.Structure):
    """struct taco_rollout_bufs (include/taco_env.h)"""
    _fields_ = [(k, C.c_void_p) for k in ("obs_store", "states_store", "act_buf", "act_env", "rew_buf", "done_buf", "value_buf", "logp_buf",
                                          "mu_buf", "sigma_buf", "timeout_buf", "last_value", "critic_ws")]


class TacoError(RuntimeError):
    pass


_lib = None
_hooks_lib = None


class _Missing:
    """stand-in for an entry point an OLDER A/B build (TACO_ENV_LIB override) does not export"""
    argtypes = restype = None


def _declare(lib, ab_build=False):
    """prototypes of every entry point of include/taco_env.h.  ab_build (TACO_ENV_LIB override AND TACO_ENV_LIB_SKIP_ABI=1): the library may be of
    another ABI revision -- missing symbols are stubbed and the version check is skipped; an override alone is held to this binding's ABI."""
    if ab_build:  # A/B comparisons across ABI revisions: tolerate missing newer symbols, skip the version check
        for name in EXPORTS:
            if not hasattr(lib, name):
                setattr(lib, name, _Missing())
    lib.taco_abi_version.restype = C.c_int
    lib.taco_source_hash.restype = C.c_char_p
    lib.taco_last_error.restype = C.c_char_p
    lib.taco_step_kernel_name.restype = C.c_char_p
    lib.taco_workspace_bytes.argtypes = [C.POINTER(TacoCfg)]
    lib.taco_workspace_bytes.restype = C.c_size_t
    lib.taco_create.argtypes = [C.POINTER(TacoCfg), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.taco_create.restype = C.c_int
    lib.taco_destroy.argtypes = [C.c_void_p]
    lib.taco_destroy.restype = None
    lib.taco_step.argtypes = [C.c_void_p] * 8
    lib.taco_step.restype = C.c_int
    lib.taco_gather_row_floats.argtypes = [C.c_int]
    lib.taco_gather_row_floats.restype = C.c_int
    lib.taco_bind_gather_block.argtypes = [C.c_void_p, C.c_void_p]
    lib.taco_bind_gather_block.restype = C.c_int
    lib.taco_set_difficulty.argtypes = [C.c_void_p, C.c_double]
    lib.taco_set_difficulty.restype = C.c_int
    lib.taco_get_step_count.argtypes = [C.c_void_p]
    lib.taco_get_step_count.restype = C.c_int64
    lib.taco_peek_step_count.argtypes = [C.c_void_p]
    lib.taco_peek_step_count.restype = C.c_int64
    lib.taco_set_step_count.argtypes = [C.c_void_p, C.c_int64]
    lib.taco_set_step_count.restype = C.c_int
    lib.taco_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_get_state.restype = C.c_int
    lib.taco_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_set_state.restype = C.c_int
    lib.taco_launch_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.taco_launch_geometry.restype = C.c_int
    lib.taco_set_kernel_form.argtypes = [C.c_void_p, C.c_int]
    lib.taco_set_kernel_form.restype = C.c_int
    lib.taco_get_kernel_form.argtypes = [C.c_void_p]
    lib.taco_get_kernel_form.restype = C.c_int
    lib.taco_check.argtypes = [C.c_void_p, C.c_void_p]
    lib.taco_check.restype = C.c_int
    lib.taco_get_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.taco_get_field.restype = C.c_int
    lib.taco_policy_blob_floats.argtypes = [C.c_void_p]
    lib.taco_policy_blob_floats.restype = C.c_size_t
    lib.taco_policy_act.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_policy_act.restype = C.c_int
    lib.taco_policy_act_stamped.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_policy_act_stamped.restype = C.c_int
    lib.taco_critic_values.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_critic_values.restype = C.c_int
    lib.taco_critic_values_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_critic_values_ring.restype = C.c_int
    lib.taco_critic_workspace_bytes.argtypes = [C.c_void_p, C.c_int64]
    lib.taco_critic_workspace_bytes.restype = C.c_size_t
    lib.taco_critic_clamped_words.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
    lib.taco_critic_clamped_words.restype = C.c_int
    lib.taco_rollout_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RolloutBufs), C.c_int, C.c_uint64, C.c_uint32, C.c_double,
                                     C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lib.taco_rollout_run.restype = C.c_int
    lib.taco_set_rollout_fusion.argtypes = [C.c_void_p, C.c_int]
    lib.taco_set_rollout_fusion.restype = C.c_int
    lib.taco_bind_rollout_stamps.argtypes = [C.c_void_p, C.c_void_p]
    lib.taco_bind_rollout_stamps.restype = C.c_int
    lib.taco_bind_states_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.taco_bind_states_ring.restype = C.c_int
    lib.taco_graph_mode.argtypes = [C.c_void_p]
    lib.taco_graph_mode.restype = C.c_int
    lib.taco_release_graphs.argtypes = [C.c_void_p]
    lib.taco_release_graphs.restype = C.c_int
    lib.taco_states_ring_row.argtypes = [C.c_void_p]
    lib.taco_states_ring_row.restype = C.c_int
    lib.taco_step_ring.argtypes = [C.c_void_p, C.POINTER(RolloutIO), C.c_void_p, C.POINTER(C.c_int32)]
    lib.taco_step_ring.restype = C.c_int
    lib.taco_bind_phase_stamps.argtypes = [C.c_void_p, C.c_void_p]
    lib.taco_bind_phase_stamps.restype = C.c_int
    lib.taco_occupancy.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.taco_occupancy.restype = C.c_int
    lib.taco_reset_done.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.taco_reset_done.restype = C.c_int
    lib.taco_step_rollout.argtypes = [C.c_void_p, C.POINTER(RolloutIO), C.c_void_p]
    lib.taco_step_rollout.restype = C.c_int
    lib.taco_gae_workspace_bytes.argtypes = []
    lib.taco_gae_workspace_bytes.restype = C.c_size_t
    lib.taco_gae.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.taco_gae.restype = C.c_int
    if not ab_build and lib.taco_abi_version() != ABI_VERSION:
        raise TacoError(f"libtaco_env.so ABI {lib.taco_abi_version()} != binding ABI {ABI_VERSION}")
    return lib


def _open(path, test_hooks=False):
    """(re)build `path` if it is missing or was built from other sources than the ones on disk, then dlopen it.  An explicit
    TACO_ENV_LIB override (A/B builds of the same ABI) is taken as it is."""
    overridden = not test_hooks and "TACO_ENV_LIB" in os.environ
    if not overridden:
        from . import build as _build
        have_sources = os.path.isdir(_build.CSRC) and all(os.path.exists(os.path.join(_build.CSRC, d)) for d in _build.DEPS)
        if not have_sources:
            # a deployment without csrc/ next to the package: the binary is taken as it is, if there is one (nothing to compare it with)
            if not os.path.exists(path):
                raise TacoError(f"{path} is missing and its sources are not next to the package; build it from a source checkout with "
                                "`python -m taco_amd.build` (hipcc, gfx950).  There is no CPU fallback for the step path.")
        else:
            try:   # sources present: an edited csrc/ never runs through a stale binary -- a failing build (hipcc absent included) is an error
                if _build.needs_build(path):
                    _build.build(test_hooks=test_hooks)
            except Exception as e:  # noqa: BLE001 -- report the build failure, never fall back to anything else
                raise TacoError(f"{path} is missing or stale (built from other sources than csrc/) and building it failed ({e}); run "
                                "`python -m taco_amd.build` (hipcc, gfx950).  There is no CPU fallback for the step path.") from e
    if not os.path.exists(path):
        raise TacoError(f"{path} is missing: build it with `python -m taco_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback for the step path.")
    # PyTorch-ROCm ships its own HIP runtime (libamdhip64, same SONAME as /opt/rocm's).  The process must hold ONE runtime, and it has to be
    # the one torch's other bundled libraries were built against: load torch first, so that libtaco_env.so binds to the runtime already
    # in the process.  (The other order leaves torch on /opt/rocm's runtime and HIP reports "no ROCm-capable device".)
    import torch  # noqa: F401
    return _declare(C.CDLL(path), ab_build=overridden and SKIP_ABI_CHECK)


def load():
    """dlopen libtaco_env.so and declare its prototypes; raises if the HIP extension is not built and cannot be built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def load_test_hooks():
    """the -DTACO_TEST_HOOKS build of the same sources (taco_amd/build.py --test-hooks): adds taco_test_slow_battery_server.  Tests only."""
    global _hooks_lib
    if _hooks_lib is None:
        from . import build as _build
        lib = _open(_build.LIB_HOOKS, test_hooks=True)
        lib.taco_test_slow_battery_server.argtypes = [C.c_void_p, C.c_int]
        lib.taco_test_slow_battery_server.restype = C.c_int
        _hooks_lib = lib
    return _hooks_lib


def check(rc, lib=None):
    if rc != 0:
        raise TacoError(f"libtaco_env: status {rc}: {(lib or load()).taco_last_error().decode()}")


def make_cfg(d):
    """flat dict (taco_amd.config.flat_cfg) -> TacoCfg"""
    c = TacoCfg()
    flags = 0
    for k, bit in FLAG_BITS.items():
        if d.get(k, False):
            flags |= 1 << bit
    c.num_envs = int(d["num_envs"])
    c.env_offset = int(d.get("env_offset", 0))
    c.num_envs_global = int(d.get("num_envs_global", d["num_envs"]))
    c.task_mode = TASKS[d["task_mode"]] if isinstance(d["task_mode"], str) else int(d["task_mode"])
    c.len_obs = int(d.get("len_obs", 1))
    c.len_states = int(d.get("len_states", 1))
    c.control_freq_inv = int(d.get("control_freq_inv", 10))
    c.substeps = int(d.get("substeps", 2))
    c.max_episode_length = int(d.get("max_episode_length", 1000))
    c.delay_time = int(d.get("delay_time", 20))
    c.flags = flags
    c.seed = int(d.get("seed", 0))
    c.dt = float(d.get("dt", 0.001))
    c.rotor_response_time = float(d.get("rotor_response_time", 0.017))
    c.difficulty = float(d.get("difficulty", 1.0))
    c.clip_actions = float(d.get("clip_actions", float("inf")))
    c.clip_obs = float(d.get("clip_obs", float("inf")))
    c.clip_states = float(d.get("clip_states", float("inf")))
    c.mass = float(d["mass"])
    for i in range(3):
        c.inertia[i] = float(d["inertia"][i])
    c.arm_x = float(d["arm_x"])
    c.arm_y = float(d["arm_y"])
    c.gravity_z = float(d.get("gravity_z", -9.81))
    return c


_raw_stream = None


def stream_ptr(device):
    """the HIP stream torch is on for `device`, as the pointer the C ABI takes (torch's own raw accessor where it exists: 0.1 us per call
    instead of 1.8 us for building a torch.cuda.Stream object -- a quarter of step()'s host-side cost)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

