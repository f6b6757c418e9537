"""Default configuration for the FPV tasks, in the nested layout the reference's task constructor reads
(`cfg["env"]`, `cfg["sim"]`, top-level task flags: fpv_asymmetry.py:57-115, vec_task_asymmetry.py:82-100), plus the
flattening to the C ABI's `taco_cfg`.

The reference's YAML files are not in its repository (IsaacGymEnvs/.gitignore:37 ignores *.yaml), so the values below
are AUTHORED here; where the reference's code pins a value it is cited, otherwise the value is a documented assumption:
  controlFrequencyInv = 10   (one action spans ten 1-ms slots: fpv_asymmetry.py:326, :378-379)
  dt = 0.001                 (rotor model hard-codes sample_time = 0.001: control/thrust_dynamics.py:34)
  substeps = 2               (default of vec_task_asymmetry.py:432)
  maxEpisodeLength = 1000    (> 500 because commands are re-drawn at progress 500, fpv_asymmetry.py:152; the test mode
                              forces 1000, train_fpv_asymmetry_ppo.py:342)
  delay_time_max = 100       (torch.arange(100) is hard-coded at fpv_asymmetry.py:329)
  rotor_response_time = 0.017, delay_time = 20, lenObservations = 1, lenStates = 5 (README.md:60-66)
"""
import copy
import math

# assets/xml/fpv_without_duct.xml:4-39 (chassis line 6; rotor arms at (+-0.047, +-0.059, 0), rotors 0.02 above them,
# every arm/rotor body: mass 1e-7, inertia 1e-7 * I).  fpv_with_duct.xml differs in line 6 only.
_CHASSIS = {"without_duct": (0.46, (0.0005, 0.0007, 0.0008)), "with_duct": (0.570, (0.00109, 0.00147, 0.001226))}
ARM_X, ARM_Y, ROTOR_Z = 0.047, 0.059, 0.02
_SMALL_M, _SMALL_J = 1e-7, 1e-7


def composite_body(asset="without_duct"):
    """Mass and principal inertia (about the chassis origin) of the 9 bodies welded by fixed joints: the actor behaves
    as ONE rigid body (no <joint> in the MJCF, collapse_fixed_joints unset: fpv_asymmetry.py:250-257).  Products of
    inertia cancel by symmetry; the centre-of-mass offset (~1.7e-8 m) is neglected."""
    m, J = _CHASSIS[asset]
    J = list(J)
    for sx in (1, -1):
        for sy in (1, -1):
            for z in (0.0, ROTOR_Z):
                x, y = sx * ARM_X, sy * ARM_Y
                m += _SMALL_M
                J[0] += _SMALL_J + _SMALL_M * (y * y + z * z)
                J[1] += _SMALL_J + _SMALL_M * (x * x + z * z)
                J[2] += _SMALL_J + _SMALL_M * (x * x + y * y)
    return m, tuple(J)


def airframe(asset="without_duct"):
    """cfg overrides selecting one of the reference's two MJCF airframes (assets/xml/fpv_without_duct.xml, fpv_with_duct.xml:6: 0.570 kg,
    J = (1.09e-3, 1.47e-3, 1.226e-3)): `default_cfg(..., **airframe("with_duct"))`.  The composite mass / inertia (chassis + 8 welded arm /
    rotor bodies) follow from `composite_body`."""
    if asset not in _CHASSIS:
        raise ValueError(f"unknown airframe {asset!r}: one of {sorted(_CHASSIS)}")
    m, J = composite_body(asset)
    return {"asset": asset, "mass": m, "inertia": J}


_TASK_FLAGS_TRAIN = dict(
    random_copter_pos=True, random_copter_quat=True, random_copter_vel=True, random_target_pos=True, random_target_yaw=True,
    battery_consumption=True, random_voltage=True,
    rotor_response_time=0.017, rotor_noise=False, rotor_delay=True, rotor_response=True, random_rotordynamic_coe=False,
    random_rotor_delay=False, random_rotor_response=False, random_rotor_speed=True, random_aerodynamic_coe=False,
    delay_time_max=100, delay_time=20, ramdom_delay_time=False, ramdom_deploy_time=False,
    random_command=True, difficulty=1.0, observation_noise=False, record_flag=False, record_path="",
)


def default_cfg(task_mode="pos", num_envs=4096, **overrides):
    """cfg['Task'] as the reference's FpvBase.__init__ reads it.  Keyword overrides replace top-level task flags, or
    `env_*` / `sim_*` prefixed keys for the nested dicts (env_lenStates=5, sim_substeps=4, ...)."""
    cfg = {
        "name": {"pos": "Fpv_pos", "rotate": "Fpv_rotate", "flip": "Fpv_flip", "mix": "Fpv_mix"}[task_mode],
        "task_mode": task_mode,
        "physics_engine": "physx",
        "seed": 0,
        "env": {"numEnvs": num_envs, "maxEpisodeLength": 1000, "lenObservations": 1, "lenStates": 1, "enableDebugVis": False,
                "envSpacing": 5.0, "controlFrequencyInv": 10, "clipObservations": math.inf, "clipStates": math.inf,
                "clipActions": math.inf},
        "sim": {"dt": 0.001, "substeps": 2, "up_axis": "z", "gravity": [0.0, 0.0, -9.81], "use_gpu_pipeline": True},
        "task": {"randomization_params": {}},
        "asset": "without_duct",
    }
    cfg.update(copy.deepcopy(_TASK_FLAGS_TRAIN))
    for k, v in overrides.items():
        if k.startswith("env_"):
            cfg["env"][k[4:]] = v
        elif k.startswith("sim_"):
            cfg["sim"][k[4:]] = v
        else:
            cfg[k] = v
    return cfg


def flat_cfg(cfg, env_offset=0, num_envs_local=None):
    """nested cfg -> the flat dict `_lib.make_cfg` (and the test oracle's make_cfg) understand."""
    env, sim = cfg["env"], cfg["sim"]
    mass, inertia = composite_body(cfg.get("asset", "without_duct"))
    n_global = int(env["numEnvs"])
    d = {
        "num_envs": int(num_envs_local if num_envs_local is not None else n_global),
        "env_offset": int(env_offset),
        "num_envs_global": n_global,
        "task_mode": cfg["task_mode"],
        "len_obs": int(env.get("lenObservations", 1)),
        "len_states": int(env.get("lenStates", env.get("lenObservations", 1))),
        "control_freq_inv": int(env.get("controlFrequencyInv", 1)),
        "substeps": int(sim.get("substeps", 2)),
        "max_episode_length": int(env["maxEpisodeLength"]),
        "delay_time": int(cfg["delay_time"]),
        "seed": int(cfg.get("seed", 0)),
        "dt": float(sim["dt"]),
        "rotor_response_time": float(cfg["rotor_response_time"]),
        "difficulty": float(cfg["difficulty"]),
        "clip_actions": float(env.get("clipActions", math.inf)),
        "clip_obs": float(env.get("clipObservations", math.inf)),
        "clip_states": float(env.get("clipStates", math.inf)),
        "mass": float(cfg.get("mass", mass)),
        "inertia": tuple(cfg.get("inertia", inertia)),
        "arm_x": ARM_X, "arm_y": ARM_Y,
        "gravity_z": float(sim.get("gravity", [0, 0, -9.81])[2]),
    }
    for k in ("random_copter_pos", "random_copter_quat", "random_copter_vel", "random_target_pos", "random_target_yaw",
              "battery_consumption", "random_voltage", "rotor_noise", "rotor_delay", "rotor_response", "random_rotordynamic_coe",
              "random_rotor_delay", "random_rotor_response", "random_rotor_speed", "random_aerodynamic_coe", "ramdom_delay_time",
              "ramdom_deploy_time", "random_command", "observation_noise"):
        d[k] = bool(cfg[k])
    # the angular rate's data flow: True (default) = the reference's own -- the body rates are re-derived from the world-frame root state after
    # every simulate() (fpv_asymmetry.py:350); False = carried across the ten substeps (one rounding per substep apart: outside 1e-5 per step,
    # tests/util.py teacher_forced_one_step)
    d["world_rate_roundtrip"] = bool(cfg.get("world_rate_roundtrip", True))
    d["record_flag"] = bool(cfg.get("record_flag", False))                    # all envs track copter_rpy_continuous (fpv_asymmetry.py:113, :339-347)
    if int(cfg.get("delay_time_max", 100)) != 100:
        raise ValueError("delay_time_max must be 100 (hard-coded in the reference, fpv_asymmetry.py:329)")
    return d


# BASELINE.json `configs`, as concrete cfgs
def baseline_config(index, num_envs=None):
    if index == 0:
        return default_cfg("pos", num_envs or 64, random_copter_pos=False, random_copter_quat=False, random_copter_vel=False,
                           random_target_pos=False, random_target_yaw=False, battery_consumption=False, random_voltage=False,
                           random_rotor_speed=False, random_command=False)
    if index == 1:
        return default_cfg("pos", num_envs or 4096, rotor_response_time=0.017, delay_time=20)
    if index == 2:
        return default_cfg("rotate", num_envs or 16384, delay_time=20, random_command=True)
    if index == 3:
        return default_cfg("flip", num_envs or 65536, delay_time=20)
    if index == 4:
        return default_cfg("mix", num_envs or 262144, random_rotordynamic_coe=True, random_rotor_response=True, random_aerodynamic_coe=True,
                           observation_noise=True, rotor_noise=True, ramdom_delay_time=True, ramdom_deploy_time=True, env_lenStates=5)
    raise IndexError(index)
