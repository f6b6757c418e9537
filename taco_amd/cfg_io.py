"""Config front-end: YAML in the layout the reference's launcher loads
(`cfg['Task']`, train_fpv_asymmetry_ppo.py:257-271 -- the reference's own YAML files are git-ignored and absent) and the
launcher's command-line overrides for the keys the environment reads (train_fpv_asymmetry_ppo.py:88-237, 273-356):
booleans arrive as the strings 'True' / 'False', anything else leaves the YAML value; train / testmodel force a set of
task flags."""
import argparse
import copy

import yaml

from .config import default_cfg

BOOL_KEYS = ("random_copter_pos", "random_copter_quat", "random_copter_vel", "random_target_pos", "random_target_yaw",
             "battery_consumption", "random_voltage", "rotor_noise", "rotor_delay", "rotor_response", "random_rotordynamic_coe",
             "random_rotor_delay", "random_rotor_response", "random_rotor_speed", "random_aerodynamic_coe", "ramdom_delay_time",
             "ramdom_deploy_time", "random_command", "observation_noise")
VALUE_KEYS = (("rotor_response_time", float), ("delay_time_max", int), ("delay_time", int))
ENV_KEYS = (("num_envs", "numEnvs"), ("num_episodes", "maxEpisodeLength"), ("lenObservations", "lenObservations"), ("lenStates", "lenStates"))


def write_default_yaml(path, task_mode="pos", **overrides):
    """a documented default `Fpv_asymmetry_PPO_<mode>.yaml` (Task section only: the env does not read the PPO section)"""
    task = default_cfg(task_mode, **overrides)
    task["name"] = "Fpv"            # the launcher appends "_<task_mode>" (train_fpv_asymmetry_ppo.py:288)
    task["env"] = {k: (v if v != float("inf") else ".inf") for k, v in task["env"].items()}
    with open(path, "w") as f:
        yaml.safe_dump({"Task": task}, f, sort_keys=False)
    return path


def load_yaml(path):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    task = cfg["Task"]
    for k, v in list(task["env"].items()):
        if v in (".inf", "inf"):
            task["env"][k] = float("inf")
    return cfg


def add_env_arguments(parser):
    """the launcher's env-related options, same names and types (train_fpv_asymmetry_ppo.py:96-160)"""
    parser.add_argument("--train_mode", type=str, default="train")
    parser.add_argument("--task_mode", type=str)
    parser.add_argument("--load_task_mode", type=str, default="pos")
    parser.add_argument("--seed", type=int)
    for cli, _ in ENV_KEYS:
        parser.add_argument("--" + cli, type=int)
    for k in BOOL_KEYS:
        parser.add_argument("--" + k, type=str)
    for k, typ in VALUE_KEYS:
        parser.add_argument("--" + k, type=typ)
    return parser


def apply_overrides(cfg, args, diff_value=(0.0, 1.0), record_path=""):
    """train_fpv_asymmetry_ppo.py:273-356 for cfg['Task']; returns the Task dict ready for the env constructor"""
    task = copy.deepcopy(cfg["Task"])
    if getattr(args, "seed", None) is not None:
        task["seed"] = args.seed
    for cli, key in ENV_KEYS:
        v = getattr(args, cli, None)
        if v is not None:
            task["env"][key] = v
    mode = args.load_task_mode if args.train_mode == "testmodel" else args.task_mode
    if mode is not None:
        task["task_mode"] = mode
    task["name"] = task.get("name", "Fpv").split("_")[0] + "_" + task["task_mode"]
    for k in BOOL_KEYS:
        s = getattr(args, k, None)
        task[k] = True if s == "True" else False if s == "False" else task[k]
    for k, _ in VALUE_KEYS:
        v = getattr(args, k, None)
        if v is not None:
            task[k] = v
    if args.train_mode == "train":                       # :322-326
        task["difficulty"] = diff_value[0]
        task["record_flag"] = False
        task["record_path"] = ""
    elif args.train_mode == "testmodel":                 # :328-354
        task["env"]["numEnvs"] = 12
        task["difficulty"] = 1
        task["record_flag"] = True
        task["record_path"] = record_path
        task["observation_noise"] = True
        task["rotor_noise"] = False
        task["random_voltage"] = True
        task["env"]["maxEpisodeLength"] = 1000
        task.update(random_copter_pos=False, random_copter_quat=True, random_copter_vel=True, random_target_pos=False,
                    random_target_yaw=False, random_command=True, random_rotordynamic_coe=False, random_rotor_delay=False,
                    random_rotor_response=False, random_rotor_speed=False, random_aerodynamic_coe=False, battery_consumption=False)
    else:
        raise ValueError("Invalid train mode")
    return task


def parse(argv=None, yaml_path=None):
    ap = add_env_arguments(argparse.ArgumentParser(description="taco_amd env launcher"))
    args = ap.parse_args(argv)
    cfg = load_yaml(yaml_path) if yaml_path else {"Task": default_cfg(args.task_mode or "pos")}
    return apply_overrides(cfg, args), args
