"""taco_amd -- MI355X-native drop-in for the fpv_asymmetry.step() hot path of yinzikang/taco."""
from .config import default_cfg, flat_cfg, baseline_config, composite_body  # noqa: F401

__all__ = ["default_cfg", "flat_cfg", "baseline_config", "composite_body", "isaacgym_task_map", "FpvPos", "FpvRotate", "FpvFlip", "FpvMix"]


def __getattr__(name):  # the env classes import torch + dlopen the HIP library: load them lazily
    if name in ("isaacgym_task_map", "FpvBase", "FpvPos", "FpvRotate", "FpvFlip", "FpvMix"):
        from . import vec_env
        return getattr(vec_env, name)
    raise AttributeError(name)
