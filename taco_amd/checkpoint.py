"""Env-state checkpoint / restore.

The reference never checkpoints environment state (ppo_asymmetry.py:452-456 saves the agent only).  Here the whole
simulation state is one blob (taco_get_state) + the step counter that keys the random streams + the caller-owned
buffers, so a run can be resumed bit-exactly -- also on a different number of GPUs, because everything is keyed by
global env id."""
import torch

from . import _lib

FORMAT = 1


def state_dict(env):
    """everything needed to continue `env` exactly (all tensors on CPU)"""
    return {
        "format": FORMAT, "abi": _lib.ABI_VERSION, "flat_cfg": dict(env._flat), "step_count": int(env.step_count),
        "blob": env.get_state().cpu(), "obs_buf": env.obs_buf.cpu(), "states_buf": env.states_buf.cpu(), "rew_buf": env.rew_buf.cpu(),
        "reset_buf": env.reset_buf.cpu(), "timeout_buf": env.timeout_buf.cpu(), "difficulty": float(env.difficulty),
    }


def load_state_dict(env, sd, strict=True):
    if sd.get("format") != FORMAT:
        raise ValueError(f"unknown checkpoint format {sd.get('format')}")
    if strict:
        for k in ("num_envs", "env_offset", "num_envs_global", "task_mode", "len_obs", "len_states", "seed"):
            if sd["flat_cfg"][k] != env._flat[k]:
                raise ValueError(f"checkpoint was taken with {k}={sd['flat_cfg'][k]!r}, this env has {env._flat[k]!r}")
    env.difficulty = sd["difficulty"]
    env.step_count = sd["step_count"]
    env.set_state(sd["blob"])
    for name in ("obs_buf", "states_buf", "rew_buf", "reset_buf", "timeout_buf"):
        getattr(env, name).copy_(sd[name].to(env.device))


def save(env, path):
    torch.save(state_dict(env), path)


def load(env, path, strict=True):
    load_state_dict(env, torch.load(path, map_location="cpu", weights_only=False), strict=strict)
