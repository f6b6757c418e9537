"""Env-state checkpoint / restore.

The reference never checkpoints environment state (ppo_asymmetry.py:452-456 saves the agent only).  Here the whole
simulation state is one blob (taco_get_state) + the step counter that keys the random streams + the caller-owned
buffers, so a run can be resumed bit-exactly -- also on a different number of GPUs: every per-env word is a column of the blob
and every random stream is keyed by the GLOBAL env id, so a checkpoint that covers an env's [env_offset, env_offset + num_envs)
range restores it whatever the sharding was when it was written (load_state_dict slices; merge() joins the shards of a
multi-rank run into one checkpoint).  tests/test_api_gpu.py::test_checkpoint_restores_across_shardings."""
import torch

from . import _lib

FORMAT = 1
# Revision of the step's ARITHMETIC (what decides whether a restored run continues bit for bit).  1: ABI <= 6.  2: ABI 7 on -- the stored world-frame
# rate is what the tenth simulate() left (w_rt) instead of being rebuilt from the round-tripped body rates: 1-4 ulp apart (round 5).  A checkpoint that
# carries no "arith_rev" is given the revision of the ABI that wrote it.
ARITH_REV = 2


def _arith_rev(sd):
    return sd.get("arith_rev", 1 if sd.get("abi", 0) <= 6 else 2)


# What makes two states "the same run": identity keys, every task flag, and every numeric cfg value the step's arithmetic reads (round 5's advisor: rotor
# response time, airframe, gravity, clips and the control decimation were not compared).  load_state_dict(strict=True) and merge() use the same list.
def _identity_keys():
    return ("num_envs_global", "task_mode", "len_obs", "len_states", "seed", "delay_time", "max_episode_length", "dt", "substeps", "control_freq_inv",
            "rotor_response_time", "mass", "inertia", "arm_x", "arm_y", "gravity_z", "clip_actions", "clip_obs", "clip_states") + tuple(_lib.FLAG_BITS)


def _same(a, b):
    if isinstance(a, (tuple, list)) or isinstance(b, (tuple, list)):
        return a is not None and b is not None and tuple(a) == tuple(b)
    return a == b


def state_dict(env):
    """everything needed to continue `env` exactly (all tensors on CPU)"""
    return {
        "format": FORMAT, "abi": _lib.ABI_VERSION, "arith_rev": ARITH_REV, "flat_cfg": dict(env._flat), "step_count": int(env.step_count),
        "blob": env.get_state().cpu(), "obs_buf": env.obs_buf.cpu(), "states_buf": env.states_buf.cpu(), "rew_buf": env.rew_buf.cpu(),
        "reset_buf": env.reset_buf.cpu(), "timeout_buf": env.timeout_buf.cpu(), "difficulty": float(env.difficulty),
    }


_PER_ENV = ("obs_buf", "states_buf", "rew_buf", "reset_buf", "timeout_buf")


def merge(state_dicts):
    """the shards of one multi-rank run (same step, adjoining env ranges) -> one checkpoint covering their union"""
    sds = sorted(state_dicts, key=lambda d: d["flat_cfg"]["env_offset"])
    first = sds[0]
    at = first["flat_cfg"]["env_offset"]
    for d in sds:
        for k in ("format", "abi", "step_count", "difficulty"):
            if d[k] != first[k]:
                raise ValueError(f"shards differ in {k}: {d[k]!r} vs {first[k]!r}")
        if _arith_rev(d) != _arith_rev(first):
            raise ValueError(f"shards differ in arith_rev: {_arith_rev(d)} vs {_arith_rev(first)}")
        for k in _identity_keys():   # (the list load_state_dict(strict=True) checks: shards taken under different flags or constants are not one run)
            if k in d["flat_cfg"] or k in first["flat_cfg"]:
                if not _same(d["flat_cfg"].get(k), first["flat_cfg"].get(k)):
                    raise ValueError(f"shards differ in {k}: {d['flat_cfg'].get(k)!r} vs {first['flat_cfg'].get(k)!r}")
        if d["flat_cfg"]["env_offset"] != at:
            raise ValueError(f"shards do not adjoin: expected a shard starting at env {at}, got {d['flat_cfg']['env_offset']}")
        at += d["flat_cfg"]["num_envs"]
    out = dict(first)
    out["flat_cfg"] = dict(first["flat_cfg"], num_envs=at - first["flat_cfg"]["env_offset"])
    out["blob"] = torch.cat([d["blob"] for d in sds], dim=1)
    for name in _PER_ENV:
        out[name] = torch.cat([d[name] for d in sds], dim=0)
    return out


def load_state_dict(env, sd, strict=True):
    """Restore `env` from a checkpoint that covers its env range (the same range, or a larger one: the env's columns are sliced out).
    strict=False skips the check that the run's identity (task, stack lengths, seed, global env count) and its arithmetic (every task flag,
    world_rate_roundtrip included; delay / episode length; dt; sub-iterations; control decimation; rotor response time; airframe; gravity; clips;
    the library's arithmetic revision, ARITH_REV) are the same, and that a recording env does not take over stale rpy rows."""
    if sd.get("format") != FORMAT:
        raise ValueError(f"unknown checkpoint format {sd.get('format')}")
    if strict:
        # the run's identity AND its arithmetic: every task flag (_lib.FLAG_BITS -- the randomisation switches and world_rate_roundtrip, the
        # angular rate's data flow, whose default changed with ABI 6) decides what the next step computes; a key an older checkpoint does not
        # carry is compared with the default it was written under (world_rate_roundtrip: False before ABI 6)
        old_defaults = {"world_rate_roundtrip": sd.get("abi", 0) >= 6, "record_flag": False}
        for k in _identity_keys():
            if k not in env._flat:
                continue
            if k == "record_flag":
                # which envs keep copter_rpy_old / _continuous current.  A checkpoint taken WITHOUT it holds stale rows for the pos / rotate envs: an env
                # that records them (episode recorder, FA:662) must not take those over; the other direction loses nothing
                if env._flat[k] and not sd["flat_cfg"].get(k, False):
                    raise ValueError("checkpoint was taken with record_flag=False (rpy_continuous of the pos / rotate envs is stale in it), this env has "
                                     "record_flag=True (strict=False loads it anyway)")
                continue
            if k not in sd["flat_cfg"] and k not in old_defaults:
                continue   # (a numeric key an older checkpoint did not store: nothing to compare it with)
            have = sd["flat_cfg"].get(k, old_defaults.get(k))
            if not _same(have, env._flat[k]):
                raise ValueError(f"checkpoint was taken with {k}={have!r}, this env has {env._flat[k]!r} (strict=False loads it anyway: the run then "
                                 f"continues under THIS env's settings and is no longer the one that was saved)")
        if _arith_rev(sd) != ARITH_REV:
            raise ValueError(f"checkpoint was written under arithmetic revision {_arith_rev(sd)} (ABI {sd.get('abi')}), this library computes revision {ARITH_REV}: "
                             f"since ABI 7 the stored world-frame rate is the tenth simulate()'s own (1-4 ulp from what ABI <= 6 stored), so the run would not "
                             f"continue bit for bit (strict=False loads it anyway)")
        if sd.get("abi") != _lib.ABI_VERSION:
            import warnings
            warnings.warn(f"checkpoint written by ABI {sd.get('abi')}, loading into ABI {_lib.ABI_VERSION}: same state layout (format {FORMAT}) and the same "
                          f"arithmetic revision ({ARITH_REV}): the run continues bit for bit")
    lo = env._flat["env_offset"] - sd["flat_cfg"]["env_offset"]
    n = env._flat["num_envs"]
    if lo < 0 or lo + n > sd["flat_cfg"]["num_envs"]:
        raise ValueError(f"checkpoint covers envs [{sd['flat_cfg']['env_offset']}, {sd['flat_cfg']['env_offset'] + sd['flat_cfg']['num_envs']}), this env needs "
                         f"[{env._flat['env_offset']}, {env._flat['env_offset'] + n}): merge() the shards first")
    env.difficulty = sd["difficulty"]
    env.step_count = sd["step_count"]
    env.set_state(sd["blob"][:, lo:lo + n].contiguous())
    env.load_stacks(sd["obs_buf"][lo:lo + n], sd["states_buf"][lo:lo + n])   # (the state stack may live in a frame ring: vec_env.load_stacks)
    for name in ("rew_buf", "reset_buf", "timeout_buf"):
        getattr(env, name).copy_(sd[name][lo:lo + n].to(env.device))


def save(env, path):
    torch.save(state_dict(env), path)


def load(env, path, strict=True):
    load_state_dict(env, torch.load(path, map_location="cpu", weights_only=False), strict=strict)
