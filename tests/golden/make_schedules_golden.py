#!/usr/bin/env python3
"""Golden values for SURVEY 8f row N3 (the per-epoch schedules), produced by EXECUTING the reference's own code:

    python tests/golden/make_schedules_golden.py [/root/reference]

`IsaacGymEnvs/algorithms/ppo_asymmetry.py` cannot be imported here (tensorboard, the env), so the file is parsed (ast) at run time and two
pieces of it are compiled and run as they are: the `self.<name> = ...` statements of `PPO.__init__` that depend on constructor arguments only
(they resolve the `None` defaults of the schedule knots, :78-97), and the preamble of `PPO.update(epoch)` up to the first use of the replay
buffer (:142-175: learning rate, Lipschitz constant, difficulty).  They run on a stub `self` (no agent, no optimizer state) for every epoch;
only the numbers are stored (tests/golden/schedules.npz).  Build container only: the GPU box never sees /root/reference.
"""
import ast
import pathlib
import sys
import types

import numpy as np

REF = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = pathlib.Path(__file__).resolve().parent
SRC = REF / "IsaacGymEnvs/algorithms/ppo_asymmetry.py"

tree = ast.parse(SRC.read_text())
cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PPO")
init = next(f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name == "__init__")
upd = next(f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name == "update")

# constructor arguments and their literal defaults
names = [a.arg for a in init.args.args]
defaults = {}
for a, d in zip(names[len(names) - len(init.args.defaults):], init.args.defaults):
    try:
        defaults[a] = ast.literal_eval(d)
    except ValueError:
        pass   # (class / dict() defaults: not schedule parameters)


def only_uses(node, allowed):
    return all(n.id in allowed for n in ast.walk(node) if isinstance(n, ast.Name))


# `self.x = <expression over constructor arguments>` statements of __init__, in order
assigns = [st for st in init.body if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Attribute)
           and isinstance(st.targets[0].value, ast.Name) and st.targets[0].value.id == "self" and only_uses(st.value, set(defaults))]
ctor = ast.FunctionDef(name="ctor", args=ast.arguments(posonlyargs=[], args=[ast.arg("self")] + [ast.arg(k) for k in defaults], kwonlyargs=[],
                                                        kw_defaults=[], defaults=[ast.Constant(None)] * 0), body=assigns, decorator_list=[])
# update(): everything before the replay buffer is touched
pre = []
for st in upd.body:
    if any(isinstance(n, ast.Attribute) and n.attr == "replay_buffer" for n in ast.walk(st)):
        break
    pre.append(st)
ret = ast.Return(ast.Tuple([ast.Name("learning_rate", ast.Load()), ast.Name("lipschitz_para", ast.Load()), ast.Name("difficulty", ast.Load())], ast.Load()))
preamble = ast.FunctionDef(name="preamble", args=upd.args, body=pre + [ret], decorator_list=[])
mod = ast.fix_missing_locations(ast.Module(body=[ctor, preamble], type_ignores=[]))
ns = {}
exec(compile(mod, "<ppo_asymmetry.py: __init__ parameter block + update() preamble>", "exec"), ns)


def run(**kw):
    args = dict(defaults, **kw)
    me = types.SimpleNamespace(agent=types.SimpleNamespace(train=lambda: None), optimizer=types.SimpleNamespace(param_groups=[{}]),
                               env=types.SimpleNamespace(difficulty=None))
    ns["ctor"](me, **args)
    out = []
    for epoch in range(me.epochs + 1):
        lr, lip, diff = ns["preamble"](me, epoch)
        assert me.optimizer.param_groups[0]["lr"] == lr and me.env.difficulty == diff
        out.append((lr, lip, diff))
    return np.array(out, np.float64)


CASES = {
    "defaults": {},
    "long": dict(epochs=2000, lr=1e-3, lr_ratio=0.1, lr_epoch_index=900, lip_lp_index=[0.2, 0.6], lip_epoch_index=[300, 1500], diff_value=[0.0, 0.8],
                 diff_epoch_index=[150, 1100]),
    "short": dict(epochs=120, lipschitz_para=3.5, lip_ratio=[1.2, 0.5], diff_lp_index=[0.1, 0.9]),
    "off": dict(learning_rate_schedule=False, lipschitz_schedule=False, difficulty_schedule=False),
}
save = {}
for name, kw in CASES.items():
    save[name] = run(**kw)
    save[name + "_kw"] = np.array(repr(sorted(kw.items())))
np.savez_compressed(OUT / "schedules.npz", **save)
print({k: v.shape for k, v in save.items() if not k.endswith("_kw")}, "defaults epoch 250:", save["defaults"][250])
