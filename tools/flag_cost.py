"""What each of config 5's flags costs the step kernel: `python tools/flag_cost.py [envs] [len_states]` -- BASELINE configs[4] (mix, every
randomisation + both noises) at `envs` (default 262 144) with all flags, then with each of them switched off alone, then with none (us per step,
bench.py's steady-state protocol, each variant in this one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taco_amd import config
from taco_amd.vec_env import FpvBase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
FL = ["random_rotordynamic_coe", "random_rotor_response", "random_aerodynamic_coe", "observation_noise", "rotor_noise", "ramdom_delay_time", "ramdom_deploy_time", "random_command"]
acts = bench.make_actions(n, 8, 0, "cuda")


def run(off):
    cfg = config.baseline_config(4, num_envs=n)
    cfg["env"]["lenStates"] = ls
    for k in off:
        cfg[k] = False
    env = FpvBase(cfg, copy_outputs=False)
    med, ws, _ = bench.steady_windows(env.step_raw, acts, torch, 0.25, 5, 200 if n > 65536 else 1000)
    form = env.kernel_form
    del env
    torch.cuda.empty_cache()
    return med, form


base, form = run([])
print(f"config 5 at {n} envs x {ls} state frames, every flag: {base:.2f} us ({form})", flush=True)
for k in FL:
    m, f = run([k])
    print(f"  without {k:26s} {m:8.2f} us  ({base - m:+.2f})  {f}", flush=True)
m, f = run(FL)
print(f"  without all of them        {m:8.2f} us  ({base - m:+.2f})  {f}")
