"""Same-box A/B of step-kernel builds: `python tools/ab_step.py libA.so libB.so ... [--cases 1:4096,1:262144,...] [--rounds 3] [--roundtrip]`.
Each (library, case) is measured in its own child process (TACO_ENV_LIB override; the binding caches one library per process), the
libraries are visited round-robin `rounds` times so that clock drift of the box hits all of them alike; prints the median of the
rounds' medians in us per step (bench.py's steady-state protocol: 0.25 s of load, then 5 windows)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(case, roundtrip, record_flag):
    import torch
    import bench
    from taco_amd import config
    from taco_amd.vec_env import FpvBase
    parts = case.split(":")     # config index : envs [: kernel form]
    idx, n = int(parts[0]), int(parts[1])
    cfg = config.baseline_config(idx, num_envs=n)
    cfg["world_rate_roundtrip"] = bool(roundtrip)
    cfg["record_flag"] = bool(record_flag)
    env = FpvBase(cfg, copy_outputs=False, kernel_form=parts[2] if len(parts) > 2 else "auto")
    acts = bench.make_actions(n, 8, 0, "cuda")
    med, ws, first = bench.steady_windows(env.step_raw, acts, torch, 0.25, 5, 2000 if n <= 16384 else (400 if n <= 131072 else 200))
    print(json.dumps({"us": med, "spread": (ws[-1] - ws[0]) / med, "form": env.kernel_form}))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3] == "1", sys.argv[4] == "1")
        sys.exit(0)
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {a.split("=")[0]: (a.split("=") + ["1"])[1] for a in sys.argv[1:] if a.startswith("--")}
    cases = opt.get("--cases", "1:4096,1:262144").split(",")
    rounds = int(opt.get("--rounds", "3"))
    res = {}
    for r in range(rounds):
        for lib in libs:
            name, _, mode = lib.partition("@")   # lib.so@carried = run that library WITHOUT the world-rate round trip (the default since round 4); @rec = record_flag
            for case in cases:
                env = dict(os.environ, TACO_ENV_LIB=os.path.abspath(name), TACO_ENV_LIB_SKIP_ABI="1")
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case, "0" if mode == "carried" else "1", "1" if mode == "rec" else "0"],
                                     env=env, capture_output=True, text=True, cwd=ROOT)
                if out.returncode != 0:
                    print(lib, case, "FAILED", out.stderr[-500:], flush=True)
                    continue
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res.setdefault((lib, case), []).append(d["us"])
                print(f"round {r} {lib:40s} {case:24s} {d['us']:8.2f} us  spread {d['spread']:.3f}  {d['form']}", flush=True)
    print("---- median of rounds")
    for (lib, case), v in res.items():
        v.sort()
        print(f"{lib:40s} {case:24s} {v[len(v) // 2]:8.2f} us   {v}")
