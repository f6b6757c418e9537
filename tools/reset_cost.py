"""What a resetting env costs the wavefront that holds it: `python tools/reset_cost.py [config index, default 1] [envs, default 4096]`.
The step kernel's shader-clock stamps are those of workgroup 0; a launch whose workgroup 0 holds a resetting env (reset_buf[0] planted
before it) is compared with launches where it holds none (reset_buf[0:64] cleared before them), phase by phase.  Prints the medians of
the stamped launches in clocks (entry -> loads landed -> [reset_idx: stamp 9] -> pre-phase -> ten substeps -> state stores -> end; then the
three role wavefronts' ends, entry-relative)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from taco_amd import config
    from taco_amd.vec_env import FpvBase
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    cfg = config.baseline_config(idx, num_envs=n)
    env = FpvBase(cfg, copy_outputs=False)
    acts = bench.make_actions(n, 8, 0, "cuda")
    for k in range(300):
        env.step_raw(acts[k % 8])
    names = ["loads", "reset", "pre_phase", "ten_substeps", "state_stores"]
    out = {}
    for label, plant in (("no reset in workgroup 0", 0), ("env 0 resets", 1)):
        rows, ends = [], []
        for t in range(41):
            for k in range(3):   # a few warm launches back to back, then the stamped one
                env.step_raw(acts[(t + k) % 8])
            env.reset_buf[:64] = 0
            env.reset_buf[0] = plant
            ph = env.phase_stamps(acts[t % 8], steps=1, back_to_back=True)
            st = env._last_stamps.cpu().tolist()
            rows.append([st[1] - st[0], st[9] - st[1], st[2] - st[9], st[3] - st[2], st[4] - st[3]])
            ends.append([st[5] - st[0]] + [s - st[0] for s in st[6:9]])
        med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(5)]
        mend = [sorted(e[i] for e in ends)[len(ends) // 2] for i in range(4)]
        out[label] = (med, mend)
        print(f"{label:26s} " + "  ".join(f"{nm} {v:7.0f}" for nm, v in zip(names, med)) +
              f"   | step wavefront end {mend[0]:6.0f}  roles end {mend[1]:6.0f} {mend[2]:6.0f} {mend[3]:6.0f}   ({env.kernel_form})", flush=True)
    a, b = out["no reset in workgroup 0"], out["env 0 resets"]
    print(f"a resetting env adds {b[1][0] - a[1][0]:.0f} clocks to its step wavefront ({b[0][1] - a[0][1]:.0f} of them in reset_idx)")


if __name__ == "__main__":
    main()
