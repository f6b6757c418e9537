"""What resets cost a large launch: `python tools/reset_share.py [config index, default 1] [envs, default 262144]`.
Steady state of bench.py's protocol; prints the resets per step (fraction of envs, fraction of 64-env wavefronts holding one, histogram of
resetting lanes per wavefront) and the launch time (a) as it is, (b) with reset_buf cleared ahead of every launch (no reset_idx runs: the
envs that ended keep stepping from where they are) -- both with the same extra memset launch between steps."""
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
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    cfg = config.baseline_config(idx, num_envs=n)
    env = FpvBase(cfg, copy_outputs=False)
    acts = bench.make_actions(n, 8, 0, "cuda")
    for k in range(3000):
        env.step_raw(acts[k % 8])
    fr, wf, hist = [], [], torch.zeros(65, dtype=torch.long)
    for k in range(20):
        env.step_raw(acts[k % 8])
        r = (env.reset_buf != 0).view(-1, 64).sum(1).cpu()
        fr.append(float(r.sum()) / n)
        wf.append(float((r > 0).float().mean()))
        hist += torch.bincount(r, minlength=65)
    print(f"config {idx}, {n} envs ({env.kernel_form}): resets per step {sum(fr) / len(fr):.4%} of envs, {sum(wf) / len(wf):.1%} of wavefronts hold one")
    print("resetting lanes per wavefront (count: share of wavefronts): " + ", ".join(f"{k}: {float(hist[k]) / float(hist.sum()):.3%}" for k in range(65) if hist[k]))
    dummy = torch.zeros_like(env.reset_buf)
    steps = 2000 if n <= 16384 else (400 if n <= 131072 else 200)

    def as_is(a):
        dummy.zero_()
        env.step_raw(a)

    def cleared(a):
        env.reset_buf.zero_()
        env.step_raw(a)

    a_us, _, _ = bench.steady_windows(as_is, acts, torch, 0.25, 5, steps)
    c_us, _, _ = bench.steady_windows(cleared, acts, torch, 0.25, 5, steps)
    a2_us, _, _ = bench.steady_windows(as_is, acts, torch, 0.25, 5, steps)
    print(f"launch + memset: as it is {a_us:.2f} us, reset_buf cleared ahead of every launch {c_us:.2f} us, as it is again (the envs re-seeded by their resets) {a2_us:.2f} us")


if __name__ == "__main__":
    main()
