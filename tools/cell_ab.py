"""A/B of libtaco_env builds on the batched critic: `python tools/cell_ab.py libA.so libB.so ...` — each library in its own child process
(TACO_ENV_LIB), same seeded policy and frame ring [33+4, 4096, 26]; prints ms per values_ring() and the largest |value difference| to the first."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(out):
    import numpy as np, torch
    from taco_amd import policy as P
    import test_policy_gpu as TP
    rng = np.random.default_rng(7)
    pol = P.ActorCritic(TP._random_policy(rng, 1, 5, [128, 128, 128], 128, [128, 128]), 1, 5)
    fr = torch.from_numpy(rng.standard_normal((37, 4096, 26)).astype(np.float32) * float(os.environ.get("CELL_AB_SCALE", "1"))).cuda()
    v = pol.values_ring(fr)
    for _ in range(10):
        pol.values_ring(fr)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            pol.values_ring(fr)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8)
    np.save(out, v.cpu().numpy())
    print(f"{sorted(ts)[3]:.4f}")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    import numpy as np
    base = None
    for k, lib in enumerate(sys.argv[1:]):
        out = f"/tmp/cell_ab_{k}.npy"
        env = dict(os.environ, TACO_ENV_LIB=os.path.abspath(lib), TACO_ENV_LIB_SKIP_ABI="0")
        r = subprocess.run([sys.executable, __file__, "--child", out], env=env, capture_output=True, text=True)
        if r.returncode:
            print(lib, "FAILED", r.stderr[-2000:])
            continue
        v = np.load(out)
        base = v if base is None else base
        print(f"{lib}: {r.stdout.strip()} ms per values_ring(37 frames x 4096 envs); values in [{v.min():.3f}, {v.max():.3f}], max |diff to first| {np.abs(v - base).max():.3e}", flush=True)
