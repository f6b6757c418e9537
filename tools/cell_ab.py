"""The batched critic with the hardware-transcendental LSTM cell (default) vs the op-for-op one (exact_critic=True / TACO_P_EXACT_CELL):
`python tools/cell_ab.py [lib.so ...]` -- per library (default: the in-tree build; each in its own child process via TACO_ENV_LIB), same
seeded policy and frame ring [33 + 4, 4096, 26] (CELL_AB_SCALE multiplies the inputs): ms per values_ring() with either cell and the
largest |value difference| between them."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    import numpy as np, torch
    from taco_amd import policy as P
    import test_policy_gpu as TP
    rng = np.random.default_rng(7)
    sd = TP._random_policy(rng, 1, 5, [128, 128, 128], 128, [128, 128])
    fr = torch.from_numpy(rng.standard_normal((37, 4096, 26)).astype(np.float32) * float(os.environ.get("CELL_AB_SCALE", "1"))).cuda()
    res = {}
    kinds = (True, False, "f16", "bf16", True, False, "f16", "bf16") if os.environ.get("CELL_AB_SPLIT", "1") == "1" else (True, False, True, False)
    for exact in kinds:
        pol = P.ActorCritic(sd, 1, 5, exact_critic=exact is True, critic_split=exact if isinstance(exact, str) else None)
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
        res.setdefault(exact, []).append(sorted(ts)[3])
        res[("v", exact)] = v.cpu().numpy()
    ve, vf = res[("v", True)], res[("v", False)]
    for kind in ("f16", "bf16"):
        if kind in res:
            vs = res[("v", kind)]
            print(f"split {kind}: {min(res[kind]):.4f} ms per values_ring; max |difference| to the exact cell {np.abs(ve - vs).max():.3e}, to the hardware cell {np.abs(vf - vs).max():.3e}, "
                  f"all finite: {bool(np.isfinite(vs).all())}")
    print(f"exact cell {min(res[True]):.4f} ms, hardware cell {min(res[False]):.4f} ms per values_ring(37 frames x 4096 envs); values in "
          f"[{ve.min():.3f}, {ve.max():.3f}], max |difference| {np.abs(ve - vf).max():.3e}, all finite: {bool(np.isfinite(vf).all())}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    for lib in (sys.argv[1:] or [None]):
        env = dict(os.environ)
        if lib:
            env["TACO_ENV_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{lib or 'in-tree build'}: {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-1500:]}", flush=True)
