#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused VecTask.step() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one VecTask.step() of the whole batch = ONE taco_step launch through the C ABI.  Envs are independent, so
for N > 1 the timed region holds no collective (each rank steps its slice); the variant with the north-star's single RCCL
all-gather of the packed obs|reward|done|time-out block per step is timed right after and reported as "with_allgather".  Workload at N = 1: BASELINE configs[1]
(task_mode=pos, 4 096 envs, rotor_response_time=0.017).  For N > 1 every rank holds 4 096 envs (weak scaling; global
env ids are disjoint contiguous slices), value = N * 4096 * K / max-over-ranks time.
Inputs (the action stream a_t = clamp(0.3 N(0,1) + (-0.45,0,0,0), -1, 1)) are resident in HBM before the timed region.

Extra objects on the JSON line: "roofline" (step kernel vs HBM peak, 820 algorithmic bytes per env-step),
"cpu_baseline" (the CPU oracle timed on this host's cores on a bounded sample; rank 0, N = 1 only), "parity"
(HIP vs oracle on the first steps of this very workload), "large_n" (the same kernel at 262 144 and 1 048 576 envs, where
the HBM roofline is the meaningful bound).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 820.0   # 205 fp32 words (88 read + 117 written), len_obs = len_states = 1
HBM_PEAK_GBPS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s"


def make_actions(n, steps, seed, device):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = 0.3 * torch.randn((steps, n, 4), generator=g, dtype=torch.float32)
    a[:, :, 0] -= 0.45
    return a.clamp_(-1, 1).to(device)


def time_kernel_launches(env, acts, steps, torch):
    """Duration of the step kernel from HIP events on the launch stream (torch.cuda.Event records on torch's current
    stream, which is the stream taco_step launches on).  Two estimates, microseconds:
      back_to_back : one event pair around `steps` consecutive launches / steps  (kernel + inter-kernel gap: an upper bound
                     that does not pay the event-record overhead per launch; this is what rocprofv3's average tracks)
      bracketed    : median of per-launch event pairs (adds ~2 us of event overhead at this kernel size)"""
    na = acts.shape[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        env.step_raw(acts[t % na])
    e1.record()
    torch.cuda.synchronize()
    back_to_back = e0.elapsed_time(e1) * 1e3 / steps
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for t in range(steps):
        starts[t].record()
        env.step_raw(acts[t % na])
        ends[t].record()
    torch.cuda.synchronize()
    d = sorted(s.elapsed_time(e) * 1e3 for s, e in zip(starts, ends))
    return back_to_back, d[len(d) // 2]


def time_kernel_settled(env, acts, steps, torch, tries=3):
    """time_kernel_launches, repeated (at most `tries` times) while the back-to-back average sits more than 30 % above the median of
    the bracketed launches: on a shared host a multi-millisecond stall of the GPU now and then lands inside a 100-launch window and
    doubles its average.  Returns (average, bracketed median, windows measured).  Only the informational large-N / stacked entries
    use it; the headline window is never re-measured."""
    for k in range(tries):
        avg, med = time_kernel_launches(env, acts, steps, torch)
        if avg <= 1.3 * med:
            break
    return avg, med, k + 1


def pmc_traffic(n_envs):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC summary (FETCH_SIZE x2 per the gfx950 note in
    MI355X_MICROARCH.md + WRITE_SIZE, separate passes; profiles/README.md).  bench.py cannot run rocprofv3 on itself, so the
    figure is the one measured for the build that produced profiles/; None if no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1])).get("derived", {})
    key = f"hbm_bytes_per_launch_{n_envs}"
    return d.get(key), os.path.basename(files[-1])


def cpu_baseline(cfg, budget_s=12.0):
    """The CPU oracle (oracle/, kind 'port') on this host: all cores through OpenMP, bounded by wall time."""
    import numpy as np
    from taco_amd import config
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    flat = config.flat_cfg(cfg)
    n = flat["num_envs"]
    orc = O.OracleEnv(flat, threads=threads)
    rng = np.random.default_rng(0)
    acts = np.clip(0.3 * rng.standard_normal((64, n, 4)) + np.array([-0.45, 0, 0, 0]), -1, 1).astype(np.float32)
    for t in range(5):
        orc.step(acts[t])
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        for _ in range(10):
            orc.step(acts[steps % 64])
            steps += 1
    dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # the same oracle on ONE thread (SURVEY 8d asks for both ends), a 3 s sample
    orc1 = O.OracleEnv(flat, threads=1)
    orc1.step(acts[0])
    t1 = time.perf_counter()
    steps1 = 0
    while time.perf_counter() - t1 < min(3.0, budget_s):
        orc1.step(acts[steps1 % 64])
        steps1 += 1
    dt1 = time.perf_counter() - t1
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps x {n} envs of the same workload through the CPU oracle (OpenMP over envs), {dt:.1f} s on {model or 'host CPU'}",
            "single_thread": {"value": n * steps1 / dt1, "unit": "env-steps/s", "cores": 1, "sample": f"{steps1} steps x {n} envs, {dt1:.1f} s"}}


def parity_check(cfg, steps=60):
    import numpy as np
    import torch
    from taco_amd import config
    from taco_amd.vec_env import FpvBase
    from oracle import oracle as O
    flat = config.flat_cfg(cfg)
    n = flat["num_envs"]
    env = FpvBase(cfg, copy_outputs=False)
    orc = O.OracleEnv(flat, threads=min(os.cpu_count() or 1, 16))
    acts = make_actions(n, steps, 1234, "cuda")
    acts_h = acts.cpu().numpy()
    linf = 0.0
    flags_equal = True
    bits_equal = True
    for t in range(steps):
        env.step_raw(acts[t])
        orc.step(acts_h[t])
        if t % 10 == 9 or t == steps - 1:
            g = env.get_state().cpu().numpy()
            o = orc.get_state().view(np.float32)
            fin = np.isfinite(g[:13]) & np.isfinite(o[:13])
            linf = max(linf, float(np.abs(g[:13] - o[:13])[fin].max()))
            keep = np.r_[0:20, 26:g.shape[0]]   # rows 20..25 (rpy_old / rpy_continuous) are tracked for flip envs only
            bits_equal &= bool((g.view(np.uint32)[keep] == o.view(np.uint32)[keep]).all())
            flags_equal &= bool((env.reset_buf.cpu().numpy() == orc.reset_buf).all())
            bits_equal &= bool((env.obs_buf.cpu().numpy().view(np.uint32) == orc.obs_buf.view(np.uint32)).all())
            bits_equal &= bool((env.rew_buf.cpu().numpy().view(np.uint32) == orc.rew_buf.view(np.uint32)).all())
    return {"steps": steps, "envs": n, "traj_linf_vs_oracle": linf, "done_flags_equal": flags_equal, "all_words_bit_equal": bits_equal}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--gather", action="store_true", help="N > 1: put the per-step RCCL all-gather inside the main timed region "
                    "(default: the sharded path alone is timed, the gathered variant is timed separately and reported as with_allgather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the all-cores CPU baseline sample")
    ap.add_argument("--no-large-n", action="store_true")
    args = ap.parse_args()

    import torch
    from taco_amd import config
    from taco_amd.vec_env import FpvBase

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal knobs (not used by the driver): run N ranks on ONE GPU over gloo to exercise the N > 1 code path on a 1-GPU box
    backend = os.environ.get("TACO_BENCH_BACKEND", "nccl")
    if os.environ.get("TACO_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_local = args.envs
    n_global = n_local * world
    cfg = config.baseline_config(1, num_envs=n_global)
    if world > 1:
        from taco_amd.dist import ShardedEnv
        env = ShardedEnv(cfg, rank=rank, world_size=world, device=dev, gather=args.gather)
        step = env.step_gathered
    else:
        env = FpvBase(cfg, sim_device=str(dev), rl_device=str(dev), copy_outputs=False)
        step = env.step_raw
    n_act = 64
    acts = make_actions(n_local, n_act, 1000 + rank, dev)

    for t in range(args.warmup):
        step(acts[t % n_act])
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for t in range(args.steps):
        step(acts[t % n_act])
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    value = n_global * args.steps / elapsed

    with_gather = None
    if dist and not args.gather:
        # the same K steps with the north-star's single all-gather of [obs|rew|done|timeout] per step (one learner sees all envs).
        # The headline above is already measured: a failure of this optional leg is reported, it does not cost the result line.
        try:
            env.gather = True
            for t in range(min(args.warmup, 50)):
                step(acts[t % n_act])
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for t in range(args.steps):
                step(acts[t % n_act])
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            el = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            with_gather = {"value": n_global * args.steps / float(el.item()), "unit": "env-steps/s", "ms_per_step": float(el.item()) / args.steps * 1e3,
                           "collective": "1 all_gather_into_tensor of the kernel-filled [obs|rew|done|timeout] block per step",
                           "bytes_per_rank": int(env.block.numel() * 4)}
        except Exception as e:  # noqa: BLE001
            with_gather = {"error": repr(e)[:300]}
        env.gather = False

    out = None
    if rank == 0:
        base = env.env if world > 1 else env
        k_avg_us, k_med_us = time_kernel_launches(base, acts, min(args.steps, 500), torch)
        achieved = ALGO_BYTES_PER_ENV_STEP * n_local / (k_avg_us * 1e-6) / 1e9
        grid, block = base.launch_geometry()
        traffic, traffic_src = pmc_traffic(n_local)
        out = {
            "metric": "env-steps/s at 4096 envs per GPU, fpv_asymmetry.step() hot path (task_mode=pos)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: task_mode=pos, 4096 envs per GPU, rotor_response_time=0.017, delay_time=20, "
                                   "dt=0.001 x 10 substeps, PhysX-substeps=2, random pose/vel/target, battery on",
                       "envs_per_gpu": n_local, "envs_total": n_global, "parallelism": f"env-sharded x{world}",
                       "collective": ("1 RCCL all-gather of [obs|rew|done|timeout] per step" if (world > 1 and args.gather) else
                                      "none in the timed region: envs are independent, each rank steps its own slice"),
                       "kernel": base.lib.taco_step_kernel_name().decode(), "grid": grid, "block": block,
                       "lanes_per_env": 4 if grid * block >= 4 * n_local else 1,
                       "role_wavefronts": bool(block == 256 and grid * 64 >= 4 * n_local)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_avg_us": k_avg_us, "kernel_bracketed_median_us": k_med_us,
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                         "note": "4096 envs = 256 step wavefronts (4 lanes per env, one per CU) + 3 post-phase role wavefronts each: instruction-latency regime, see large_n for the throughput regime"},
            "gpu_event_ms_per_step": ev0.elapsed_time(ev1) / args.steps,
        }
        if with_gather is not None:
            out["with_allgather"] = with_gather
        if world == 1:
            if not args.no_large_n:
                out["large_n"] = []
                for big_n in (262144, 1048576):   # one full residency round (BASELINE configs[4]'s env count) / four rounds (steady state)
                    benv = FpvBase(config.baseline_config(1, num_envs=big_n), sim_device=str(dev), rl_device=str(dev), copy_outputs=False)
                    bacts = make_actions(big_n, 4, 7, dev)
                    for t in range(20):
                        benv.step_raw(bacts[t % 4])
                    torch.cuda.synchronize()
                    b_avg, b_med, b_win = time_kernel_settled(benv, bacts, 100, torch)
                    bg, bb = benv.launch_geometry()
                    ach = ALGO_BYTES_PER_ENV_STEP * big_n / (b_avg * 1e-6) / 1e9
                    tr, _ = pmc_traffic(big_n)
                    out["large_n"].append({"envs": big_n, "kernel_avg_us": b_avg, "env_steps_per_s": big_n / (b_avg * 1e-6), "achieved_GBps": ach,
                                           "frac_of_hbm_peak": ach / HBM_PEAK_GBPS, "traffic": tr, "grid": bg, "block": bb,
                                           "kernel_bracketed_median_us": b_med, "windows_measured": b_win})
                    del benv, bacts
                    torch.cuda.empty_cache()
                # the documented training configuration stacks 5 state frames for the LSTM critic (README.md:60-66 of the reference):
                # + (len - 1) * 104 B read and len * 104 B written per env-step for the stack (936 B at len 5)
                out["stacked_states"] = []
                for sn in (n_local, 262144):
                    scfg = config.baseline_config(1, num_envs=sn)
                    scfg["env"]["lenStates"] = 5
                    senv = FpvBase(scfg, sim_device=str(dev), rl_device=str(dev), copy_outputs=False)
                    sacts = make_actions(sn, 4, 7, dev)
                    for t in range(20):
                        senv.step_raw(sacts[t % 4])
                    torch.cuda.synchronize()
                    s_avg, s_med, s_win = time_kernel_settled(senv, sacts, 100, torch)
                    sbytes = ALGO_BYTES_PER_ENV_STEP + 4 * 104 + 5 * 104 - 104   # the len-1 state frame is already in the 820 B
                    out["stacked_states"].append({"envs": sn, "len_states": 5, "kernel_avg_us": s_avg, "env_steps_per_s": sn / (s_avg * 1e-6),
                                                  "algorithmic_bytes_per_env_step": sbytes, "achieved_GBps": sbytes * sn / (s_avg * 1e-6) / 1e9,
                                                  "frac_of_hbm_peak": sbytes * sn / (s_avg * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                                  "kernel_bracketed_median_us": s_med, "windows_measured": s_win})
                    del senv, sacts
                    torch.cuda.empty_cache()
            out["parity"] = parity_check(config.baseline_config(1, num_envs=n_local))
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(config.baseline_config(1, num_envs=n_local), budget_s=args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- the result line is out; a failing teardown must not turn the run into an error
            pass


if __name__ == "__main__":
    main()
