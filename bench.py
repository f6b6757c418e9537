#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused VecTask.step() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 without torchrun's environment: bench.py launches its own N ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...` as a child process, BEFORE anything in this process touches the GPU), relays rank 0's JSON line and exits with
the children's status.  Under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

A "step" is one VecTask.step() of the whole batch = ONE taco_step launch through the C ABI.  Workload at N = 1: BASELINE configs[1]
(task_mode=pos, 4 096 envs, rotor_response_time=0.017).  For N > 1 every rank holds 4 096 envs (weak scaling; global env ids are disjoint
contiguous slices) and EVERY timed step carries north_star's collective: one RCCL all-gather of the kernel-filled [obs|reward|done|time-out]
block per step, overlapped as SURVEY 8(e) prescribes (ShardedEnv.step_async: the gather of step t runs on the process group's stream
under the kernel of step t + 1 and is waited for right after that launch, before gather t + 1 is issued; the last gather is waited for inside the timed region).
value = N * 4096 * K / max-over-ranks time (every rank: W warm-up steps, synchronize + barrier + synchronize, t0, K steps + the drain of
the last gather, synchronize, t1, barrier; before the warm-up a twin env is stepped for 0.2 s so that a short region does not time the
process's first calls).  Reported beside it: "without_allgather" (the gather-free ShardedEnv.step_local of the same K steps: what a
data-parallel learner that never needs the global view pays) and "with_allgather_serial" (step + gather waited for before the next step: a
single learner that needs obs(t) before action(t + 1)); --no-gather makes the gather-free leg the headline instead.  Then a strong-scaling
leg (4 096 envs in TOTAL).  Inputs (the action stream a_t = clamp(0.3 N(0,1) + (-0.45,0,0,0), -1, 1)) are resident in HBM before any
timed region.

Extra objects on the JSON line (rank 0, N = 1): "roofline" (the step kernel at the bench workload vs the HBM peak, 820 algorithmic bytes
per env-step), "steady_state" (median of 5 x 2 000-step windows), "step_api" (VecTask.step(), the clamped copies included),
"configs" (BASELINE configs 2-5 at their per-GPU shapes, config 5's rollout incl. the policy), "large_n" (the same kernel at 262 144 and
1 048 576 envs after a time-based warm-up: the GPU needs ~40 ms of sustained load to reach its steady clock), "parity" (HIP vs oracle
on the first steps of this very workload) and "cpu_baseline" (the CPU oracle on this host's cores, bounded samples, configs 1-3).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 820.0   # 205 fp32 words (88 read + 117 written), len_obs = len_states = 1
EXTRA_BYTES_PER_STATE_FRAME = 104.0   # + 104 B per extra materialised states frame (config 5 => 1 236 B)
HBM_PEAK_GBPS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s"


def algo_bytes(len_states):
    return ALGO_BYTES_PER_ENV_STEP + EXTRA_BYTES_PER_STATE_FRAME * (len_states - 1)


def self_launch(args, argv):
    """--gpus N > 1 outside torchrun: start the ranks as a child process tree.  Nothing here touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(p.stdout[-4000:])
    sys.exit(p.returncode if p.returncode != 0 or lines else 1)


def make_actions(n, steps, seed, device):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = 0.3 * torch.randn((steps, n, 4), generator=g, dtype=torch.float32)
    a[:, :, 0] -= 0.45
    return a.clamp_(-1, 1).to(device)


def time_kernel_launches(env, acts, steps, torch):
    """Duration of the step kernel from HIP events on the launch stream (torch.cuda.Event records on torch's current
    stream, which is the stream taco_step launches on).  Two estimates, microseconds:
      back_to_back : one event pair around max(1 000, steps) consecutive launches, after 200 untimed ones  (kernel + inter-kernel gap: an upper bound
                     that does not pay the event-record overhead per launch; this is what rocprofv3's average tracks)
      bracketed    : median of per-launch event pairs (adds ~2 us of event overhead at this kernel size)"""
    na = acts.shape[0]
    for t in range(200):   # (untimed: the region below must not start on an idle GPU whatever --steps was)
        env.step_raw(acts[t % na])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    long_run = max(1000, steps)   # never fewer than 1 000 launches: the event pair's own cost (~1 us) must vanish in the average
    e0.record()
    for t in range(long_run):
        env.step_raw(acts[t % na])
    e1.record()
    torch.cuda.synchronize()
    back_to_back = e0.elapsed_time(e1) * 1e3 / long_run
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for t in range(steps):
        starts[t].record()
        env.step_raw(acts[t % na])
        ends[t].record()
    torch.cuda.synchronize()
    d = sorted(s.elapsed_time(e) * 1e3 for s, e in zip(starts, ends))
    return back_to_back, d[len(d) // 2]


def steady_windows(step_fn, acts, torch, warm_seconds, windows, steps_per_window):
    """Back-to-back launches: `warm_seconds` of untimed steps (the GPU reaches its steady clock only after ~40 ms of sustained load: a
    100-launch window right after creating a large env reads 10-25 % slow, profiles/r02_c_step_time_vs_step_index.txt), then `windows`
    event-bracketed windows.  -> (median us/step, all windows sorted, us/step of the first 100 launches)"""
    na = acts.shape[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(100):
        step_fn(acts[t % na])
    e1.record()
    torch.cuda.synchronize()
    first = e0.elapsed_time(e1) * 10.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_seconds:
        for t in range(200):
            step_fn(acts[t % na])
        torch.cuda.synchronize()
    ws = []
    for _ in range(windows):
        e0.record()
        for t in range(steps_per_window):
            step_fn(acts[t % na])
        e1.record()
        torch.cuda.synchronize()
        ws.append(e0.elapsed_time(e1) * 1e3 / steps_per_window)
    ws.sort()
    return ws[len(ws) // 2], ws, first


def pmc_summary():
    """-> (summary dict or None, file name or reason).  The committed rocprofv3 PMC summary (profiles/r*_pmc_summary.json) whose `source_hash` is
    THIS build's -- chosen by hash, not by file name (round 5's advisor: the lexicographically last file happened to be the right one) -- newest first
    among several.  No match: the reason is reported on the line instead of a silent None."""
    import glob
    from taco_amd import build
    want = build.source_hash()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=os.path.getmtime, reverse=True)
    seen = []
    for f in files:
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("source_hash") == want:
            return d, os.path.basename(f)
        seen.append(d.get("source_hash"))
    return None, (f"no profiles/r*_pmc_summary.json carries this build's source hash {want} ({len(files)} summaries of other builds: run tools/profile_all.sh)"
                  if files else "no PMC summary under profiles/")


def pmc_traffic(n_envs):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC summary (FETCH_SIZE x2 per the gfx950 note in
    MI355X_MICROARCH.md + WRITE_SIZE, separate passes; profiles/README.md).  bench.py cannot run rocprofv3 on itself, so the figure is
    the one measured for the build that produced the summary: it is reported only if that build's source hash is THIS build's."""
    d, name = pmc_summary()
    if d is None:
        return None, name
    return d.get("derived", {}).get(f"hbm_bytes_per_launch_{n_envs}"), name


def rocprof_row(shape, needle="taco_step_kernel", by="TotalDurationNs"):
    """the size-isolated `rocprofv3 --kernel-trace --stats` table of this build for one shape (tools/profile_all.sh step 5d:
    profiles/<tag>_kernel_stats_<shape>.csv, <tag> = the PMC summary's): the row of the kernel that dominates it -- what the figure next to it
    can be checked against.  None if this build has not been profiled."""
    import csv
    d, name = pmc_summary()
    if d is None:
        return None
    f = os.path.join(ROOT, "profiles", name.replace("_pmc_summary.json", f"_kernel_stats_{shape}.csv"))
    if not os.path.exists(f):
        return None
    rows = [r for r in csv.DictReader(open(f)) if needle in r["Name"]]
    if not rows:
        return None
    r = max(rows, key=lambda r: float(r[by]))
    return {"csv": os.path.relpath(f, ROOT), "kernel": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) * 1e-3,
            "min_us": float(r["MinNs"]) * 1e-3}


# VALU issue roof.  One SIMD issues one wave64 VALU instruction per ~1.2 ns in a realistic mix at 2-4 wavefronts per SIMD (fma / fmac / mul / add
# with VGPR operands 1.05-1.3 ns, SGPR-operand and VOP3 forms up to 2 ns, selects ~1.15 inside a mix: tools/ubench/intmul_cndmask.hip,
# profiles/r05_b_ubench_intmul_cndmask.txt; the step loop's own census priced that way averages 1.23 ns, tools/isa_audit.py) -- so
# floor = SQ_INSTS_VALU per launch / 1 024 SIMDs x 1.2 ns, the instruction count from the committed PMC summary of THIS build.
VALU_NS_PER_INST = 1.2
N_SIMDS = 1024


def pmc_valu(n_envs, kernel_us):
    d, name = pmc_summary()
    if d is None:
        return {"missing": name}
    key = {262144: "sq_262144", 4096: "sq_4096"}.get(n_envs)
    insts = d.get(key, {}).get("SQ_INSTS_VALU", {}).get("mean_per_launch") if key else None
    if not insts:
        return {"missing": f"{name} holds no SQ_INSTS_VALU for {n_envs} envs"}
    floor_us = insts / N_SIMDS * VALU_NS_PER_INST * 1e-3
    return {"insts_per_launch": insts, "ns_per_inst": VALU_NS_PER_INST, "simds": N_SIMDS, "floor_us": floor_us, "frac": floor_us / kernel_us,
            "source": name + " (SQ_INSTS_VALU) x tools/ubench/intmul_cndmask (ns per instruction in a realistic mix)"}


def cpu_sample(cfg, threads, budget_s):
    """the CPU oracle (oracle/, kind 'port') stepping `cfg` on `threads` OpenMP threads for about budget_s seconds"""
    import numpy as np
    from taco_amd import config
    from oracle import oracle as O
    flat = config.flat_cfg(cfg)
    n = flat["num_envs"]
    orc = O.OracleEnv(flat, threads=threads)
    rng = np.random.default_rng(0)
    acts = np.clip(0.3 * rng.standard_normal((16, n, 4)) + np.array([-0.45, 0, 0, 0]), -1, 1).astype(np.float32)
    orc.step(acts[0])
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        orc.step(acts[steps % 16])
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": threads, "sample": f"{steps} steps x {n} envs, {dt:.1f} s"}


def cpu_baseline(budget_s=12.0):
    """SURVEY 8(d) configs 1-3 on this host: all cores (bounded by wall time) and one thread.  The headline object is config 2 (pos, 4 096 envs) on all cores."""
    from taco_amd import config
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    main = cpu_sample(config.baseline_config(1, num_envs=4096), threads, budget_s)
    out = {"value": main["value"], "unit": "env-steps/s", "cores": threads, "kind": "port",
           "sample": f"{main['sample']} of the same workload through the CPU oracle (OpenMP over envs) on {model or 'host CPU'}",
           "single_thread": cpu_sample(config.baseline_config(1, num_envs=4096), 1, min(3.0, budget_s)), "configs": [],
           # context, NOT measured in this run: the reference's OWN torch modules (rows C-H'', O, P: no rigid-body integrate, no glue) timed by the survey in the
           # build container (48.1 ms per step at 4 096 envs on 8 cores) -- the reference cannot travel to the GPU box
           "reference_torch_partial_path_in_build_container": {"value": 85e3, "unit": "env-steps/s", "envs": 4096, "cores": 8, "ms_per_step": 48.1,
                                                               "source": "BASELINE.md section 2 (survey's measurement; partial path, not this run)"}}
    side = max(1.0, min(3.0, budget_s / 4))
    for label, idx, n in ((1, 0, 64), (3, 2, 16384)):   # SURVEY 8(d) numbering: 1 = pos @ 64 (plumbing), 2 = the headline above, 3 = rotate @ 16 384
        cfg = config.baseline_config(idx, num_envs=n)
        out["configs"].append({"config": label, "task_mode": cfg["task_mode"], "envs": n, "all_cores": cpu_sample(cfg, min(threads, max(1, n // 16)), side),
                               "single_thread": cpu_sample(cfg, 1, side)})
    return out


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
    env.check()
    return {"steps": steps, "envs": n, "traj_linf_vs_oracle": linf, "done_flags_equal": flags_equal, "all_words_bit_equal": bits_equal,
            "note": "HIP vs the CPU oracle; the oracle is pinned to the reference by fixtures from its own code, except row I (the rigid-body "
                    "integrate standing in for the closed PhysX binary): parity unpinned, error bounded in DESIGN.md section 5"}


def config_entry(idx, n, dev, torch, warm=0.1):
    """one BASELINE config at its per-GPU shape: steady-state kernel time, env-steps/s, fraction of the HBM roofline"""
    from taco_amd import config
    from taco_amd.vec_env import FpvBase
    cfg = config.baseline_config(idx, num_envs=n)
    env = FpvBase(cfg, sim_device=str(dev), rl_device=str(dev), copy_outputs=False)
    acts = make_actions(n, 8, 11, dev)
    med, ws, _ = steady_windows(env.step_raw, acts, torch, warm, 5, 400)
    b = algo_bytes(env.len_states)
    grid, block = env.launch_geometry()
    extra = {}
    if env.len_states > 1:
        # ... that was the MATERIALISED stack (step_raw in place: 2 len - 1 frames moved per env-step).  VecTask.step() keeps the stack in a frame
        # ring (taco_bind_states_ring): one frame written, the algorithmic 820 B -- the figure of this entry; the materialised one beside it
        extra = {"materialised_stack": {"kernel_us": med, "algorithmic_bytes_per_env_step": b, "frac_of_hbm_peak": b * n / (med * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                        "kernel_form": env.kernel_form}}
        del env
        env = FpvBase(cfg, sim_device=str(dev), rl_device=str(dev))
        med, ws, _ = steady_windows(env.step, acts, torch, warm, 5, 400)
        b = ALGO_BYTES_PER_ENV_STEP
        extra["path"] = "VecTask.step(): state stack = strided view of the frame ring, one frame written per env-step"
    flags = [k for k in ("random_rotordynamic_coe", "random_rotor_response", "random_aerodynamic_coe", "observation_noise", "rotor_noise",
                         "ramdom_delay_time", "ramdom_deploy_time", "random_command") if cfg.get(k)]
    return {"config": idx + 1, "task_mode": cfg["task_mode"], "envs": n, "len_states": env.len_states, "flags_on": flags, "kernel_us": med,
            "env_steps_per_s": n / (med * 1e-6), "algorithmic_bytes_per_env_step": b, "achieved_GBps": b * n / (med * 1e-6) / 1e9,
            "frac_of_hbm_peak": b * n / (med * 1e-6) / 1e9 / HBM_PEAK_GBPS, "kernel_form": env.kernel_form, "grid": grid, "block": block,
            "rocprof": rocprof_row({(2, 16384): "rotate_16384", (3, 16384): "flip_16384", (4, 32768): "mix_32768x5"}.get((idx, n), "none")), **extra}


def graph_entry(n, acts, dev, torch, steps_in_graph=64):
    """the same workload as ONE HIP graph of `steps_in_graph` steps (each step = ONE kernel node: every step wavefront reads and rewrites the
    device-resident clock pair of its 16 envs), replayed back to back: what the per-step kernel boundary costs when the host is out of the loop"""
    from taco_amd import config
    from taco_amd.vec_env import FpvBase
    env = FpvBase(config.baseline_config(1, num_envs=n), sim_device=str(dev), rl_device=str(dev), copy_outputs=False)
    for t in range(8):
        env.step_raw(acts[t % acts.shape[0]])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(steps_in_graph):
            env.step_raw(acts[t % acts.shape[0]])
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        g.replay()
    torch.cuda.synchronize()
    ws = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ws.append(e0.elapsed_time(e1) * 1e3 / (20 * steps_in_graph))
    ws.sort()
    env.check()
    return {"steps_per_graph": steps_in_graph, "us_per_step": ws[2], "env_steps_per_s": n / (ws[2] * 1e-6), "windows_us": ws,
            "what": "taco_step captured into a HIP graph (ONE kernel node per step: the step kernel advances the device-resident clock itself), 20 replays per window"}


def documented_policy(rng, hd=128):
    """random-init weights of the documented architecture (README.md:60-66): actor MLP 26-128-128-128-4, critic LSTM 26 -> 128 + MLP 128-128-128-1"""
    import numpy as np
    sd = {"log_std": np.zeros(4, np.float32)}
    dims = [26, hd, hd, hd, 4]
    for i in range(4):
        sd[f"actor_mlp.layers.{2 * i}.weight"] = (rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)
        sd[f"actor_mlp.layers.{2 * i}.bias"] = np.zeros(dims[i + 1], np.float32)
    sd["critic_encoder.layers.weight_ih_l0"] = (rng.standard_normal((4 * hd, 26)) * 0.2).astype(np.float32)
    sd["critic_encoder.layers.weight_hh_l0"] = (rng.standard_normal((4 * hd, hd)) / np.sqrt(hd)).astype(np.float32)
    sd["critic_encoder.layers.bias_ih_l0"] = np.zeros(4 * hd, np.float32)
    sd["critic_encoder.layers.bias_hh_l0"] = np.zeros(4 * hd, np.float32)
    dims = [hd, hd, hd, 1]
    for i in range(3):
        sd[f"critic_mlp.layers.{2 * i}.weight"] = (rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)
        sd[f"critic_mlp.layers.{2 * i}.bias"] = np.zeros(dims[i + 1], np.float32)
    return sd


def rollout_entry(n, horizon, dev, torch):
    """config 5's 'LSTM-critic rollout': one taco_rollout_run call = horizon x (actor forward -> clipped action -> env step writing the next
    replay slot), then the critic over all horizon + 1 slots in one batched pass + time-out bootstrap, then GAE.  Random-init weights of the
    documented architecture (actor MLP 26-128-128-128-4, critic LSTM 26->128 over 5 frames + MLP 128-128-128-1).  The critic is timed alone
    as well (over the replay store's frame ring) in BOTH of its forms: the host layer's default since round 5 -- the ring-form LSTM on the
    16-bit matrix pipe with split-f16 operands (TACO_P_SPLIT_F16: values within 1e-6 of the f32 kernel's) -- and the
    f32 MFMA kernel (ActorCritic(critic_split=None), the C ABI's own default), priced against the f32 MFMA peak."""
    import numpy as np
    from taco_amd import config, policy as P
    from taco_amd.rollout import RolloutBuffer
    from taco_amd.vec_env import FpvBase
    cfg = config.baseline_config(4, num_envs=n)
    env = FpvBase(cfg, sim_device=str(dev), rl_device=str(dev), copy_outputs=False)
    buf = RolloutBuffer(n, 26, 1, 26, env.len_states, 4, horizon, 4, 0.99, 0.95, str(dev))
    hd = 128
    sd = documented_policy(np.random.default_rng(0))
    pol = P.ActorCritic(sd, 1, env.len_states)                        # default: critic_split = "f16"
    pol32 = P.ActorCritic(sd, 1, env.len_states, critic_split=None)   # the f32 MFMA critic

    def time_rollout(p, reps=5, inner=4):
        def run():
            buf.reset()
            last = buf.run(env, p)
            buf.compute_returns_and_advantage(last)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(inner):
                run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / inner)
        ts.sort()
        return ts[len(ts) // 2]

    def time_critic(p, st):
        for _ in range(3):
            p.values_ring(st)
        torch.cuda.synchronize()
        cs = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                p.values_ring(st)
            e1.record()
            torch.cuda.synchronize()
            cs.append(e0.elapsed_time(e1) * 1e-3 / 4)
        cs.sort()
        return cs[2]

    fused_s = time_rollout(pol)
    fused32_s = time_rollout(pol32)
    env.set_rollout_fusion(False)   # the launch-per-step path of the same call (2 x horizon + 3 launches), for comparison
    lps_s = time_rollout(pol, reps=3)
    env.set_rollout_fusion(True)
    forced_ms = None
    if n > 8192:   # the persistent kernel FORCED above its limit (its workgroups queue: 16 envs hold a whole CU): the record of why the limit stands
        try:
            env.set_rollout_fusion("force")
            forced_ms = time_rollout(pol, reps=3, inner=2) * 1e3
        finally:
            env.set_rollout_fusion(True)
    rows, T, hd2 = (horizon + 1) * n, env.len_states, hd
    st = buf._frames   # the replay store's frame ring [horizon + T][n][26]: what taco_rollout_run's critic pass reads
    c16, c32 = time_critic(pol, st), time_critic(pol32, st)
    v16, v32 = pol.values_ring(st), pol32.values_ring(st)
    dv, vmax = float((v32 - v16).abs().max()), float(v32.abs().max())
    flops = 2 * rows * (T * 4 * hd2 * (26 + hd2) + 2 * hd2 * hd2 + hd2)
    # what the f32 kernels actually issue (v_mfma_f32_16x16x4_f32 = 2 048 flop): the paired-slot LSTM kernel runs blocks of 16 envs x 2 slots, per
    # wavefront 64 MFMAs at the first timestep (x chains only: h_-1 = 0) and 288 at each later one (shared input projection); the MLP 544 per
    # 16 rows.  K is padded 26 -> 32 in the x chains, so this is not a subset of the model count either; SQ_INSTS_MFMA in profiles/ agrees.
    issued = 2048 * (((n + 15) // 16) * ((horizon + 2) // 2) * 8 * (64 + (T - 1) * 288) + ((rows + 15) // 16) * 544) if T >= 2 else None
    return {"config": 5, "what": "taco_rollout_run (actor forward + env step per step, replay store fused; critic batched over all slots afterwards) + GAE",
            "envs": n, "horizon": horizon, "len_states": T, "ms_per_rollout": fused_s * 1e3, "env_steps_per_s": n * horizon / fused_s,
            "ms_per_rollout_f32_critic": fused32_s * 1e3, "env_steps_per_s_f32_critic": n * horizon / fused32_s,
            "per_step_chain": ("ONE persistent kernel (a workgroup owns 16 envs for the whole horizon; the actor's MFMAs run under the substeps): 4 launches per rollout"
                               if n <= 8192 else "actor launch + step launch per step (above 8 192 envs)"),
            "ms_per_rollout_launch_per_step": lps_s * 1e3,
            "ms_per_rollout_persistent_forced": forced_ms,
            "critic": {"rows": rows, "ms": c16 * 1e3, "rocprof": rocprof_row(f"critic_{rows}", "critic_lstm"), "kernel": "ring-form LSTM on v_mfma_f32_16x16x32_f16 with split-f16 operands (the host layer's default, TACO_P_SPLIT_F16) + the MLP's two hidden layers likewise (f32 head on the VALU)",
                       "tflops_model": flops / c16 / 1e12, "speedup_vs_f32_critic": c32 / c16,
                       "max_abs_value_difference_to_f32_critic": dv, "max_abs_value": vmax,
                       "bound": "VALU issue (the cells' activations and the operand splits), not the matrix pipe: DESIGN.md section 4.3",
                       "note": "every f32 operand of the LSTM's chains as two f16 halves (11 + 11 bits), three MFMA chains per product chain into one f32 accumulator; "
                               "trajectories / actions / log-probs / dones bit-identical to the f32 critic's rollout"},
            "critic_f32": {"rows": rows, "ms": c32 * 1e3, "tflops": flops / c32 / 1e12, "frac_of_f32_mfma_peak": flops / c32 / 157.3e12,
                           "mfma_tflops_issued": issued / c32 / 1e12 if issued else None,
                           "frac_of_f32_mfma_peak_issued": issued / c32 / 157.3e12 if issued else None,
                           "lstm_cell": "hardware 2^x / reciprocal (TACO_P_EXACT_CELL keeps the oracle's operations, +23 % critic time)",
                           "note": "ActorCritic(critic_split=None), the C ABI's default. tflops / frac_of_f32_mfma_peak: MODEL flops 2 rows (T 4 H (26 + H) + 2 H H + H) over the LSTM + MLP kernels' time -- the "
                                   "kernels skip the h chain of the first timestep and share input projections between neighbouring slots, so the matrix "
                                   "pipe's own utilisation is the *_issued pair (MFMA instructions issued x 2 048 flop); peak 157.3 TFLOP/s (MI355X_MICROARCH.md)"}}


def shader_clock_mhz():
    """peak shader clock of the GPU from sysfs (pp_dpm_sclk's highest level); 2400 (MI355X_MICROARCH.md) if it cannot be read"""
    import glob
    import re
    best = 0
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                m = re.search(r"(\d+)\s*Mhz", line, re.I)
                if m:
                    best = max(best, int(m.group(1)))
        except OSError:
            pass
    return (best, "sysfs pp_dpm_sclk") if best > 0 else (2400, "MI355X_MICROARCH.md peak engine clock (sysfs not readable)")


def latency_floor(env, acts, kernel_avg_us, torch):
    """What bounds the 4 096-env launch: not bytes but the instruction stream of its SLOWEST wavefront (every step wavefront has a SIMD to
    itself and issues one instruction per ~5 clocks whatever the dependencies: tools/ubench/valu_single).  Measured, not modelled: the
    kernel's own shader-clock stamps (taco_bind_phase_stamps) of workgroup 0 give the critical path in clocks -- step wavefront entry -> its
    last store, and the latest role wavefront's end; at the GPU's peak shader clock that is the floor the launch cannot beat without
    shortening the stream.  instr_per_step_wavefront comes from the committed PMC summary of the same build when there is one."""
    # (workgroup 0 holds 16 envs: at ~1 % resets per env-step one stamped launch in six sees reset_env in its `loads` phase, ~4 400 clocks; the
    # floor is the reset-free path, so the shortest of seven stamped launches is reported and the longest beside it)
    runs = []
    for k in range(7):
        ph_k = env.phase_stamps(acts[k % acts.shape[0]], steps=40 + k, back_to_back=True)
        st_k = env._last_stamps.cpu().tolist()
        runs.append((st_k[5] - st_k[0], ph_k, st_k))
    runs.sort(key=lambda r: r[0])
    _, ph, st = runs[0]
    wave0 = st[5] - st[0]
    roles_end = max(st[6:9]) - st[0] if max(st[6:9]) > 0 else wave0
    mhz, src = shader_clock_mhz()
    ticks = max(wave0, roles_end)
    out = {"critical_path_clocks": ticks, "step_wavefront_clocks": wave0, "phases_clocks": {"loads": ph[0], "pre_phase": ph[1], "ten_substeps": ph[2],
           "state_stores_and_frames": ph[3], "tail": ph[4]}, "shader_clock_mhz": mhz, "clock_source": src, "us": ticks / mhz,
           "frac_of_kernel_avg": ticks / mhz / kernel_avg_us, "step_wavefront_clocks_longest_of_7": runs[-1][0],
           "note": "kernel_avg_us - us = dispatch of 256 workgroups x 4 wavefronts + the drain of the last stores (GPU-side launch boundary)"}
    try:
        import glob
        from taco_amd import build
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        d = json.load(open(files[-1])) if files else {}
        if d.get("source_hash") == build.source_hash() and "instr_per_step_wavefront_4096" in d.get("derived", {}):
            out["instr_per_whole_step_wavefront"] = d["derived"]["instr_per_step_wavefront_4096"]   # counters of the quad form WITHOUT role wavefronts: one
            # wavefront runs the whole step (pre-phase, ten substeps, post-phase); the headline form's step wavefront hands the post-phase and the battery to its roles
    except Exception:  # noqa: BLE001
        pass
    return out


def init_dist(world, rank, dev, backend, torch):
    """-> (torch.distributed or None, backend in use, error text or None).  The requested backend first ("nccl" = RCCL); if its
    initialisation fails, gloo (the timed region holds no data-path collective: only the barriers and the max-over-ranks reduction need
    a process group); if that fails too the ranks run unsynchronised and the line says so."""
    if world == 1:
        return None, None, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    err = None
    for b in ([backend, "gloo"] if backend != "gloo" else ["gloo"]):
        try:
            if b == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(b, rank=rank, world_size=world)
            probe = torch.ones(1, device=dev if b == "nccl" else "cpu")
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
            return dist, b, err
        except Exception as e:  # noqa: BLE001
            err = (err + " | " if err else "") + f"{b}: {repr(e)[:240]}"
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
    return None, None, err


BASELINE_TOTALS = {2: ("pos", 1, None), 3: ("rotate", 2, 16384), 4: ("flip", 3, 65536), 5: ("mix", 4, 262144)}   # BASELINE.json configs[1..4] (1-based numbers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="which BASELINE.json config (1-based) the MAIN timed leg runs: 2 = pos, 4 096 envs per GPU (weak scaling, the metric's config); "
                         "3 = rotate 16 384 / 4 = flip 65 536 / 5 = mix 262 144 envs IN TOTAL, sharded over the N ranks (strong scaling)")
    ap.add_argument("--gather", action="store_true", help="(the default since round 6; kept so that old command lines still parse)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the per-step all-gather out of the main timed region (the gather-free "
                    "sharded path becomes the headline; the gathered variants are still timed and reported)")
    ap.add_argument("--direct-rccl", action="store_true", help="N > 1 over RCCL: the per-step all-gather as ONE ncclAllGather called directly on a comm stream "
                    "(taco_amd/rccl.py; opt-in: its multi-rank initialisation has never run on real hardware) instead of torch.distributed's collective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall-time budget of the all-cores CPU baseline sample of config 1")
    ap.add_argument("--no-large-n", action="store_true")
    ap.add_argument("--no-configs", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args, sys.argv[1:])   # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from taco_amd import config
    from taco_amd.vec_env import FpvBase

    # rehearsal knobs (not used by the driver): run N ranks on ONE GPU over gloo to exercise the N > 1 code path on a 1-GPU box
    backend = os.environ.get("TACO_BENCH_BACKEND", "nccl")
    if os.environ.get("TACO_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, backend_used, dist_error = init_dist(world, rank, dev, backend, torch)
    solo = world > 1 and dist is None    # no process group could be formed: every rank measures alone, rank 0 reports its own rate x N

    task, cfg_idx, total = BASELINE_TOTALS[args.config]
    weak = args.config == 2
    n_global = args.envs * world if weak else total
    cfg = config.baseline_config(cfg_idx, num_envs=n_global)
    gathered_main = world > 1 and not args.no_gather and not solo   # north_star: "a single RCCL all-gather of the (obs, reward, done) buffers per step"
    finish_main = None
    if world > 1:
        from taco_amd.dist import ShardedEnv
        env = ShardedEnv(cfg, rank=rank, world_size=world, device=dev, gather=gathered_main, direct=bool(args.direct_rccl and backend_used == "nccl" and not solo))
        n_local = env.hi - env.lo
        pend = []

        def wait_previous():
            while pend:
                pend.pop(0).wait()

        def step_overlapped(a):   # the gather of step t is waited for once the KERNEL of step t + 1 is in flight, then gather t + 1 is issued (two alternating blocks)
            pend.append(env.step_async(a, before_gather=wait_previous))

        def finish_overlapped():  # ... and the last one inside the timed region
            while pend:
                pend.pop(0).wait()
            env.drain()

        # --no-gather: no global view, every rank makes the launch a single-GPU VecTask.step() makes (no gather block bound: filling it
        # costs 0.4 us per step at 4 096 envs, tools/host_cost.py)
        step = step_overlapped if gathered_main else env.step_local
        finish_main = finish_overlapped if gathered_main else None
    else:
        env = FpvBase(cfg, sim_device=str(dev), rl_device=str(dev))   # copy_outputs=True: VecTask.step() as the PPO loop calls it
        step = env.step
        n_local = n_global
    n_act = 64 if n_local <= 65536 else 8
    acts = make_actions(n_local, n_act, 1000 + rank, dev)

    def sync_ranks():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist:
            return x
        t = torch.tensor([x], device=dev if backend_used == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(x):
        if not dist:
            return [x]
        t = torch.tensor([x], device=dev if backend_used == "nccl" else "cpu", dtype=torch.float64)
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [float(o.item()) for o in outs]

    def close_interval(t0):
        """every rank's interval ends at ITS synchronize; the closing barrier follows (it is part of the bracket, not of any rank's K steps:
        an RCCL or gloo barrier is 50-200 us, a fifth of a K = 20 region); the result is the MAX over ranks"""
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        return max_over_ranks(el)

    def timed(step_fn, a, steps, warmup, finish=None):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; max over ranks"""
        na = a.shape[0]
        for t in range(warmup):
            step_fn(a[t % na])
        if finish:
            finish()
        batch = [a[t % na] for t in range(steps)]   # (the views of the synthetic action batches are made ahead: not part of a step)
        sync_ranks()
        t0 = time.perf_counter()
        for x in batch:
            step_fn(x)
        if finish:
            finish()
        return close_interval(t0)

    # ---- before it: the host path and the GPU brought to their running state on a TWIN env (the measured env's trajectory stays W + K steps
    # from its reset).  A short region (the driver runs K = 20, W = 5) otherwise measures the process's first step() calls -- on a fresh box
    # the interpreter and libraries still page in: 17.0 us per step instead of 14.0 in the same process a moment later (tools/short_run.py).
    pre_warm = None
    try:
        twin = FpvBase(config.baseline_config(cfg_idx, num_envs=n_local), sim_device=str(dev), rl_device=str(dev))
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.2:
            for t in range(n_act):
                twin.step(acts[t])
            k += n_act
        torch.cuda.synchronize()
        pre_warm = {"steps": k, "seconds": time.perf_counter() - t0, "on": "a twin env of the same configuration (not the measured one)"}
        del twin
    except Exception as e:  # noqa: BLE001
        pre_warm = {"error": repr(e)[:200]}
    import gc
    gc.collect()

    # ---- the main timed leg.  It holds no data-path collective (envs are independent); a failure inside it still yields a line.
    main_error = None
    try:
        elapsed = timed(step, acts, args.steps, args.warmup, finish_main)
    except Exception as e:  # noqa: BLE001
        main_error = repr(e)[:300]
        dist = None
        solo = world > 1
        gathered_main = False
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(args.steps):
            (env.env.step_raw if world > 1 else env.step_raw)(acts[t % n_act])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    value = (n_local * world if solo else n_global) * args.steps / elapsed

    base = env.env if world > 1 else env
    k_avg_us, k_med_us = time_kernel_launches(base, acts, min(args.steps, 500), torch)
    multi = None
    if world > 1:
        multi = {"backend": backend_used, "kernel_us_per_rank": gather_floats(k_avg_us), "envs_per_rank": gather_floats(float(n_local))}
        if dist_error:
            multi["collective_init_error"] = dist_error
        if solo:
            multi["error"] = "no process group: the ranks ran unsynchronised; value = rank 0's own rate x N (" + (dist_error or main_error or "") + ")"
            multi["ranks_seen"] = [rank]
        if main_error:
            multi["main_leg_error"] = main_error
    if dist:
        # which ranks the collective library actually connected (one all-gather of rank ids)
        ids_dev = dev if backend_used == "nccl" else "cpu"
        ids = [torch.zeros(1, dtype=torch.int64, device=ids_dev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank], device=ids_dev, dtype=torch.int64))
        multi["ranks_seen"] = sorted(int(x.item()) for x in ids)
        # The same K steps in the two other forms (the headline above is already measured: a failure of these legs is reported, it does not cost
        # the result line): gather-free (step_local), and with the gather waited for before the next step (serial: one learner sees all envs)
        try:
            if gathered_main:
                el_local = timed(env.step_local, acts, args.steps, min(args.warmup, 50))
                multi["without_allgather"] = {"value": n_global * args.steps / el_local, "unit": "env-steps/s", "ms_per_step": el_local / args.steps * 1e3,
                                              "what": "ShardedEnv.step_local: every rank steps its own slice, no gather block bound, no collective"}
            env.gather = True
            el_serial = timed(env.step_gathered, acts, args.steps, min(args.warmup, 50))
            multi["with_allgather_serial"] = {"value": n_global * args.steps / el_serial, "unit": "env-steps/s", "ms_per_step": el_serial / args.steps * 1e3,
                                              "what": "step + gather, waited for before the next step is launched"}
            if not gathered_main:
                el_over = timed(step_overlapped, acts, args.steps, min(args.warmup, 50), finish_overlapped)
                multi["with_allgather"] = {"value": n_global * args.steps / el_over, "unit": "env-steps/s", "ms_per_step": el_over / args.steps * 1e3,
                                           "what": "overlapped: step t + 1 is launched before the gather of step t is waited for (two alternating blocks)"}
            multi["allgather"] = {"collective": "1 all_gather_into_tensor of the kernel-filled [obs|rew|done|timeout] block per step",
                                  "bytes_per_rank": int(env.block.numel() * 4), "backend": backend_used,
                                  "path": ("ncclAllGather called directly on a comm stream (--direct-rccl, taco_amd/rccl.py)" if env.direct_path else
                                           "torch.distributed all_gather_into_tensor (async_op)"), "direct_reason": env.direct_reason,
                                  "host_cost_note": "torch.distributed's collective costs 24-38 us of host time per call on this stack (profiles/r06_l_gather_host_cost.txt, one rank): "
                                                    "at 4 096 envs per GPU the gathered headline is bound by that, not by the 11-us kernel"}
        except Exception as e:  # noqa: BLE001
            multi["allgather_legs_error"] = repr(e)[:300]
        env.gather = gathered_main
        short = max(50, min(args.steps, 300))
        if weak:
            # strong scaling: BASELINE's "4 096 envs" spread over the N GPUs (512-2 048 envs per rank: the latency regime, expected to lose)
            try:
                from taco_amd.dist import ShardedEnv
                n_s = 4096
                senv = ShardedEnv(config.baseline_config(1, num_envs=n_s), rank=rank, world_size=world, device=dev, gather=False)
                sacts = make_actions(senv.hi - senv.lo, n_act, 2000 + rank, dev)
                el_s = timed(senv.step_gathered, sacts, args.steps, min(args.warmup, 50))
                senv.gather = True
                el_sg = timed(senv.step_gathered, sacts, args.steps, min(args.warmup, 50))
                multi["strong_scaling"] = {"envs_total": n_s, "envs_per_rank": senv.hi - senv.lo, "value": n_s * args.steps / el_s, "unit": "env-steps/s",
                                           "ms_per_step": el_s / args.steps * 1e3, "with_allgather_value": n_s * args.steps / el_sg,
                                           "with_allgather_ms_per_step": el_sg / args.steps * 1e3}
                del senv
            except Exception as e:  # noqa: BLE001
                multi["strong_scaling"] = {"error": repr(e)[:300]}

    # BASELINE.json's configs 3, 4 and 5 at their REAL shapes: the total env count sharded over the N ranks of this run (rotate 16 384; flip
    # 65 536; mix 262 144 with every flag, 5 state frames and the policy rollout), each leg wrapped: a failure costs only its own entry
    baseline_legs = None
    if not args.no_configs and not solo:
        short = max(50, min(args.steps, 300))
        baseline_legs = [sharded_config_leg(number, rank, world, dev, torch, timed, gather_floats, short) for number in (3, 4, 5) if number != args.config]
        baseline_legs.append(sharded_rollout_leg(262144, 16, rank, world, dev, torch, sync_ranks, close_interval))
        if dist:
            # ... and rank 0 alone on configs[1] while the others wait: the N = 1 number of THIS run, for agreement with the 1-GPU record
            try:
                sync_ranks()
                if rank == 0:
                    e1 = FpvBase(config.baseline_config(1, num_envs=args.envs), sim_device=str(dev), rl_device=str(dev))
                    a1 = make_actions(args.envs, 64, 1000, dev)
                    med1, _, _ = steady_windows(e1.step, a1, torch, 0.05, 3, 1000)
                    multi["n1_on_rank0"] = {"us_per_step": med1, "value": args.envs / (med1 * 1e-6), "what": "VecTask.step() at 4 096 envs on rank 0's GPU alone"}
                    del e1
                sync_ranks()
            except Exception as e:  # noqa: BLE001
                multi["n1_on_rank0"] = {"error": repr(e)[:300]}

    out = None
    if rank == 0:
        nbytes = algo_bytes(base.len_states)
        achieved = nbytes * n_local / (k_avg_us * 1e-6) / 1e9
        grid, block = base.launch_geometry()
        traffic, traffic_src = pmc_traffic(n_local)
        what = {2: "task_mode=pos, 4096 envs per GPU, rotor_response_time=0.017, delay_time=20, dt=0.001 x 10 substeps, PhysX-substeps=2, random pose/vel/target, battery on",
                3: "task_mode=rotate, 16 384 envs in total, delay_time=20, random_command", 4: "task_mode=flip, 65 536 envs in total",
                5: "task_mode=mix, 262 144 envs in total, every random_* flag, observation / rotor noise, random delay + deploy time, 5 state frames"}[args.config]
        out = {
            "metric": ("env-steps/s at 4096 envs per GPU, fpv_asymmetry.step() hot path (task_mode=pos); VecTask.step() = one taco_step launch per step"
                       if weak else f"env-steps/s of BASELINE configs[{args.config - 1}] ({task}, {n_global} envs in total) sharded over the GPUs; one taco_step launch per step and rank"),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: " + what,
                       "envs_per_gpu": n_local, "envs_total": n_global, "parallelism": f"env-sharded x{world}",
                       "api": "VecTask.step() (taco_amd.vec_env: returns the reference's (obs dict, rew, done, extras))" if world == 1 else
                              ("ShardedEnv.step_async() (taco_step on this rank's slice, which fills the gather block, + the all-gather; waited for after the next launch)"
                               if gathered_main else "ShardedEnv.step_local() = VecTask.step() on this rank's slice"),
                       "collective": ("none" if world == 1 else
                                      (f"1 {'RCCL' if backend_used == 'nccl' else backend_used} all_gather_into_tensor of the [obs|rew|done|timeout] block per step, inside the timed "
                                       "region, overlapped with the next step's launch" if gathered_main else
                                       "none in the timed region (--no-gather, or the collective could not be set up: see collective_init_error / main_leg_error)")),
                       "kernel": base.lib.taco_step_kernel_name().decode(), "kernel_form": base.kernel_form, "grid": grid, "block": block,
                       "pre_warm": pre_warm,
                       "timing": "per rank: W warm-up steps, synchronize + barrier + synchronize, t0, K steps, synchronize, t1, barrier; elapsed = MAX over ranks of t1 - t0"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_frac": (traffic / (k_avg_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "traffic_source": traffic_src, "kernel_avg_us": k_avg_us, "kernel_bracketed_median_us": k_med_us,
                         "algorithmic_bytes_per_env_step": nbytes,
                         "valu": pmc_valu(n_local, k_avg_us) if (world == 1 and n_local == 4096) else None,   # the other roof: SQ_INSTS_VALU of this launch (steady state) / 1 024 SIMDs x the mix's issue cost
                         "bound_is": "the roof SURVEY 8(d) prescribes for this path and `frac` is priced against; see `regime` for what limits THIS launch",
                         "regime": "latency" if n_local <= 16384 else "valu-issue",
                         "binds": ("at 4 096 envs per GPU neither roof is reached: instruction LATENCY of the slowest wavefront (latency_floor) + the launch boundary; "
                                   "the HBM roof prices the throughput regime (large_n: >= 60 % of it at 262 144 envs, where VALU issue is the nearer roof)"),
                         # (the bench run's table holds every instantiation: the headline's is the <block, lanes per env, ...> one with the most calls)
                         "rocprof": rocprof_row("bench_4096", f"taco_step_kernel<{block}, {4 if base.kernel_form.startswith('quad') else 1}, ", by="Calls"),
                         "note": "frac = ALGORITHMIC bytes (SURVEY 8d) / kernel time / 8 TB/s; traffic_frac = counter-measured HBM bytes / kernel time / 8 TB/s. "
                                 "4096 envs = 256 step wavefronts (4 lanes per env, one per CU) + 3 post-phase role wavefronts each: instruction-latency "
                                 "regime (SURVEY 8d): what binds is the instruction stream of the slowest wavefront (latency_floor: ~8.5 us in-kernel, "
                                 "+ ~0.7 us when some wavefront holds a resetting env, + ~1.6 us launch boundary incl. the write-back of 1.5 MB: "
                                 "profiles/r05_b_ubench_launch_boundary.txt), not HBM (3-4 %) and not VALU issue (roofline.valu: ~15 %). The throughput regime is large_n, "
                                 "where the VALU issue roof binds (large_n[].valu, binding_roof)"},
        }
        if multi:
            out.update(multi)
        if baseline_legs is not None:
            out["baseline_configs"] = baseline_legs
        if world == 1 and weak:
            try:
                out["latency_floor"] = latency_floor(base, acts, k_avg_us, torch)
            except Exception as e:  # noqa: BLE001
                out["latency_floor"] = {"error": repr(e)[:300]}
            med, ws, _ = steady_windows(env.step, acts, torch, 0.05, 5, 2000)
            out["steady_state"] = {"us_per_step": med, "env_steps_per_s": n_local / (med * 1e-6), "windows_us": ws, "api": "VecTask.step()",
                                   "protocol": "median of 5 event-bracketed windows of 2000 back-to-back steps after 50 ms of warm-up"}
            rmed, rws, _ = steady_windows(env.step_raw, acts, torch, 0.05, 5, 2000)
            out["step_raw"] = {"us_per_step": rmed, "env_steps_per_s": n_local / (rmed * 1e-6), "windows_us": rws,
                               "what": "taco_step in place on the current buffers (no return values built): the same kernel launch without VecTask.step()'s host-side dict"}
            out["step_api"] = {"us_per_step": med, "env_steps_per_s": n_local / (med * 1e-6),
                               "what": "VecTask.step() = the headline (ping-pong buffer pairs: the launch step_raw makes, no copy, no torch op)"}
            try:
                cenv = FpvBase(config.default_cfg("pos", n_local, env_clipObservations=5.0, env_clipStates=5.0), sim_device=str(dev), rl_device=str(dev))
                cmed, _, _ = steady_windows(cenv.step, acts, torch, 0.05, 5, 1000)
                out["step_api_finite_clip"] = {"us_per_step": cmed, "what": "VecTask.step() with finite clipObservations / clipStates: the kernel's OUT instantiation also writes the clamped copies"}
                del cenv
            except Exception as e:  # noqa: BLE001
                out["step_api_finite_clip"] = {"error": repr(e)[:300]}
            try:
                out["graph_replay"] = graph_entry(n_local, acts, dev, torch)
            except Exception as e:  # noqa: BLE001
                out["graph_replay"] = {"error": repr(e)[:300]}
            if not args.no_configs:
                # SURVEY 8(d)'s numbering: 3 = rotate @ 16 384, 4 = flip @ 16 384 per rank, 5 = mix @ 32 768 per rank with every flag + 5 state frames
                out["configs"] = [config_entry(2, 16384, dev, torch), config_entry(3, 16384, dev, torch), config_entry(4, 32768, dev, torch)]
                try:
                    out["configs"].append(rollout_entry(32768, 16, dev, torch))
                    out["configs"].append(dict(rollout_entry(4096, 32, dev, torch), config="5 at 4096 envs x 32 steps"))
                except Exception as e:  # noqa: BLE001
                    out["configs"].append({"config": 5, "error": repr(e)[:300]})
                torch.cuda.empty_cache()
            if not args.no_large_n:
                out["large_n"] = []
                # BASELINE configs[4]'s env count; four residency rounds; 5 state frames as stacks (VecTask.step) and on the ROLLOUT path, where the
                # replay store keeps the stacks as a frame ring and a step writes one frame (taco_rollout_io.states_newest_only)
                for big_n, ls, ring in ((262144, 1, False), (1048576, 1, False), (262144, 5, "api"), (262144, 5, False), (262144, 5, True)):
                    bcfg = config.baseline_config(1, num_envs=big_n)
                    bcfg["env"]["lenStates"] = ls
                    benv = FpvBase(bcfg, sim_device=str(dev), rl_device=str(dev), copy_outputs=(ring == "api"))
                    bacts = make_actions(big_n, 4, 7, dev)
                    step_fn = benv.step if ring == "api" else benv.step_raw
                    if ring is True:
                        frames = torch.zeros(2, big_n, 26, device=dev)
                        brew = torch.zeros(big_n, device=dev)
                        flip = [0]

                        def step_fn(a, benv=benv, frames=frames, brew=brew, flip=flip):
                            flip[0] ^= 1
                            benv.step_into(a, benv.obs_buf, benv.obs_buf, None, None, brew, None, states_newest=frames[flip[0]])
                    b_med, b_ws, b_first = steady_windows(step_fn, bacts, torch, 0.25, 5, 200)
                    bg, bb = benv.launch_geometry()
                    nb = ALGO_BYTES_PER_ENV_STEP if ring else algo_bytes(ls)
                    ach = nb * big_n / (b_med * 1e-6) / 1e9
                    tr, _ = pmc_traffic(big_n) if (ls == 1 and not ring) else (None, None)
                    valu = pmc_valu(big_n, b_med) if (ls == 1 and not ring) else None
                    out["large_n"].append({"envs": big_n, "len_states": ls,
                                           "path": {True: "rollout (replay store's frame ring: one states frame written per step)",
                                                    "api": "VecTask.step() (frame ring behind states_buf: one states frame written per step, the stack a strided view)",
                                                    False: "step_raw in place" + (" (stack materialised: 2 len - 1 frames moved per env-step)" if ls > 1 else "")}[ring],
                                           "kernel_avg_us": b_med, "env_steps_per_s": big_n / (b_med * 1e-6),
                                           "algorithmic_bytes_per_env_step": nb, "achieved_GBps": ach, "frac_of_hbm_peak": ach / HBM_PEAK_GBPS,
                                           "traffic": tr, "traffic_frac_of_hbm_peak": (tr / (b_med * 1e-6) / 1e9 / HBM_PEAK_GBPS) if tr else None,
                                           "valu": valu,
                                           "binding_roof": ("valu" if (valu and "frac" in valu and tr and valu["frac"] > tr / (b_med * 1e-6) / 1e9 / HBM_PEAK_GBPS) else
                                                            ("hbm" if (valu and "frac" in valu and tr) else None)),
                                           "rocprof": rocprof_row(str(big_n)) if (ls == 1 and not ring) else None,
                                           "kernel_form": benv.kernel_form, "grid": bg, "block": bb, "windows_us": b_ws,
                                           "first_100_launches_us": b_first,
                                           "protocol": "median of 5 windows of 200 back-to-back launches after 0.25 s of warm-up launches"})
                    del benv, bacts
                    torch.cuda.empty_cache()
            out["parity"] = parity_check(config.baseline_config(1, num_envs=n_local))
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- the result line is out; a failing teardown must not turn the run into an error
            pass


def sharded_config_leg(number, rank, world, dev, torch, timed, gather_floats, steps):
    """one BASELINE config at its real total env count, sharded over the ranks of this run: the sharded path alone, then with the per-step all-gather"""
    from taco_amd import config
    from taco_amd.dist import ShardedEnv
    task, idx, total = BASELINE_TOTALS[number]
    try:
        senv = ShardedEnv(config.baseline_config(idx, num_envs=total), rank=rank, world_size=world, device=dev, gather=False)
        n_local = senv.hi - senv.lo
        sacts = make_actions(n_local, 8, 3000 + rank, dev)
        el = timed(senv.step_gathered, sacts, steps, 30)
        k_us, _ = time_kernel_launches(senv.env, sacts, min(steps, 200), torch)
        entry = {"config": number, "task_mode": task, "envs_total": total, "envs_per_rank": n_local, "len_states": senv.env.len_states, "steps": steps,
                 "value": total * steps / el, "unit": "env-steps/s", "ms_per_step": el / steps * 1e3, "kernel_us_per_rank": gather_floats(k_us),
                 "rocprof": rocprof_row({(5, 262144): "mix_262144x5", (3, 16384): "rotate_16384"}.get((number, n_local), "none")) if rank == 0 else None,
                 "kernel_form": senv.env.kernel_form, "states_path": "frame ring (one frame per env-step)" if senv.env._ring_on else "in place",
                 # (a state stack lives in the frame ring: the algorithmic 820 B; SURVEY 8d's + 104 B per frame is for a MATERIALISED stack)
                 "frac_of_hbm_peak_per_gpu": (ALGO_BYTES_PER_ENV_STEP if senv.env._ring_on else algo_bytes(senv.env.len_states)) * n_local / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
        try:
            senv.gather = True
            el_g = timed(senv.step_gathered, sacts, steps, 10)
            entry["with_allgather"] = {"value": total * steps / el_g, "ms_per_step": el_g / steps * 1e3, "bytes_per_rank": int(senv.block.numel() * 4)}
        except Exception as e:  # noqa: BLE001
            entry["with_allgather"] = {"error": repr(e)[:300]}
        del senv
        torch.cuda.empty_cache()
        return entry
    except Exception as e:  # noqa: BLE001
        return {"config": number, "envs_total": total, "error": repr(e)[:300]}


def sharded_rollout_leg(total, horizon, rank, world, dev, torch, sync_ranks, close_interval):
    """BASELINE configs[4]'s "LSTM-critic rollout" at its real shape: every rank rolls out its slice of the 262 144 envs (actor + env step per
    step, critic batched afterwards, GAE) -- data-parallel, no collective inside; value = total env-steps / max-over-ranks time."""
    import numpy as np
    from taco_amd import config, policy as P
    from taco_amd.dist import shard_bounds
    from taco_amd.rollout import RolloutBuffer
    from taco_amd.vec_env import FpvBase
    try:
        lo, hi = shard_bounds(total, world, rank)
        n = hi - lo
        env = FpvBase(config.baseline_config(4, num_envs=total), sim_device=str(dev), rl_device=str(dev), copy_outputs=False, env_offset=lo, num_envs_local=n)
        buf = RolloutBuffer(n, 26, 1, 26, env.len_states, 4, horizon, 4, 0.99, 0.95, str(dev))
        pol = P.ActorCritic(documented_policy(np.random.default_rng(0)), 1, env.len_states, device=str(dev))

        def run():
            buf.reset()
            buf.compute_returns_and_advantage(buf.run(env, pol))

        for _ in range(2):
            run()
        sync_ranks()
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            run()
        el = close_interval(t0) / reps
        del buf, env, pol
        torch.cuda.empty_cache()
        return {"config": 5, "what": "taco_rollout_run + GAE on every rank's slice (policy forward included), data-parallel", "envs_total": total, "envs_per_rank": n,
                "horizon": horizon, "ms_per_rollout": el * 1e3, "value": total * horizon / el, "unit": "env-steps/s"}
    except Exception as e:  # noqa: BLE001
        return {"config": 5, "what": "rollout", "envs_total": total, "error": repr(e)[:300]}


if __name__ == "__main__":
    main()
