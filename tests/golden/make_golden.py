#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own
importable torch sub-models on CPU.

    python tests/golden/make_golden.py [/root/reference]

Only numbers (inputs + the reference's outputs) are written; no reference source is
copied.  The reference files are loaded by path under empty in-memory parent packages
(`isaacgym`, `isaacgymenvs`, `isaacgymenvs.utils`) so that their
`from isaacgym.torch_utils import ...` lines resolve to the reference's own files.  Two
files hard-code device 'cuda:0' (control/fpv_dynamics.py:33, control/task_reward.py:61,77);
for those the module text is exec'd from memory with that literal replaced by 'cpu'.

`fpv_asymmetry.py` itself cannot be imported (needs the Isaac Gym binary), so the glue
between sub-models in the "chain" / "obs" fixtures is restated here in this script's own
words, calling the reference's functions for all arithmetic the reference delegates to them
(fpv_asymmetry.py:334-360 refresh_state, :362-372 mid_physics_step, :608-650 control,
:390-421 observation pack).

This script runs only in the build container; the GPU box never sees /root/reference.
"""
import sys, types, importlib.util, linecache, pathlib, math
import numpy as np
import torch

REF = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "/root/reference")
OUT = pathlib.Path(__file__).resolve().parent
CTRL = REF / "IsaacGymEnvs/isaacgymenvs/tasks/control"

torch.set_num_threads(1)
torch.manual_seed(0)


def _load(name, path, device_patch=False):
    path = str(path)
    if not device_patch:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    src = open(path).read().replace("'cuda:0'", "'cpu'")
    fn = f"<patched:{name}>"
    linecache.cache[fn] = (len(src), None, src.splitlines(True), fn)
    mod = types.ModuleType(name)
    mod.__file__ = fn
    sys.modules[name] = mod
    exec(compile(src, fn, "exec"), mod.__dict__)
    return mod


for n in ("isaacgym", "isaacgymenvs", "isaacgymenvs.utils"):
    m = types.ModuleType(n)
    m.__path__ = []
    sys.modules[n] = m
TU = _load("isaacgym.torch_utils", REF / "python/isaacgym/torch_utils.py")
TJ = _load("isaacgymenvs.utils.torch_jit_utils", REF / "IsaacGymEnvs/isaacgymenvs/utils/torch_jit_utils.py")
BAT = _load("ref_battery", CTRL / "battery_dynamics.py")
THR = _load("ref_thrust", CTRL / "thrust_dynamics.py")
PID = _load("ref_pid", CTRL / "angvel_control.py")
REW = _load("ref_reward", CTRL / "task_reward.py", device_patch=True)
ALLOC = _load("ref_alloc", CTRL / "fpv_dynamics.py", device_patch=True)

f32 = torch.float32


def npy(t):
    return t.detach().cpu().numpy().copy()


def rand_unit_quat(g, n):
    q = torch.randn(n, 4, generator=g, dtype=f32)
    return q / q.norm(dim=1, keepdim=True)


def save(name, **arrs):
    np.savez_compressed(OUT / f"{name}.npz", **arrs)
    print(f"{name}.npz:", {k: v.shape for k, v in arrs.items()})


# --------------------------------------------------------------------------- (1) quaternion helpers
def gen_quat():
    g = torch.Generator().manual_seed(100)
    n = 64 * 3
    q = rand_unit_quat(g, n)
    p = rand_unit_quat(g, n)
    # edge rows: identity, -identity, gimbal +-90deg pitch (|sinp| >= 1), pure-roll, q and -q pairs
    s = math.sqrt(0.5)
    edge = torch.tensor([
        [0, 0, 0, 1], [0, 0, 0, -1], [0, s, 0, s], [0, -s, 0, s], [s, 0, 0, s], [0, 0, s, s],
        [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0.5, 0.5, 0.5, 0.5], [-0.5, 0.5, -0.5, 0.5],
        [0.0, 0.70710678, 0.0, 0.70710678], [0.0, 0.70710683, 0.0, 0.70710683],
    ], dtype=f32)
    q = torch.cat([q, edge, -edge], 0)
    p = torch.cat([p, edge.flip(0), edge], 0)
    v = torch.randn(q.shape[0], 3, generator=g, dtype=f32) * 3
    e = (torch.rand(q.shape[0], 3, generator=g, dtype=f32) * 2 - 1) * math.pi
    e[-5:] = torch.tensor([[0, 0, 0], [math.pi, 0, 0], [0, math.pi / 2, 0], [0.3, -0.2, 1.0], [-math.pi, math.pi, -math.pi]], dtype=f32)
    r, pch, y = TU.get_euler_xyz_v1(q)
    save("quat",
         q=npy(q), p=npy(p), v=npy(v), e=npy(e),
         mul=npy(TU.quat_mul(q, p)),
         conj=npy(TU.quat_conjugate(q)),
         rot_inv=npy(TU.quat_rotate(TU.quat_conjugate(q), v)),
         rel=npy(TU.quat_mul(TU.quat_conjugate(q), p)),
         rpy=npy(torch.stack([r, pch, y], -1)),
         from_euler=npy(TU.quat_from_euler_xyz(e[:, 0], e[:, 1], e[:, 2])),
         diff_rad=npy(TJ.quat_diff_rad(q, p)),
         mat=npy(TJ.quaternion_to_matrix(q).reshape(-1, 9)))


# --------------------------------------------------------------------------- (2) rate PID
def gen_pid():
    g = torch.Generator().manual_seed(200)
    n, T = 64, 50
    dt = 0.001
    c = PID.angvel_control(1, 0, n, "cpu", dt)
    des = torch.randn(T, n, 3, generator=g, dtype=f32) * 10
    cur = torch.randn(T, n, 3, generator=g, dtype=f32) * 8
    des[:, 0:4] *= 60          # saturate error clip (400) and D clip (150)
    des[3, 8:12] = cur[3, 8:12]  # exact zero error -> previous_error re-seeded on the next call
    des[4, 8:12] = cur[4, 8:12]
    des[:, 12:16] = des[0:1, 12:16]  # constant command, slowly varying rate
    cur[:, 12:16] = cur[0:1, 12:16] + 0.01 * torch.arange(T, dtype=f32).view(T, 1, 1)
    out = torch.zeros(T, n, 3)
    prev = torch.zeros(T, n, 3)
    integ = torch.zeros(T, n, 3)
    for t in range(T):
        if t == 25:   # mid-sequence reset of some envs (angvel_control.py:90-94)
            c.reset(torch.tensor([1, 5, 9, 13]))
        out[t] = c.compute(des[t], cur[t])
        prev[t] = c.previous_error
        integ[t] = c.integral
    save("pid", dt=np.float64(dt), des=npy(des), cur=npy(cur), out=npy(out), prev=npy(prev), integ=npy(integ),
         reset_step=np.int64(25), reset_ids=np.array([1, 5, 9, 13]))


# --------------------------------------------------------------------------- (3) allocator, (7) real->sim
def gen_alloc():
    g = torch.Generator().manual_seed(300)
    a = ALLOC.FpvDynamicsReal2Sim()
    n = 64 * 3
    u = torch.zeros(n, 4, dtype=f32)
    u[:, 0] = torch.rand(n, generator=g) * 1000
    u[:, 1:] = torch.randn(n, 3, generator=g) * 150
    u[0:8, 0] = torch.tensor([0, 1e-3, 1, 50, 1000, 999.9, 1200, -5.0])
    u[8:16, 1:] *= 10
    u[16] = torch.tensor([500., 10, -20, 400])
    u[17] = torch.tensor([900., 300, 300, 300])
    u[18] = torch.tensor([50., 0, 0, 0])
    u[19] = torch.tensor([1000., 0, 0, 0])
    u_in = u.clone()
    f = a.control_allocator(u)          # clips u[:,3] in place
    forces = torch.rand(n, 4, generator=g, dtype=f32) * 5
    torques = torch.rand(n, 4, generator=g, dtype=f32)
    forces[0] = torch.tensor([0., 1, 2, 3]); torques[0] = torch.tensor([10., 11, 12, 13])
    fi, ti = forces.clone(), torques.clone()
    fs, ts = a.sim_process(forces, torques)
    save("alloc", u=npy(u_in), u_after=npy(u), thr=npy(f), f_real=npy(fi), t_real=npy(ti), f_sim=npy(fs), t_sim=npy(ts))


# --------------------------------------------------------------------------- (4) battery
def gen_battery():
    T = 2000
    pm_levels = [0.0, 50.0, 300.0, 1500.0, 2500.0, 4000.0, 30000.0]   # last ones drive the radicand negative -> NaN
    n = len(pm_levels) * 2
    dt = 0.001
    b = BAT.Battery_Dynamics(n, "cpu", True, dt)
    b.reset(torch.arange(n), False)
    E0 = torch.zeros(n, 1)
    E0[len(pm_levels):, 0] = torch.tensor([0.3, 0.9, 1.5, 2.1, 2.2, 0.05, 1.0])
    b.E_c[:] = E0
    g = torch.Generator().manual_seed(400)
    base = torch.tensor(pm_levels * 2, dtype=f32).view(n, 1)
    V = torch.zeros(T, n)
    Pm = torch.zeros(T, n)
    for t in range(T):
        pm = base * (1 + 0.1 * torch.rand(n, 1, generator=g))
        Pm[t] = pm[:, 0]
        V[t] = b.sim_process(pm)[:, 0]
    off = BAT.Battery_Dynamics(4, "cpu", False, dt)
    v_off = off.sim_process(torch.ones(4, 1))
    keep = np.r_[0:20, 100:110, 990:1000, 1990:2000]
    save("battery", dt=np.float64(dt), E0=npy(E0[:, 0]), Pm=npy(Pm), V_keep_idx=keep, V=npy(V)[keep],
         E_end=npy(b.E_c[:, 0]), u1_end=npy(b.u_1[:, 0]), t_end=npy(b.time[:, 0]), V_off=npy(v_off[:, 0]))


# --------------------------------------------------------------------------- (5) rotor first-order lag
def gen_rotor():
    taus = [0.016, 0.017, 0.018, 0.001]
    volts = [22.0, 24.2, 26.1]
    scales = [1.0, 0.95, 1.05]
    cases = [(tau, v, s) for tau in taus for v in volts for s in scales]
    n, T = len(cases), 500
    r = THR.RotorDynamics(n, "cpu", 0.017)
    r.reset(torch.arange(n), 0, False, False, True, False, False, False, False)
    g = torch.Generator().manual_seed(500)
    tau = torch.tensor([c[0] for c in cases], dtype=f32).view(n, 1) * (1 + 0.01 * torch.randn(n, 4, generator=g))
    tau[[c[0] == 0.001 for c in cases]] = 0.001
    r.response_time[:] = tau
    r.omega_para[:] = r.omega_para_init * torch.tensor([c[2] for c in cases], dtype=f32).view(n, 1) * (1 + 0.02 * torch.rand(n, 5, generator=g))
    V = torch.tensor([c[1] for c in cases], dtype=f32).view(n, 1)
    thr = torch.zeros(T, n, 4, dtype=f32)
    thr[:] = 100
    thr[50:] = 500                                   # step
    thr[250:] = torch.rand(T - 250, n, 4, generator=g) * 900 + 100   # then random throttle
    om = torch.rand(n, 4, generator=g) * 400
    om0 = om.clone()
    omega = torch.zeros(T, n, 4)
    for t in range(T):
        om = r.sim_process(V + 0.001 * t, thr[t], om)
        omega[t] = om
    keep = np.r_[0:12, 48:62, 245:260, 490:500]
    save("rotor", tau=npy(tau), para=npy(r.omega_para), V0=npy(V[:, 0]), thr=npy(thr), om0=npy(om0),
         keep_idx=keep, omega=npy(omega)[keep])


# --------------------------------------------------------------------------- (6) aero
def gen_aero():
    g = torch.Generator().manual_seed(600)
    n = 64 * 3
    a = THR.AeroDynamics(n, "cpu")
    a.para_force_torque[:] = a.para_force_torque_init * (1 + 0.05 * (2 * torch.rand(n, 2, generator=g) - 1))
    a.para_d[:] = a.para_d_init * (1 + 0.05 * (2 * torch.rand(n, 2, generator=g) - 1))
    a.para_t[:] = a.para_t_init * (1 + 0.05 * (2 * torch.rand(n, 1, generator=g) - 1))
    a.para_force_torque[0] = a.para_force_torque_init[0]; a.para_d[0] = a.para_d_init[0]; a.para_t[0] = a.para_t_init[0]
    vb = torch.randn(n, 3, generator=g, dtype=f32) * 4
    om = torch.rand(n, 4, generator=g, dtype=f32) * 800
    vb[0] = torch.tensor([3., -4, 1]); om[0] = 316.0
    vb[1] = 0
    rf, rt, bf, bt = a.sim_process(vb, om)
    save("aero", cf_ct=npy(a.para_force_torque), d=npy(a.para_d), kt=npy(a.para_t[:, 0]), vb=npy(vb), om=npy(om),
         rf=npy(rf), rt=npy(rt), bf=npy(bf), bt=npy(bt))


# --------------------------------------------------------------------------- (8) rewards + done flags
def gen_reward():
    g = torch.Generator().manual_seed(800)
    n = 64 * 3
    max_len = 1000.0
    rel_pos = torch.randn(n, 3, generator=g, dtype=f32) * 2
    rel_pos_b = torch.randn(n, 3, generator=g, dtype=f32) * 2
    rel_v = torch.randn(n, 3, generator=g, dtype=f32) * 3
    pos = torch.randn(n, 3, generator=g, dtype=f32) * 2 + torch.tensor([0, 0, 2.5])
    q = rand_unit_quat(g, n)
    qt = rand_unit_quat(g, n)
    relq = TU.quat_mul(TU.quat_conjugate(q), qt)
    cmd = torch.zeros(n, 2, dtype=f32)
    cmd[:, 1] = (torch.rand(n, generator=g) * 2 - 1) * 6
    prog = torch.randint(0, 1000, (n,), generator=g)
    # boundary flags
    eps = 1e-6
    pos[0:6, 2] = torch.tensor([0.1, 0.1 - eps, 0.1 + eps, 0.0999999, 0.09, -1.0])
    rel_pos_b[6:12] = torch.tensor([[10, 0, 0], [10 + 1e-5, 0, 0], [10 - 1e-5, 0, 0], [6, 8, 0], [6, 8, 1e-3], [0, 0, 0]], dtype=f32)
    rel_pos[6:12] = torch.tensor([[11.2, 0, 0], [11.2 + 1e-5, 0, 0], [8, 0, 6], [0, 0, 0], [1.2, 0, 0], [0, 1.2, 10]], dtype=f32)
    prog[12:18] = torch.tensor([997, 998, 999, 1000, 500, 0])
    pos[12:18, 2] = torch.tensor([2.0, 2.0, 2.0, 2.0, 0.05, 0.05])
    q[18] = torch.tensor([0., 0, 0, 1]); qt[18] = torch.tensor([0., 0, 0, 1])
    q[19] = torch.tensor([0., 0, 0, 1]); qt[19] = torch.tensor([0., 0, 0, -1])
    rb = torch.zeros(n, dtype=torch.long)
    r_pos, d_pos = REW.compute_pos_reward(rel_pos_b, pos, q, qt, rb, prog, max_len)
    r_rot, d_rot = REW.compute_rotating_reward(rel_pos, rel_v, pos, q, cmd, rb, prog, max_len)
    r_flip, d_flip = REW.compute_flip_reward(rel_pos_b, relq, pos, cmd, rb, prog, max_len)
    save("reward", max_len=np.float64(max_len), rel_pos=npy(rel_pos), rel_pos_b=npy(rel_pos_b), rel_v=npy(rel_v), pos=npy(pos),
         q=npy(q), qt=npy(qt), relq=npy(relq), cmd=npy(cmd), prog=npy(prog),
         r_pos=npy(r_pos), d_pos=npy(d_pos), r_rot=npy(r_rot), d_rot=npy(d_rot), r_flip=npy(r_flip), d_flip=npy(d_flip))


# --------------------------------------------------------------------------- (9) composed substep chain C -> H''
def gen_chain():
    """Restated glue of mid_physics_step (fpv_asymmetry.py:334-372, 608-650) driving the reference's
    sub-models for T substeps over a PRESCRIBED rigid-body trajectory (row I is not in the reference)."""
    g = torch.Generator().manual_seed(900)
    n, T = 16, 120
    dt = 0.001
    pid = PID.angvel_control(1, 0, n, "cpu", dt)
    bat = BAT.Battery_Dynamics(n, "cpu", True, dt)
    rot = THR.RotorDynamics(n, "cpu", 0.017)
    aero = THR.AeroDynamics(n, "cpu")
    r2s = ALLOC.FpvDynamicsReal2Sim()
    ids = torch.arange(n)
    pid.reset(ids); bat.reset(ids, False)
    rot.reset(ids, 1.0, False, True, True, False, False, False, False)
    bat.E_c[:] = torch.rand(n, 1, generator=g) * 2.2
    rot.response_time[:] = 0.017 + (torch.rand(n, 4, generator=g) * 2 - 1) * 0.001
    rot.omega_para[:] = rot.omega_para_init * (1 + 0.05 * (2 * torch.rand(n, 5, generator=g) - 1))
    aero.para_force_torque[:] = aero.para_force_torque_init * (1 + 0.05 * (2 * torch.rand(n, 2, generator=g) - 1))
    aero.para_d[:] = aero.para_d_init * (1 + 0.05 * (2 * torch.rand(n, 2, generator=g) - 1))
    aero.para_t[:] = aero.para_t_init * (1 + 0.05 * (2 * torch.rand(n, 1, generator=g) - 1))
    rotor_speed = torch.rand(n, 4, generator=g) * 400
    om0 = rotor_speed.clone()
    E0 = bat.E_c.clone()

    # prescribed trajectory: attitude spins fast enough to cross the +-pi euler wrap and the gimbal region
    q = rand_unit_quat(g, n)
    w_world = torch.randn(n, 3, generator=g) * 25
    v_world = torch.randn(n, 3, generator=g) * 4
    qs, vs, ws, acts = [], [], [], []
    for t in range(T):
        qs.append(q.clone()); vs.append(v_world.clone()); ws.append(w_world.clone())
        dq = TU.quat_mul(torch.cat([w_world * dt * 0.5, torch.ones(n, 1)], 1), q)
        q = dq / dq.norm(dim=1, keepdim=True)
        w_world = w_world + torch.randn(n, 3, generator=g) * 0.5
        v_world = v_world + torch.randn(n, 3, generator=g) * 0.05
        a = torch.randn(n, 4, generator=g) * 0.4
        a[:, 0] -= 0.3
        if t % 10 != 0 and acts:
            a = acts[-1]
        acts.append(a.clamp(-1, 1))
    qs, vs, ws, acts = (torch.stack(x) for x in (qs, vs, ws, acts))

    r0, p0, y0 = TU.get_euler_xyz_v1(qs[0])
    rpy_old = torch.stack([r0, p0, y0], -1)
    rpy_cont = rpy_old.clone()
    rec = {k: [] for k in ("rpy", "rpy_cont", "vb", "wb", "u", "thr", "V", "om", "f_sim", "t_sim", "body_f")}
    for t in range(T):
        q, v, w = qs[t], vs[t], ws[t]
        r, p, y = TU.get_euler_xyz_v1(q)
        rpy = torch.stack([r, p, y], -1)
        d = rpy - rpy_old
        d = torch.where(d > 1, d - 2 * np.pi, d)
        d = torch.where(d < -1, d + 2 * np.pi, d)
        rpy_cont = rpy_cont + d
        rpy_old = rpy.clone()
        vb = TU.quat_rotate(TU.quat_conjugate(q), v)
        wb = TU.quat_rotate(TU.quat_conjugate(q), w)
        a = acts[t]
        temp = torch.zeros(n, 4, dtype=f32)
        temp[:, 0] = (a[:, 0] + 1) / 2 * 1000
        temp[:, 1:] = a[:, 1:] * 20
        u = torch.zeros(n, 4, dtype=f32)
        u[:, 0] = temp[:, 0]
        u[:, 1:] = pid.compute(temp[:, 1:], wb)
        thr = r2s.control_allocator(u)
        P_m = torch.sum(400 * (rotor_speed * 2 * torch.pi / 4500) ** 3, dim=1).unsqueeze(1)
        V = bat.sim_process(P_m)
        rotor_speed = rot.sim_process(V, thr, rotor_speed)
        rf, rt, bf, bt = aero.sim_process(vb, rotor_speed)
        fs, ts = r2s.sim_process(rf, rt)
        for k, val in (("rpy", rpy), ("rpy_cont", rpy_cont), ("vb", vb), ("wb", wb), ("u", u), ("thr", thr), ("V", V[:, 0]),
                       ("om", rotor_speed), ("f_sim", fs), ("t_sim", ts), ("body_f", bf)):
            rec[k].append(npy(val))
    save("chain", dt=np.float64(dt), q=npy(qs), v=npy(vs), w=npy(ws), act=npy(acts), om0=npy(om0), E0=npy(E0[:, 0]),
         tau=npy(rot.response_time), para=npy(rot.omega_para), cf_ct=npy(aero.para_force_torque), d=npy(aero.para_d),
         kt=npy(aero.para_t[:, 0]), E_end=npy(bat.E_c[:, 0]), u1_end=npy(bat.u_1[:, 0]), t_end=npy(bat.time[:, 0]),
         **{k: np.stack(v) for k, v in rec.items()})


# --------------------------------------------------------------------------- (10) observation pack (row O), noise-free
def gen_obs():
    """Restated glue of refresh_state's relative quantities (fpv_asymmetry.py:354-360) and the
    noise-free 26-D frame (fpv_asymmetry.py:415-421 + task tails :713-714, :768-771, :831-838)."""
    g = torch.Generator().manual_seed(1000)
    n = 64 * 3
    p = torch.randn(n, 3, generator=g) * 2 + torch.tensor([0, 0, 2.5])
    p[0:6, 2] = torch.tensor([0.0, 0.5, 0.25, -0.3, 0.7, 0.49999])
    q = rand_unit_quat(g, n)
    v = torch.randn(n, 3, generator=g) * 3
    w = torch.randn(n, 3, generator=g) * 6
    pt = torch.randn(n, 3, generator=g) * 2 + torch.tensor([0, 0, 3.0])
    yaw = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    z = torch.zeros(n)
    qt = TU.quat_from_euler_xyz(z, z, yaw)
    V = 22 + torch.rand(n, generator=g) * 4.1
    act = (torch.rand(n, 4, generator=g) * 2 - 1)
    cmd = torch.stack([torch.ones(n), (torch.rand(n, generator=g) * 2 - 1) * 6], -1)
    flip_radian = 2 * math.pi * torch.randint(-3, 4, (n,), generator=g).float()
    roll_cont = torch.randn(n, generator=g) * 6

    cq = TU.quat_conjugate(q)
    rel_pos = pt - p
    rel_pos_b = TU.quat_rotate(cq, rel_pos)
    rel_q_b = TU.quat_mul(cq, qt)
    rel_v = torch.zeros_like(v) - v
    rel_w = torch.zeros_like(w) - w
    rel_v_b = TU.quat_rotate(cq, rel_v)
    rel_w_b = TU.quat_rotate(cq, rel_w)
    fr = torch.zeros(n, 26)
    fr[:, 0:3] = rel_pos_b / 3
    fr[:, 3:12] = TJ.quaternion_to_matrix(rel_q_b).reshape(n, 9)
    fr[:, 12:15] = rel_v_b / 2
    fr[:, 15:18] = rel_w_b / torch.pi
    fr[:, 18] = (V - 23) / 3
    fr[:, 19:23] = act
    fr[:, 23] = 4 * torch.clamp(p[:, 2], 0, 0.5) - 1
    f_pos = fr.clone(); f_pos[:, -2:] = 0
    f_rot = fr.clone(); f_rot[:, -2] = cmd[:, 0]; f_rot[:, -1] = cmd[:, 1] / 6
    c1 = torch.clamp(flip_radian - roll_cont, min=-2 * torch.pi, max=2 * torch.pi)
    f_flip = fr.clone(); f_flip[:, -2] = -1; f_flip[:, -1] = c1 / 2 / torch.pi
    save("obs", p=npy(p), q=npy(q), v=npy(v), w=npy(w), pt=npy(pt), qt=npy(qt), V=npy(V), act=npy(act), cmd=npy(cmd),
         flip_radian=npy(flip_radian), roll_cont=npy(roll_cont), rel_pos_b=npy(rel_pos_b), rel_q_b=npy(rel_q_b),
         rel_v_b=npy(rel_v_b), rel_w_b=npy(rel_w_b), flip_cmd=npy(c1), f_pos=npy(f_pos), f_rot=npy(f_rot), f_flip=npy(f_flip))


# --------------------------------------------------------------------------- (11) reset helpers given the uniforms
def gen_reset():
    """torch_rand_float's affine map and the euler->quat used by rand_quat / reset_target_idx, evaluated by the
    reference functions on KNOWN uniforms (the global torch generator is re-seeded to replay them), so the restatement's
    own counter-based generator can be checked for identical post-processing (torch_utils.py:199-219;
    fpv_asymmetry.py:698-704, 523-548)."""
    n = 64
    outs = {}
    cols = []
    for col, (name, lo, hi) in enumerate((("pm2", -2.0, 2.0), ("pmpi", -math.pi, math.pi), ("u0_400", 0.0, 400.0),
                                          ("u0_2p2", 0.0, 2.2), ("pm6", -6.0, 6.0), ("dr095", 1 - 0.05 * 0.7, 1 + 0.05 * 0.7),
                                          ("tau", 0.017 - 0.001, 0.017 + 0.001), ("noise", 1 - 10 / 700, 1 + 10 / 700))):
        # torch_rand_float is TorchScript (calls aten::rand itself): replay the uniforms by re-seeding the global generator
        torch.manual_seed(1100 + col)
        cols.append(torch.rand(n, 1))
        torch.manual_seed(1100 + col)
        outs[name] = npy(TU.torch_rand_float(lo, hi, (n, 1), "cpu")[:, 0])
    u = torch.cat(cols, 1)
    e = (u[:, 0:3] * 2 - 1) * math.pi
    outs["quat_xyz"] = npy(TU.quat_from_euler_xyz(e[:, 0], e[:, 1], e[:, 2]))
    z = torch.zeros(n)
    outs["quat_roll_only"] = npy(TU.quat_from_euler_xyz(e[:, 0], z, z))
    outs["quat_yaw_only"] = npy(TU.quat_from_euler_xyz(z, z, e[:, 2]))
    save("reset", u=npy(u), e=npy(e), **outs)


# --------------------------------------------------------------------------- (12) replay buffer: GAE + advantage normalisation
def gen_gae():
    """PPOReplayBuffer.store + compute_returns_and_advantage (algorithms/buffer_asymmetry.py:49-68, 93-132) run on CPU."""
    _load("nets_asymmetry", REF / "IsaacGymEnvs/algorithms/nets_asymmetry.py")  # the buffer does `import nets_asymmetry as core`
    BUF = _load("ref_buffer", REF / "IsaacGymEnvs/algorithms/buffer_asymmetry.py")
    g = torch.Generator().manual_seed(1200)
    H, N = 24, 96
    gamma, lam = 0.99, 0.95
    buf = BUF.PPOReplayBuffer(N, 26, 1, 26, 5, 4, H, 4, gamma, lam, "cpu")
    rew = torch.rand(H, N, generator=g) * 0.02
    done = (torch.rand(H, N, generator=g) < 0.08).long()
    val = torch.randn(H, N, 1, generator=g) * 0.3
    last = torch.randn(N, 1, generator=g) * 0.3
    z4 = torch.zeros(N, 4)
    for t in range(H):
        buf.store(torch.zeros(N, 1, 26), torch.zeros(N, 5, 26), z4, rew[t], torch.zeros(N), done[t], val[t], z4, z4)
    buf.compute_returns_and_advantage(last)
    # un-normalised advantage = ret - value
    save("gae", gamma=np.float64(gamma), lam=np.float64(lam), rew=npy(rew), done=npy(done), value=npy(val[:, :, 0]), last_value=npy(last[:, 0]),
         ret=npy(buf.ret_buf[:, :, 0]), adv_norm=npy(buf.adv_buf[:, :, 0]), adv_raw=npy(buf.ret_buf[:, :, 0] - buf.value_buf[:, :, 0]),
         done_f32=npy(buf.done_buf[:, :, 0]), rew_buf=npy(buf.rew_buf[:, :, 0]))


# --------------------------------------------------------------------------- (13) policy forward: PPO_ActorCritic.act / evaluate
def gen_policy():
    """PPO_ActorCritic (algorithms/nets_asymmetry.py:270-378) in the documented training configuration (README.md:60-66:
    actor = MLP on obs, critic = 1-layer LSTM over the 5-frame state stack + MLP), run on CPU.  Small widths keep the fixture small;
    the weights are whatever the reference's own initialisation produced under the seed below (they are data)."""
    import torch.nn as nn
    NETS = _load("nets_asymmetry", REF / "IsaacGymEnvs/algorithms/nets_asymmetry.py")
    torch.manual_seed(1300)
    para = {
        "actor_critic_mlp_dict": {"actor_input_dim": 26, "actor_output_dim": 4, "critic_input_dim": 26 * 5, "critic_output_dim": 1,
                                  "actor_hidden_sizes": [64, 40], "critic_hidden_sizes": [48], "activation": nn.ReLU},
        "use_actor_encoder": False, "use_critic_encoder": True, "share_encoder": False, "critic_encoder_type": "LSTM",
        "critic_encoder_dict": {"encoder_type": "LSTM", "input_size": 26, "output_size": 24, "num_layers": 1, "bidirectional": False},
    }
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        agent = NETS.PPO_ActorCritic(para)
    with torch.no_grad():
        agent.log_std.copy_(torch.tensor([-0.5, -0.2, 0.1, 0.3]))
        # the reference's para_init leaves the output layers tiny; scale the weights up so that every nonlinearity is exercised
        for p_ in agent.parameters():
            if p_.dim() >= 2:
                p_.mul_(3.0)
    g = torch.Generator().manual_seed(1301)
    N = 96
    obs = torch.randn(N, 1, 26, generator=g)
    states = torch.randn(N, 5, 26, generator=g)
    eps = torch.randn(N, 4, generator=g)
    with torch.no_grad():
        action, logp_det, value, mu, sigma = agent.act(obs, states, deterministic=True)
        scale = agent.log_std.exp() * agent.log_std.exp()
        action_s = mu + scale * eps
        logp_s, entropy, value_e, mu_e, sigma_e = agent.evaluate(obs, states, action_s)
        fwd = agent.forward(obs)
    assert torch.equal(mu, mu_e) and torch.equal(value, value_e) and torch.equal(fwd, mu)
    sd = {"sd." + k: npy(v) for k, v in agent.state_dict().items()}
    save("policy", obs=npy(obs), states=npy(states), eps=npy(eps), mu=npy(mu), value=npy(value[:, 0]), sigma=npy(sigma), logp_det=npy(logp_det),
         action_s=npy(action_s), logp_s=npy(logp_s), actor_hidden=np.array([64, 40]), critic_hidden=np.array([48]), lstm_hidden=np.array(24), **sd)


def gen_policy_documented():
    """The same module at the DOCUMENTED widths (README.md:60-66: actor MLP 26-128-128-128-4, critic LSTM 26 -> 128 over 5 frames + MLP
    128-128-128-1) -- the architecture the batched critic kernels (taco_critic_lstm_kernel / taco_critic_mlp_kernel) are specialised for."""
    import torch.nn as nn
    NETS = _load("nets_asymmetry", REF / "IsaacGymEnvs/algorithms/nets_asymmetry.py")
    torch.manual_seed(1400)
    para = {
        "actor_critic_mlp_dict": {"actor_input_dim": 26, "actor_output_dim": 4, "critic_input_dim": 26 * 5, "critic_output_dim": 1,
                                  "actor_hidden_sizes": [128, 128, 128], "critic_hidden_sizes": [128, 128], "activation": nn.ReLU},
        "use_actor_encoder": False, "use_critic_encoder": True, "share_encoder": False, "critic_encoder_type": "LSTM",
        "critic_encoder_dict": {"encoder_type": "LSTM", "input_size": 26, "output_size": 128, "num_layers": 1, "bidirectional": False},
    }
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        agent = NETS.PPO_ActorCritic(para)
    with torch.no_grad():
        agent.log_std.copy_(torch.tensor([-0.4, -0.1, 0.0, 0.2]))
        for p_ in agent.parameters():
            if p_.dim() >= 2:
                p_.mul_(4.0)
    g = torch.Generator().manual_seed(1401)
    N = 100
    obs = torch.randn(N, 1, 26, generator=g)
    states = torch.randn(N, 5, 26, generator=g)
    states[7] = 0.0                      # an all-zero stack
    states[11, :3] = 0.0                 # a stack whose first frames are zero (a freshly reset env)
    with torch.no_grad():
        action, logp_det, value, mu, sigma = agent.act(obs, states, deterministic=True)
    sd = {"sd." + k: npy(v) for k, v in agent.state_dict().items()}
    save("policy_documented", obs=npy(obs), states=npy(states), mu=npy(mu), value=npy(value[:, 0]), sigma=npy(sigma), logp_det=npy(logp_det), **sd)


if __name__ == "__main__":
    if "--only-policy-documented" in sys.argv:
        gen_policy_documented()
        sys.exit(0)
    gen_gae()
    gen_policy()
    gen_policy_documented()
    if "--only-gae" in sys.argv or "--only-next" in sys.argv:
        sys.exit(0)
    gen_quat(); gen_pid(); gen_alloc(); gen_battery(); gen_rotor(); gen_aero(); gen_reward(); gen_chain(); gen_obs(); gen_reset()
