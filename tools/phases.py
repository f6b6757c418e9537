"""Phase breakdown of one step launch (shader-clock ticks of workgroup 0, see taco_bind_phase_stamps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
for n in [int(x) for x in sys.argv[1:]] or (4096,):
    env = FpvBase(config.baseline_config(1, num_envs=n), copy_outputs=False)
    a = (0.3 * torch.randn((n, 4)) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
    for _ in range(30): env.step_raw(a)
    ph = env.phase_stamps(a, 50)
    # calibrate ticks: time a long run of steps with events and compare with the stamped total of the same launches
    tot = sum(ph)
    print(f"N={n} geometry={env.launch_geometry()} ticks: loads={ph[0]:.0f} pre={ph[1]:.0f} substeps={ph[2]:.0f} stores+frames={ph[3]:.0f} reward+out={ph[4]:.0f} total={tot:.0f}", flush=True)
    pb = env.phase_stamps(a, 200, back_to_back=True)
    print(f"   back to back (last of 200 launches): loads={pb[0]:.0f} pre={pb[1]:.0f} substeps={pb[2]:.0f} stores+frames={pb[3]:.0f} reward+out={pb[4]:.0f} total={sum(pb):.0f}", flush=True)
    t = env._last_stamps.cpu().tolist()
    if t[6]:
        print(f"   role wavefronts end (ticks after the substeps' end, last launch): reward {t[6] - t[3]}  obs {t[7] - t[3]}  states {t[8] - t[3]}; wavefront 0 {t[5] - t[3]}", flush=True)
