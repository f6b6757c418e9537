import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config, _lib
from taco_amd.vec_env import FpvBase
n = 4096
env = FpvBase(config.baseline_config(1, num_envs=n), copy_outputs=False)
a = (0.3 * torch.randn((n, 4)) + torch.tensor([-0.45, 0, 0, 0])).clamp(-1, 1).cuda()
for _ in range(30): env.step_raw(a)
st = torch.zeros(16, dtype=torch.int64, device="cuda")
_lib.check(env.lib.taco_bind_phase_stamps(env._h, st.data_ptr()))
acc = torch.zeros(16, dtype=torch.float64)
for _ in range(50):
    env.step_raw(a); acc += st.cpu().double()
acc /= 50
names = [("loads", 0, 1), ("pre", 1, 2), ("substeps", 2, 3), ("sandwich+state stores+queue pop", 3, 8), ("rel quantities + frame", 8, 9), ("put_frame states", 9, 10),
         ("noise + put_frame obs", 10, 4), ("reward", 4, 11), ("outputs", 11, 5)]
print({k: round(float(acc[b] - acc[a_])) for k, a_, b in names})
