import sys; sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np, torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
from oracle import oracle as O
cfg=config.baseline_config(0); flat=config.flat_cfg(cfg)
env=FpvBase(cfg,copy_outputs=False); orc=O.OracleEnv(flat)
g0=env.get_state().cpu().numpy(); o0=orc.get_state().view(np.float32)
print("init mismatch rows:", np.unique(np.argwhere(g0.view(np.uint32)!=o0.view(np.uint32))[:,0]))
a=np.zeros((64,4),np.float32)
env.step_raw(torch.from_numpy(a).cuda()); orc.step(a)
g=env.get_state().cpu().numpy(); o=orc.get_state().view(np.float32)
bad=np.unique(np.argwhere((g.view(np.uint32)!=o.view(np.uint32)))[:,0])
print("step1 mismatch rows:", bad[:40])
for r in bad[:12]: print(r, g[r,:3], o[r,:3])
print("rew", env.rew_buf[:3].cpu().numpy(), orc.rew_buf[:3])
print("obs", env.obs_buf[0,0].cpu().numpy(), orc.obs_buf[0,0])
