import sys; sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np, torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
from oracle import oracle as O
from test_parity_gpu import action_stream
n=65536+77
cfg=config.baseline_config(4,num_envs=n); flat=config.flat_cfg(cfg)
for trial in range(3):
    env=FpvBase(cfg,copy_outputs=False); orc=O.OracleEnv(flat,threads=16)
    acts=action_stream(n,3,0)
    for t in range(2):
        env.step_raw(torch.from_numpy(acts[t]).cuda()); orc.step(acts[t])
        g=env.get_state().cpu().numpy(); o=orc.get_state().view(np.float32)
        keep=np.r_[0:20,26:g.shape[0]]
        bad=np.argwhere(g.view(np.uint32)[keep]!=o.view(np.uint32)[keep])
        print("trial",trial,"step",t,"n bad",len(bad))
        if len(bad):
            rows=np.unique(keep[bad[:,0]]); envs=np.unique(bad[:,1])
            print(" rows",rows[:20]," envs",envs[:20], " lanes", envs[:20]%64, "blocks", envs[:20]//256)
            e=envs[0]; print(" env",e,"act_old gpu",g[44:48,e],"orc",o[44:48,e],"act gpu",g[40:44,e],"a_in",acts[t][e])
