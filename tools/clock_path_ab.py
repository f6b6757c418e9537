import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from taco_amd import config
from taco_amd.vec_env import FpvBase
dev=torch.device("cuda:0"); n=4096
acts=bench.make_actions(n,64,1000,dev)
env=FpvBase(config.baseline_config(1,num_envs=n),copy_outputs=False)
med,_,_=bench.steady_windows(env.step_raw,acts,torch,0.1,5,2000); print("eager, clock in kernel args      %.2f us"%med)
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    env.step_raw(acts[0])
med,_,_=bench.steady_windows(env.step_raw,acts,torch,0.1,5,2000); print("eager, clock on device (sticky)   %.2f us"%med)
for K in (1,16,64):
    e2=FpvBase(config.baseline_config(1,num_envs=n),copy_outputs=False)
    for t in range(8): e2.step_raw(acts[t])
    torch.cuda.synchronize()
    g2=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for t in range(K): e2.step_raw(acts[t%64])
    for _ in range(2000//K): g2.replay()
    torch.cuda.synchronize()
    ws=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(max(1,1280//K)): g2.replay()
        e1.record(); torch.cuda.synchronize()
        ws.append(e0.elapsed_time(e1)*1e3/(max(1,1280//K)*K))
    ws.sort(); print("graph of %d steps                 %.2f us/step"%(K,ws[2]))
