import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taco_amd import config
from taco_amd.vec_env import FpvBase
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from sweep5 import run
for ls in (2, 5, 8, 9, 10, 12, 16):
    print(ls, [round(run(4096, 1, ls), 1) for _ in range(2)], flush=True)
