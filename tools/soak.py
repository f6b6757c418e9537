"""One-off soak: long bit-exact comparison HIP vs oracle on small launches (the role-wavefront / mailbox instantiation), all randomisation on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from taco_amd import config
import test_parity_gpu as T
# (the last two: the one-lane forms, whose resetting lanes share their Philox draws -- round 5)
for task, n, steps, form in (("mix", 2048, 3000, "auto"), ("flip", 777, 2000, "auto"), ("rotate", 4096, 1500, "auto"), ("mix", 5000, 1200, "lane_throughput"), ("flip", 3001, 1000, "lane_roles")):
    cfg = config.default_cfg(task, n, env_maxEpisodeLength=300, rotor_noise=True, observation_noise=True, ramdom_deploy_time=True, ramdom_delay_time=True,
                             random_rotordynamic_coe=True, random_rotor_response=True, random_aerodynamic_coe=True, env_lenStates=3, delay_time=60, seed=123)
    done = T.run_pair(cfg, steps, seed=7, check_every=25, form=form)
    print(task, n, steps, form, "episodes finished:", done, flush=True)
