"""SURVEY section 8f rows N2 / N4, CPU parts: the Logger-compatible recorder and the YAML / CLI front-end."""
import argparse
import math
import os

import numpy as np
import pytest

from taco_amd import cfg_io, config
from taco_amd.recorder import EpisodeRecorder


def test_recorder_matches_the_reference_logger_behaviour(tmp_path):
    r = EpisodeRecorder(str(tmp_path / "rec"))
    assert not r.dump_buffer()                                   # empty buffer: nothing written (logger.py:39)
    for ep in range(7):
        for t in range(4):
            r.store_buffer(copter_pos=np.array([t, ep, 1.0]), reward=np.array([0.1 * t]))
        wrote = r.dump_buffer()
        r.reset_buffer()
        assert wrote == (ep < 5)                                 # at most the first five episodes (logger.py:36)
    files = sorted(os.listdir(tmp_path / "rec"))
    assert "copter_pos0.npy" in files and "copter_pos4.csv" in files and "copter_pos5.npy" not in files
    a = np.load(tmp_path / "rec" / "copter_pos3.npy")
    assert a.shape == (4, 3) and a[2, 0] == 2 and a[0, 1] == 3
    assert np.loadtxt(tmp_path / "rec" / "reward1.csv", delimiter=",").shape == (4,)


def test_yaml_roundtrip_and_cli_overrides(tmp_path):
    p = cfg_io.write_default_yaml(str(tmp_path / "Fpv_asymmetry_PPO_rotate.yaml"), "rotate", delay_time=40)
    cfg = cfg_io.load_yaml(p)
    assert cfg["Task"]["task_mode"] == "rotate" and cfg["Task"]["delay_time"] == 40 and math.isinf(cfg["Task"]["env"]["clipObservations"])
    ap = cfg_io.add_env_arguments(argparse.ArgumentParser())
    # booleans are strings; anything but 'True' / 'False' keeps the YAML value (train_fpv_asymmetry_ppo.py:282 ff.)
    args = ap.parse_args(["--train_mode=train", "--task_mode=flip", "--num_envs=512", "--lenStates=5", "--rotor_noise=True",
                          "--random_voltage=False", "--battery_consumption=maybe", "--rotor_response_time=0.02", "--delay_time=20", "--seed=7"])
    task = cfg_io.apply_overrides(cfg, args, diff_value=(0.2, 1.0))
    assert task["name"] == "Fpv_flip" and task["task_mode"] == "flip" and task["env"]["numEnvs"] == 512 and task["env"]["lenStates"] == 5
    assert task["rotor_noise"] is True and task["random_voltage"] is False and task["battery_consumption"] is True
    assert task["rotor_response_time"] == 0.02 and task["delay_time"] == 20 and task["seed"] == 7
    assert task["difficulty"] == 0.2 and task["record_flag"] is False
    flat = config.flat_cfg(task)
    assert flat["task_mode"] == "flip" and flat["num_envs"] == 512 and flat["rotor_noise"] and not flat["random_voltage"]
    # testmodel forces the evaluation flag set (:328-354)
    args = ap.parse_args(["--train_mode=testmodel", "--load_task_mode=pos"])
    task = cfg_io.apply_overrides(cfg, args, record_path="/tmp/x")
    assert task["env"]["numEnvs"] == 12 and task["difficulty"] == 1 and task["observation_noise"] and not task["battery_consumption"]
    assert task["name"] == "Fpv_pos" and task["record_flag"] and task["env"]["maxEpisodeLength"] == 1000
    with pytest.raises(ValueError):
        cfg_io.apply_overrides(cfg, ap.parse_args(["--train_mode=other"]))
