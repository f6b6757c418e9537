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


def test_schedules_match_the_reference_formulas():
    """Row N3 (ppo_asymmetry.py:142-175), defaults of the reference constructor (epochs=500): values at the knots and between them,
    computed by hand from the reference's expressions."""
    from taco_amd.train_utils import Schedules
    s = Schedules()
    # learning rate: ratio 1 -> 0.3 over learning process [0, 0.7] (= epoch 350 of 500, both clocks coincide), then flat
    assert s.learning_rate(0) == 3e-4
    assert abs(s.learning_rate(175) - ((0.3 - 1) / 350 * 175 + 1) * 3e-4) < 1e-18
    assert s.learning_rate(350) == 0.3 * 3e-4 and s.learning_rate(499) == 0.3 * 3e-4
    # lipschitz: ratio 1 until lp 0.3 / epoch 100, down to 0.3 at lp 0.7 / epoch 500; min of the two clocks, times 5
    assert s.lipschitz(0) == 5 and s.lipschitz(100) == 5
    lp = 250 / 500
    p0 = (0.3 - 1) / (0.7 - 0.3) * (lp - 0.3) + 1
    p1 = (0.3 - 1) / (500 - 100) * (250 - 100) + 1
    assert abs(s.lipschitz(250) - min(p0, p1) * 5) < 1e-12 and min(p0, p1) == p0
    assert abs(s.lipschitz(400) - 0.3 * 5) < 1e-12
    # difficulty: 0.1 -> 1, max of the two clocks
    assert s.difficulty(0) == 0.1 and s.difficulty(99) == 0.1 and abs(s.difficulty(149) - ((1 - 0.1) / 400 * 49 + 0.1)) < 1e-12  # the epoch clock leads
    d0 = (1 - 0.1) / (0.7 - 0.3) * (lp - 0.3) + 0.1
    d1 = (1 - 0.1) / (500 - 100) * (250 - 100) + 0.1
    assert abs(s.difficulty(250) - max(d0, d1)) < 1e-12 and max(d0, d1) == d0 and d0 > d1
    assert s.difficulty(351) == 1 and s.at(351) == (s.learning_rate(351), s.lipschitz(351), 1)
    off = Schedules(learning_rate_schedule=False, lipschitz_schedule=False, difficulty_schedule=False)
    assert off.at(3) == (0.3 * 3e-4, 0.3 * 5, 1)
    # monotone: lr and lipschitz never rise, difficulty never falls
    vals = [s.at(e) for e in range(500)]
    assert all(a[0] >= b[0] - 1e-18 and a[1] >= b[1] - 1e-12 and a[2] <= b[2] + 1e-12 for a, b in zip(vals, vals[1:]))  # (the ramp overshoots its end value by 1 ulp at the knot, as the reference's does)


def test_schedules_equal_the_references_update_preamble():
    """Row N3 pinned by fixture: tests/golden/schedules.npz holds what the reference's OWN `PPO.update(epoch)` preamble (ppo_asymmetry.py:142-175,
    executed on a stub self by tests/golden/make_schedules_golden.py) yields for every epoch -- constructor defaults, two other knot sets and all
    schedules off.  `Schedules.at(epoch)` must return the same three doubles, bit for bit."""
    import ast
    from taco_amd.train_utils import Schedules
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "schedules.npz"))
    for case in ("defaults", "long", "short", "off"):
        s = Schedules(**dict(ast.literal_eval(str(g[case + "_kw"]))))
        ref = g[case]
        assert ref.shape == (s.epochs + 1, 3)
        got = np.array([s.at(e) for e in range(s.epochs + 1)], np.float64)
        assert np.array_equal(got, ref), f"{case}: first difference at epoch {np.argwhere(got != ref)[0]}"
    assert np.allclose(g["defaults"][250], [1.5e-4, 3.25, 0.55], rtol=1e-12)


def test_spectral_clamp_and_torchscript_export(tmp_path):
    """Row N3: spectral_normalize_actors (:398-404) and save_actor_as_pt (:458-468) on the golden policy's actor."""
    import torch
    from taco_amd.train_utils import actor_module_from_state_dict, save_actor_as_pt, spectral_normalize_actor_
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "policy.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    actor = actor_module_from_state_dict(sd, 1)
    mu = actor(torch.from_numpy(g["obs"])).detach().numpy()
    np.testing.assert_allclose(mu, g["mu"], rtol=0, atol=1e-6)          # same forward as the reference module
    traced, out0 = save_actor_as_pt(sd, str(tmp_path / "actor_1.pt"), 1)
    loaded = torch.jit.load(str(tmp_path / "actor_1.pt"))
    np.testing.assert_allclose(loaded(torch.from_numpy(g["obs"])).detach().numpy(), mu, rtol=0, atol=1e-7)
    assert out0.shape == (1, 4)
    norms = [float(torch.linalg.matrix_norm(p.detach(), ord=2)) for p in actor.parameters() if p.ndim > 1]
    assert norms[0] > 4 and norms[1] > 4 and norms[2] < 0.1                # the reference's orthogonal init (x 3 in the fixture)
    n = spectral_normalize_actor_(actor, 1.5)
    after = [float(torch.linalg.matrix_norm(p.detach(), ord=2)) for p in actor.parameters() if p.ndim > 1]
    assert n == 2 and after[0] == pytest.approx(1.5, rel=1e-5) and after[1] == pytest.approx(1.5, rel=1e-5) and after[2] == norms[2]
