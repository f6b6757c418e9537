"""Shared helpers for the parity tests."""
import numpy as np


def ulp_diff(got, exp):
    """|got - exp| in units of the fp32 spacing at exp; NaN==NaN counts as 0."""
    got = np.asarray(got, np.float32)
    exp = np.asarray(exp, np.float32)
    both_nan = np.isnan(got) & np.isnan(exp)
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    sp = np.spacing(np.abs(np.where(np.isfinite(exp), exp, 1)).astype(np.float32)).astype(np.float64)
    out = err / sp
    out[both_nan] = 0
    return out


def assert_bits_equal(got, exp, what=""):
    got = np.ascontiguousarray(got)
    exp = np.ascontiguousarray(exp)
    assert got.shape == exp.shape, f"{what}: shape {got.shape} vs {exp.shape}"
    if got.dtype == np.float32:
        g, e = got.view(np.uint32), exp.astype(np.float32).view(np.uint32)
        nan_ok = np.isnan(got) & np.isnan(exp)
        bad = (g != e) & ~nan_ok & ~((got == 0) & (exp == 0))  # +0 / -0 compare equal (torch.eq semantics)
    else:
        bad = got != exp
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} words differ, first at {np.argwhere(bad)[0]}: {got[tuple(np.argwhere(bad)[0])]} vs {exp[tuple(np.argwhere(bad)[0])]}"


def assert_ulp(got, exp, max_ulp, what="", abs_floor=0.0):
    d = ulp_diff(got, exp)
    if abs_floor:
        d = np.where(np.abs(np.asarray(got, np.float64) - np.asarray(exp, np.float64)) <= abs_floor, 0, d)
    assert np.nanmax(d) <= max_ulp, f"{what}: max {np.nanmax(d):.1f} ulp > {max_ulp}"
    assert (np.isnan(got) == np.isnan(exp)).all(), f"{what}: NaN pattern differs"


# ---------------------------------------------------------------------------------------------------------------------------------------
# Comparison with the GLUE fixtures (tests/golden/glue_*.npz = traces of the reference's own VecTask.step(), tests/golden/make_glue_golden.py),
# shared by the CPU test of the oracle (tests/test_oracle_golden.py) and the GPU test of the HIP kernel (tests/test_parity_gpu.py):
#   EXACT (bit for bit, every recorded step, every env): reset_buf, time_outs, progress_buf, actions_remained_length, the whole
#     actions_remained_buffer [4 x 100], actions / actions_old, rotor and aero parameters, command[0], pos / rotate command[1];
#   TOLERANCE for the continuous state: the closed loop runs 150-520 steps through torch's libm-backed sin / cos / atan2 / asin / sqrt on the
#     reference's side and the build's own polynomial / correctly rounded versions on the other (<= 2-3 ulp apart per call by design,
#     tests/test_oracle_math.py), so trajectories drift apart at the 1e-5 ... 2e-3 level; any glue error shows up at the 1e-1 level.
GLUE_TOL = {"root state": 2e-3, "target pose": 1e-6, "rpy_old / rpy_continuous": 1e-3, "PID memory": 1e-3, "battery state / voltage": 1e-4,
            "rotor speeds (~300 rev/s)": 2e-2, "command[1] / flip_radian": 2e-4, "obs stack": 1e-3, "states stack": 1e-3, "reward": 5e-6}


def glue_case(g):
    """(cfg, actions [T,n,4], {step: record index}) of one glue fixture"""
    import ast
    from taco_amd import config
    task, n, seed = str(g["cfg_task"]), int(g["cfg_n"]), int(g["cfg_seed"])
    kw = dict(ast.literal_eval(str(g["cfg_kw"])))
    return config.default_cfg(task, n, seed=seed, **kw), g["actions"], {int(t): k for k, t in enumerate(g["step"])}


def check_against_glue_fixture(g, k, what, blob, obs, states, rew, done, tmo, maxima, seen, rpy_envs=None):
    """one recorded step of a glue fixture against (state blob as float32 [467, n], obs, states, rew, done, time-outs) of an implementation.
    `maxima` collects the largest deviation per continuous group, `seen` what the trace exercised.  rpy_envs: envs whose rpy rows the
    implementation maintains (None = all)."""
    task = str(g["cfg_task"])
    mine, ref = np.ascontiguousarray(blob).view(np.float32), g["blob"][k]
    assert_bits_equal(np.asarray(done), g["reset"][k], what + " reset_buf")
    assert_bits_equal(np.asarray(tmo).astype(np.uint8), g["timeout"][k], what + " time_outs")
    assert_bits_equal(mine[65:67], ref[65:67], what + " progress_buf / actions_remained_length")
    assert_bits_equal(mine[67:], ref[67:], what + " actions_remained_buffer")
    assert_bits_equal(mine[40:48], ref[40:48], what + " actions / actions_old")
    assert_bits_equal(mine[51:65], ref[51:65], what + " rotor / aero parameters")
    assert_bits_equal(mine[48], ref[48], what + " command[0] (task id)")
    if task in ("pos", "rotate"):
        assert_bits_equal(mine[49], ref[49], what + " command[1]")

    def close(a, b, name):
        a, b = np.asarray(a), np.asarray(b)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what} {name}: NaN pattern differs"
        m = float(d[np.isfinite(d)].max()) if np.isfinite(d).any() else 0.0
        maxima[name] = max(maxima.get(name, 0.0), m)
        assert m <= GLUE_TOL[name], f"{what} {name}: max |diff| {m:.3e} > {GLUE_TOL[name]}"

    close(mine[0:13], ref[0:13], "root state")
    close(mine[13:20], ref[13:20], "target pose")
    if rpy_envs is None:
        close(mine[20:26], ref[20:26], "rpy_old / rpy_continuous")
    elif np.any(rpy_envs):
        close(mine[20:26][:, rpy_envs], ref[20:26][:, rpy_envs], "rpy_old / rpy_continuous")
    close(mine[26:32], ref[26:32], "PID memory")
    close(mine[32:36], ref[32:36], "battery state / voltage")
    close(mine[36:40], ref[36:40], "rotor speeds (~300 rev/s)")
    close(mine[49:51], ref[49:51], "command[1] / flip_radian")
    close(obs, g["obs"][k], "obs stack")
    close(states, g["states"][k], "states stack")
    close(rew, g["rew"][k], "reward")
    seen["reset"] = seen.get("reset", 0) + int(np.asarray(done).sum())
    seen["timeout"] = seen.get("timeout", 0) + int(np.asarray(tmo).sum())
    seen["dense"] = seen.get("dense", 0) + int((ref[66].view(np.int32) + 10 > 90).sum())   # this step's write reached the tail slots [90, 100)
    seen["at500"] = seen.get("at500", 0) + int((ref[65].view(np.int32) == 500).sum())


def check_glue_trace_content(case, seen):
    """the traces really contain what they are there for"""
    assert seen["reset"] > (5 if case in ("pos", "overflow", "deploy", "rotate") else 0)   # (flip / mix record only ~70 of their 520 steps)
    if case in ("pos", "overflow", "deploy", "rotate"):
        assert seen["timeout"] > 10
    if case in ("overflow", "deploy"):
        assert seen["dense"] > 100, "the delay line never reached the tail slots (stale-tail / truncated-write regime)"
    if case in ("flip", "mix"):
        assert seen["at500"] >= 10, "no env crossed progress 500 (command re-draw)"
