"""The DISTRIBUTIONS of the build's random draws against the reference's own torch expressions (RNG parity is distributional).

The glue fixtures feed the build's own uniforms / integers / normals through the reference's code, so they pin WHERE every draw lands but cannot
see a wrong threshold or a biased normal.  This file closes that hole on the CPU, at fixed seeds, >= 10^6 samples per test:

  * `rounded_normal` (inverse CDF on one uniform; oracle/taco_oracle.c, taco_step.hpp rounded_normal) against
    `torch.clamp(torch.round(torch.normal(0, 1, size)), -lim, lim)` -- the deploy length (lim 1, fpv_asymmetry.py:324) and the delay length
    (lim 3, :576): thresholds to the last bit of fp32, chi-square against the exact probabilities AND against torch's own samples;
  * the observation noise's standard normals (Box-Muller, obs_normals) against N(0,1) / `torch.normal(0, sigma, size)` (:402-410): moments,
    Kolmogorov-Smirnov, tails, independence across the twelve slots, env ids and steps;
  * the 24-bit uniform behind every `torch_rand_float` / `torch.rand` (torch_utils.py:217-219): half-open [0, 1), grid 2^-24 (torch.rand's own for
    fp32), moments, chi-square over bins, lag correlation across env / step / stream / index.

The generator is counter-based (Philox4x32-10; key = seed, counter = (env id, step, stream, block)): the HIP kernel's draws are the oracle's bit
for bit (every `-m gpu` parity test with a reset or noise in it), so what holds here holds on the GPU.
All p-value bounds are 1e-4 on FIXED seeds: deterministic, and a real defect (a threshold off by 3e-3, a normal with sigma 1.01) fails by many
orders of magnitude -- each test ends with such a deliberately broken variant to show its power.
"""
import numpy as np
import pytest
import torch
from scipy import stats

from oracle import oracle as O

P_MIN = 1e-4
STREAMS = {"reset": 1, "cmd": 2, "deploy": 3, "rotor": 4, "obs": 5}


def _chi2_p(counts, probs):
    exp = np.asarray(probs, np.float64) * counts.sum()
    return stats.chisquare(counts, exp).pvalue


def _torch_rounded(n, lim, seed):
    g = torch.Generator().manual_seed(seed)
    # the reference's expression (fpv_asymmetry.py:324 / :576), generator pinned
    return torch.clamp(torch.round(torch.normal(0, 1, size=(n, 1), generator=g)), -lim, lim).long().numpy().ravel()


@pytest.mark.parametrize("lim,stream,idx", [(1, "deploy", 0), (3, "reset", 36)])
def test_rounded_normal_against_the_references_torch_expression(lim, stream, idx):
    n_env, n_step = 4096, 512   # 2.1e6 draws: every env x step of the stream / index the env really uses (deploy 0; RU_DELAY = 36)
    u = O.uniform_block(7, n_env, n_step, STREAMS[stream], idx + 1)[:, :, idx].ravel()
    k = O.rounded_normal(u, lim)
    cats = np.arange(-lim, lim + 1)
    edges = np.concatenate([[-np.inf], cats[:-1] + 0.5, [np.inf]])
    probs = np.diff(stats.norm.cdf(edges))
    counts = np.array([(k == c).sum() for c in cats])
    assert counts.sum() == u.size
    p_exact = _chi2_p(counts, probs)
    t = _torch_rounded(u.size, lim, 1234)
    tc = np.array([(t == c).sum() for c in cats])
    p_torch_exact = _chi2_p(tc, probs)                       # torch's own samples against the same probabilities (sanity of the test itself)
    p_two = stats.chi2_contingency(np.stack([counts, tc])).pvalue
    print(f"rounded_normal lim {lim}: counts {counts.tolist()} torch {tc.tolist()}  p(exact) {p_exact:.3f}  p(torch vs exact) {p_torch_exact:.3f}  p(two-sample) {p_two:.3f}")
    assert p_exact > P_MIN and p_torch_exact > P_MIN and p_two > P_MIN
    # the thresholds are Phi(-2.5 ... 2.5) rounded to fp32, and a uniform ON a threshold goes up (u >= T), as round() maps z = +-x.5 ... measure-zero
    T = np.array([0.0062096653, 0.0668072013, 0.3085375387, 0.6914624613, 0.9331927987, 0.9937903347], np.float32)
    assert np.array_equal(T, stats.norm.cdf(np.arange(-2.5, 3.0, 1.0)).astype(np.float32))
    for i, th in enumerate(T):
        below, at = np.nextafter(th, np.float32(0)), th
        want = lambda z: int(np.clip(z, -lim, lim))
        assert O.rounded_normal([below], lim)[0] == want(-3 + i) and O.rounded_normal([at], lim)[0] == want(-3 + i + 1)
    assert O.rounded_normal([0.0], lim)[0] == -lim and O.rounded_normal([np.float32(1) - np.float32(2 ** -24)], lim)[0] == lim
    # power: thresholds shifted by 3e-3 (a wrong table) are rejected outright
    kb = np.clip(-3 + (u[:, None] >= (T + np.float32(3e-3))[None, :]).sum(1), -lim, lim)
    assert _chi2_p(np.array([(kb == c).sum() for c in cats]), probs) < 1e-12


def test_observation_noise_normals_against_n01_and_torch_normal():
    n_env, n_step = 2048, 64     # 131 072 env-steps x 12 slots = 1.57e6 normals (the reference draws eleven of the twelve per env-step)
    z = O.obs_normals_block(11, n_env, n_step).astype(np.float64)
    flat = z.reshape(-1)
    n = flat.size
    m, v = flat.mean(), flat.var()
    sk, ku = stats.skew(flat), stats.kurtosis(flat)
    ks = stats.kstest(flat, "norm")
    print(f"obs normals: n {n} mean {m:+.2e} var {v:.5f} skew {sk:+.2e} excess kurtosis {ku:+.2e} KS p {ks.pvalue:.3f} max |z| {np.abs(flat).max():.3f}")
    assert abs(m) < 4 / np.sqrt(n) and abs(v - 1) < 4 * np.sqrt(2 / n)
    assert abs(sk) < 4 * np.sqrt(6 / n) and abs(ku) < 4 * np.sqrt(24 / n) + 2e-3   # (a 24-bit uniform truncates the tail at 5.77 sigma: -1e-4 of kurtosis)
    assert ks.pvalue > P_MIN
    # tails: counts beyond 2, 3, 4 sigma against the normal law (Poisson bounds), and the hard limit of a 24-bit Box-Muller radius
    for s in (2.0, 3.0, 4.0):
        c, e = int((np.abs(flat) > s).sum()), 2 * stats.norm.sf(s) * n
        assert abs(c - e) < 5 * np.sqrt(e) + 1, f"beyond {s} sigma: {c} vs {e:.1f} expected"
    assert np.abs(flat).max() <= np.sqrt(-2 * np.log(2.0 ** -24)) + 1e-3
    # against torch.normal itself (the reference's call, fpv_asymmetry.py:403: torch.normal(0, 0.06 / 3 / 3, size)): two-sample KS on the scaled values
    g = torch.Generator().manual_seed(99)
    sig = 0.06 / 3 / 3
    tn = torch.normal(0, sig, size=(400000,), generator=g).numpy().astype(np.float64)
    mine = (flat[:400000].astype(np.float32) * np.float32(sig) + np.float32(0.0)).astype(np.float64)   # z * std + mean, what torch.normal computes
    p2 = stats.ks_2samp(mine, tn).pvalue
    print(f"  two-sample KS against torch.normal(0, {sig:.5f}): p {p2:.3f}")
    assert p2 > P_MIN
    # independence: the twelve slots of an env-step (Box-Muller pairs share a radius: cos / sin of one angle must still be uncorrelated, also in
    # their squares), neighbouring env ids, consecutive steps
    rows = z.reshape(-1, 12)
    bound = 5 / np.sqrt(rows.shape[0])
    c1 = np.corrcoef(rows, rowvar=False)
    c2 = np.corrcoef(rows ** 2, rowvar=False)
    off = ~np.eye(12, dtype=bool)
    print(f"  max |corr| between slots {np.abs(c1[off]).max():.2e}, between their squares {np.abs(c2[off]).max():.2e} (bound {bound:.2e})")
    assert np.abs(c1[off]).max() < bound and np.abs(c2[off]).max() < bound
    lag_env = np.corrcoef(z[:-1].reshape(-1), z[1:].reshape(-1))[0, 1]
    lag_step = np.corrcoef(z[:, :-1].reshape(-1), z[:, 1:].reshape(-1))[0, 1]
    assert abs(lag_env) < 5 / np.sqrt(n) and abs(lag_step) < 5 / np.sqrt(n)
    # power: a sigma of 1.005 fails the variance bound (2x), a mean of 0.01 the mean bound (3x) and the KS test
    assert abs((flat * 1.005).var() - 1) > 4 * np.sqrt(2 / n) and 0.01 > 3 * 4 / np.sqrt(n) and stats.kstest(flat + 0.01, "norm").pvalue < 1e-12


@pytest.mark.parametrize("stream", ["reset", "cmd", "rotor", "obs"])
def test_uniforms_are_24_bit_half_open_and_uncorrelated(stream):
    n_env, n_step, n_idx = 1024, 32, 40    # 1.3e6 uniforms: indices 0 .. 39 cover every draw of a reset (RU_COUNT = 37) and the rotor noise (4 x 10)
    u32 = O.uniform_block(5, n_env, n_step, STREAMS[stream], n_idx)
    u = u32.astype(np.float64)
    flat = u.reshape(-1)
    n = flat.size
    # grid and range: k / 2^24 with k in [0, 2^24): half-open like torch.rand (which has the same fp32 grid -- checked here on torch itself)
    k = flat * 2.0 ** 24
    assert np.array_equal(k, np.round(k)) and flat.min() >= 0.0 and flat.max() < 1.0
    tr = torch.rand(1 << 20, generator=torch.Generator().manual_seed(3)).numpy().astype(np.float64) * 2.0 ** 24
    assert np.array_equal(tr, np.round(tr)) and tr.max() < 2.0 ** 24
    assert abs(flat.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n) and abs(flat.var() - 1 / 12) < 4 * np.sqrt(1 / 180 / n)
    p = stats.chisquare(np.bincount((flat * 256).astype(np.int64), minlength=256)).pvalue
    lo = stats.chisquare(np.bincount((k.astype(np.int64) & 255), minlength=256)).pvalue      # the LOW byte of the 24 bits is as uniform as the high one
    ks = stats.kstest(flat, "uniform").pvalue
    print(f"uniform [{stream}]: n {n} mean {flat.mean():.5f} var {flat.var():.6f} chi2 p {p:.3f} low-byte chi2 p {lo:.3f} KS p {ks:.3f}")
    assert p > P_MIN and lo > P_MIN and ks > P_MIN
    # lag correlations along every axis of the counter: env id, step, index inside the stream (incl. across Philox blocks: lag 1 and lag 4)
    b = 5 / np.sqrt(n)
    cc = lambda a, c: np.corrcoef(a.reshape(-1), c.reshape(-1))[0, 1]
    lags = {"env": cc(u[:-1], u[1:]), "step": cc(u[:, :-1], u[:, 1:]), "index +1": cc(u[:, :, :-1], u[:, :, 1:]), "index +4": cc(u[:, :, :-4], u[:, :, 4:])}
    print("  lag correlations: " + ", ".join(f"{a} {v:+.2e}" for a, v in lags.items()) + f" (bound {b:.2e})")
    assert all(abs(v) < b for v in lags.values())
    # ... and against another stream and another seed at the same (env, step, index)
    other = O.uniform_block(5, n_env, n_step, STREAMS["deploy"], n_idx).astype(np.float64)
    seed2 = O.uniform_block(6, n_env, n_step, STREAMS[stream], n_idx).astype(np.float64)
    assert abs(cc(u, other)) < b and abs(cc(u, seed2)) < b and not np.array_equal(u32, seed2.astype(np.float32))
    # counter-based: the same (seed, env, step, stream, index) gives the same number wherever the block starts (what makes sharding invisible)
    assert np.array_equal(O.uniform_block(5, 16, 4, STREAMS[stream], n_idx, gid0=500, step0=20), u32[500:516, 20:24])


def test_torch_rand_float_is_the_references_affine_map_on_the_uniform():
    """torch_rand_float(lower, upper) = (upper - lower) * torch.rand + lower (torch_utils.py:217-219): the build applies the same fp32 expression to
    its uniform -- end points, so that e.g. E_c ~ U(0, 2.2) (battery_dynamics.py:45) can reach 0 and never 2.2"""
    u = np.array([0.0, 2.0 ** -24, 0.5, 1 - 2.0 ** -24], np.float32)
    for lo, hi in ((0.0, 2.2), (-2.0, 2.0), (-6.0, 6.0), (1 - 10.0 / 700, 1 + 10.0 / 700), (0.0, 400.0)):
        got = O.rand_float(lo, hi, u)
        want = (torch.tensor(hi - lo) * torch.from_numpy(u) + lo).numpy()   # python-float bounds, fp32 tensor: the reference's expression
        assert np.array_equal(got, want.astype(np.float32)), (lo, hi, got, want)
        assert got[0] == np.float32(lo) and got[-1] < np.float32(hi) + (0 if hi != 1 + 10.0 / 700 else 1e-7)
