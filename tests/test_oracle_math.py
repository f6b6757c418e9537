"""The oracle's own sin/cos/atan2/asin/log (bit-exact twins of taco_amd/csrc/taco_math.hpp) against numpy, and its
Philox4x32-10 against the published known-answer vectors of Random123 (kat_vectors)."""
import numpy as np

from oracle import oracle as O
from util import ulp_diff


def test_sincos_accuracy():
    x = np.concatenate([np.linspace(-7, 7, 20001), np.linspace(-100, 100, 5001), [0.0, -0.0, 1e-8, 3.14159265, 1.57079633]]).astype(np.float32)
    s, c = O.vec("orc_sinf", x), O.vec("orc_cosf", x)
    # absolute error: near the zeros of sin/cos a relative bound is meaningless for any 3-term Cody-Waite scheme
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 1.2e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 1.2e-7
    assert O.lib().orc_sinf(0.0) == 0.0 and O.lib().orc_cosf(0.0) == 1.0
    assert np.isnan(O.lib().orc_sinf(float("inf"))) and np.isnan(O.lib().orc_cosf(float("nan")))


def test_atan2_accuracy_and_special_cases():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(40000).astype(np.float32) * rng.choice([1e-3, 1, 50], 40000).astype(np.float32)
    x = rng.standard_normal(40000).astype(np.float32) * rng.choice([1e-3, 1, 50], 40000).astype(np.float32)
    r = O.vec("orc_atan2f", y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(r - ref).max() < 4e-7
    assert ulp_diff(r, ref.astype(np.float32)).max() <= 3
    f = O.lib().orc_atan2f
    assert f(0.0, 1.0) == 0.0 and f(0.0, 0.0) == 0.0
    assert abs(f(0.0, -1.0) - np.pi) < 3e-7 and abs(f(-0.0, -1.0) + np.pi) < 3e-7
    assert abs(f(1.0, 0.0) - np.pi / 2) < 2e-7 and abs(f(-1.0, 0.0) + np.pi / 2) < 2e-7
    assert np.isnan(f(float("nan"), 1.0))


def test_asin_accuracy():
    x = np.linspace(-1, 1, 40001).astype(np.float32)
    r = O.vec("orc_asinf", x)
    ref = np.arcsin(x.astype(np.float64))
    assert np.abs(r - ref).max() < 2.5e-7
    assert ulp_diff(r, ref.astype(np.float32)).max() <= 3
    assert np.isnan(O.lib().orc_asinf(1.0000001)) and O.lib().orc_asinf(0.0) == 0.0


def test_log_accuracy():
    x = np.concatenate([np.linspace(2.0 ** -24, 1.0, 30001), np.geomspace(1e-7, 1e3, 5001)]).astype(np.float32)
    r = O.vec("orc_logf", x)
    ref = np.log(x.astype(np.float64))
    assert (np.abs(r - ref) <= 2e-7 + 2.5e-7 * np.abs(ref)).all()


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert O.philox(0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox(0xffffffffffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox(0x299f31d0a4093822, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_is_24_bit_half_open():
    u = O.lib().orc_uniform
    assert u(0) == 0.0 and u(0xff) == 0.0 and u(0x100) == 2.0 ** -24
    assert u(0xffffffff) == 1.0 - 2.0 ** -24
