"""tests/stdrandom.py restates the random streams the reference's tests draw from; these are the known answers that
pin it: the C++ standard's check values for both engines ([rand.predef]: the 10000th consecutive invocation of a
default-constructed minstd_rand0 is 1043618065, of mt19937 4123659995)."""
import numpy as np

from stdrandom import MinStdRand0, Mt19937


def test_minstd_rand0_check_value():
    rng = MinStdRand0(1)  # default_seed
    for _ in range(9999):
        rng.draw()
    assert rng.draw() == 1043618065


def test_mt19937_check_value():
    raw = Mt19937().raw(10000)
    assert int(raw[0]) == 3499211612 and int(raw[9999]) == 4123659995


def test_uniform_real_stays_in_range_and_uses_two_draws():
    rng = Mt19937()
    u = rng.uniform(-50.0, 50.0, 1000)
    assert u.min() >= -50.0 and u.max() < 50.0
    assert int(Mt19937().raw(2001)[2000]) == int(rng.raw(1)[0])
    m = MinStdRand0(1153297050)
    v = [m.uniform(0.01, 1.99) for _ in range(100)]
    assert min(v) >= 0.01 and max(v) < 1.99
