"""The random stream the reference's own tests draw from, restated: `std::default_random_engine` of libstdc++ is
`minstd_rand0` (x <- 16807 x mod 2^31 - 1) and its `std::uniform_real_distribution<double>` is
`generate_canonical<double, 53>` -- two engine draws per value, (d1 - 1) + (d2 - 1) * R over R^2 with R = 2147483646 --
scaled to [a, b).  With it the parity tests integrate the SAME samples as tests/ohmtest/NdtTests.cpp does on a
GCC / Linux build (the platform its expected values were taken on), so the reference's expected probabilities can be held
to the reference's own tolerances instead of "a statistically similar cloud".  Test infrastructure only."""
import numpy as np


class MinStdRand0:
    MODULUS = 2147483647

    def __init__(self, seed):
        self.state = seed % self.MODULUS or 1

    def draw(self):
        self.state = (16807 * self.state) % self.MODULUS
        return self.state

    def canonical(self):
        r = 2147483646.0  # max() - min() + 1
        total = float(self.draw() - 1)
        total += float(self.draw() - 1) * r
        value = total / (r * r)
        return value if value < 1.0 else float(np.nextafter(1.0, 0.0))

    def uniform(self, a, b):
        """One draw of a std::uniform_real_distribution<double>(a, b)."""
        return self.canonical() * (b - a) + a


class Mt19937:
    """`std::mt19937` (default seed 5489) with libstdc++'s `std::uniform_real_distribution<double>`: two 32-bit draws per
    value, (d1 + d2 * 2^32) / 2^64 in double arithmetic.  numpy's legacy RandomState is the same generator with the same
    `init_genrand` seeding; drawing the full 32-bit range returns its raw output stream (checked in
    tests/test_stdrandom.py against the C++ standard's own check value: the 10000th draw of a default-constructed engine
    is 4123659995)."""

    def __init__(self, seed=5489):
        self._rs = np.random.RandomState(seed)

    def raw(self, n):
        return self._rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)

    def uniform(self, a, b, n):
        """n consecutive draws of a std::uniform_real_distribution<double>(a, b)."""
        d = self.raw(2 * n).astype(np.float64)
        total = d[0::2] + d[1::2] * 4294967296.0
        value = total / 18446744073709551616.0
        value = np.where(value < 1.0, value, np.nextafter(1.0, 0.0))
        return value * (b - a) + a
