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
