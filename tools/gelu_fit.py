"""Coefficients of the GELU used by csrc/ffn.hip:  gelu(x) = max(x, 0) - |x| r(|x|),  r(z) = erfc(z / sqrt 2) / 2 ~ q(z)^-P
with q a polynomial (the Abramowitz-Stegun 7.1.28 form with the 1/2 and the 1/sqrt 2 folded in).  Prints, for a few
(degree, power) pairs, the least-squares fit weighted by z (what enters the GELU is z r(z)) and its maximum error.
The kernel uses (5, 16): |gelu error| < 5e-6."""
import numpy as np
from scipy.special import erfc
from scipy.optimize import least_squares

xs = np.linspace(0, 9, 4001)
target = 0.5 * erfc(xs / np.sqrt(2))


def fit(n, P):
    m = xs < 5
    c = np.polyfit(xs[m], target[m] ** (-1.0 / P), n)[::-1]

    def res(c):
        q = np.maximum(np.polyval(c[::-1], xs), 1e-3)
        return (q ** (-P) - target) * np.maximum(xs, 0.3)
    c = least_squares(res, c, method="lm", max_nfev=4000).x
    return np.abs(res(c)).max(), c


for n, P in [(6, 16), (5, 16), (6, 8), (5, 8)]:
    e, c = fit(n, P)
    print(f"degree {n}, power {P}: max |z err| = {e:.3e}; c0..c{n} =", " ".join(f"{v:.8e}" for v in c))
