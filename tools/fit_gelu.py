#!/usr/bin/env python3
"""Coefficients of csrc/common.h:gelu_fast -- erfc(z) ~= 2^P(z), P(z) = z (c1 + z (c2 + ... + z c6)), iteratively reweighted
least squares towards the minimax fit on [0, 4.2]; prints the coefficients, the erf error and the fp32 error of the whole
gelu against the erf form in float64."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf


def model(c, z):
    p = np.zeros_like(z)
    for ck in c[::-1]:
        p = (p + ck) * z
    return 1 - np.exp2(p)


def main(deg=6, zmax=4.2):
    z = np.linspace(1e-6, zmax, 6000)
    target = erf(z)
    c = np.zeros(deg)
    c[:2] = (-1.6, -0.9)
    c = least_squares(lambda cc: model(cc, z) - target, c, method="lm", xtol=1e-15, ftol=1e-15).x
    w = np.ones_like(z)
    best = (1.0, c)
    for _ in range(150):
        e = model(c, z) - target
        m = np.abs(e).max()
        if m < best[0]:
            best = (m, c.copy())
        w = w * (1 + 3 * np.abs(e) / m)
        w /= w.mean()
        c = least_squares(lambda cc: (model(cc, z) - target) * w, c, method="lm", xtol=1e-15, ftol=1e-15).x
    err, c = best
    print("max |erf error| on the fit interval:", err)
    print("coefficients c1..c%d:" % deg, ", ".join("%.9e" % x for x in c))
    f = np.float32
    x = np.linspace(-8, 8, 2000001).astype(f)
    a = np.abs(x)
    zf = np.minimum(a * f(0.70710678118654752440), f(zmax))
    cf = c.astype(f)
    t = cf[-1]
    for ck in cf[-2::-1]:
        t = (t * zf + ck).astype(f)
    g = (f(-0.5) * a) * np.exp2((t * zf).astype(f)).astype(f) + np.maximum(x, f(0))
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    print("fp32 gelu: max abs error %.3e, max relative error where |gelu| > 1e-3: %.3e"
          % (np.abs(g - ref).max(), (np.abs(g - ref) / np.maximum(np.abs(ref), 1e-3)).max()))


if __name__ == "__main__":
    main()
