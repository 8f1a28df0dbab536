"""Fit of bd_common.h:gelu_fast -- x / (1 + 2^(-x p(x^2))), p cubic in x^2 -- against the exact erf GELU (build-container tool)."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf

x = np.linspace(-9, 9, 20001)
g = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def model(c, x):
    x2 = x * x
    return x / (1 + np.exp2(-np.clip(x * (c[0] + x2 * (c[1] + x2 * c[2])), -80, 80)))


c = least_squares(lambda c: model(c, x) - g, [2.3022, 0.1029, 0.0]).x
for _ in range(30):   # push towards minimax by re-weighting the largest residuals
    e = model(c, x) - g
    w = 1 + 20 * np.abs(e) / np.abs(e).max()
    c = least_squares(lambda c: (model(c, x) - g) * w, c).x
e = model(c, x) - g
print("coefficients", c, "max abs err", np.abs(e).max(), "at x =", x[np.abs(e).argmax()])
xs = np.linspace(-30, 30, 60001)
xc = np.clip(xs, -8, 8)
fast = xs / (1 + np.exp2(-(xc * (2.30034092 + xc * xc * (1.07380689e-01 + xc * xc * -1.10189899e-03)))))
print("shipped constants, clamp 8: max abs err on [-30, 30]", np.abs(fast - xs * 0.5 * (1 + erf(xs / np.sqrt(2)))).max())


# ---- gelu_poly2: x * (0.5 + xc * P(xc^2)), xc = clamp(x, -4, 4), P of degree 7
x = np.linspace(-10, 10, 80001)
g = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def poly(c, x):
    xc = np.clip(x, -4, 4)
    x2 = xc * xc
    p = c[-1]
    for k in range(len(c) - 2, -1, -1):
        p = p * x2 + c[k]
    return x * (0.5 + xc * p)


c = np.zeros(8)
c[0] = 0.3989
c = least_squares(lambda c: poly(c, x) - g, c, xtol=1e-15, ftol=1e-15).x
for _ in range(80):
    e = poly(c, x) - g
    w = 1 + 40 * np.abs(e) / np.abs(e).max()
    c = least_squares(lambda c: (poly(c, x) - g) * w, c, xtol=1e-15, ftol=1e-15).x
print("gelu_poly2 coefficients (low to high):", ", ".join("%.9e" % v for v in c), "max abs err", np.abs(poly(c, x) - g).max())
