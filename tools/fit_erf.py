#!/usr/bin/env python
"""Fit the polynomial of csrc/common.h: erf_poly_f — erf(z) ~ z P(z^2) on [0, zmax], minimax by linear programming (scipy HiGHS) on a
dense grid, then evaluated in fp32 Horner arithmetic against scipy's erf on [0, 8] (clamped argument).  Prints the coefficients
(lowest order first) and the errors of erf and of gelu(x) = 0.5 x (1 + erf(x / sqrt 2)).  Usage: python tools/fit_erf.py [zmax] [degree]"""
import sys

import numpy as np
from scipy.optimize import linprog
from scipy.special import erf

zmax = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
z = np.linspace(1e-4, zmax, 3000)
V = np.vander(z * z, deg + 1, increasing=True) * z[:, None]
f = erf(z)
n = deg + 1
obj = np.zeros(n + 1); obj[-1] = 1
one = np.ones((len(z), 1))
r = linprog(obj, A_ub=np.vstack([np.hstack([V, -one]), np.hstack([-V, -one])]), b_ub=np.concatenate([f, -f]),
            bounds=[(None, None)] * n + [(0, None)], method="highs")
c = r.x[:n].astype(np.float32)
zz = np.linspace(0, 8, 400001).astype(np.float32)
zc = np.minimum(zz, np.float32(zmax)); u = zc * zc
acc = np.full_like(zz, c[-1])
for k in range(deg - 1, -1, -1):
    acc = acc * u + c[k]
e = np.abs((zc * acc).astype(np.float64) - erf(zz.astype(np.float64)))
x = zz.astype(np.float64) * np.sqrt(2)
print("coefficients (z^0, z^2, ...):", ", ".join("%.9ef" % v for v in c))
print("minimax error (LP, fp64): %.3e; fp32 Horner: erf %.3e, gelu abs %.3e" % (r.x[-1], e.max(), (0.5 * x * e).max()))
