#!/usr/bin/env python3
"""Coefficients of common.h gelu_erf: erfc(t) ~ 2^(-t q(t)) on [0, 4.2], q of degree 6, fitted to g(t) = -log2(erfc(t)) / t by least squares on Chebyshev nodes
with Lawson reweighting towards the minimax error of erfc itself (weight = d erfc / d g = erfc(t) ln 2 t).  Prints the fp32 coefficients and the largest
|erf error| when the form is evaluated in fp32.  CPU only (numpy, scipy)."""
import numpy as np
from scipy.special import erf, erfc, log_ndtr

T, N = 4.2, 6
t = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * T / 2 + T / 2
g = lambda x: -(np.log(2.0) + log_ndtr(-x * np.sqrt(2.0))) / np.log(2.0) / x       # log erfc(x) = log 2 + log Phi(-x sqrt 2)
A, y, sens, lw = np.vander(t, N + 1, increasing=True), g(t), erfc(t) * np.log(2) * t, np.ones_like(t)
for _ in range(60):
    W = sens * lw
    coef, *_ = np.linalg.lstsq(A * W[:, None], y * W, rcond=None)
    err = np.abs((A @ coef - y) * sens)
    lw = lw * (0.5 + err / err.max())
    lw /= lw.max()
c32 = coef.astype(np.float32)
tt = np.linspace(1e-6, T, 400001).astype(np.float32)
q = np.full_like(tt, c32[-1])
for c in c32[-2::-1]:
    q = q * tt + c
approx = np.float32(1) - np.exp2((-tt * q).astype(np.float32))
print("coefficients (constant first):", [float(c) for c in c32])
print("max |erf error|, fp32 evaluation: %.3e" % np.abs(approx.astype(np.float64) - erf(tt.astype(np.float64))).max())
