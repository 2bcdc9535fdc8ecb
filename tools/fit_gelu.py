"""Coefficients of the polynomial erf-GELU of the bf16 GEMM epilogues (csrc/common.h: gelu_poly2, gelu_grad_poly2) and
their error in fp32 Horner arithmetic against the erf form.  usage: python tools/fit_gelu.py"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf

f32 = np.float32
Phi = lambda x: 0.5 * (1 + erf(x / np.sqrt(2)))          # noqa: E731
phi = lambda x: np.exp(-x * x / 2) / np.sqrt(2 * np.pi)    # noqa: E731
n = 9                                                      # coefficients: degree 8 in z = x^2 / 8 - 1 on |x| <= 4
k = np.arange(2000)
zn = np.cos(np.pi * (k + 0.5) / 2000)
un = (zn + 1) * 8
xn = np.sqrt(un)
Pn = np.where(un > 1e-12, xn * (Phi(xn) - 0.5) / np.maximum(un, 1e-300), phi(0))
Rn = np.where(xn > 1e-6, (Phi(xn) - 0.5 + xn * phi(xn)) / np.maximum(xn, 1e-300), 2 * phi(0))
cp = C.cheb2poly(C.chebfit(zn, Pn, n - 1))
cr = C.cheb2poly(C.chebfit(zn, Rn, n - 1))
print('P (z^0 ..):', ', '.join(f'{v:.9e}f' for v in cp))
print('R (z^0 ..):', ', '.join(f'{v:.9e}f' for v in cr))


def horner32(co, z):
    acc = np.full_like(z, f32(co[-1]))
    for a in co[-2::-1]:
        acc = (acc * z + f32(a)).astype(f32)
    return acc


xs = np.linspace(-8, 8, 800001).astype(f32)
xc = np.clip(xs, -4, 4).astype(f32)
z = (xc * (xc * f32(0.125)) - f32(1)).astype(f32)
g = (f32(0.5) * (xs + (np.abs(xs) - np.abs(xc))) + (xc * xc) * horner32(cp.astype(f32), z)).astype(f32)
d = (xc * horner32(cr.astype(f32), z) + f32(0.5)).astype(f32)
x64 = xs.astype(np.float64)
print('gelu  max |err|', np.abs(g - x64 * Phi(x64)).max(), ' inside |x| < 3.9:', np.abs(g - x64 * Phi(x64))[np.abs(xs) < 3.9].max())
print("gelu' max |err|", np.abs(d - (Phi(x64) + x64 * phi(x64))).max(), ' inside |x| < 3.9:',
      np.abs(d - (Phi(x64) + x64 * phi(x64)))[np.abs(xs) < 3.9].max())
