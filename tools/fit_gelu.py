#!/usr/bin/env python3
"""Minimax fit of the packed GELU used by the GEMM epilogues (crh_encoder.hip: gelu_erf2): x / (1 + 2^q(x)), q odd of degree 9,
fitted with |x| clamped to 8 inside q; the kernel evaluates it unclamped, which is the same function on the fit range and runs
to the right limits beyond it (q(x)/x < 0 everywhere: checked at the end).  Prints the f32 coefficients (c1, c3, c5, c7, c9) and the error of an f32 evaluation."""
import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
x = np.linspace(-10, 10, 80001)
g = x*0.5*(1+erf(x/np.sqrt(2)))
def model(c, x, f32=False):
    dt = np.float32 if f32 else np.float64
    x = x.astype(dt); c = np.asarray(c, dtype=dt)
    xc = np.clip(x, dt(-8), dt(8))
    x2 = xc*xc
    h = np.full_like(x, c[-1])
    for ck in c[-2::-1]:
        h = h*x2 + ck
    q = h*xc
    with np.errstate(over='ignore'):
        e = np.exp2(q) + dt(1)
    return x*(dt(1)/e)
c = -np.log2(np.e)*np.array([1.59565628e+00, 7.29375740e-02, -2.49721000e-04, -6.11622809e-05, 2.23817605e-06])
w = np.ones_like(x)
for it in range(80):
    r = least_squares(lambda c: w*(model(c,x)-g), c, method='lm', xtol=1e-15, ftol=1e-15)
    c = r.x
    e = np.abs(model(c,x)-g)
    w = w*(1+2*e/e.max()); w /= w.mean()
c32 = c.astype(np.float32)
print([float(v) for v in c32])
xx = np.linspace(-12, 12, 2000001)
gg = xx*0.5*(1+erf(xx/np.sqrt(2)))
e = np.abs(model(c32, xx, f32=True).astype(np.float64) - gg)
print("max abs err f32 eval %.3e at %.3f" % (e.max(), xx[e.argmax()]))
rel = e/np.maximum(np.abs(gg), 1e-30)
m = np.abs(gg) > 1e-3
print("max rel err where |gelu|>1e-3: %.3e" % rel[m].max())

# the unclamped polynomial keeps its sign: h(t) = q(x)/x as a function of t = x^2 is negative for all t >= 0
t = np.linspace(0, 1e4, 2000001)
h = np.zeros_like(t) + float(c32[-1])
for ck in c32[-2::-1]:
    h = h * t + float(ck)
print("max of q(x)/x over x^2 in [0, 1e4]: %.4f (must be < 0)" % h.max())
