"""CPU: fits the polynomial of csrc/pf_common.h gelu_erf (erf(a) = 1 - 2^(-a P(a)), a = min(|x| / sqrt 2, 4)) and prints the float32-evaluated GELU's
maximum error against float64 beside that of a correctly rounded float32 erf.  usage: python tools/fit_erf.py"""
import numpy as np
from scipy.special import erf, erfc
from numpy.polynomial import chebyshev as C
# g(a) = -log2(1 - erf(a)) / a  on (0, A]
A = 4.0
a = np.cos(np.pi*(np.arange(4000)+0.5)/4000)*A/2 + A/2
g = -np.log2(erfc(a))/a
for deg in (6,7,8,9):
    ch = C.Chebyshev.fit(a, g, deg, domain=[0,A])
    pol = ch.convert(kind=np.polynomial.Polynomial)
    co = pol.coef.astype(np.float32)
    # float32 evaluation
    x = np.linspace(-6, 6, 2000001).astype(np.float32)
    ax = np.minimum(np.abs(x)*np.float32(0.70710678), np.float32(A)).astype(np.float32)
    p = np.full_like(ax, co[-1])
    for c in co[-2::-1]:
        p = (p*ax + c).astype(np.float32)
    e = (np.float32(1) - np.exp2((-(p*ax)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    e = np.copysign(e, x)
    gel = (np.float32(0.5)*x*(np.float32(1)+e)).astype(np.float32)
    ref = 0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
    ref32 = (np.float32(0.5)*x*(np.float32(1)+erf(x.astype(np.float64)*np.float64(np.float32(0.70710678))).astype(np.float32))).astype(np.float32)
    err = np.abs(gel-ref)
    err32 = np.abs(ref32.astype(np.float64)-ref)
    print(deg, "max abs err fast", err.max(), "at", x[err.argmax()], " f32-erf path", err32.max(), " rel-to-max(|x|,1e-3):", (err/np.maximum(np.abs(ref),1e-3)).max(), (err32/np.maximum(np.abs(ref),1e-3)).max())
    print("   coefs", [float(c) for c in co])
