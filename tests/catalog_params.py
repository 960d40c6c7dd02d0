"""Parameter sets shared by the tests (data, not code under test)."""
import math

import numpy as np


def bubble_para():
    # example/bubble.jl:10-22
    rs, beta, spin, Qsize, dim, me = 1.0, 25.0, 2, 4, 3, 0.5
    kF = (9 * math.pi / (2 * spin)) ** (1.0 / 3) / rs
    extQ = [q for q in np.linspace(0.0 * kF, 1.5 * kF, Qsize)]
    beta_s = beta / (kF ** 2 / 2 / me)
    return dict(kF=kF, beta=beta_s, me=me, spin=spin, dim=dim, Qsize=Qsize, extQ=extQ)


def bubble_userdata():
    p = bubble_para()
    return [p["kF"], p["beta"], p["me"], float(p["spin"]), float(p["dim"]), float(p["Qsize"])] + list(p["extQ"])


def lindhard(q, p):
    # example/bubble.jl:24-36
    me, kF, spin = p["me"], p["kF"], p["spin"]
    density = me * kF / (2 * math.pi ** 2)
    if q < 1e-6:
        q = 1e-6
    x = q / 2 / kF
    if abs(q - 2 * kF) > 1e-6:
        Pi = 1 + (1 - x ** 2) * math.log1p(4 * x / ((1 - x) ** 2)) / 4 / x
    else:
        Pi = 1.0
    return -Pi * density * spin / 2


def bubble_exact():
    p = bubble_para()
    return [lindhard(q, p) for q in p["extQ"]]


def bubble_exact_finite_T(nk=1000, nc=500, kmax=5.0):
    """The static polarisation the bubble integrand actually integrates to: example/bubble.jl:24-36 is the T = 0 closed form,
    the integrand (bubble.jl:38-75) runs at beta*EF = 25 with mu = EF.  Pi(q) = spin * int d^3k/(2pi)^3 (f(w1)-f(w2))/(w1-w2),
    Gauss-Legendre in (k, cos theta); converged to 1e-10 at the default orders (differs from T = 0 by ~7e-5..1.5e-4)."""
    p = bubble_para()
    kF, beta, me, spin = p["kF"], p["beta"], p["me"], p["spin"]
    f = lambda w: 0.5 * (1 - np.tanh(0.5 * beta * w))
    df = lambda w: -beta * 0.25 / np.cosh(np.clip(0.5 * beta * w, -300, 300)) ** 2
    kk, wk = np.polynomial.legendre.leggauss(nk)
    cc, wc = np.polynomial.legendre.leggauss(nc)
    k = 0.5 * kmax * kF * (kk + 1)
    K, Cc = np.meshgrid(k, cc, indexing="ij")
    out = []
    for q in p["extQ"]:
        w1 = (K * K - kF * kF) / (2 * me)
        w2 = (K * K + 2 * K * q * Cc + q * q - kF * kF) / (2 * me)
        d = w1 - w2
        small = np.abs(d) < 1e-7
        val = np.where(small, df(0.5 * (w1 + w2)), (f(w1) - f(w2)) / np.where(small, 1.0, d))
        out.append(float(np.einsum("i,j,ij->", 0.5 * kmax * kF * wk * k * k, wc, val) * 2 * math.pi / (2 * math.pi) ** 3 * spin))
    return out


def genz_userdata(D=32, a=5.0):
    u = [0.3 + 0.4 * i / (D - 1) for i in range(D)]
    return [float(D), a] + u


def genz_exact(D=32, a=5.0):
    u = [0.3 + 0.4 * i / (D - 1) for i in range(D)]
    out = 1.0
    for ui in u:
        out *= a * (math.atan(a * (1 - ui)) + math.atan(a * ui))
    return out


def gaussian_exact(D=16, L=math.sqrt(50.0)):
    return math.erf(L / math.sqrt(2.0)) ** D
