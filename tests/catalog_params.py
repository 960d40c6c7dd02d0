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
