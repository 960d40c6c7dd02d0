#!/usr/bin/env python3
"""The reference's flagship example -- the polarisation bubble of free electrons, example/bubble.jl = test/bubble.jl:12-133 -- written in
Python the way it is written in Julia (needs an MI355X): a struct of parameters in `userdata`, the external momentum looked up by the
Discrete draw, Green's functions with branches on sampled values, a histogram over the Discrete draw.  Both closures are traced into
the kernels (printed below); indices are 0-based, the Discrete draw runs 1 .. Qsize like the reference's."""
import math
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcintegration_jl_amd as mci
from mcintegration_jl_amd import Continuous, Discrete, integrate

PI = math.pi
rs, beta, spin, Qsize, dim, me = 1.0, 25.0, 2, 4, 3, 0.5                    # Para (test/bubble.jl:12-22)
kF = (9 * PI / (2 * spin)) ** (1 / 3) / rs
para = types.SimpleNamespace(kF=kF, beta=beta / (kF ** 2 / 2 / me), me=me, spin=spin, dim=dim, Qsize=Qsize,
                             extQ=[np.array([q, 0.0, 0.0]) for q in np.linspace(0.0, 1.5 * kF, Qsize)])


def lindhard(q, para):                                                       # :24-38  the zero-temperature closed form
    density = para.me * para.kF / (2 * PI ** 2)
    q = max(q, 1e-6)
    x = q / 2 / para.kF
    Pi = 1 + (1 - x ** 2) * math.log1p(4 * x / ((1 - x) ** 2)) / 4 / x if abs(q - 2 * para.kF) > 1e-6 else 1.0
    return -Pi * density * para.spin / 2


def green(tau, omega, beta):                                                 # :40-51
    if tau >= 0.0:
        return np.exp(-omega * tau) / (1 + np.exp(-omega * beta)) if omega > 0.0 else np.exp(omega * (beta - tau)) / (1 + np.exp(omega * beta))
    return -np.exp(-omega * (tau + beta)) / (1 + np.exp(-omega * beta)) if omega > 0.0 else -np.exp(-omega * tau) / (1 + np.exp(omega * beta))


def integrand(vars, config):                                                 # :53-78
    R, Theta, Phi, T, Ext = vars
    para = config.userdata
    kF, beta, me = para.kF, para.beta, para.me
    r = R[0] / (1 - R[0])
    theta, phi = Theta[0], Phi[0]
    k = np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])
    factor = 1.0 / (2 * PI) ** para.dim
    factor *= r ** 2 / (1 - R[0]) ** 2 * np.sin(theta)
    Tin, Tout = 0.0, T[0]
    q = para.extQ[Ext[0] - 1]                                                # external momentum
    kq = k + q
    tau = Tout - Tin
    g1 = green(tau, (np.dot(k, k) - kF ** 2) / (2 * me), beta)
    g2 = green(-tau, (np.dot(kq, kq) - kF ** 2) / (2 * me), beta)
    return g1 * g2 * para.spin * factor


def measure(vars, obs, weight, config):                                      # :84-88
    Ext = vars[-1]
    obs[0][Ext[0] - 1] += weight[0]


for alg in ("vegas", "vegasmc", "mcmc"):                                     # run(Steps, alg), :94-133
    f = (lambda idx, v, c: integrand(v, c)) if alg == "mcmc" else integrand
    m = (lambda idx, v, obs, w, c: measure(v, obs, [w], c)) if alg == "mcmc" else measure
    var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
           Continuous(0.0, para.beta, alpha=3.0), Discrete(1, Qsize, adapt=False))
    kw = dict(measure=m, userdata=para, var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(Qsize)], solver=alg, print=-1)
    result = integrate(f, neval=1e6, block=8, **kw)
    t0 = time.time()
    result = integrate(f, neval=1e8, block=64, niter=1, config=result.config, solver=alg, measure=m, print=-1)
    eng = result.config._engine
    print("Algorithm : %s   (%s integrand, %s measure; 1e8 evaluations in %.3f s)" % (alg, type(eng.integrand).__name__, type(eng.measure).__name__, time.time() - t0))
    print("%10s  %10s   %10s  %10s" % ("q/kF", "avg", "err", "T = 0"))
    for i, q in enumerate(para.extQ):
        print("%10.6f  %10.6f +- %10.6f  %10.6f" % (q[0] / kF, result.mean[0][i], result.stdev[0][i], lindhard(q[0], para)))
