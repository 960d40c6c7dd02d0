#!/usr/bin/env python3
"""Generates tests/golden/golden.json.

The reference is pure Julia and `julia` is absent from the build image, so these vectors cannot
come from running the reference.  They are of two kinds:

 (A) literal known answers copied from the reference's own tests (data, not code):
       locate cases          /root/reference/test/utility.jl:2-9
       _maxdof case          /root/reference/test/utility.jl:14-15
       doReweight! fixpoint  /root/reference/test/mpi_test.jl:148-169
       hand grids            /root/reference/test/utility.jl:31-33
 (B) hand-derived vectors: an independent pure-Python evaluation (this file, written from the
     formulas at src/distribution/common.jl:43-82, src/distribution/variable.jl:206-239,:369-382,
     src/distribution/sampler.jl:293-305, src/main.jl:296-320, src/statistics.jl:186-220) on
     small inputs.  They pin the C oracle and the HIP/C++ product against a second implementation.
 (C) Philox4x32-10 known-answer vectors (Random123 kat_vectors).

Run:  python tests/golden/make_golden.py
"""
import json
import math
import os


def smooth(d, factor=6.0):
    n = len(d)
    if n <= 1:
        return list(d)
    new = list(d)
    new[0] = (d[0] * (factor + 1) + d[1]) / (factor + 2)
    new[-1] = (d[-1] * (factor + 1) + d[-2]) / (factor + 2)
    for i in range(1, n - 1):
        new[i] = (d[i - 1] + d[i] * factor + d[i + 1]) / (factor + 2)
    return new


def seqsum(v):
    s = 0.0
    for x in v:
        s += x
    return s


def rescale(d, alpha):
    if len(d) == 1:
        return list(d)
    s = seqsum(d)
    out = [x / s for x in d]
    for i, x in enumerate(out):
        if 0 < x <= 0.99999999:
            out[i] = (-(1 - x) / math.log(x)) ** alpha
    return out


def train_continuous(grid, hist, alpha):
    dist = rescale(smooth(hist, 6.0), alpha)
    n = len(grid)
    new = [0.0] * n
    new[0], new[-1] = grid[0], grid[-1]
    j, acc = 0, 0.0
    f_ninc = seqsum(dist) / (n - 1)
    for i in range(2, n):  # 1-based 2..n-1
        while acc < f_ninc:
            j += 1
            acc += dist[j - 1]
        acc -= f_ninc
        new[i - 1] = grid[j] - (acc / dist[j - 1]) * (grid[j] - grid[j - 1])
    new[-1] = grid[-1]
    return new


def train_discrete(hist, alpha):
    d = rescale(list(hist), alpha)
    s = seqsum(d)
    d = [x / s for x in d]
    acc, run = [0.0], 0.0
    for x in d:
        run += x
        acc.append(run)
    return d, acc


def map_draw(grid, y):
    N = len(grid) - 1
    iy = int(math.floor(y * N)) + 1
    dy = y * N - (iy - 1)
    x = grid[iy - 1] + dy * (grid[iy] - grid[iy - 1])
    prob = 1.0 / (N * (grid[iy] - grid[iy - 1]))
    return dict(y=y, x=x, gidx=iy, prob=prob)


def mean_std(obs_sum, obs_sq, block):
    mean = [s / block for s in obs_sum]
    std = []
    for q, m in zip(obs_sq, mean):
        v = (q / block - m * m) / (block - 1)
        std.append(math.sqrt(v) if v > 0 else 0.0)
    return mean, std


def average(means, stds, init, mx):
    if mx <= init:
        return means[0], stds[0], 0.0
    w = [1.0 / (stds[i - 1] + 1.0e-10) ** 2 for i in range(init, mx + 1)]
    d = [means[i - 1] for i in range(init, mx + 1)]
    ws = seqsum(w)
    mea = seqsum(di * wi / ws for di, wi in zip(d, w))
    chi2 = seqsum(wi * (di - mea) ** 2 for di, wi in zip(d, w))
    return mea, 1.0 / math.sqrt(ws), chi2 / ((mx - init + 1) - 1)


def main():
    out = {}
    # (A) literals from the reference's tests
    eps = 2.220446049250313e-16
    out["locate"] = dict(grid=[0.0, 0.1, 0.3, 0.5],
                         cases=[[eps, 1], [0.5 - eps, 3], [0.0, 1], [0.05, 1], [0.2, 2], [0.31, 3]])
    out["maxdof"] = dict(dof=[[1, 2, 3, 5], [3, 1, 2, 7], [2, 4, 1, 2]], expect=[3, 4, 3, 7])
    out["doreweight"] = dict(visited=[1, 2, 3, 4], goal=[1.0, 2.0, 3.0, 4.0], gamma=1.0, n_iterations=5,
                             reweight0=[0.25, 0.25, 0.25, 0.25], expect=[0.25, 0.25, 0.25, 0.25], rtol=1e-3)
    gx, gy = [0.0, 0.1, 0.4, 1.0], [0.0, 0.2, 0.6, 1.0]
    out["hand_grids"] = dict(X=gx, Y=gy, Z_bounds=[1, 6], dof=[[1, 1, 1], [2, 3, 3]])
    # (B) hand-derived
    hists = {
        "h5": [1.0, 4.0, 2.0, 0.5, 0.25],
        "h8_peaked": [1e-10, 1e-10, 3.0, 50.0, 7.0, 1e-10, 1e-10, 0.2],
        "h3": [0.3, 0.3, 0.4],
    }
    out["smooth"] = {k: dict(inp=v, out=smooth(v)) for k, v in hists.items()}
    out["rescale"] = {k + "_a%g" % a: dict(inp=v, alpha=a, out=rescale(v, a))
                      for k, v in hists.items() for a in (1.5, 2.0, 3.0)}
    grids = {
        "h5": [0.0, 0.2, 0.4, 0.6, 0.8, 1.0],
        "h8_peaked": [-1.0, -0.9, -0.5, 0.0, 0.1, 0.3, 0.7, 1.5, 2.0],
        "h3": gx,
    }
    out["train_continuous"] = {k + "_a%g" % a: dict(grid=grids[k], hist=hists[k], alpha=a,
                                                    out=train_continuous(grids[k], hists[k], a))
                               for k in hists for a in (2.0, 3.0)}
    out["train_discrete"] = {}
    for k, v in hists.items():
        d, acc = train_discrete(v, 2.0)
        out["train_discrete"][k] = dict(hist=v, alpha=2.0, distribution=d, accumulation=acc)
    ys = [0.0, 0.05, 0.3333333333333333, 0.5, 0.6666666666666666, 0.999999999999]
    out["map_draw"] = dict(X=[map_draw(gx, y) for y in ys], Y=[map_draw(gy, y) for y in ys])
    series1 = [0.12, 0.55, 0.31, 0.98, 0.44, 0.07, 0.63, 0.29, 0.81, 0.5]
    series2 = [1.5, -0.2, 0.7, 0.1, 0.9, 1.1, -0.6, 0.3, 0.0, 0.45]
    s = [seqsum(series1), seqsum(series2)]
    q = [seqsum(x * x for x in series1), seqsum(x * x for x in series2)]
    m, e = mean_std(s, q, 10)
    out["mean_std"] = dict(series=[series1, series2], obs_sum=s, obs_sq=q, block=10, mean=m, std=e)
    # docs/src/index.md:40-49 per-iteration table of the reference (statistical sample output, used
    # here only as INPUT data to pin average() against the table's own "wgt average" column)
    it_mean = [-3.8394711, -3.889894, -4.0258398, -4.0010193, -3.990754, -4.000744, -4.0021542, -3.9979708,
               -3.994137, -3.9999099]
    it_std = [0.12101621, 0.04161423, 0.016628525, 0.0097242712, 0.0055248673, 0.0025751679, 0.005940518,
              0.0034603885, 0.0026675679, 0.0033455927]
    doc_avg = [[-3.8394711, 0.12101621, 0.0], [-3.8394711, 0.12101621, 0.0], [-4.007122, 0.015441393, 9.2027],
               [-4.0027523, 0.0082285382, 4.6573], [-3.9944823, 0.0045868638, 3.5933],
               [-3.9992433, 0.0022454867, 3.0492], [-3.9996072, 0.0021004392, 2.4814],
               [-3.9991666, 0.0017955468, 2.0951], [-3.9975984, 0.0014895459, 2.1453],
               [-3.9979808, 0.0013607691, 1.9269]]
    out["average_docs_table"] = dict(iter_mean=it_mean, iter_std=it_std, init=2, printed=doc_avg,
                                     computed=[list(average(it_mean, it_std, 2, mx)) for mx in range(1, 11)])
    # (C) Philox4x32-10 KAT (Random123 kat_vectors: ctr[4] key[2] -> out[4])
    out["philox4x32_10"] = [
        dict(ctr=[0, 0, 0, 0], key=[0, 0], out=[0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        dict(ctr=[0xffffffff] * 4, key=[0xffffffff] * 2, out=[0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        dict(ctr=[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], key=[0xa4093822, 0x299f31d0],
             out=[0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    # Philox4x32-7 KAT (Random123 kat_vectors, the `philox4x32 7` lines): the opt-in cheaper stream, mci_set_rng_rounds(7)
    out["philox4x32_7"] = [
        dict(ctr=[0, 0, 0, 0], key=[0, 0], out=[0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
        dict(ctr=[0xffffffff] * 4, key=[0xffffffff] * 2, out=[0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
        dict(ctr=[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], key=[0xa4093822, 0x299f31d0],
             out=[0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
    ]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
