"""Randomised problem layouts shared by tests/ and tools/fuzz_layouts.py (data generators, not code under test)."""
import mcintegration_jl_amd as mci


def pipe_case(rng):
    """layouts the software-pipelined :vegas loop takes (mci_device.h pipe_eligible): Continuous pools only, 8..16 draws in all, grids small
    enough for LDS pair tables; ragged dof tables (padding probabilities), adapt on/off, 1-4 integrands"""
    while True:
        npool = int(rng.integers(1, 4))
        ni = int(rng.integers(1, 5))
        dof = [[int(rng.integers(0, 9)) for _ in range(npool)] for _ in range(ni)]
        for i in range(ni):
            if sum(dof[i]) == 0:
                dof[i][int(rng.integers(0, npool))] = 1
        maxdof = [max(d[v] for d in dof) for v in range(npool)]
        if 8 <= sum(maxdof) <= 16 and min(maxdof) > 0:
            break
    var, oleaves = [], []
    for v in range(npool):
        lo, hi = float(rng.uniform(-2, 0)), float(rng.uniform(0.5, 3))
        ninc = int(rng.choice([17, 100, 257, 1000]))
        alpha = float(rng.choice([1.0, 2.0, 3.0]))
        adapt = bool(rng.integers(0, 5) > 0)
        var.append(mci.Continuous(lo, hi, alpha=alpha, ninc=ninc, adapt=adapt))
        oleaves.append(dict(kind=0, pool=v, lower=lo, upper=hi, npts=ninc, alpha=alpha, adapt=adapt))
    draws = [(v, s) for v in range(npool) for s in range(maxdof[v])]
    lines = []
    for i in range(ni):
        own = [k for k, (v, s) in enumerate(draws) if s < dof[i][v]]
        coef = rng.uniform(0.2, 1.5, size=len(own))
        arg = " + ".join("%.6f * x[%d]" % (c, k) for c, k in zip(coef, own))
        sign = "-" if rng.integers(0, 4) == 0 else ""
        lines.append("w[%d] = %s(%.3f + 0.5 * cos(%s) + 0.05 * x[%d] * x[%d]);" % (i, sign, 0.4 + 0.3 * i, arg, own[0], own[-1]))
    return tuple(var), oleaves, dof, "\n".join(lines), len(draws)
