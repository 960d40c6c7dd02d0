"""The arithmetic claim behind train!'s serial walk on the device (mci_train.h train_leaf, DESIGN.md section 5 "Refinement walk"), checked
on the CPU in plain IEEE doubles: the reference's recurrence (variable.jl:227-234, bin-major)

    rec[j] = acc_f;   while acc_f >= f_ninc: acc_f -= f_ninc;   acc_f += avg_f[j + 1]

equals a list of SLOTS  acc_f += e[s]  built from decisions c[j] = floor(C[j] / f_ninc) - floor(C[j-1] / f_ninc) taken from prefix sums
C[] of ANY summation order -- per bin c[j] times -f_ninc, then +avg_f[j + 1], the last subtraction merged with the addition into
avg_f[j + 1] - f_ninc where f_ninc / 2 <= avg_f[j + 1] <= 2 f_ninc (both roundings it replaces are exact: Sterbenz) -- WHENEVER
every decision agrees with the count `while rec[j] >= f_ninc` taken from the slots' own record; and that disagreements are rare
(exact ties).  This is a restatement of the device algorithm for the test, not the product path."""
import math

import numpy as np
import pytest


def recurrence(d, f):
    n = len(d)
    rec, acc = [], 0.0 + d[0]
    for j in range(n):
        rec.append(acc)
        while acc >= f:
            acc -= f
        acc += d[j + 1] if j + 1 < n else 0.0
    return rec


def slots(d, f, csum):
    n = len(d)
    inv = 1.0 / f
    c = []
    for j in range(n):
        if j >= n - 1:
            c.append(0)
            continue
        v = math.floor(csum[j] * inv) - (math.floor(csum[j - 1] * inv) if j > 0 else 0)
        c.append(int(v) if v > 0 else 0)
    e, first = [], []
    for j in range(n):
        first.append(len(e))
        nxt = d[j + 1] if j + 1 < n else 0.0
        e.extend([-f] * max(c[j] - 1, 0))
        if c[j] >= 1 and 0.5 * f <= nxt <= 2.0 * f:
            e.append(nxt - f)
        else:
            if c[j] >= 1:
                e.append(-f)
            e.append(nxt)
    acc, vals = 0.0 + d[0], []
    for x in e:
        vals.append(acc)
        acc = acc + x
    rec = [vals[s] for s in first]
    ok = True
    for j in range(n - 1):
        k, a = 0, rec[j]
        while a >= f:
            a -= f
            k += 1
        ok = ok and k == c[j]
    return rec, ok, len(e)


def histograms(rng, n):
    """rescaled distributions the way train! meets them: nearly flat (a converged grid), a slope, heavy tails, narrow peaks, tiny bins"""
    kind = rng.integers(0, 6)
    x = np.linspace(0.0, 1.0, n)
    if kind == 0:
        d = 1.0 + 1e-3 * rng.standard_normal(n)
    elif kind == 1:
        d = 1.0 + rng.uniform(-0.5, 0.5) * x + 1e-2 * rng.standard_normal(n)
    elif kind == 2:
        d = np.exp(rng.normal(0.0, 2.0, n))
    elif kind == 3:
        d = 1e-6 + np.exp(-((x - rng.uniform()) * rng.choice([30.0, 300.0, 3000.0])) ** 2)
    elif kind == 4:
        d = rng.choice([1e-12, 1.0, 3.0, 40.0], size=n, p=[0.3, 0.5, 0.15, 0.05]) * rng.uniform(0.9, 1.1, n)
    else:
        d = np.abs(rng.standard_cauchy(n)) + 1e-9
    return np.abs(d).astype(np.float64) * float(rng.choice([1e-8, 1.0, 1e5]))


@pytest.mark.parametrize("n", [2, 17, 100, 999, 1025])
def test_slots_with_agreeing_decisions_are_the_recurrence_bit_for_bit(n):
    rng = np.random.default_rng(1000 + n)
    held = 0
    cases = 120 if n <= 100 else 40
    for case in range(cases):
        d = histograms(rng, n)
        f = float(np.sum(d)) / n                       # (any f_ninc > 0 will do for the claim; the device uses Julia's sum order)
        order = rng.integers(0, 3)                     # prefix sums in different association orders: the decisions may differ, the claim holds
        csum = np.cumsum(d) if order == 0 else np.cumsum(d[::-1])[::-1][0] - np.concatenate([np.cumsum(d[::-1])[::-1][1:], [0.0]]) if order == 1 \
            else np.cumsum(d.astype(np.longdouble)).astype(np.float64)
        ref = recurrence(list(d), f)
        rec, ok, nslot = slots(list(d), f, list(csum))
        assert nslot <= 2 * n - 1 or n == 2
        if ok:
            held += 1
            assert rec == ref, (n, case)
    assert held >= 0.9 * cases, (held, cases)          # decisions fail on exact ties only


def test_an_exact_tie_is_caught_by_the_check():
    """a flat histogram: acc_f == f_ninc at every bin, the decisions taken from prefix sums with a rounding error in them disagree
    with the record somewhere -- the check says so (the device then walks the general form)"""
    n = 100
    d = [0.1] * n
    f = 0.1
    csum = [0.1 * (j + 1) * (1.0 - 1e-15) for j in range(n)]   # prefix sums a hair low: floor() loses a point at every bin
    ref = recurrence(d, f)
    rec, ok, _ = slots(d, f, csum)
    assert not ok or rec == ref
    csum = list(np.cumsum(np.array(d)))
    rec, ok, _ = slots(d, f, csum)
    assert not ok or rec == ref
