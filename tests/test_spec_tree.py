"""CPU tests of the speculation trees behind "several lanes per chain" (csrc/mci_spec.h; host side: mci_api.hip spec_build, exported as
mci_speculation_tree): the group's lanes are the most probable nodes of a chain's accept / reject outcome tree, numbered
ancestors-first, and the masks every lane carries select exactly one root-to-leaf path for any outcome of the accept tests."""
import ctypes as C
import itertools
import random

import numpy as np
import pytest

from mcintegration_jl_amd._lib import lib, MCIError, check


def tree(lanes, accept, max_accepts=-1):
    d = (C.c_int32 * lanes)()
    a = (C.c_int32 * lanes)()
    n = (C.c_int32 * lanes)()
    na = (C.c_uint64 * lanes)()
    nr = (C.c_uint64 * lanes)()
    check(lib().mci_speculation_tree(lanes, float(accept), max_accepts, d, a, n, na, nr))
    return [dict(depth=d[i], anc=a[i], nacc=n[i], needacc=int(na[i]), needrej=int(nr[i])) for i in range(lanes)]


def walk(t, ok):
    """the sequential chain over the tree: start at the root; an accepted node moves to its accept child, a rejected one to its reject
    child; returns the lanes visited"""
    child = {}
    for i, nd in enumerate(t):
        for j, other in enumerate(t):
            if other["depth"] == nd["depth"] + 1 and j != i:
                if other["needacc"] == nd["needacc"] | (1 << i) and other["needrej"] == nd["needrej"]:
                    child[(i, True)] = j
                if other["needacc"] == nd["needacc"] and other["needrej"] == nd["needrej"] | (1 << i):
                    child[(i, False)] = j
    path, cur = [], 0
    while cur is not None:
        path.append(cur)
        cur = child.get((cur, bool(ok[cur])))
    return path


@pytest.mark.parametrize("lanes", [2, 4, 8, 16, 32, 64])
def test_a_small_acceptance_gives_the_reject_chain(lanes):
    t = tree(lanes, 1e-3)
    for i, nd in enumerate(t):
        assert nd == dict(depth=i, anc=-1, nacc=0, needacc=0, needrej=(1 << i) - 1)


def test_one_half_gives_the_complete_binary_tree():
    t = tree(64, 0.5)
    depths = sorted(nd["depth"] for nd in t)
    assert depths == sorted(sum(([d] * 2 ** d for d in range(6)), [])) + [6]
    assert max(nd["nacc"] for nd in t) == 6 or max(nd["nacc"] for nd in t) == 5
    t = tree(64, 0.5, 2)
    assert max(nd["nacc"] for nd in t) == 2


@pytest.mark.parametrize("lanes,accept,limit", [(64, 0.5, -1), (64, 0.35, 2), (64, 0.2, 1), (32, 0.7, -1), (16, 0.5, -1), (8, 0.3, 1), (64, 0.9, 3), (4, 0.5, -1), (2, 0.5, -1)])
def test_masks_select_the_path_of_the_sequential_chain(lanes, accept, limit):
    t = tree(lanes, accept, limit)
    # ancestors first; the root is lane 0; an accept edge's ancestor is the nearest one
    assert t[0] == dict(depth=0, anc=-1, nacc=0, needacc=0, needrej=0)
    for i, nd in enumerate(t):
        assert (nd["needacc"] | nd["needrej"]) < (1 << i) and nd["needacc"] & nd["needrej"] == 0
        assert bin(nd["needacc"] | nd["needrej"]).count("1") == nd["depth"] and bin(nd["needacc"]).count("1") == nd["nacc"]
        assert nd["anc"] == (nd["needacc"].bit_length() - 1 if nd["needacc"] else -1)
        if limit >= 0:
            assert nd["nacc"] <= limit
    rng = random.Random(lanes * 1000 + int(accept * 100))
    for trial in range(300):
        p = rng.choice([0.02, 0.2, 0.5, 0.8, 0.98])
        ok = [rng.random() < p for _ in range(lanes)]
        okm = sum(1 << i for i in range(lanes) if ok[i])
        onpath = [i for i, nd in enumerate(t) if (okm & nd["needacc"]) == nd["needacc"] and (okm & nd["needrej"]) == 0]
        assert onpath == walk(t, ok)
        # the path is a chain of ancestors: depths 0, 1, 2, ... and the deepest lane is the highest one
        assert [t[i]["depth"] for i in onpath] == list(range(len(onpath))) and onpath[-1] == max(onpath)


def test_bad_arguments_are_refused():
    with pytest.raises(MCIError):
        tree(65, 0.5)
    with pytest.raises(MCIError):
        tree(8, 1.5)
