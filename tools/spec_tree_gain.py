#!/usr/bin/env python3
"""Development tool (no GPU): what a WIDER group of lanes per chain would buy (a workgroup per chain, 256 lanes).
A trip of a group advances its chain by the number of tree nodes that lie on the path the chain takes; for a chain whose steps change the
configuration with probability q that is, in expectation, the sum over the tree's nodes of q^(accept edges) (1 - q)^(reject edges).  The
trees are built like the library builds them (csrc/mci_host_jit.h spec_build: the `lanes` most probable nodes for an assumed acceptance,
at most `limit` accept edges on a way through the tree -- every accept LEVEL costs a trip one more exchange of configurations); for each
true acceptance the best tree of the solver's family is taken, as a group's adaptation does (mci_spec.h spec_adapt).
usage: python tools/spec_tree_gain.py"""
import heapq

FAMILY = (0.03, 0.12, 0.3, 0.5, 0.7, 0.85, 0.93)      # mci_host_jit.h fam_vegasmc


def expected_steps(lanes, build_acc, limit, q):
    front, seq, tab = [(-1.0, 0, -1, False)], 1, []
    while len(tab) < lanes and front:
        negp, _, parent, via = heapq.heappop(front)
        nd = (0, 0, 0) if parent < 0 else (tab[parent][0] + 1, tab[parent][1] + (1 if via else 0), tab[parent][2] + (0 if via else 1))
        me = len(tab)
        tab.append(nd)
        heapq.heappush(front, (negp * (1.0 - build_acc), seq, me, False))
        seq += 1
        if limit < 0 or nd[1] + 1 <= limit:
            heapq.heappush(front, (negp * build_acc, seq, me, True))
            seq += 1
    return sum(q ** a * (1.0 - q) ** r for _, a, r in tab), max(d for d, _, _ in tab), max(a for _, a, _ in tab)


if __name__ == "__main__":
    print("expected chain steps per trip (best tree of the family; depth / accept levels of that tree)")
    print("%-16s %s" % ("true acceptance", "".join("%28s" % ("%d lanes" % n) for n in (16, 64, 256, 1024))))
    for limit, label in ((12, "<= 12 accept levels (the shipped cap)"), (-1, "no cap on accept levels")):
        print(label)
        for q in (0.94, 0.74, 0.46, 0.25, 0.05):
            row = []
            for lanes in (16, 64, 256, 1024):
                e, d, a = max(expected_steps(lanes, b, limit, q) for b in FAMILY)
                row.append("%10.2f  (%3d deep, %2d lv)" % (e, d, a))
            print("  %-14.2f %s" % (q, "".join("%28s" % r for r in row)))
