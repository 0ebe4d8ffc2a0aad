#!/usr/bin/env python3
"""find_pairs.py — searches for short (query, reference) pairs on which the oracle's output changes when ONE FO_RULE_* bit
is flipped and no other single bit reproduces that change: the pairs tests/golden/rule_pairs.in.tsv holds for the rules that
the random pairs of sw_pairs.tsv do not exercise (A.4 traceback priority, A.4 gap-tie).  Deterministic (seeded); the chosen
pairs are committed, this script is how they were found:  python tools/pin_kit/find_pairs.py > tests/golden/rule_pairs.in.tsv"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

BITS = (1, 2, 4, 8, 16, 32, 64)


def fields(res):
    return (res["score"], res["end_query"], res["end_ref"], res["beg_query"], res["beg_ref"], res["n_ops"], O.cigar_str(res["ops"][:16]))


def dview(f):  # what the D kit prints: end_query / end_ref / beg_query are -1 there
    return (f[0], f[4], f[5], f[6])


def under(q, r, off):
    return fields(O.sw(q, r, params=O.default_params(rules=0x7f & ~off)))


def main():
    O.build()
    rng = np.random.default_rng(20261005)
    want = {2: 8, 4: 8}
    found = {b: [] for b in want}

    def mut(s):
        s = list(s)
        for _ in range(int(rng.integers(1, 4))):
            k, p = int(rng.integers(0, 3)), int(rng.integers(0, len(s)))
            if k == 0:
                s[p] = str(rng.choice(list("ACGT")))
            elif k == 1:
                del s[p:p + int(rng.integers(1, 4))]
            else:
                s[p:p] = [str(c) for c in rng.choice(list("ACGT"), size=int(rng.integers(1, 4)))]
        return "".join(s)

    it = 0
    while any(len(found[b]) < 4 * want[b] for b in want) and it < 400000:
        it += 1
        alpha = "ACGT"[:int(rng.integers(2, 5))]
        core = "".join(rng.choice(list(alpha), size=int(rng.integers(12, 40))))
        q = mut(core)
        r = "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 6)))) + mut(core) + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 6))))
        if len(q) < 4 or len(r) < 4:
            continue
        base = fields(O.sw(q, r))
        for b in want:
            alt = under(q, r, b)
            if dview(alt) == dview(base):
                continue  # (must show in the D kit's columns too)
            if any(dview(under(q, r, o)) == dview(alt) for o in BITS if o != b):
                continue  # (another single switch gives the same line: this pair would not name the rule)
            found[b].append((q, r))
    print("#rule_bit\tquery\tref   (found by tools/pin_kit/find_pairs.py; outputs are in sw_pairs.tsv)")
    for b in want:
        for q, r in sorted(found[b], key=lambda x: len(x[0]) + len(x[1]))[:want[b]]:
            print("%d\t%s\t%s" % (b, q, r))


if __name__ == "__main__":
    main()
