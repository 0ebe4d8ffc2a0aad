#!/usr/bin/env python3
"""compare.py <kit output .tsv> — lines of a pin-kit run against tests/golden/sw_pairs.tsv (what the oracle under the
default rules gives), and for each line that differs the single FO_RULE_* switch, if any, under which the oracle gives
the kit's line.  Needs the oracle built (make -C oracle).  Columns printed as -1 by the kit are skipped."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

RULES = {1: "END_MIN_REF_THEN_QUERY (A.3)", 2: "HDIR_DIAG_F_E (A.4)", 4: "GAP_TIE_EXTENDS (A.4)", 8: "EQ_BY_CHAR (A.4)",
         16: "SAM_GAP_LETTERS (A.5)", 32: "PAD_SOFTCLIP (A.6)", 64: "N_MATCHES_N (A.1)"}


def fields(res):
    return [res["score"], res["end_query"], res["end_ref"], res["beg_query"], res["beg_ref"], res["n_ops"], O.cigar_str(res["ops"][:16])]


def main():
    kit = [l.rstrip("\n").split("\t") for l in open(sys.argv[1]) if not l.startswith("#")]
    O.build()
    bad = 0
    for f in kit:
        q, r = f[0], f[1]
        got = [int(x) for x in f[2:8]] + [f[8]]
        exp = fields(O.sw(q, r))
        same = all(g == e for g, e in zip(got, exp) if g != -1)
        if same:
            continue
        bad += 1
        fix = [name for bit, name in RULES.items()
               if all(g == e for g, e in zip(got, fields(O.sw(q, r, params=O.default_params(rules=0x7f & ~bit)))) if g != -1)]
        print("DIFFERS  q=%s.. r=%s..  kit=%s  oracle=%s  reconciled by switching off: %s" % (q[:24], r[:24], got, exp, ", ".join(fix) or "no single switch"))
    print("%d of %d lines differ from the oracle under the default rules" % (bad, len(kit)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
