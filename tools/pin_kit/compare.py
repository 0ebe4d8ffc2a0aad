#!/usr/bin/env python3
"""compare.py <kit output .tsv> — lines of a pin-kit run against tests/golden/sw_pairs.tsv (what the oracle under the
default rules gives), and for each line that differs the single FO_RULE_* switch, if any, under which the oracle gives
the kit's line.  Needs the oracle built (make -C oracle).  Columns printed as -1 by the kit are skipped.

One run decides every switch: tests/test_golden.py::test_one_pin_kit_run_decides_every_rule checks, for each of the seven
FO_RULE_* bits, that at least five of the committed pairs change under the flip IN THE COLUMNS THE D KIT PRINTS and that
this script then names exactly that bit for them; the eighth assumption (dhtslib's Cigar.alignedLength) is the probe line of
pin_dparasail.d (22 or 24)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

RULES = {1: "END_MIN_REF_THEN_QUERY (A.3)", 2: "HDIR_DIAG_F_E (A.4)", 4: "GAP_TIE_EXTENDS (A.4)", 8: "EQ_BY_CHAR (A.4)",
         16: "SAM_GAP_LETTERS (A.5)", 32: "PAD_SOFTCLIP (A.6)", 64: "N_MATCHES_N (A.1)"}


def fields(res):
    return [res["score"], res["end_query"], res["end_ref"], res["beg_query"], res["beg_ref"], res["n_ops"], O.cigar_str(res["ops"][:16])]


def same(got, exp):
    return all(g == e for g, e in zip(got, exp) if g != -1)


def diagnose(kit_rows):
    """kit_rows: [query, ref, score, end_query, end_ref, beg_query, position, n_ops, cigar] (strings, as in the file).
    Returns one entry per row that differs from the oracle under the default rules: (query, ref, kit fields, oracle
    fields, [rule bits whose flip alone reproduces the kit's line])."""
    out = []
    for f in kit_rows:
        q, r = f[0], f[1]
        got = [int(x) for x in f[2:8]] + [f[8]]
        exp = fields(O.sw(q, r))
        if same(got, exp):
            continue
        fix = [bit for bit in RULES if same(got, fields(O.sw(q, r, params=O.default_params(rules=0x7f & ~bit))))]
        out.append((q, r, got, exp, fix))
    return out


def main():
    kit = [l.rstrip("\n").split("\t") for l in open(sys.argv[1]) if not l.startswith("#")]
    O.build()
    diffs = diagnose(kit)
    votes = {}
    for q, r, got, exp, fix in diffs:
        print("DIFFERS  q=%s.. r=%s..  kit=%s  oracle=%s  reconciled by switching off: %s" %
              (q[:24], r[:24], got, exp, ", ".join(RULES[b] for b in fix) or "no single switch"))
        if len(fix) == 1:
            votes[fix[0]] = votes.get(fix[0], 0) + 1
    print("%d of %d lines differ from the oracle under the default rules" % (len(diffs), len(kit)))
    for bit, n in sorted(votes.items()):
        print("  -> %d lines are reconciled by switching off %s and by no other single switch" % (n, RULES[bit]))
    return 1 if diffs else 0


if __name__ == "__main__":
    sys.exit(main())
