"""The forced-diagonal shortcut of the device's selection step (fade_amd/csrc/fadehip_kernels.hpp select_one, DESIGN.md
§3.2), checked as a statement about the ORACLE alone (no GPU): walk back along the end cell's diagonal with P0 = score,
P(k+1) = P(k) - W(d_k); whenever the partial sums stay positive and reach exactly 0, the oracle's traceback must be
that gap-free diagonal — begin cell, op runs and all — under the default rules and under every A.4 variant (the
argument says the direction the traceback tries first is forced, whatever the tie rules are).  The converse is not
claimed (a gap-free path may also exist when the walk gives up; those go to the traced pass)."""
import numpy as np
import pytest

from helpers import concat, make_pairs

MATCH, MISMATCH = 2, -3
ALPHA = set(b"ACGTN")


def _w(a, b, n_eq_n=True):
    if a not in ALPHA or b not in ALPHA:
        return 0  # parasail's wildcard column / row
    if a == b:
        return MISMATCH if (a == ord("N") and not n_eq_n) else MATCH
    return MISMATCH


def _walk(q, r, score, end_q, end_r, n_eq_n=True, eq_by_char=True):
    """Returns (beg_q, beg_r, runs) when the walk is forced, else None.  runs: [(len, '=' | 'X'), ...] in CIGAR order."""
    p, k, runs = score, 0, []
    while end_q - k >= 0 and end_r - k >= 0:
        a, b = q[end_q - k], r[end_r - k]
        w = _w(a, b, n_eq_n)
        op = "=" if ((a == b) if eq_by_char else w > 0) else "X"
        if runs and runs[-1][1] == op:
            runs[-1][0] += 1
        else:
            runs.append([1, op])
        p -= w
        k += 1
        if p == 0:
            return end_q - k + 1, end_r - k + 1, [(n, o) for n, o in reversed(runs)]
        if p < 0:
            return None
    return None


RULES = [0x7f, 0x7f & ~2, 0x7f & ~4, 0x7f & ~(2 | 4), 0x7f & ~8, 0x7f & ~64, 0x7f & ~1]


@pytest.mark.parametrize("rules", RULES, ids=["default", "E_before_F", "ties_open", "both_A4", "eq_by_sign", "N_ne_N", "other_end_rule"])
def test_forced_diagonal_is_what_the_oracle_traces(oracle, rules):
    rng = np.random.default_rng(rules)
    qs, rs = make_pairs(rng, 2500, kinds=("planted", "related", "nrich", "iupac", "tandem", "homopolymer", "lowcomplexity", "random"))
    qc, qo = concat(qs)
    rc, ro = concat(rs)
    res, ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=64, params=oracle.default_params(rules=rules))
    forced = with_x = 0
    for k in range(len(qs)):
        score, end_q, end_r, beg_q, beg_r, n_ops = (int(x) for x in res[k])
        if score <= 0:
            continue
        got = _walk(bytes(qs[k]), bytes(rs[k]), score, end_q, end_r, n_eq_n=bool(rules & 64), eq_by_char=bool(rules & 8))
        if got is None:
            continue
        forced += 1
        bq, br, runs = got
        assert (bq, br) == (beg_q, beg_r), (k, got, res[k])
        cig = [(int(o) >> 4, "MIDNSHP=X"[int(o) & 15]) for o in ops[k][:min(n_ops, 64)]]
        lead = [(bq, "S")] if bq > 0 else []
        tail = [(len(qs[k]) - 1 - end_q, "S")] if end_q < len(qs[k]) - 1 else []
        assert cig == lead + runs + tail, (k, cig, lead + runs + tail)
        with_x += any(o == "X" for _, o in runs)
    assert forced > 500 and with_x > 50, (forced, with_x)
