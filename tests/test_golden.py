"""Committed golden fixtures (tests/golden/, made by make_golden.py from the scalar oracle).

CPU: the oracle still reproduces them (pins the oracle against accidental drift).
GPU: the HIP path reproduces them through the C ABI, reading the same SAM + FASTA text a user would."""
import os

import numpy as np
import pytest

import samutil

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["anno_c1", "anno_c2", "anno_c5", "anno_floor0"]


def _load_sw():
    rows = []
    for line in open(os.path.join(GOLD, "sw_pairs.tsv")):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        rows.append((f[0], f[1], tuple(int(x) for x in f[2:8]), f[8]))
    return rows


def _load_case(tag):
    names, seqs = samutil.read_fasta(open(os.path.join(GOLD, tag + ".fa")).read())
    cn, cl, batch, qnames = samutil.sam_to_batch(open(os.path.join(GOLD, tag + ".sam")).read())
    assert cn == names and cl == [len(s) for s in seqs]
    exp, params = [], {}
    for line in open(os.path.join(GOLD, tag + ".expected.tsv")):
        if line.startswith("#floor_len"):
            params = dict(kv.split("=") for kv in line[1:].split())
            continue
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        exp.append((f[0], int(f[1]), int(f[2]), f[3], f[4], f[5], f[6]))
    return names, seqs, batch, qnames, exp, int(params["floor_len"]), int(params["window"])


def test_oracle_reproduces_golden_sw(oracle):
    rows = _load_sw()
    assert len(rows) == 226  # 210 random / adversarial pairs + 16 that exercise the A.4 rules (rule_pairs.in.tsv)
    for q, r, nums, cig in rows:
        res = oracle.sw(q, r)
        assert (res["score"], res["end_query"], res["end_ref"], res["beg_query"], res["beg_ref"], res["n_ops"]) == nums
        assert oracle.cigar_str(res["ops"][:16]) == cig


def test_one_pin_kit_run_decides_every_rule(oracle):
    """tools/pin_kit: a maintainer with libparasail / dparasail prints sw_pairs.tsv's columns for the committed pairs and
    compare.py names the FO_RULE_* switch a mismatch points to.  That run must DECIDE every switch (SURVEY Appendix A.1,
    A.3-A.6): for each of the seven bits, at least five committed pairs change under the flip — in the columns the D kit
    prints (score, position, n_ops, cigar; end_query / end_ref / beg_query are -1 there) — and compare.py, fed the oracle's
    output under the flipped rule as if it were the kit's, names exactly that bit for them and no other.  (The eighth
    assumption, dhtslib's Cigar.alignedLength, is the probe line of pin_dparasail.d: 22 or 24.)"""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(GOLD))
    spec = importlib.util.spec_from_file_location("pin_compare", os.path.join(root, "tools", "pin_kit", "compare.py"))
    cmp_ = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cmp_)
    rows = _load_sw()
    for bit in cmp_.RULES:
        flipped = oracle.default_params(rules=0x7f & ~bit)
        for d_kit in (False, True):
            kit = []
            for q, r, nums, cig in rows:
                f = cmp_.fields(oracle.sw(q, r, params=flipped))
                if d_kit:
                    f[1] = f[2] = f[3] = -1
                kit.append([q, r] + [str(x) for x in f[:6]] + [f[6]])
            diffs = cmp_.diagnose(kit)
            named = [d for d in diffs if d[4] == [bit]]
            assert len(named) >= 5, (bit, d_kit, len(diffs), len(named))
            assert all(bit in d[4] for d in diffs), (bit, d_kit)  # every line that moved is explained by this flip


@pytest.mark.parametrize("tag", CASES)
def test_oracle_reproduces_golden_annotate(oracle, tag):
    names, seqs, batch, qnames, exp, floor_len, window = _load_case(tag)
    G = oracle.GenomeHolder(names, seqs)
    reads, keep = oracle.make_reads(batch)
    n_art = 0
    for i, e in enumerate(exp):
        a = oracle.annotate_one(G, reads[i], floor_len, window)
        got = (qnames[i], int(batch["flag"][i]), a["rs"]) + tuple((a[k] if a["has_tags"] else "") for k in
                                                                 ("am", "as_", "ar", "ab"))
        assert got == e
        n_art += a["has_tags"]
    assert n_art >= 5


@pytest.mark.gpu
def test_gpu_reproduces_golden_sw(ctx):
    from fade_amd.api import cigar_str
    rows = _load_sw()
    out = ctx.sw_batch([r[0] for r in rows], [r[1] for r in rows])
    for k, (q, r, nums, cig) in enumerate(rows):
        g = out[k]
        assert tuple(int(g[f]) for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops")) == nums, k
        assert cigar_str(g["ops"][:min(nums[5], 16)]) == cig, k


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_gpu_reproduces_golden_annotate(ctx, tag):
    from fade_amd import format_tags
    names, seqs, batch, qnames, exp, floor_len, window = _load_case(tag)
    ctx.genome_upload(names, [s.encode() for s in seqs])
    rs, aln, stats = ctx.annotate(batch, floor_len, window)
    tags = format_tags(batch, names, rs, aln)
    for i, e in enumerate(exp):
        t = tags.get(i)
        got = (qnames[i], int(batch["flag"][i]), int(rs[i])) + ((t["am"], t["as_"], t["ar"], t["ab"]) if t else
                                                                ("", "", "", ""))
        assert got == e, i
    assert int(stats[0]) == len(exp)
