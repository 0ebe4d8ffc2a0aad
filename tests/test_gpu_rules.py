"""The rule switches of ABI 2 (fadehip_params.rules, include/fadehip.h FADEHIP_RULE_*): every assumption about the
un-vendored libparasail / dparasail behaviour is a device-side switch with the same bit as the
oracle's FO_RULE_*.  Each non-default setting is compared with the oracle run under the same setting, at the SW seam
(level 1: score, end cell, begin, CIGAR) and through annotateTask (level 2: rs, am)."""
import numpy as np
import pytest

import fade_amd
from fade_amd import format_tags, synth
from helpers import concat, make_pairs

pytestmark = pytest.mark.gpu

END_MIN_REF, HDIR_F_E, TIE_EXTENDS, EQ_BY_CHAR, SAM_LETTERS, PAD_S, N_EQ_N = (1 << k for k in range(7))
DEFAULT = 0x7f
SETTINGS = [
    ("end_cell_first_in_row_major_order", DEFAULT & ~END_MIN_REF),
    ("traceback_prefers_E_over_F", DEFAULT & ~HDIR_F_E),
    ("gap_ties_open", DEFAULT & ~TIE_EXTENDS),
    ("eq_by_matrix_sign", DEFAULT & ~EQ_BY_CHAR),
    ("swapped_gap_letters", DEFAULT & ~SAM_LETTERS),
    ("no_softclip_padding", DEFAULT & ~PAD_S),
    ("n_mismatches_n", DEFAULT & ~N_EQ_N),
    ("all_of_A3_A4_A5_flipped", DEFAULT & ~(END_MIN_REF | HDIR_F_E | TIE_EXTENDS | EQ_BY_CHAR | SAM_LETTERS)),
]


def test_rule_bits_match_the_oracles(oracle):
    import re
    hdr = open(oracle.__file__.replace("pyoracle.py", "fade_oracle.h")).read()
    inc = open(fade_amd._lib.HERE + "/../include/fadehip.h").read()
    fo = dict(re.findall(r"FO_RULE_(\w+) = 1u << (\d)", hdr))
    fh = dict(re.findall(r"FADEHIP_RULE_(\w+) = 1u << (\d)", inc))
    assert fo == fh and len(fo) == 7


@pytest.mark.parametrize("name,rules", SETTINGS, ids=[s[0] for s in SETTINGS])
def test_level1_under_rule(oracle, name, rules):
    rng = np.random.default_rng(rules)
    # ties are what the A.3 / A.4 rules decide: homopolymers, tandem repeats and related pairs with gaps make many
    qs, rs = make_pairs(rng, 1400, kinds=("random", "planted", "homopolymer", "tandem", "nrich", "iupac", "related"))
    qs2, rs2 = make_pairs(rng, 300, lq_range=(120, 260), lr_range=(300, 900), kinds=("tandem", "related", "homopolymer"))
    qs3, rs3 = make_pairs(rng, 4000, kinds=("lowcomplexity",))
    qc, qo = concat(qs + qs2 + qs3)
    rc, ro = concat(rs + rs2 + rs3)
    c = fade_amd.Context(device=0, rules=rules)
    try:
        got = c.sw_batch_packed(qc, qo, rc, ro)
    finally:
        c.close()
    exp, exp_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16, params=oracle.default_params(rules=rules))
    differs_from_default = 0
    dflt, dflt_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16, striped=True)
    for k in range(len(qo) - 1):
        g = tuple(int(got[k][f]) for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops"))
        assert g == tuple(int(x) for x in exp[k]), (name, k, g, exp[k])
        m = min(int(exp[k][5]), 16)
        assert list(got[k]["ops"][:m]) == list(exp_ops[k][:m]), (name, k)
        differs_from_default += (tuple(exp[k]) != tuple(dflt[k])) or list(exp_ops[k][:m]) != list(dflt_ops[k][:m])
    assert differs_from_default > 0, "the inputs never exercise rule %s" % name


@pytest.mark.parametrize("name,rules", SETTINGS, ids=[s[0] for s in SETTINGS])
def test_level2_under_rule(oracle, name, rules):
    g = synth.Genome(2, 300_000, 9)
    # low-complexity inserts so that ties happen inside real windows
    parts = [synth.make_reads(g, 3000, 5, read_len=150, window=100, p_sc=0.6, clip_min=4, clip_max=60, p_sub=0.01),
             synth.make_reads(g, 1500, 6, read_len=250, window=300, p_sc=0.6, clip_min=4, clip_max=70, p_sub=0.02)]
    seqs = [a.tobytes().decode() for a in g.ascii_contigs()]
    c = fade_amd.Context(device=0, rules=rules)
    try:
        c.genome_upload(g.names, g.ascii_contigs())
        for b, w in zip(parts, (100, 300)):
            rs, aln, stats = c.annotate(b, 5, w)
            tags = format_tags(b, g.names, rs, aln)
            G = oracle.GenomeHolder(g.names, seqs)
            ors, oam = oracle.annotate_batch_soa(G, b, 5, w, threads=8, params=oracle.default_params(rules=rules))
            assert np.array_equal(rs, ors), (name, np.nonzero(rs != ors)[0][:10])
            for i in range(len(ors)):
                if oam[i] is None:
                    assert i not in tags, (name, i)
                else:
                    assert tags[i]["am"] == oam[i], (name, i)
            if not rules & PAD_S:
                assert not (rs & 6).any()  # analysis.d:78-80 / 102-104 need S ops in the result
            else:
                assert (rs & 6).any()
    finally:
        c.close()


def test_rules_are_refused_where_they_do_not_exist(monkeypatch):
    with pytest.raises(fade_amd.FadeHipError):
        fade_amd.Context(device=0, rules=0x80)  # unknown bit
    monkeypatch.setenv("FADEHIP_KERNEL", "pk")
    with pytest.raises(fade_amd.FadeHipError) as e:
        fade_amd.Context(device=0, rules=DEFAULT & ~HDIR_F_E)  # the single-pass A/B kernels carry the default rules only
    assert e.value.code == -5
