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


@pytest.mark.parametrize("rules", [DEFAULT, DEFAULT & ~END_MIN_REF], ids=["default", "end_min_query_then_ref"])
def test_full_length_matches_at_each_row_class(oracle, rules):
    """The largest scores each key layout of the score pass has to hold (fadehip_kernels.hpp: 64 * score + 6 bits for
    row classes <= 14, 32 * score + 5 bits and 16-step windows for 16 .. 24, one key per row for 32): the whole query
    matches the window exactly, at window ends that fall on every position of a key window, twice (a tie between two
    end cells of the same score), under both A.3 rules."""
    rng = np.random.default_rng(2024)
    qs, rs = [], []
    for lq in (150, 160, 176, 192, 208, 223, 224, 225, 240, 255, 256, 257, 300, 319, 320, 321, 352, 383, 384, 385, 448, 511, 512):
        for rep in range(6):
            q = rng.integers(0, 4, lq)
            q_txt = np.frombuffer(b"ACGT", np.uint8)[q]
            lead = int(rng.integers(0, 70))
            gap = int(rng.integers(1, 40))
            flank = lambda n: np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
            copies = [q_txt] if rep % 2 == 0 else [q_txt, flank(gap), q_txt]  # one exact copy, or two (equal end cells)
            r_txt = np.concatenate([flank(lead)] + copies + [flank(int(rng.integers(0, 50)))])
            qs.append(q_txt.tobytes())
            rs.append(r_txt.tobytes())
    qc, qo = concat([np.frombuffer(x, np.uint8) for x in qs])
    rc, ro = concat([np.frombuffer(x, np.uint8) for x in rs])
    c = fade_amd.Context(device=0, rules=rules)
    try:
        got = c.sw_batch_packed(qc, qo, rc, ro)
    finally:
        c.close()
    exp, exp_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16, params=oracle.default_params(rules=rules))
    for k in range(len(qs)):
        g = tuple(int(got[k][f]) for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops"))
        assert g == tuple(int(x) for x in exp[k]), (k, len(qs[k]), g, exp[k])
        assert g[0] == 2 * len(qs[k])
        m = min(int(exp[k][5]), 16)
        assert list(got[k]["ops"][:m]) == list(exp_ops[k][:m]), k
