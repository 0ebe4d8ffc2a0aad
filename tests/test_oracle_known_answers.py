"""The oracle against construction-known answers and the literal constants
the reference tree holds.  The reference has no tests or golden vectors of its own (PARITY UNPINNED),
so these are answers forced by the scoring scheme, independent of any tie rule."""
import numpy as np
import pytest

from helpers import rand_seq, revcomp


def _other(c):
    return ord("A") if c != ord("A") else ord("C")


def _unique_planted(rng, lq, L, lr, left, flank=8):
    """query whose L-base end matches the reference exactly once; the `flank` bases beyond the
    match all mismatch, so extending the alignment can only lose score."""
    r = rand_seq(rng, lr)
    q = rand_seq(rng, lq)
    s = int(rng.integers(flank + 1, lr - L - flank - 1))
    seg = r[s:s + L]
    if left:   # match at the END of the query (what a left-clip artifact gives, analysis.d:74-80)
        q[lq - L:] = seg
        for k in range(1, flank + 1):
            q[lq - L - k] = _other(r[s - k])
    else:      # match at the START of the query (right-clip artifact, analysis.d:98-104)
        q[:L] = seg
        for k in range(flank):
            q[L + k] = _other(r[s + L + k])
    return q, r, s


@pytest.mark.parametrize("left", [True, False])
def test_planted_exact_match(oracle, left):
    rng = np.random.default_rng(5)
    for _ in range(40):
        lq, L, lr = 150, int(rng.integers(30, 60)), 340
        q, r, s = _unique_planted(rng, lq, L, lr, left)
        res = oracle.sw(q.tobytes(), r.tobytes())
        # 2 per matched base; a longer random alignment cannot reach 60+
        assert res["score"] == 2 * L
        assert res["beg_ref"] == s
        exp = "%dS%d=" % (lq - L, L) if left else "%d=%dS" % (L, lq - L)
        assert oracle.cigar_str(res["ops"]) == exp


def test_scoring_constants(oracle):
    # README.md:139-140 / anno.d:36: match 2, mismatch 3, gap open 10, gap extend 2
    assert oracle.sw("ACGTACGTAC", "ACGTACGTAC")["score"] == 20
    core = "ACGTTGCATGCCGATAGCTAGGCTAACG"
    # one mismatch in the middle: 27 matches - 3
    mm = core[:14] + ("A" if core[14] != "A" else "C") + core[15:]
    assert oracle.sw(mm, core)["score"] == 2 * 27 - 3
    # a 1-base deletion from the query costs 10, a 3-base one 10 + 2*2 (first gap base = open)
    left, right = "ACGTTGCATGCCGATAGCTAGGCTAACGAT", "TTGACCGTAGGCTAGCTAGGATCGATCCGA"
    for k, cost in ((1, 10), (3, 14)):
        q = left + right
        r = left + "CAC"[:k] + right  # neighbours differ from the gap's end bases: placement is unique
        res = oracle.sw(q, r)
        assert res["score"] == 2 * len(q) - cost
        assert oracle.cigar_str(res["ops"]) == "%d=%dD%d=" % (len(left), k, len(right))
    # insertion in the query
    res = oracle.sw(left + "CAC" + right, left + right)
    assert oracle.cigar_str(res["ops"]) == "%d=3I%d=" % (len(left), len(right))


def test_matrix_wildcard_and_n(oracle):
    # Appendix A.1: N vs N scores +2; non-ACGTN letters hit the wildcard (0) but emit '=' when equal
    assert oracle.sw("ACGTNNACGT", "ACGTNNACGT")["score"] == 20
    res = oracle.sw("ACGTACGTRRACGTACGT", "ACGTACGTRRACGTACGT")
    assert res["score"] == 32 and oracle.cigar_str(res["ops"]) == "18="
    res = oracle.sw("ACGTACGTRYACGTACGT", "ACGTACGTYRACGTACGT")
    assert res["score"] == 32 and oracle.cigar_str(res["ops"]) == "8=2X8="


def test_end_cell_tie_rule(oracle):
    # two equally good hits: the one ending at the smaller reference index wins (Appendix A.3)
    hit = "ACGTTGCATGCC"
    r = "TTTT" + hit + "TTTTTTTT" + hit + "TTTT"
    res = oracle.sw("GGGGGG" + hit, r)
    assert res["score"] == 24 and res["beg_ref"] == 4 and res["end_ref"] == 4 + len(hit) - 1


def test_util_restatements(oracle):
    # util.d:18-34 on every nt16 code; BAM packs the first base in the high nibble
    codes = list(range(16))
    packed = np.array([(codes[i] << 4) | codes[i + 1] for i in range(0, 16, 2)], dtype=np.uint8)
    rc = oracle.reverse_complement_packed(packed, 16)
    comp = "=TGKCYSBAWRDMHVN"
    assert rc == "".join(comp[c] for c in reversed(codes)).encode()
    # util.d:37-62 parse_clips: H is skipped, first S is left, later S is right
    S, M, H = 4, 0, 5
    op = lambda n, o: (n << 4) | o
    assert oracle.parse_clips([op(5, H), op(7, S), op(90, M), op(3, S)]) == (op(7, S), op(3, S))
    assert oracle.parse_clips([op(90, M), op(3, S), op(2, H)]) == (0, op(3, S))
    assert oracle.parse_clips([op(100, M)]) == (0, 0)
    assert oracle.parse_clips([op(7, S), op(90, M)]) == (op(7, S), 0)


def test_cutoff_integer_form():
    # analysis.d:43,76: `score > float(clip_len*0.9*2)` == 5*score > 9*clip_len for every reachable pair
    L = np.arange(1, 2001)[:, None]
    S = np.arange(0, 4001)[None, :]
    assert np.array_equal(S.astype(np.float32) > (L * 0.9 * 2).astype(np.float32), 5 * S > 9 * L)


def test_rs_reachable_values(oracle):
    """readstatus.d:5-26 bit layout through annotateTask on hand-made records."""
    from fade_amd import synth
    cfg, g, b = synth.make_config("C1", 3000, contig_len=150_000)
    G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
    rs, am = oracle.annotate_batch_soa(G, b, cfg["floor_len"], cfg["window"], threads=4)
    assert set(int(v) for v in np.unique(rs)) <= {0, 1, 3, 5, 33, 35, 37}
    t = b["_truth"]
    # planted artifacts with clips well above the noise floor are called on the planted side
    big_l = t["plantL"] & (t["clipL"] >= 20) & (t["clipR"] == 0)
    big_r = t["plantR"] & (t["clipR"] >= 20) & (t["clipL"] == 0)
    assert ((rs[big_l] >> 1) & 1).mean() > 0.97
    assert ((rs[big_r] >> 2) & 1).mean() > 0.97
    i = int(np.nonzero(big_l)[0][0])
    name, pos, cig = am[i].split(";")[0].split(",")
    assert int(pos) == int(t["segL"][i]) or cig.endswith("=")


def test_two_clip_read_runs_sw_twice(oracle):
    """anno.d:79-91: both clips qualify -> the reference aligns twice; the result is one alignment (F5/F6)."""
    from fade_amd import synth
    g = synth.Genome(1, 100_000, 3)
    b = synth.make_reads(g, 4000, 9, read_len=150, window=100, p_sc=1.0, clip_min=8, clip_max=30)
    G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
    reads, keep = oracle.make_reads(b)
    t = b["_truth"]
    both = np.nonzero((t["clipL"] > 5) & (t["clipR"] > 5) & ((b["flag"] & 4) == 0))[0][:50]
    assert len(both) > 10
    for i in both:
        a = oracle.annotate_one(G, reads[int(i)], 5, 100)
        assert a["n_sw_calls"] == 2
        assert (a["rs"] >> 1) & 3 != 3  # left and right calls are mutually exclusive
