"""Level-1 parity: fadehip_sw_batch (analysis.d:67 replacement) vs the scalar oracle, bit-exact."""
import os

import numpy as np
import pytest

from helpers import concat, make_pairs

pytestmark = pytest.mark.gpu


def _compare(ctx, oracle, qs, rs):
    qc, qo = concat(qs)
    rc, ro = concat(rs)
    got = ctx.sw_batch_packed(qc, qo, rc, ro)
    exp, exp_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16)
    bad = []
    for k in range(len(qs)):
        g = got[k]
        e = exp[k]
        tup_g = (int(g["score"]), int(g["end_query"]), int(g["end_ref"]), int(g["beg_query"]), int(g["beg_ref"]),
                 int(g["n_ops"]))
        tup_e = tuple(int(x) for x in e)
        n = min(tup_e[5], 16)
        if tup_g != tup_e or list(g["ops"][:n]) != list(exp_ops[k][:n]):
            bad.append((k, tup_g, tup_e, oracle.cigar_str(g["ops"][:n]), oracle.cigar_str(exp_ops[k][:n])))
    assert not bad, "%d/%d mismatches, first: %r" % (len(bad), len(qs), bad[:3])


def test_sw_small_known(ctx, oracle):
    q = np.frombuffer(b"GGGGGGGGGGACGTACGGTCAGGC", dtype=np.uint8)
    r = np.frombuffer(b"TTTTTTTTTTACGTACGGTCAGGCTTTTTTTTTT", dtype=np.uint8)
    got = ctx.sw_batch([q.tobytes()], [r.tobytes()])[0]
    assert int(got["score"]) == 28 and int(got["beg_ref"]) == 10
    assert oracle.cigar_str(got["ops"][:got["n_ops"]]) == "10S14="


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sw_random_families(ctx, oracle, seed):
    rng = np.random.default_rng(seed)
    qs, rs = make_pairs(rng, 1400)
    _compare(ctx, oracle, qs, rs)


@pytest.mark.parametrize("kinds", [("planted", "related", "random"), ("refspecial", "planted", "refspecial"),
                                   ("nrich", "iupac", "related")], ids=["acgt", "refspecial", "qspecial"])
def test_sw_all_row_classes(ctx, oracle, kinds):
    """Every R class of the kernel (Lq 1..512) and ragged window lengths.  The packed kernel has two sweeps: one for
    octets whose queries are pure A,C,G,T (ids acgt, refspecial) and one for octets with N / IUPAC query rows."""
    rng = np.random.default_rng(11)
    qs, rs = [], []
    for lq in list(range(1, 40)) + [63, 64, 65, 95, 96, 97, 127, 128, 129, 150, 159, 160, 161, 191, 192, 193, 223,
                                     224, 225, 250, 255, 256, 257, 300, 319, 320, 321, 383, 384, 385, 500, 511, 512]:
        for kind in kinds:
            q, r = make_pairs(rng, 1, lq_range=(lq, lq), lr_range=(1, 1200), kinds=(kind,))
            qs += q
            rs += r
    _compare(ctx, oracle, qs, rs)


def test_sw_empty_inputs(ctx):
    out = ctx.sw_batch([b"", b"ACGT", b""], [b"ACGT", b"", b""])
    assert [int(x) for x in out["n_ops"]] == [0, 1, 0]
    assert int(out["ops"][1][0]) == (4 << 4 | 4)
    assert len(ctx.sw_batch([], [])) == 0


def test_sw_limits_fail_loudly(ctx):
    import fade_amd
    with pytest.raises(fade_amd.FadeHipError):
        ctx.sw_batch([b"A" * 32769], [b"ACGT"])  # FADEHIP_MAX_LONG_QUERY
    small = fade_amd.Context(device=0, max_ref_len=8192)
    try:
        with pytest.raises(fade_amd.FadeHipError):
            small.sw_batch([b"ACGT"], [b"A" * 9000])
    finally:
        small.close()


def test_sw_long_queries(ctx, oracle):
    """Queries past the 16-lane kernels' 512 bases: one alignment per wavefront with 12 / 16 / 24 / 32 / 48 / 64 rows per lane
    (up to 768 / 1,024 / 1,536 / 2,048 / 3,072 / 4,096 bases); beyond that, and whenever such a query shares its batch's long list, a thread per alignment.
    Same results from all three, and from the thread kernel alone (FADEHIP_LONG_THREAD=1)."""
    import fade_amd
    rng = np.random.default_rng(23)
    groups = {}
    for name, lqs in (("r12", (513, 514, 600, 768)), ("r16", (769, 777, 1000, 1024)), ("r24", (1025, 1536)), ("r32", (1537, 2047, 2048)), ("r48", (2049, 3072)), ("r64", (3073, 4096)),
                      ("thread", (4097, 4500))):
        qs, rs = [], []
        for lq in lqs:
            for kind in ("planted", "related", "random", "nrich", "tandem"):
                q, r = make_pairs(rng, 1, lq_range=(lq, lq), lr_range=(lq // 2, 2 * lq + 300), kinds=(kind,))
                qs += q
                rs += r
        groups[name] = (qs, rs)
    # mixed with short pairs in one batch: several kernels serve the same call
    q, r = make_pairs(rng, 40, lq_range=(30, 512), lr_range=(30, 900))
    for name, (qs, rs) in groups.items():
        _compare(ctx, oracle, qs + q, rs + r)
    allq = sum((g[0] for g in groups.values()), [])
    allr = sum((g[1] for g in groups.values()), [])
    _compare(ctx, oracle, allq + q, allr + r)
    os.environ["FADEHIP_LONG_THREAD"] = "1"
    try:
        c2 = fade_amd.Context(device=0)
        wq = sum((groups[k][0] for k in ("r12", "r16", "r24", "r32", "r48", "r64")), [])
        wr = sum((groups[k][1] for k in ("r12", "r16", "r24", "r32", "r48", "r64")), [])
        a = c2.sw_batch([x.tobytes() for x in wq], [x.tobytes() for x in wr])
        c2.close()
    finally:
        del os.environ["FADEHIP_LONG_THREAD"]
    b = ctx.sw_batch([x.tobytes() for x in wq], [x.tobytes() for x in wr])
    assert a.tobytes() == b.tobytes()


def test_sw_neighbour_independence(ctx, oracle):
    """One N-bearing query switches its whole octet to the general sweep: the other seven results must not move."""
    rng = np.random.default_rng(5)
    qs, rs = make_pairs(rng, 64, lq_range=(150, 150), lr_range=(300, 340), kinds=("planted", "related", "random"))
    base = ctx.sw_batch([q.tobytes() for q in qs], [r.tobytes() for r in rs])
    qs2 = [q.copy() for q in qs]
    for k in range(3, 64, 8):
        qs2[k][rng.integers(0, len(qs2[k]), size=4)] = ord("N")
    got = ctx.sw_batch([q.tobytes() for q in qs2], [r.tobytes() for r in rs])
    for k in range(64):
        if k % 8 != 3:
            assert got[k].tobytes() == base[k].tobytes(), k
    _compare(ctx, oracle, qs2, rs)


@pytest.mark.parametrize("open_,ext,match,mismatch", [(6, 1, 1, -4), (4, 4, 2, -4), (12, 3, 3, -2), (5, 1, 8, -4), (9, 2, 5, -9)])
def test_sw_other_scoring_schemes(oracle, open_, ext, match, mismatch):
    """fadehip_params is not tied to FADE's 10/2/2/-3.  match <= 2 keeps the two-pass path; larger match scores leave
    the 16-bit key ranges of the score pass and take the single-pass packed kernel (match <= 7) or the int32 kernel."""
    import fade_amd
    c = fade_amd.Context(device=0, open=open_, ext=ext, match=match, mismatch=mismatch)
    try:
        rng = np.random.default_rng(open_ * 100 + match)
        qs, rs = make_pairs(rng, 700, lq_range=(1, 512), lr_range=(1, 1200))
        # long exact matches drive the score to match * Lq, the top of the value ranges
        for lq in (224, 300, 512):
            r = make_pairs(rng, 1, lq_range=(lq, lq), lr_range=(900, 900), kinds=("random",))[1][0]
            qs.append(r[100:100 + lq].copy())
            rs.append(r)
        qc, qo = concat(qs)
        rc, ro = concat(rs)
        p = oracle.default_params()
        p.open, p.ext, p.match, p.mismatch = open_, ext, match, mismatch
        got = c.sw_batch_packed(qc, qo, rc, ro)
        exp, exp_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16, params=p)
        for k in range(len(qs)):
            g = got[k]
            tup = tuple(int(g[f]) for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops"))
            assert tup == tuple(int(x) for x in exp[k]), (k, tup, exp[k])
            n = min(tup[5], 16)
            assert list(g["ops"][:n]) == list(exp_ops[k][:n]), k
    finally:
        c.close()


def test_sw_scoring_outside_the_profile_range_is_rejected():
    import fade_amd
    with pytest.raises(fade_amd.FadeHipError):
        fade_amd.Context(device=0, open=10, ext=2, match=6, mismatch=-3)   # match + open > 15
    with pytest.raises(fade_amd.FadeHipError):
        fade_amd.Context(device=0, open=4, ext=1, match=2, mismatch=-5)    # mismatch + open < 0
    with pytest.raises(fade_amd.FadeHipError):
        fade_amd.Context(device=0, open=2, ext=3, match=2, mismatch=-1)    # ext > open


def test_sw_long_windows(oracle):
    """Windows stream through the wave kernels' LDS a chunk of 2,048 columns at a time (up to 32,000 columns: the end-cell
    key's 16 bits); beyond that the thread-per-alignment kernel serves them.  Lengths around every boundary: one chunk
    (2,044 columns with the sweep's 15 steps of skew), the old 8,000-column limit, 32,000."""
    import fade_amd
    c = fade_amd.Context(device=0, max_ref_len=40000)
    try:
        rng = np.random.default_rng(77)
        qs, rs = [], []
        for lr in (2030, 2044, 2045, 2060, 4090, 4100, 7990, 8001, 12000, 20000, 31990, 32000, 32001, 36000):
            for lq, kind in ((50, "planted"), (150, "related"), (250, "random"), (600, "random")):
                q, r = make_pairs(rng, 1, lq_range=(lq, lq), lr_range=(lr, lr), kinds=(kind,))
                qs += q
                rs += r
        # planted matches at the very end of long windows (the last chunk), and across a chunk boundary
        for lr, at in ((20000, 19900), (6000, 2040), (10000, 4090)):
            r = helpers_rand(rng, lr)
            q = helpers_rand(rng, 150)
            q[40:140] = r[at - 50:at + 50]
            qs.append(q)
            rs.append(r)
        _compare(c, oracle, qs, rs)
        with pytest.raises(fade_amd.FadeHipError):
            c.sw_batch([b"ACGT"], [b"A" * 40001])
    finally:
        c.close()


def helpers_rand(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()


def test_annotate_long_windows_take_the_wave_kernels(oracle):
    """Level 2 with a window size that puts every window beyond one staged chunk (-w 1500: ~3,100 columns) and beyond the old
    8,000-column limit (-w 5000): rs, tags and stats against the oracle, and the device profile shows no thread-per-alignment
    fallback (the score pass ran: forward_ms of the wave kernels covers all alignments)."""
    import fade_amd
    from fade_amd import format_tags, synth
    cfg, g, b = synth.make_config("C2", 3000, contig_len=300_000)
    c = fade_amd.Context(device=0)
    try:
        c.genome_upload(g.names, g.ascii_contigs())
        G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
        for w in (1500, 5000):
            rs, aln, st = c.annotate(b, cfg["floor_len"], w)
            tags = format_tags(b, g.names, rs, aln)
            ors, oam = oracle.annotate_batch_soa(G, b, cfg["floor_len"], w, threads=8)
            assert np.array_equal(rs, ors), w
            for i in range(len(ors)):
                assert (tags[i]["am"] if i in tags else None) == oam[i], (w, i)
            assert len(tags) > 50
    finally:
        c.close()
