"""Edge cases of the annotate path on the GPU, each compared with the oracle's annotateTask:
hard clips, mid-CIGAR S, clip lengths around --min-length, contig-edge windows, unmapped records,
empty sequences, soft-masked / IUPAC FASTA, N-rich reads, mixed read lengths (several row classes),
trace chunking, batch-split invariance, and the limits that must fail loudly."""
import numpy as np
import pytest

import fade_amd
import samutil
from fade_amd import format_tags, synth

pytestmark = pytest.mark.gpu


def _compare(ctx, oracle, names, seqs, batch, floor_len, window):
    ctx.genome_upload(names, [s.encode() if isinstance(s, str) else s for s in seqs])
    rs, aln, stats = ctx.annotate(batch, floor_len, window)
    tags = format_tags(batch, names, rs, aln)
    G = oracle.GenomeHolder(names, seqs)
    ors, oam = oracle.annotate_batch_soa(G, batch, floor_len, window, threads=8)
    assert np.array_equal(rs, ors), np.nonzero(rs != ors)[0][:10]
    for i in range(len(ors)):
        if oam[i] is None:
            assert i not in tags
        else:
            assert tags[i]["am"] == oam[i], i
    return rs, aln, tags


def _sam(lines, contigs):
    hdr = "".join("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)) for n, s in contigs)
    return hdr + "\n".join(lines) + "\n"


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def test_handmade_records(ctx, oracle):
    rng = np.random.default_rng(3)
    ref = "".join("ACGT"[k] for k in rng.integers(0, 4, size=3000))
    contigs = [("c1", ref), ("c2", ref[::-1])]
    q = "I" * 150
    lines = []

    def add(name, flag, rname, pos, cigar, seq, extra=""):
        lines.append("\t".join([name, str(flag), rname, str(pos + 1), "60", cigar, "*", "0", "0", seq, q[:len(seq)]]) + extra)

    def read_with_left_artifact(pos, L, seg_start, lq=150):
        clip = _rc(ref[seg_start:seg_start + L])
        return clip + ref[pos:pos + lq - L]

    # left artifacts with hard clip in front, clip lengths around the floor, SA tag
    for k, L in enumerate([4, 5, 6, 7, 20, 50]):
        pos = 600 + 200 * k
        add("lh%d" % L, 0, "c1", pos, "10H%dS%dM" % (L, 150 - L), read_with_left_artifact(pos, L, pos - 60))
        add("ls%d" % L, 16, "c1", pos, "%dS%dM" % (L, 150 - L), read_with_left_artifact(pos, L, pos - 60), "\tSA:Z:c2,5,+,50M,60,0;")
    # right artifacts near the contig end (window clamps at targetLength)
    for L in (8, 30):
        pos = len(ref) - (150 - L) - 20
        seg_end = len(ref) - 2
        seq = ref[pos:pos + 150 - L] + _rc(ref[seg_end - L:seg_end])
        add("re%d" % L, 0, "c1", pos, "%dM%dS" % (150 - L, L), seq)
    # left artifact at the contig start (window clamps at 0)
    add("l0", 0, "c1", 30, "20S130M", read_with_left_artifact(30, 20, 2))
    # both clips, mid-CIGAR S (parse_clips sees none at the ends? it sees a right clip), S after H at the end
    add("both", 0, "c1", 1500, "12S120M18S", _rc(ref[1420:1432]) + ref[1500:1620] + _rc(ref[1660:1678]))
    add("mid", 0, "c1", 1700, "50M10S90M", ref[1700:1750] + "ACGTACGTAC" + ref[1750:1840], "\tSA:Z:c1,9,+,50M,60,0;")
    add("endh", 0, "c1", 1900, "130M20S5H", ref[1900:2030] + _rc(ref[2050:2070]))
    # unmapped with a CIGAR that has S (anno.d:61: !isMapped wins), mapped without S, empty SEQ with S
    add("un", 4, "c1", 100, "20S130M", read_with_left_artifact(100, 20, 60))
    add("nos", 0, "c1", 100, "150M", ref[100:250])
    lines.append("\t".join(["noseq", "0", "c1", "301", "60", "20S130M", "*", "0", "0", "*", "*"]))
    # N-rich read and clip of Ns
    nr = list(read_with_left_artifact(2200, 30, 2150))
    for p in range(0, 150, 7):
        nr[p] = "N"
    add("nrich", 0, "c1", 2200, "30S120M", "".join(nr))
    # second contig
    add("c2l", 0, "c2", 500, "25S125M", _rc(contigs[1][1][430:455]) + contigs[1][1][500:625])
    text = _sam(lines, contigs)
    names, lens, batch, qnames = samutil.sam_to_batch(text)
    for floor_len, window in ((5, 300), (5, 100), (0, 40), (6, 1000)):
        rs, aln, tags = _compare(ctx, oracle, names, [c[1] for c in contigs], batch, floor_len, window)
    by = dict(zip(qnames, rs))
    assert by["un"] == 0 and by["nos"] == 0
    assert by["mid"] == 33  # parse_clips finds the mid-CIGAR S as a right clip: sc | sup
    assert by["lh50"] == 3 and by["ls50"] == 35 and by["re30"] == 5


def test_softmasked_and_iupac_reference(ctx, oracle):
    cfg, g, b = synth.make_config("C2", 3000, contig_len=60_000)
    rng = np.random.default_rng(8)
    seqs = []
    for a in g.ascii_contigs():
        s = bytearray(a.tobytes())
        for p in rng.integers(0, len(s), size=len(s) // 50):
            s[p] = rng.choice(list(b"NRYKMSWnryacgt"))
        lo = bytes(s).lower()
        s[10_000:20_000] = lo[10_000:20_000]
        seqs.append(bytes(s).decode())
    _compare(ctx, oracle, g.names, seqs, b, cfg["floor_len"], cfg["window"])


def test_mixed_read_lengths_and_classes(ctx, oracle):
    g = synth.Genome(2, 80_000, 5)
    parts = []
    # 513+ bases: past the 16-lane kernels, served in the same batch by the one-alignment-per-wavefront kernel (here with 64
    # rows per lane: the long list's longest read has 3,000 bases)
    for k, lq in enumerate([36, 76, 101, 150, 151, 200, 250, 300, 400, 512, 513, 600, 1000, 2048, 3000]):
        parts.append(synth.make_reads(g, 400 if lq <= 512 else 60 if lq <= 1000 else 24, 50 + k, read_len=lq, window=120, p_sc=0.5, clip_min=6,
                                      clip_max=min(30, lq // 3), insert_mu=max(350, lq + 100)))
    keys = ("tid", "pos", "flag", "has_sa", "l_seq")
    b = {k: np.concatenate([p[k] for p in parts]) for k in keys}
    for off, data in (("cigar_off", "cigar_ops"), ("seq_off", "seq_packed"), ("qual_off", "qual")):
        b[data] = np.concatenate([p[data] for p in parts])
        offs, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            offs.append(p[off][1:].astype(np.int64) + base)
            base += int(p[off][-1])
        b[off] = np.concatenate(offs)
    rs, aln, tags = _compare(ctx, oracle, g.names, [a.tobytes().decode() for a in g.ascii_contigs()], b, 5, 120)
    assert len(tags) > 100


def test_trace_chunking_and_batch_split_invariance(oracle, monkeypatch):
    monkeypatch.setenv("FADEHIP_NO_SHORTCUT", "1")  # every candidate through pass 2, so that every chunk has a plan
    cfg, g, b = synth.make_config("C2", 20_000, contig_len=200_000)
    contigs = g.ascii_contigs()
    big = fade_amd.Context(device=0)
    big.genome_upload(g.names, contigs)
    rs, aln, stats = big.annotate(b, cfg["floor_len"], cfg["window"])
    key = lambda a: {int(x["read_idx"]): x["sw"].tobytes() + bytes([int(x["art"])]) for x in a}
    ref = key(aln)
    # a 4 MB trace budget forces dozens of forward/traceback chunks per class
    small = fade_amd.Context(device=0, trace_bytes=4 << 20)
    small.genome_upload(g.names, contigs)
    rs2, aln2, stats2 = small.annotate(b, cfg["floor_len"], cfg["window"])
    assert np.array_equal(rs, rs2) and key(aln2) == ref and list(stats) == list(stats2)
    # trace_bytes of the profile = the largest pass-2 plan of the run: a chunk's is smaller than the whole list's
    assert 0 < small.last_profile(0)["trace_bytes"] <= big.last_profile(0)["trace_bytes"]
    # the same reads in two halves, on the two slots: every read's result is independent of its batch
    n = len(rs)
    h1, h2 = synth.take(b, np.arange(0, n // 2)), synth.take(b, np.arange(n // 2, n))
    big.annotate_upload(0, h1)
    big.annotate_upload(1, h2)
    big.annotate_run(0, cfg["floor_len"], cfg["window"])
    big.annotate_run(1, cfg["floor_len"], cfg["window"])
    r1, a1, s1 = big.annotate_collect(0)
    r2, a2, s2 = big.annotate_collect(1)
    assert np.array_equal(np.concatenate([r1, r2]), rs)
    k2 = {k + n // 2: v for k, v in key(a2).items()}
    assert {**key(a1), **k2} == ref
    assert list(s1 + s2) == list(stats)
    # idempotence: running the same slot again gives the same bytes
    big.annotate_run(1, cfg["floor_len"], cfg["window"])
    r2b, a2b, _ = big.annotate_collect(1)
    assert np.array_equal(r2, r2b) and key(a2b) == key(a2)
    big.close()
    small.close()


def test_limits_fail_loudly(ctx):
    g = synth.Genome(1, 50_000, 2)
    ctx.genome_upload(g.names, g.ascii_contigs())
    ok = synth.make_reads(g, 50, 1, read_len=150, window=100, p_sc=1.0, clip_min=10, clip_max=20)
    # a window beyond max_ref_len does not fail the batch: the read stays un-re-aligned (rs keeps sc / sup) and is counted
    small = fade_amd.Context(device=0, max_ref_len=8192)
    try:
        small.genome_upload(g.names, g.ascii_contigs())
        rs, aln, stats = small.annotate(ok, 5, 9000)
        n_clipped = int(((ok["flag"] & 4) == 0).sum())
        assert small.last_oversize == n_clipped and len(aln) == 0
        assert set(int(x) for x in rs) <= {0, 1, 33} and int(stats[0]) == 50 and int(stats[4]) == 0
    finally:
        small.close()
    # the default (2^20) sends such windows to the thread-per-alignment kernel instead
    rs, aln, stats = ctx.annotate(synth.take(ok, np.arange(6)), 5, 9000)
    assert ctx.last_oversize == 0 and len(aln) == int(((ok["flag"][:6] & 4) == 0).sum())
    bad = dict(ok)
    bad["tid"] = np.where((ok["flag"] & 4) == 0, 7, ok["tid"]).astype(np.int32)
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.annotate(bad, 5, 100)
    assert e.value.code == -1
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.genome_upload(["x"], [b"ACGT=ACGT"])
    assert e.value.code == -7
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs, aln, stats = ctx.annotate(ok, 5, 100)  # the context stays usable after the errors
    assert int(stats[0]) == 50
    empty = synth.take(ok, np.arange(0))
    rs, aln, stats = ctx.annotate(empty, 5, 100)
    assert len(rs) == 0 and len(aln) == 0 and int(stats[0]) == 0


@pytest.mark.parametrize("slack", ["-1", "0"])
def test_two_pass_rerun_paths(oracle, slack, monkeypatch):
    """The pass-2 range is a heuristic; correctness must not depend on it.  With the slack removed most paths
    leave their traced steps and go through the re-run rounds (4 snapshots back, then from step 0)."""
    monkeypatch.setenv("FADEHIP_SPAN_SLACK", slack)
    monkeypatch.setenv("FADEHIP_DEBUG", "1")
    c = fade_amd.Context(device=0)
    try:
        for name, n in (("C2", 6000), ("C3", 2500)):
            cfg, g, b = synth.make_config(name, n, contig_len=300_000)
            _compare(c, oracle, g.names, [a.tobytes().decode() for a in g.ascii_contigs()], b, cfg["floor_len"], cfg["window"])
        # level 1: every pair is traced; long related pairs make long paths
        from helpers import concat, make_pairs
        rng = np.random.default_rng(5)
        qs, rs_ = make_pairs(rng, 700, kinds=("related", "planted", "tandem", "random"))
        qc, qo = concat(qs)
        rc, ro = concat(rs_)
        got = c.sw_batch_packed(qc, qo, rc, ro)
        exp, exp_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16, striped=True)
        for k in range(len(qs)):
            assert tuple(int(got[k][f]) for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops")) == tuple(int(x) for x in exp[k]), k
            m = min(int(exp[k][5]), 16)
            assert list(got[k]["ops"][:m]) == list(exp_ops[k][:m]), k
    finally:
        c.close()


def test_single_pass_kernels_agree(oracle, monkeypatch):
    """FADEHIP_KERNEL=pk / int32 (the single-pass kernels kept for A/B runs) give the same answers."""
    cfg, g, b = synth.make_config("C2", 5000, contig_len=300_000)
    for k in ("pk", "int32", "twopass"):
        monkeypatch.setenv("FADEHIP_KERNEL", k)
        c = fade_amd.Context(device=0)
        try:
            _compare(c, oracle, g.names, [a.tobytes().decode() for a in g.ascii_contigs()], b, cfg["floor_len"], cfg["window"])
        finally:
            c.close()


def test_compact_sequences_same_result_and_missing_bases_fail(ctx):
    """The device reads bases only of mapped records with an S op: dropping all other slices changes nothing, and a
    record that must be re-aligned but arrives without its bases is an error, not an out-of-bounds read."""
    cfg, g, b = synth.make_config("C2", 20000, contig_len=400_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs0, aln0, st0 = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    c = ctx.compact_sequences(b)
    assert c["seq_packed"].nbytes < 0.25 * b["seq_packed"].nbytes
    rs1, aln1, st1 = ctx.annotate(c, cfg["floor_len"], cfg["window"])
    assert np.array_equal(rs0, rs1) and list(st0) == list(st1)
    k0, k1 = np.argsort(aln0["read_idx"]), np.argsort(aln1["read_idx"])
    assert aln0[k0].tobytes() == aln1[k1].tobytes()
    bad = dict(b)
    bad["seq_off"] = np.zeros_like(b["seq_off"])  # every slice empty
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.annotate(bad, cfg["floor_len"], cfg["window"])
    assert e.value.code == -1


def test_trace_all_gives_every_realigned_read_its_cigar(ctx):
    """params.trace_all = 1: level 2 reports beg_* and the CIGAR of every re-aligned read (no candidate selection, no
    early exit); each must equal the level-1 result for (reverse complement of the read, its window)."""
    cfg, g, b = synth.make_config("C2", 4000, contig_len=300_000)
    all_ctx = fade_amd.Context(device=0, trace_all=True)
    try:
        contigs = g.ascii_contigs()
        all_ctx.genome_upload(g.names, contigs)
        rs, aln, stats = all_ctx.annotate(b, cfg["floor_len"], cfg["window"])
        ctx.genome_upload(g.names, contigs)
        rs0, aln0, stats0 = ctx.annotate(b, cfg["floor_len"], cfg["window"])
        assert np.array_equal(rs, rs0) and list(stats) == list(stats0) and len(aln) == len(aln0) > 100
        assert (aln["sw"]["n_ops"] > 0).all() and (aln0["sw"]["n_ops"] == 0).any()
        comp = np.zeros(16, dtype=np.uint8)
        comp[[1, 2, 4, 8, 15]] = [8, 4, 2, 1, 15]
        nt16 = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
        qs, ws = [], []
        for a in aln:
            i = int(a["read_idx"])
            lq = int(b["l_seq"][i])
            o = int(b["seq_off"][i])
            pk = b["seq_packed"][o:o + (lq + 1) // 2]
            codes = np.empty(2 * len(pk), dtype=np.uint8)
            codes[0::2], codes[1::2] = pk >> 4, pk & 15
            qs.append(nt16[comp[codes[:lq]][::-1]].tobytes())
            s0 = int(a["win_start"])
            ws.append(contigs[int(b["tid"][i])][s0:s0 + int(a["win_len"])].tobytes())
        l1 = ctx.sw_batch(qs, ws)
        for k, a in enumerate(aln):
            for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops"):
                assert int(a["sw"][f]) == int(l1[k][f]), (k, f)
            n = min(int(l1[k]["n_ops"]), 16)
            assert list(a["sw"]["ops"][:n]) == list(l1[k]["ops"][:n]), k
    finally:
        all_ctx.close()


def test_two_slots_driven_by_two_threads_give_the_serial_results(ctx):
    """bench.py and the `fade` driver keep two batches in flight, one per slot, from two host threads."""
    import threading
    cfg, g, b0 = synth.make_config("C2", 30000, contig_len=400_000)
    b1 = synth.make_reads(g, 30000, 99, **cfg)
    ctx.genome_upload(g.names, g.ascii_contigs())
    ref = [ctx.annotate(b, cfg["floor_len"], cfg["window"]) for b in (b0, b1)]
    got = {0: [], 1: []}

    def loop(slot, batch):
        for _ in range(6):
            ctx.annotate_upload(slot, batch)
            ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
            got[slot].append(ctx.annotate_collect(slot))

    th = [threading.Thread(target=loop, args=(k, b)) for k, b in ((0, b0), (1, b1))]
    [t.start() for t in th]
    [t.join() for t in th]
    for slot in (0, 1):
        rs0, aln0, st0 = ref[slot]
        k0 = np.argsort(aln0["read_idx"])
        for rs, aln, st in got[slot]:
            assert np.array_equal(rs, rs0) and list(st) == list(st0)
            assert aln[np.argsort(aln["read_idx"])].tobytes() == aln0[k0].tobytes()


def test_batch_larger_than_max_batch_reads_is_rejected():
    c = fade_amd.Context(device=0, max_batch_reads=1000)
    try:
        cfg, g, b = synth.make_config("C2", 1001, contig_len=100_000)
        c.genome_upload(g.names, g.ascii_contigs())
        with pytest.raises(fade_amd.FadeHipError) as e:
            c.annotate(b, 5, 100)
        assert e.value.code == -1
        rs, aln, st = c.annotate(synth.take(b, np.arange(1000)), 5, 100)
        assert int(st[0]) == 1000
    finally:
        c.close()


def test_exact_diagonal_shortcut_equals_the_traced_path(ctx, monkeypatch):
    """The selection kernel finishes candidates whose score is match * L over an all-match diagonal without pass 2;
    FADEHIP_NO_SHORTCUT=1 sends them through the traced re-computation instead: same records, byte for byte."""
    cfg, g, b = synth.make_config("C5", 30000, contig_len=400_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs0, aln0, st0 = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    monkeypatch.setenv("FADEHIP_NO_SHORTCUT", "1")
    rs1, aln1, st1 = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    monkeypatch.delenv("FADEHIP_NO_SHORTCUT")
    assert np.array_equal(rs0, rs1) and list(st0) == list(st1) and int(st0[4]) > 1000
    assert aln0[np.argsort(aln0["read_idx"])].tobytes() == aln1[np.argsort(aln1["read_idx"])].tobytes()


def test_malformed_offsets_are_rejected_before_any_launch(ctx):
    cfg, g, b = synth.make_config("C2", 2000, contig_len=100_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    for key in ("cigar_off", "seq_off"):
        bad = dict(b)
        arr = b[key].copy()
        arr[500], arr[501] = arr[501], arr[500] + 7  # decreasing pair
        arr[500] = arr[502] + 1
        bad[key] = arr
        with pytest.raises(fade_amd.FadeHipError) as e:
            ctx.annotate(bad, 5, 100)
        assert e.value.code == -1
    bad = dict(b)
    bad["l_seq"] = b["l_seq"].copy()
    bad["l_seq"][7] = -3
    with pytest.raises(fade_amd.FadeHipError):
        ctx.annotate(bad, 5, 100)
    rs, aln, st = ctx.annotate(b, 5, 100)
    assert int(st[0]) == 2000


@pytest.mark.parametrize("read_len", [36, 40, 41, 50, 56, 57, 76, 80, 81, 100, 101, 104, 105, 150, 151, 152, 153])
def test_uniform_read_lengths_and_the_eight_lane_score_kernels(oracle, read_len, capfd):
    """Libraries of one read length: the score pass of the batch's top row class runs on eight-lane groups when the reads fit
    rows in steps of eight that the sixteen-lane class does not offer (40 / 56 / 80 / 104 / 152 rows: sw_pk_kernel<R8,1,LG=8>);
    lengths on both sides of every boundary, rs and am against the oracle, and the library says (FADEHIP_DEBUG) which
    geometry it took."""
    import os
    os.environ["FADEHIP_DEBUG"] = "1"
    try:
        c = fade_amd.Context(device=0)
    finally:
        del os.environ["FADEHIP_DEBUG"]
    try:
        cfg = synth.config("C5")
        cfg.update(read_len=read_len, contig_len=60_000, insert_mu=max(cfg["insert_mu"], read_len + 150), clip_max=min(cfg["clip_max"], read_len // 2))
        g = synth.Genome(2, cfg["contig_len"], 11)
        b = synth.make_reads(g, 3000, 900 + read_len, **{k: v for k, v in cfg.items() if k in (
            "read_len", "window", "p_sc", "clip_min", "clip_max", "insert_mu", "insert_sd")})
        b.pop("_truth", None)
        rs, aln, tags = _compare(c, oracle, g.names, [a.tobytes().decode() for a in g.ascii_contigs()], b, cfg["floor_len"], cfg["window"])
        assert len(tags) > 20
    finally:
        c.close()
    err = capfd.readouterr().err
    eight = "eight-lane groups" in err
    assert eight == (read_len in (36, 40, 41, 50, 56, 76, 80, 100, 101, 104, 150, 151, 152)), (read_len, err[-300:])
