"""BASELINE.json sizes on the GPU through size-independent properties, plus an oracle spot check.
(The oracle cannot finish 10 M reads in seconds; a random sample of the same batch is compared.)"""
import numpy as np
import pytest

import fade_amd
from fade_amd import shard, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n", [("C2", 1_000_000), ("C3", 300_000), ("C5", 500_000)])
def test_full_size_batch_properties(oracle, name, n):
    cfg = synth.config(name)
    g = synth.Genome(cfg["n_contigs"], 2_000_000, cfg["genome_seed"])
    cfg["contig_len"] = 2_000_000
    b = synth.make_reads(g, n, 1234, **cfg)
    ctx = fade_amd.Context(device=0)
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs, aln, stats = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    # readstatus.d: reachable values only; art bits exclusive
    assert set(int(v) for v in np.unique(rs)) <= {0, 1, 3, 5, 33, 35, 37}
    # device stats kernel == host Stats.parse over rs (a checksum of checksums)
    assert list(stats) == list(shard.stats_from_rs(rs))
    # exactly the reads with a clip longer than the floor were aligned, each once
    t = b["_truth"]
    mapped = (b["flag"] & 4) == 0
    want = mapped & ((t["clipL"] > cfg["floor_len"]) | (t["clipR"] > cfg["floor_len"]))
    assert len(aln) == int(want.sum())
    assert np.array_equal(np.sort(aln["read_idx"]), np.nonzero(want)[0])
    # artifact calls come only from aligned reads and agree with the per-alignment art field
    art = np.zeros(n, dtype=np.uint8)
    art[aln["read_idx"]] = aln["art"]
    assert np.array_equal((rs >> 1) & 3, art)
    # alignment invariants: score bound, coordinates inside the window, CIGAR consumes the whole query
    sw = aln["sw"]
    lq = cfg["read_len"]
    assert (sw["score"] >= 0).all() and (sw["score"] <= 2 * lq).all()
    assert (sw["end_ref"] < aln["win_len"]).all() and (sw["beg_ref"] <= sw["end_ref"] + 1).all()
    traced = sw["n_ops"] > 0  # untraced = cannot be an artifact (include/fadehip.h)
    assert (art[aln["read_idx"]][~traced] == 0).all() and traced.sum() >= (aln["art"] != 0).sum()
    assert ((sw["beg_ref"] >= 0) == traced).all()
    ok = traced & (sw["n_ops"] <= 16)
    qlen = np.zeros(len(aln), dtype=np.int64)
    for k in range(16):
        op, ln = sw["ops"][:, k] & 15, sw["ops"][:, k] >> 4
        qlen += np.where((k < sw["n_ops"]) & np.isin(op, (1, 4, 7, 8)), ln, 0)
    assert (qlen[ok] == lq).all()
    # long planted artifacts are found where they were planted
    big = t["plantL"] & (t["clipL"] >= 25) & (t["clipR"] == 0) & want
    assert ((rs[big] >> 1) & 1).mean() > 0.97
    # oracle spot check on a random sample of this very batch
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(n, size=3000, replace=False))
    sub = synth.take(b, idx)
    G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
    ors, oam = oracle.annotate_batch_soa(G, sub, cfg["floor_len"], cfg["window"], threads=8)
    assert np.array_equal(ors, rs[idx])
    tags = fade_amd.format_tags(b, g.names, rs, aln[np.isin(aln["read_idx"], idx)])
    for j, i in enumerate(idx):
        assert (oam[j] is None) == (int(i) not in tags)
        if oam[j] is not None:
            assert tags[int(i)]["am"] == oam[j]
    ctx.close()
