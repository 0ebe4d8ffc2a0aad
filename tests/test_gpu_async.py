"""The asynchronous level-2 pipeline of ABI 2 (include/fadehip.h): one-DMA pinned batch blocks, records the caller
leaves out (anno.d:61-65), host-side bounds with device-side checks, device-planned pass 2 with its scratch-overflow
re-run, several slots driven by one thread, several contexts on one device.  Everything is compared with the
plain full-batch path, which test_gpu_annotate / test_gpu_edges / test_gpu_fuzz compare with the oracle."""
import numpy as np
import pytest

import fade_amd
from fade_amd import shard, synth

pytestmark = pytest.mark.gpu


def _key(aln, idx=None):
    """read index (mapped through idx when the batch was a subset) -> the bytes FADE reads from the alignment."""
    out = {}
    for x in aln:
        i = int(x["read_idx"])
        y = x.copy()
        y["read_idx"] = 0
        out[int(idx[i]) if idx is not None else i] = y.tobytes()
    return out


@pytest.mark.parametrize("name", ["C2", "C5"])
def test_clipped_only_pinned_batch_equals_full_batch(ctx, name):
    cfg, g, b = synth.make_config(name, 40_000, contig_len=500_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs0, aln0, st0 = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    sub, idx = ctx.clipped_only(b)
    assert sub["n_skipped"] + len(idx) == len(rs0) and 0 < len(idx) < 0.4 * len(rs0)
    # anno.d:61-65: everything that was left out has rs = 0
    left_out = np.ones(len(rs0), bool)
    left_out[idx] = False
    assert not rs0[left_out].any()
    for form in ("dict", "pinned"):
        batch = ctx.pinned_batch(sub) if form == "pinned" else sub
        ctx.annotate_upload(1, batch)
        ctx.annotate_run(1, cfg["floor_len"], cfg["window"])
        rs1, aln1, st1 = ctx.annotate_collect(1)
        assert np.array_equal(rs1, rs0[idx]), form
        assert list(st1) == list(st0), form  # read_count includes the records that were not sent
        assert _key(aln1, idx) == _key(aln0), form
    # without the caller's bound the library scans the CIGARs itself: same answer
    sub2 = dict(sub)
    sub2.pop("ref_span_bound")
    rs2, aln2, st2 = ctx.annotate(sub2, cfg["floor_len"], cfg["window"], slot=2)
    assert np.array_equal(rs2, rs0[idx]) and list(st2) == list(st0) and _key(aln2, idx) == _key(aln0)


def test_ref_span_bound_too_small_is_an_error_not_a_fault(ctx):
    cfg, g, b = synth.make_config("C2", 5000, contig_len=200_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    sub, idx = ctx.clipped_only(b)
    bad = dict(sub)
    bad["ref_span_bound"] = 20  # the reads align over 100-144 reference bases
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.annotate(bad, cfg["floor_len"], cfg["window"])
    assert e.value.code == -1 and "ref_span_bound" in str(e.value)
    rs, aln, st = ctx.annotate(sub, cfg["floor_len"], cfg["window"])  # the context stays usable
    assert int(st[0]) == 5000


@pytest.mark.parametrize("waves", ["1", "7"])
def test_pass2_persistent_launch_of_any_size(monkeypatch, waves):
    """Pass 2 is one persistent launch whose waves draw octets from a ticket counter, trace them into their own scratch,
    walk the tracebacks and re-trace the paths that left their steps.  Its size is a guess made before the candidates
    are known; any size must give the same bytes — here a single wave (and seven) serving thousands of candidates."""
    cfg, g, b = synth.make_config("C5", 30_000, contig_len=400_000)
    monkeypatch.setenv("FADEHIP_NO_SHORTCUT", "1")  # every candidate goes through pass 2
    monkeypatch.setenv("FADEHIP_SPAN_SLACK", "0")    # ... and many paths leave their traced steps
    big = fade_amd.Context(device=0)
    monkeypatch.setenv("FADEHIP_P2_WAVES", waves)
    small = fade_amd.Context(device=0)
    try:
        for c in (big, small):
            c.genome_upload(g.names, g.ascii_contigs())
        rs0, aln0, st0 = big.annotate(b, cfg["floor_len"], cfg["window"])
        assert big.last_profile(0)["candidates"] > 2000
        rs1, aln1, st1 = small.annotate(b, cfg["floor_len"], cfg["window"])
        assert np.array_equal(rs0, rs1) and list(st0) == list(st1) and _key(aln0) == _key(aln1)
        assert small.last_profile(0)["trace_bytes"] < big.last_profile(0)["trace_bytes"]
        from helpers import concat, make_pairs
        qs, rs_ = make_pairs(np.random.default_rng(11), 400, kinds=("related", "planted", "random"))
        qc, qo = concat(qs)
        rc, ro = concat(rs_)
        assert small.sw_batch_packed(qc, qo, rc, ro).tobytes() == big.sw_batch_packed(qc, qo, rc, ro).tobytes()
    finally:
        small.close()
        big.close()


def test_all_slots_in_flight_from_one_thread(ctx):
    """run() returns once the work is enqueued: one host thread keeps every slot of the ctx busy."""
    cfg, g, b = synth.make_config("C2", 8 * 6000, contig_len=500_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    n_slots = fade_amd._lib.NUM_SLOTS
    parts = [synth.take(b, np.arange(k * 6000, (k + 1) * 6000)) for k in range(8)]
    ref = []
    for p in parts:
        ref.append(ctx.annotate(p, cfg["floor_len"], cfg["window"]))
    pinned = [ctx.pinned_batch(p) for p in parts]
    got = [None] * 8
    for k in range(8 + n_slots):
        slot = k % n_slots
        if k >= n_slots:
            got[k - n_slots] = ctx.annotate_collect(slot)
        if k < 8:
            ctx.annotate_upload(slot, pinned[k])
            ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
    for k in range(8):
        assert np.array_equal(got[k][0], ref[k][0]) and list(got[k][2]) == list(ref[k][2]), k
        assert _key(got[k][1]) == _key(ref[k][1]), k


def test_results_are_views_until_the_next_run(ctx):
    cfg, g, b = synth.make_config("C2", 4000, contig_len=200_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    ctx.annotate_upload(3, b)
    ctx.annotate_run(3, cfg["floor_len"], cfg["window"])
    rs, aln, st = ctx.annotate_results(3)
    rs2, aln2, st2 = ctx.annotate_results(3)  # asking again is free and gives the same block
    assert rs.ctypes.data == rs2.ctypes.data and aln.ctypes.data == aln2.ctypes.data and list(st) == list(st2)
    assert len(rs) == 4000 and int(st[0]) == 4000 and len(aln) == int((np.asarray(aln["read_idx"]) >= 0).sum())
    p = ctx.last_profile(3)
    assert p["alignments"] == len(aln) and p["forward_ms"] > 0
    half = synth.take(b, np.arange(2000))
    ctx.annotate_upload(3, half)  # the batch of the slot's NEXT run goes up; the results of the last run stay
    rs3, aln3, st3 = ctx.annotate_results(3)
    assert rs3.ctypes.data == rs.ctypes.data and len(rs3) == 4000 and ctx.last_profile(3)["alignments"] == len(aln)
    ctx.annotate_run(3, cfg["floor_len"], cfg["window"])  # ... until the slot is run again
    rs4, aln4, st4 = ctx.annotate_results(3)
    assert len(rs4) == 2000 and int(st4[0]) == 2000
    fresh = fade_amd.Context(device=0)
    try:
        with pytest.raises(fade_amd.FadeHipError) as e:
            fresh.annotate_results(0)  # nothing was ever run
        assert e.value.code == -6
        with pytest.raises(fade_amd.FadeHipError) as e:
            fresh.genome_upload(g.names, g.ascii_contigs()) or fresh.annotate_run(0, 5, 100)  # nothing uploaded
        assert e.value.code == -6
    finally:
        fresh.close()


def test_two_contexts_on_one_device_interleaved(oracle):
    """What `fade annotate --gpus 2` does when both contexts sit on one device (the test map of the CLI): batches dealt
    round-robin to two fadehip_ctx, each with its own genome copy, streams and slots; results as from one context."""
    cfg, g, b = synth.make_config("C2", 24_000, contig_len=300_000)
    one = fade_amd.Context(device=0)
    cs = [fade_amd.Context(device=0), fade_amd.Context(device=0)]
    try:
        for c in [one] + cs:
            c.genome_upload(g.names, g.ascii_contigs())
        rs0, aln0, st0 = one.annotate(b, cfg["floor_len"], cfg["window"])
        parts = [np.arange(k * 4000, (k + 1) * 4000) for k in range(6)]
        rs = np.zeros_like(rs0)
        keys, stats = {}, np.zeros(8, np.int64)
        for k, idx in enumerate(parts):  # deal: ctx k % 2, slot (k // 2) % 2
            cs[k % 2].annotate_upload((k // 2) % 2, synth.take(b, idx))
            cs[k % 2].annotate_run((k // 2) % 2, cfg["floor_len"], cfg["window"])
            if k >= 2:
                j = k - 2
                r, a, s = cs[j % 2].annotate_collect((j // 2) % 2)
                rs[parts[j]] = r
                keys.update(_key(a, parts[j]))
                stats += s
        for j in (4, 5):
            r, a, s = cs[j % 2].annotate_collect((j // 2) % 2)
            rs[parts[j]] = r
            keys.update(_key(a, parts[j]))
            stats += s
        assert np.array_equal(rs, rs0) and list(stats) == list(st0) and keys == _key(aln0)
        # the one collective of the path, over the two contexts (RCCL communicator with both ranks on device 0 is not
        # possible: the sum is what stats_allreduce must give on a multi-device node, checked here on the host)
        assert list(stats) == list(shard.stats_from_rs(rs0))
    finally:
        for c in [one] + cs:
            c.close()


def test_slot_reuse_without_fetching_results(ctx):
    """A batch whose results are never fetched does not wedge its slot: the next upload waits for the slot's stream."""
    cfg, g, b = synth.make_config("C2", 12_000, contig_len=200_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    first, second = synth.take(b, np.arange(0, 6000)), synth.take(b, np.arange(6000, 12_000))
    want = ctx.annotate(second, cfg["floor_len"], cfg["window"], slot=1)
    ctx.annotate_upload(0, first)
    ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
    ctx.annotate_upload(0, second)  # the first batch's results are dropped
    ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
    ctx.annotate_run(0, cfg["floor_len"], cfg["window"])  # ... and a run may be repeated on the uploaded batch
    got = ctx.annotate_collect(0)
    assert np.array_equal(got[0], want[0]) and list(got[2]) == list(want[2]) and _key(got[1]) == _key(want[1])
    out = ctx.sw_batch([b"ACGTACGTAC"], [b"TTACGTACGTACTT"])  # level 1 borrows slot 0
    assert int(out[0]["score"]) == 20


def test_prefetching_uploads_pipeline(ctx):
    """The streamed pattern of bench.py and of a reader-fed driver: right after run(k) on a slot, the batch of that slot's
    next run is uploaded (into the slot's other input buffer, on its copy stream) while run(k) is still in flight."""
    cfg, g, b = synth.make_config("C5", 10 * 3000, contig_len=400_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    parts = [synth.take(b, np.arange(k * 3000, (k + 1) * 3000)) for k in range(10)]
    ref = [ctx.annotate(p, cfg["floor_len"], cfg["window"], slot=3) for p in parts]
    pinned = [ctx.pinned_batch(p) for p in parts]
    n_slots, got = 2, [None] * 10
    for seq in range(10):
        slot = seq % n_slots
        if seq >= n_slots:
            got[seq - n_slots] = ctx.annotate_collect(slot)
        else:
            ctx.annotate_upload(slot, pinned[seq])
        ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
        if seq + n_slots < 10:
            ctx.annotate_upload(slot, pinned[seq + n_slots])  # beside the run just enqueued
    for seq in (8, 9):
        got[seq] = ctx.annotate_collect(seq % n_slots)
    for k in range(10):
        assert np.array_equal(got[k][0], ref[k][0]) and list(got[k][2]) == list(ref[k][2]) and _key(got[k][1]) == _key(ref[k][1]), k


@pytest.mark.parametrize("name", ["C2", "C5"])
def test_every_record_sent_with_the_callers_bounds_equals_full_batch(ctx, name):
    """ABI 3: every record of the batch is sent (the device's gate does anno.d:61-65 for all of them), the records it can
    never align without their bases, and upload sizes the run from the caller's bounds without walking the records."""
    cfg, g, b = synth.make_config(name, 40_000, contig_len=500_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs0, aln0, st0 = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    allrec = ctx.with_bounds(b)
    assert len(allrec["pos"]) == len(rs0) and 0 < allrec["n_with_seq"] < 0.4 * len(rs0)
    assert len(allrec["seq_packed"]) < 0.4 * len(b["seq_packed"])
    for form in ("dict", "pinned"):
        batch = ctx.pinned_batch(allrec) if form == "pinned" else allrec
        ctx.annotate_upload(1, batch)
        ctx.annotate_run(1, cfg["floor_len"], cfg["window"])
        rs1, aln1, st1 = ctx.annotate_collect(1)
        assert np.array_equal(rs1, rs0), form
        assert list(st1) == list(st0), form
        assert _key(aln1) == _key(aln0), form
    # reads of several lengths under one pair of bounds (three row classes)
    parts = []
    for k, L in enumerate((76, 150, 210)):
        c2 = dict(cfg, read_len=L)
        parts.append(synth.make_reads(g, 6000, 50 + k, **c2))
    mixed = synth.concat(parts)
    rs2, aln2, st2 = ctx.annotate(mixed, cfg["floor_len"], cfg["window"])
    mb = ctx.with_bounds(mixed)
    assert (mb["l_seq_min"], mb["l_seq_max"]) == (76, 210)
    rs3, aln3, st3 = ctx.annotate(mb, cfg["floor_len"], cfg["window"], slot=2)
    assert np.array_equal(rs2, rs3) and list(st2) == list(st3) and _key(aln2) == _key(aln3)


def test_bounds_too_small_or_bad_offsets_fail_the_batch_not_the_device(ctx):
    """The caller's bounds are never trusted for memory safety: the gate kernel checks list capacities and every
    offset before it reads through it; the batch fails at results and the context stays usable."""
    cfg, g, b = synth.make_config("C2", 20_000, contig_len=300_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    good = ctx.with_bounds(b)
    rs0, aln0, st0 = ctx.annotate(good, cfg["floor_len"], cfg["window"])
    few = dict(good, n_with_seq=good["n_with_seq"] // 3)
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.annotate(few, cfg["floor_len"], cfg["window"])
    assert e.value.code == -1 and "bounds" in str(e.value)
    short = dict(good, l_seq_max=64, l_seq_min=64)  # the reads have 150 bases: another row class than announced
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.annotate(short, cfg["floor_len"], cfg["window"])
    assert e.value.code == -1
    bad = dict(good)
    so = good["seq_off"].copy()
    so[1000] = so[-1] + 7_000_000  # points far beyond seq_packed
    bad["seq_off"] = so
    with pytest.raises(fade_amd.FadeHipError) as e:
        ctx.annotate(bad, cfg["floor_len"], cfg["window"])
    assert e.value.code == -1 and "seq_off" in str(e.value)
    bad = dict(good)
    co = good["cigar_off"].copy()
    co[500] = co[499] - 1 if co[499] > 0 else 5  # decreasing
    bad["cigar_off"] = co
    with pytest.raises(fade_amd.FadeHipError):
        ctx.annotate(bad, cfg["floor_len"], cfg["window"])
    rs1, aln1, st1 = ctx.annotate(good, cfg["floor_len"], cfg["window"])
    assert np.array_equal(rs0, rs1) and _key(aln0) == _key(aln1)


@pytest.mark.parametrize("name", ["C2", "C3"])
def test_snapshots_on_and_off_give_the_same_bytes(monkeypatch, name):
    """The score pass leaves wave snapshots only when the slot's previous run had many pass-2 candidates (otherwise a
    candidate is traced from step 0).  Either way, and across the switch, the results are the same bytes — with the
    forced-diagonal shortcut (few candidates) and without it (every candidate through pass 2)."""
    cfg, g, b = synth.make_config(name, 24_000, contig_len=400_000)
    outs = []
    for shortcut in (True, False):
        if not shortcut:
            monkeypatch.setenv("FADEHIP_NO_SHORTCUT", "1")
        for ck in ("0", "1", None):
            if ck is None:
                monkeypatch.delenv("FADEHIP_CKPT", raising=False)
            else:
                monkeypatch.setenv("FADEHIP_CKPT", ck)
            c = fade_amd.Context(device=0)
            try:
                c.genome_upload(g.names, g.ascii_contigs())
                for rep in range(3 if ck is None else 1):  # adaptive: the first run decides for the second
                    rs, aln, st = c.annotate(b, cfg["floor_len"], cfg["window"])
                    prof = c.last_profile(0)
                    outs.append((rs.tobytes(), _key(aln), list(st)))
                    if ck == "0":
                        assert prof["snapshot_bytes"] == 0
                    if ck == "1":
                        assert prof["snapshot_bytes"] > 0
                    if ck is None:
                        # the first run of a slot leaves none; many candidates (no shortcut: all of them go through pass 2)
                        # switch the snapshots on from the second run, few (C2 with the shortcut) leave them off
                        frac = prof["candidates"] / max(prof["alignments"], 1)
                        if rep == 0:
                            assert prof["snapshot_bytes"] == 0
                        else:
                            assert (prof["snapshot_bytes"] > 0) == (frac > 1 / 32), (rep, shortcut, frac)
                        if not shortcut:
                            assert frac > 1 / 32
                        elif name == "C2":
                            assert frac < 1 / 32
            finally:
                c.close()
    for o in outs[1:]:
        assert o == outs[0]
