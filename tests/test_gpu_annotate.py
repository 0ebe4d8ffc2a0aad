"""Level-2 parity: the device annotate path vs the oracle's annotateTask restatement (anno.d:55-110)."""
import numpy as np
import pytest

from fade_amd import format_tags, synth

pytestmark = pytest.mark.gpu


def _oracle_tags(oracle, g, b, cfg, floor_len=None, window=None):
    G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
    reads, keep = oracle.make_reads(b)
    n = len(b["pos"])
    rs = np.zeros(n, dtype=np.uint8)
    tags = {}
    for i in range(n):
        a = oracle.annotate_one(G, reads[i], floor_len=cfg["floor_len"] if floor_len is None else floor_len,
                                window=cfg["window"] if window is None else window)
        rs[i] = a["rs"]
        if a["has_tags"]:
            tags[i] = dict(rs=a["rs"], am=a["am"], as_=a["as_"], ar=a["ar"], ab=a["ab"])
    return rs, tags


@pytest.mark.parametrize("name,n", [("C1", 6000), ("C2", 6000), ("C3", 3000), ("C5", 4000), ("C6", 6000)])
def test_annotate_matches_oracle(ctx, oracle, name, n):
    cfg, g, b = synth.make_config(name, n, contig_len=300_000)
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs, aln, stats = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    tags = format_tags(b, ctx.contig_names, rs, aln)
    ors, otags = _oracle_tags(oracle, g, b, cfg)
    assert np.array_equal(rs, ors), "rs differs at %r" % np.nonzero(rs != ors)[0][:10]
    assert tags == otags
    # stats.d:45-54 over the batch
    c = np.zeros(8, dtype=np.int64)
    for v in ors:
        c += [1, v & 1, (v >> 5) & 1, ((v >> 1 | v >> 2) & 1) & (v >> 5) & 1, (v >> 1 | v >> 2) & 1,
              ((v >> 1) & (v >> 3) | (v >> 2) & (v >> 4)) & 1, (v >> 1) & 1, (v >> 2) & 1]
    assert list(stats) == list(c)
    assert len(tags) > 0


def test_stats_allreduce_over_rccl(ctx):
    """fadehip_stats_allreduce: ncclCommInitAll + ncclAllReduce(int64, sum) over the contexts' devices.  The
    GPU box has one device, so this is the one-rank case of the call the multi-device driver makes at exit."""
    import fade_amd
    c = np.arange(8, dtype=np.int64).reshape(1, 8) * 1000 + 7
    out = fade_amd.stats_allreduce([ctx], c)
    assert np.array_equal(out, c)


def test_stats_allreduce_with_two_contexts_on_the_one_device(ctx):
    """Two contexts (two ranks) of ONE process on this box's single GPU, through ncclCommInitAll.  RCCL wants one rank per
    device; what it says to a device listed twice is asserted for what it is — a sum that must be right, or a refusal that
    must be FADEHIP_E_RCCL from ncclCommInitAll — and written to gpurun_out/rccl_two_contexts.txt (DESIGN.md §0(e) quotes
    it).  The two-PROCESS form is tests/test_gpu_rccl_ranks.py.  `fade annotate --gpus N` under FADE_DEVICE_MAP=0,0 does not
    depend on either: it sums the lanes' reports in the parent."""
    import os
    import fade_amd
    other = fade_amd.Context(device=0)
    try:
        c = np.array([np.arange(8) * 10 + 1, np.arange(8) * 1000 + 5], dtype=np.int64)
        want = np.tile(c.sum(axis=0), (2, 1))
        try:
            out = fade_amd.stats_allreduce([ctx, other], c.copy())
            ran = "RCCL formed a communicator of two ranks on one device and summed"
            assert np.array_equal(out, want)
        except fade_amd.FadeHipError as e:
            ran = "refused: %s" % e
            assert e.code == -8 and "ncclCommInitAll" in str(e), e  # FADEHIP_E_RCCL from the communicator set-up: refused, not crashed
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "rccl_two_contexts.txt"), "w") as f:
            f.write(ran + "\n")
        # the contexts stay usable either way
        one = fade_amd.stats_allreduce([ctx], c[:1])
        assert np.array_equal(one, c[:1])
    finally:
        other.close()
