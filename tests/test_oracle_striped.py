"""Three implementations of SURVEY.md Appendix A must agree bit for bit (§8c item 2): the scalar
full-matrix oracle, the striped AVX2 restatement (the timed CPU baseline), and — on the GPU box —
the HIP kernels (tests/test_gpu_sw.py).  This file covers scalar vs striped on the CPU."""
import numpy as np
import pytest

from helpers import concat, make_pairs


def _agree(oracle, qs, rs):
    qc, qo = concat(qs)
    rc, ro = concat(rs)
    a, ao = oracle.sw_batch(qc, qo, rc, ro, threads=8)
    b, bo = oracle.sw_batch(qc, qo, rc, ro, threads=8, striped=True)
    bad = [k for k in range(len(qs)) if not (np.array_equal(a[k], b[k]) and
                                             np.array_equal(ao[k][:min(a[k][5], 16)], bo[k][:min(a[k][5], 16)]))]
    assert not bad, (len(bad), bad[:5])


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_striped_equals_scalar_on_adversarial_families(oracle, seed):
    rng = np.random.default_rng(seed)
    qs, rs = make_pairs(rng, 2100)
    _agree(oracle, qs, rs)


def test_striped_equals_scalar_on_every_length(oracle):
    rng = np.random.default_rng(7)
    qs, rs = [], []
    for lq in list(range(1, 70)) + [127, 128, 129, 150, 250, 255, 256, 257, 400, 512]:
        for kind in ("planted", "related", "tandem"):
            q, r = make_pairs(rng, 1, lq_range=(lq, lq), lr_range=(1, 900), kinds=(kind,))
            qs += q
            rs += r
    _agree(oracle, qs, rs)


def test_annotate_with_striped_equals_scalar(oracle):
    from fade_amd import synth
    for name in ("C2", "C5"):
        cfg, g, b = synth.make_config(name, 8000, contig_len=200_000)
        G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
        rs1, am1 = oracle.annotate_batch_soa(G, b, cfg["floor_len"], cfg["window"], threads=8)
        rs2, am2 = oracle.annotate_batch_soa(G, b, cfg["floor_len"], cfg["window"], threads=8,
                                             params=oracle.default_params(striped=True))
        assert np.array_equal(rs1, rs2) and am1 == am2
