#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the scalar oracle.

The reference (blachlylab/fade) ships no tests, fixtures or golden vectors, and cannot be built or
imported in this environment (D + un-vendored parasail/htslib), so these vectors pin the *restated*
reference semantics (oracle/) — parity against the reference itself stays
UNPINNED.  Inputs are synthetic (fade_amd.synth, tests/helpers.make_pairs); outputs come from
oracle/libfadeoracle.so.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fade_amd import synth  # noqa: E402
from helpers import make_pairs  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
import samutil  # noqa: E402


def sw_pairs():
    rng = np.random.default_rng(20261003)
    qs, rs = make_pairs(rng, 210, lq_range=(8, 260), lr_range=(20, 700))
    with open(os.path.join(HERE, "sw_pairs.tsv"), "w") as f:
        f.write("#query\tref\tscore\tend_query\tend_ref\tbeg_query\tposition\tn_ops\tcigar_front16\n")
        pairs = [(q.tobytes().decode(), r.tobytes().decode()) for q, r in zip(qs, rs)]
        # + the pairs that exercise the rules random pairs do not reach (A.4 traceback priority, A.4 gap-tie): inputs in
        # rule_pairs.in.tsv, found by tools/pin_kit/find_pairs.py — so that ONE pin-kit run decides every rule switch
        for line in open(os.path.join(HERE, "rule_pairs.in.tsv")):
            if not line.startswith("#"):
                _, q, r = line.rstrip("\n").split("\t")
                pairs.append((q, r))
        for q, r in pairs:
            res = O.sw(q, r)
            f.write("\t".join([q, r, str(res["score"]), str(res["end_query"]),
                               str(res["end_ref"]), str(res["beg_query"]), str(res["beg_ref"]), str(res["n_ops"]),
                               O.cigar_str(res["ops"][:16])]) + "\n")


def annotate_case(tag, cfg_name, n, contig_len, n_contigs, floor_len=None, window=None, lower_case=False):
    cfg = synth.config(cfg_name)
    cfg["contig_len"] = contig_len
    cfg["n_contigs"] = n_contigs
    g = synth.Genome(n_contigs, contig_len, cfg["genome_seed"])
    b = synth.make_reads(g, n, 77, **cfg)
    fa = g.fasta_bytes()
    if lower_case:  # soft-masked FASTA: analysis.d:63 upper-cases the window
        fa = b"\n".join(l if l.startswith(b">") or (k % 3) else l.lower() for k, l in enumerate(fa.split(b"\n")))
    open(os.path.join(HERE, tag + ".fa"), "wb").write(fa)
    qn = ["%s_%d" % (tag, i // 2) for i in range(n)]
    open(os.path.join(HERE, tag + ".sam"), "w").write(samutil.batch_to_sam(b, g.names, g.lengths, qn))
    names, seqs = samutil.read_fasta(fa.decode())
    G = O.GenomeHolder(names, seqs)
    reads, keep = O.make_reads(b)
    fl = cfg["floor_len"] if floor_len is None else floor_len
    w = cfg["window"] if window is None else window
    with open(os.path.join(HERE, tag + ".expected.tsv"), "w") as f:
        f.write("#floor_len=%d window=%d\n#qname\tflag\trs\tam\tas\tar\tab\n" % (fl, w))
        for i in range(n):
            a = O.annotate_one(G, reads[i], fl, w)
            f.write("\t".join([qn[i], str(int(b["flag"][i])), str(a["rs"])] +
                              [a[k] if a["has_tags"] else "" for k in ("am", "as_", "ar", "ab")]) + "\n")


if __name__ == "__main__":
    sw_pairs()
    annotate_case("anno_c1", "C1", 600, 40_000, 1)                       # default -w 300, one contig
    annotate_case("anno_c2", "C2", 600, 20_000, 3, lower_case=True)      # -w 100, contig edges, soft-masked FASTA
    annotate_case("anno_c5", "C5", 400, 20_000, 2)                       # 30 % clips incl. lengths around the floor
    annotate_case("anno_floor0", "C5", 300, 20_000, 1, floor_len=0, window=50)
    print("golden fixtures written to", HERE)
