"""One-off differential run over RANDOM COMBINATIONS of the rule switches (fadehip_params.rules / FO_RULE_*): the suite
(tests/test_gpu_rules.py) checks each switch alone and one combination; here every seed draws a 7-bit mask and compares
level 1 (score, end, begin, CIGAR) and level 2 (rs, am) with the oracle under the same mask.
GPU box: python tools/rules_fuzz.py [n_seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fade_amd  # noqa: E402
from fade_amd import format_tags, synth  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from helpers import concat, make_pairs  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    oracle.build()
    g = synth.Genome(2, 300_000, 9)
    seqs = [a.tobytes().decode() for a in g.ascii_contigs()]
    G = oracle.GenomeHolder(g.names, seqs)
    t0 = time.time()
    total = 0
    seen = set()
    for seed in range(n_seeds):
        rng = np.random.default_rng(1000 + seed)
        rules = int(rng.integers(0, 128))
        if rules == 0:
            rules = 0x7f  # 0 means "default" at the ABI
        seen.add(rules)
        p = oracle.default_params(rules=rules)
        c = fade_amd.Context(device=0, rules=rules)
        try:
            qs, rs = make_pairs(rng, 2400, kinds=("random", "planted", "homopolymer", "tandem", "nrich", "iupac", "related", "lowcomplexity"))
            qc, qo = concat(qs)
            rc, ro = concat(rs)
            got = c.sw_batch_packed(qc, qo, rc, ro)
            exp, exp_ops = oracle.sw_batch(qc, qo, rc, ro, threads=8, max_ops=16, params=p)
            for k in range(len(qs)):
                gk = tuple(int(got[k][f]) for f in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops"))
                assert gk == tuple(int(x) for x in exp[k]), (hex(rules), k, gk, exp[k])
                m = min(int(exp[k][5]), 16)
                assert list(got[k]["ops"][:m]) == list(exp_ops[k][:m]), (hex(rules), k)
            b = synth.make_reads(g, 2500, 50 + seed, read_len=int(rng.choice([100, 150, 250])), window=int(rng.choice([100, 300])),
                                 p_sc=0.6, clip_min=1, clip_max=60, p_sub=0.01)
            w = int(rng.choice([100, 300]))
            c.genome_upload(g.names, g.ascii_contigs())
            rs_, aln, _ = c.annotate(b, 5, w)
            tags = format_tags(b, g.names, rs_, aln)
            ors, oam = oracle.annotate_batch_soa(G, b, 5, w, threads=8, params=p)
            assert np.array_equal(rs_, ors), (hex(rules), np.nonzero(rs_ != ors)[0][:10])
            for i in range(len(ors)):
                assert (tags[i]["am"] if i in tags else None) == oam[i], (hex(rules), i)
            total += len(qs) + len(ors)
        finally:
            c.close()
        print("seed %d rules 0x%02x ok, %d cases, %.0f s" % (seed, rules, total, time.time() - t0), flush=True)
    print("rules fuzz: %d seeds, %d distinct masks, %d cases, 0 mismatches" % (n_seeds, len(seen), total))


if __name__ == "__main__":
    main()
