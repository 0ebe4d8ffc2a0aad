"""How much of the rs / am output rests on each assumption about the un-vendored arithmetic?
For every FADEHIP_RULE_* switch — and for the dhtslib question whether Cigar.alignedLength also counts I ops — the
fraction of reads whose rs byte or am string differs from the default rules' on C2, C5 and the repeat-rich C6:
GPU against GPU under different fadehip_params.rules (200 k reads per config), each variant cross-checked with the
oracle under the same switch on the first 20 k reads.  Parity with the real parasail / dparasail / dhtslib cannot be
pinned in this environment; this measures the blast radius of each guess.
GPU box: python tools/rule_sensitivity.py [n_reads]   -> gpurun_out/rule_sensitivity.json + a markdown table on stdout"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fade_amd  # noqa: E402
from fade_amd import format_tags, synth  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

RULES = [(1 << 0, "A.3 end cell: ties to the smallest ref index, then query index (off: first in row-major order)"),
         (1 << 1, "A.4 traceback priority DIAG > F > E (off: DIAG > E > F)"),
         (1 << 2, "A.4 a gap opens only on strict >, ties extend (off: ties open)"),
         (1 << 3, "A.4 '=' vs 'X' by residue equality (off: by the sign of the matrix entry)"),
         (1 << 4, "A.5 ref-only step 'D', query-only step 'I' (off: swapped)"),
         (1 << 5, "A.6 CIGAR padded with S for the unaligned query ends (off: no padding)"),
         (1 << 6, "A.1 N vs N scores match (off: mismatch)")]


def run(ctx, g, b, cfg):
    ctx.genome_upload(g.names, g.ascii_contigs())
    rs, aln, _ = ctx.annotate(b, cfg["floor_len"], cfg["window"])
    tags = format_tags(b, g.names, rs, aln)
    am = {i: t["am"] for i, t in tags.items()}
    return rs.copy(), am, aln.copy()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    n_check = min(n, 20_000)
    oracle.build()
    out = {"n_reads": n, "configs": {}}
    for name in ("C2", "C5", "C6"):
        cfg = synth.config(name)
        cfg["contig_len"] = 5_000_000
        g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"], kind=cfg.get("genome_kind", "uniform"))
        b = synth.make_reads(g, n, 7, **cfg)
        sub = synth.take(b, np.arange(n_check))
        G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
        base_ctx = fade_amd.Context(device=0)
        rs0, am0, aln0 = run(base_ctx, g, b, cfg)
        base_ctx.close()
        n_aligned = int(len(aln0))
        n_art = len(am0)
        rows = []
        for bit, text in RULES:
            rules = 0x7f & ~bit
            c = fade_amd.Context(device=0, rules=rules)
            rs1, am1, _ = run(c, g, b, cfg)
            c.close()
            d_rs = int((rs1 != rs0).sum())
            keys = set(am0) | set(am1)
            d_am = sum(1 for k in keys if am0.get(k) != am1.get(k))
            # the oracle under the same switch, on the first reads
            ors, oam = oracle.annotate_batch_soa(G, sub, cfg["floor_len"], cfg["window"], threads=16, params=oracle.default_params(rules=rules))
            ok = bool(np.array_equal(ors, rs1[:n_check])) and all((am1.get(i) if i in am1 else None) == oam[i] for i in range(n_check))
            rows.append(dict(rule_off=text, bit=bit, rs_changed=d_rs, am_changed=d_am, frac_reads_rs=d_rs / n, frac_reads_am=d_am / n,
                             frac_artifacts_am=d_am / max(n_art, 1), oracle_agrees_on_first=n_check if ok else -1))
            assert ok, (name, hex(rules))
        # dhtslib alignedLength counting I as well (include/fadehip.h FADEHIP_REF_CONSUMING_OPS): it enters (a) through the
        # READ's CIGAR (window end, analysis.d:53; right-clip overlap, :110) — these reads have S and M ops only, no change —
        # and (b) through the RESULT CIGAR in analysis.d:111-113, which moves the as / ar / ab slices (never rs or am) of
        # right-side artifacts whose re-alignment has an I op
        n_right_with_i = 0
        for a in aln0:
            if int(a["art"]) & 2:
                ops = a["sw"]["ops"][:min(int(a["sw"]["n_ops"]), 16)]
                n_right_with_i += int(any((int(o) & 15) == 1 for o in ops))
        rows.append(dict(rule_off="dhtslib Cigar.alignedLength also counts I ops", bit=None, rs_changed=0, am_changed=0, frac_reads_rs=0.0,
                         frac_reads_am=0.0, frac_artifacts_am=0.0, as_ar_ab_changed=n_right_with_i,
                         note="no I op in any read CIGAR of these configs; only the as/ar/ab slices of right-side artifacts with an I op in the result move"))
        out["configs"][name] = dict(reads=n, aligned=n_aligned, artifacts=n_art, rows=rows)
        print("\n### %s: %d reads, %d re-aligned, %d artifact calls under the default rules\n" % (name, n, n_aligned, n_art))
        print("| assumption switched off | reads whose rs changes | reads whose am changes | share of artifact calls |")
        print("|---|---|---|---|")
        for r in rows:
            print("| %s | %d (%.4f %%) | %d (%.4f %%) | %.3f %% |" % (r["rule_off"], r["rs_changed"], 100 * r["frac_reads_rs"], r["am_changed"],
                                                                    100 * r["frac_reads_am"], 100 * r["frac_artifacts_am"]))
        sys.stdout.flush()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rule_sensitivity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
