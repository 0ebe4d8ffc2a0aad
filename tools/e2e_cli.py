#!/usr/bin/env python3
"""End-to-end throughput of the `fade annotate` host driver (BGZF inflate -> pack -> PCIe -> kernels ->
tag formatting -> BGZF deflate), i.e. the PCIe- and I/O-inclusive rate that bench.py's `value` excludes.
Writes gpurun_out/e2e_cli.json.  Usage: python tools/e2e_cli.py [n_reads] [threads]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fade_amd import synth  # noqa: E402

NT16 = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)


def write_sam(path, batch, g):
    n = len(batch["pos"])
    lq = int(batch["l_seq"][0])
    b = batch["seq_packed"].reshape(n, -1)
    codes = np.empty((n, 2 * b.shape[1]), dtype=np.uint8)
    codes[:, 0::2] = b >> 4
    codes[:, 1::2] = b & 15
    seqs = NT16[codes[:, :lq]]
    quals = (batch["qual"].reshape(n, lq) + 33).astype(np.uint8)
    ops = "MIDNSHP=X"
    co = batch["cigar_off"]
    with open(path, "wb") as f:
        f.write(b"@HD\tVN:1.6\tSO:unsorted\n")
        for nm, ln in zip(g.names, g.lengths):
            f.write(b"@SQ\tSN:%s\tLN:%d\n" % (nm.encode(), ln))
        f.write(b"@PG\tID:synth\tPN:fade_amd.synth\n")
        out = []
        for i in range(n):
            c = batch["cigar_ops"][co[i]:co[i + 1]]
            cig = "".join("%d%s" % (int(o) >> 4, ops[int(o) & 15]) for o in c) or "*"
            tid = int(batch["tid"][i])
            out.append(b"r%d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s%s\n" % (
                i // 2, int(batch["flag"][i]), g.names[tid].encode() if tid >= 0 else b"*", int(batch["pos"][i]) + 1,
                60 if tid >= 0 else 0, cig.encode(), seqs[i].tobytes(), quals[i].tobytes(),
                b"\tSA:Z:chr1,1,+,50M,60,0;" if batch["has_sa"][i] else b""))
            if len(out) >= 100000:
                f.write(b"".join(out))
                out = []
        f.write(b"".join(out))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    tmp = os.environ.get("TMPDIR", "/tmp")
    cfg = synth.config("C2")
    g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    t0 = time.time()
    b = synth.make_reads(g, n, 100, **cfg)
    fa, sam, bam = os.path.join(tmp, "e2e.fa"), os.path.join(tmp, "e2e.sam"), os.path.join(tmp, "e2e.bam")
    open(fa, "wb").write(g.fasta_bytes())
    write_sam(sam, b, g)
    print("inputs written in %.1f s" % (time.time() - t0), flush=True)
    fade = os.path.join(ROOT, "fade_amd", "fade")
    base = [fade, "annotate", "--timing", "-t", str(threads), "-w", str(cfg["window"]), "--batch", "262144"]
    res = {}

    def run(tag, args, out):
        t = time.time()
        with open(out, "wb") as fo:
            p = subprocess.run(base + args, stdout=fo, stderr=subprocess.PIPE)
        dt = time.time() - t
        assert p.returncode == 0, p.stderr.decode()
        res[tag] = dict(seconds=dt, reads_per_s=n / dt, out_bytes=os.path.getsize(out),
                        timing=[l for l in p.stderr.decode().splitlines() if l.startswith("[timing]")])
        print(tag, res[tag], flush=True)

    run("sam_to_bam", ["-b", sam, fa], bam)  # also produces the BAM input for the next runs
    run("bam_to_bam", ["-b", bam, fa], os.path.join(tmp, "e2e.out.bam"))
    run("bam_to_ubam", ["-u", bam, fa], os.path.join(tmp, "e2e.out.ubam"))
    run("bam_to_sam", [bam, fa], os.path.join(tmp, "e2e.out.sam"))
    res["n_reads"] = n
    res["threads"] = threads
    res["note"] = "includes loading + uploading the 100 Mbp FASTA (text -> 4-bit in HBM) once per run"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_cli.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
