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


def write_sam(path, batch, g, append=False, name_base=0):
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
    with open(path, "ab" if append else "wb") as f:
        if not append:
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
name_base + i // 2, int(batch["flag"][i]), g.names[tid].encode() if tid >= 0 else b"*", int(batch["pos"][i]) + 1,
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
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s"])
    t0 = time.time()
    fa, sam, bam = os.path.join(tmp, "e2e.fa"), os.path.join(tmp, "e2e.sam"), os.path.join(tmp, "e2e.bam")
    open(fa, "wb").write(g.fasta_bytes())
    # the reads in pieces of 1 M (memory), appended to one SAM, then converted to an un-annotated BAM
    done = 0
    while done < n:
        m = min(1_000_000, n - done)
        b = synth.make_reads(g, m, 100 + done // 1_000_000, **cfg)
        write_sam(sam, b, g, append=done > 0, name_base=done // 2)
        done += m
    with open(bam, "wb") as fo:
        subprocess.check_call([os.path.join(ROOT, "tools", "sam2bam"), sam], stdout=fo)
    print("inputs written in %.1f s (%d reads, BAM %d bytes)" % (time.time() - t0, n, os.path.getsize(bam)), flush=True)
    fade = os.path.join(ROOT, "fade_amd", "fade")
    cpu = os.path.join(ROOT, "tools", "cpu_annotate")
    res = {}

    def run(tag, exe, args, out, t, env=None):
        base = [exe, "annotate", "--timing", "-t", str(t), "-w", str(cfg["window"]), "--batch", "262144"]
        if os.path.exists(out):  # (truncating the previous leg's 1.6 GB output costs 0.3 s: not part of this leg)
            os.remove(out)
        t1 = time.time()
        with open(out, "wb") as fo:
            p = subprocess.run(base + args, stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
        dt = time.time() - t1
        assert p.returncode == 0, p.stderr.decode()
        res[tag] = dict(seconds=dt, reads_per_s=n / dt, out_bytes=os.path.getsize(out), threads=t,
                        timing=[l for l in p.stderr.decode().splitlines() if l.startswith("[timing]")])
        print(tag, res[tag], flush=True)

    out_gpu, out_cpu = os.path.join(tmp, "e2e.gpu.bam"), os.path.join(tmp, "e2e.cpu.bam")
    if os.environ.get("E2E_SAM_AB"):  # SAM -> BAM with this build and with another binary (A/B of host changes)
        for rep in range(2):
            run("sam_to_bam_other_%d" % rep, os.environ["E2E_SAM_AB"], ["-b", sam, fa], out_gpu, threads)
            run("sam_to_bam_this_%d" % rep, fade, ["-b", sam, fa], out_gpu, threads)
        json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_sweep.json"), "w"), indent=1)
        return
    for t in [int(x) for x in os.environ.get("E2E_SWEEP", "").split(",") if x]:  # -t sweep of the BAM -> BAM leg only
        run("gpu_bam_to_bam_t%d" % t, fade, ["-b", bam, fa], out_gpu, t)
        if os.environ.get("E2E_SLOTS2"):  # A/B: the double-buffered form of the driver
            run("gpu_bam_to_bam_t%d_slots2" % t, fade, ["-b", bam, fa], out_gpu, t, env={"FADE_SLOTS": "2"})
    if os.environ.get("E2E_SWEEP"):
        json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_sweep.json"), "w"), indent=1)
        return
    run("gpu_bam_to_bam", fade, ["-b", bam, fa], out_gpu, threads)
    run("gpu_bam_to_bam_effort1", fade, ["-b", bam, fa], os.path.join(tmp, "e2e.out3.bam"), threads, env={"FADE_BGZF_EFFORT": "1"})
    run("gpu_bam_to_bam_effort3", fade, ["-b", bam, fa], os.path.join(tmp, "e2e.out3.bam"), threads, env={"FADE_BGZF_EFFORT": "3"})
    run("gpu_bam_to_bam_again", fade, ["-b", bam, fa], out_gpu, threads)
    run("gpu_bam_to_bam_32_threads", fade, ["-b", bam, fa], os.path.join(tmp, "e2e.out3.bam"), 32)
    run("gpu_bam_to_ubam", fade, ["-u", bam, fa], os.path.join(tmp, "e2e.out.ubam"), threads)
    run("gpu_sam_to_bam", fade, ["-b", sam, fa], os.path.join(tmp, "e2e.out2.bam"), threads)
    # the CPU comparator: the same reader / writer around the striped AVX2 restatement of annotateTask
    run("cpu_bam_to_bam", cpu, ["-b", bam, fa], out_cpu, threads)
    if threads != 32:
        run("cpu_bam_to_bam_32_threads", cpu, ["-b", bam, fa], out_cpu, 32)
    res["gpu_over_cpu_same_threads"] = res["gpu_bam_to_bam"]["reads_per_s"] / res["cpu_bam_to_bam"]["reads_per_s"]
    # same records from both (the BGZF payloads: everything but the header's @PG CL field)
    if n <= 4_000_000:
        import gzip
        a, c = gzip.decompress(open(out_gpu, "rb").read()), gzip.decompress(open(out_cpu, "rb").read())
        la, lc = int.from_bytes(a[4:8], "little"), int.from_bytes(c[4:8], "little")
        res["gpu_and_cpu_records_identical"] = a[8 + la:] == c[8 + lc:]
        assert res["gpu_and_cpu_records_identical"]
    res["n_reads"] = n
    res["note"] = ("wall time of the whole process: HIP start-up, loading the 100 Mbp FASTA, its upload (text -> 4-bit in HBM), "
                   "BGZF inflate, annotate, tags, BGZF deflate; GPU and CPU legs share reader / writer code and thread count")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_cli.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
