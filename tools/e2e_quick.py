"""End-to-end `fade annotate -b` on a synthetic BAM, a few settings side by side (wall time of the process and its own
--timing lines).  python tools/e2e_quick.py [n_reads] [label=ENV1=V1,ENV2=V2:extra args ...]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthgen as sg  # noqa: E402
from fade_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
variants = sys.argv[2:] or ["default="]
cfg = synth.config("C2")
tmp = os.environ.get("TMPDIR", "/tmp")
bam, fa = os.path.join(tmp, "e2eq.bam"), os.path.join(tmp, "e2eq.fa")
g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
g.write_fasta(fa)
w = sg.BamWriter(bam, g)
done = 0
while done < n:
    m = min(1_000_000, n - done)
    w.write(sg.make_reads(g, m, 100 + done // 1_000_000, cfg), done // 2)
    done += m
w.close()
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s"])
res = {}
for v in variants:
    label, rest = v.split("=", 1)
    envs, _, extra = rest.partition(":")
    env = dict(os.environ, **dict(kv.split("=", 1) for kv in envs.split(",") if kv))
    exe = os.path.join(ROOT, "tools", "cpu_annotate") if label.startswith("cpu") else os.path.join(ROOT, "fade_amd", "fade")
    best = None
    for rep in range(2):
        out = os.path.join(tmp, "e2eq.out.bam")
        if os.path.exists(out):
            os.remove(out)
        t0 = time.perf_counter()
        with open(out, "wb") as fo:
            fmt = [] if "-u" in extra.split() else ["-b"]  # (-u: uncompressed BGZF instead)
            p = subprocess.run([exe, "annotate", "--timing", "-t", "16", "-w", str(cfg["window"])] + fmt + extra.split() + [bam, fa], stdout=fo,
                               stderr=subprocess.PIPE, env=env)
        dt = time.perf_counter() - t0
        r = dict(seconds=dt, reads_per_s=n / dt, rc=p.returncode, out_bytes=os.path.getsize(out),
                 timing=[l for l in p.stderr.decode(errors="replace").splitlines() if l.startswith(("[timing]", "[fadehip", "[trace]"))])
        if best is None or dt < best["seconds"]:
            best = r
    res[label] = best
    print(label, "%.3f s  %.2f M reads/s  out %d" % (best["seconds"], best["reads_per_s"] / 1e6, best["out_bytes"]), flush=True)
    for l in best["timing"]:
        print("   ", l, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_quick.json"), "w"), indent=1)
