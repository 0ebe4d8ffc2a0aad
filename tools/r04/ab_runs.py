"""A/B of `fade annotate -b` settings on one synthetic BAM, the variants interleaved round by round (box noise is +-10 % between
minutes): python tools/r04/ab_runs.py <n_reads> <rounds> label=ENV=V,ENV2=V ...   Prints min / median wall time per variant."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthgen as sg  # noqa: E402
from fade_amd import synth  # noqa: E402

n, rounds = int(sys.argv[1]), int(sys.argv[2])
variants = [v.split("=", 1) for v in sys.argv[3:]]
cfg = synth.config("C2")
tmp = os.environ.get("TMPDIR", "/tmp")
bam, fa, out = (os.path.join(tmp, "ab." + x) for x in ("bam", "fa", "out.bam"))
g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
g.write_fasta(fa)
w = sg.BamWriter(bam, g)
done = 0
while done < n:
    m = min(1_000_000, n - done)
    w.write(sg.make_reads(g, m, 100 + done // 1_000_000, cfg), done // 2)
    done += m
w.close()
res = {l: [] for l, _ in variants}
for r in range(rounds):
    for label, envs in variants:
        env = dict(os.environ, **dict(kv.split("=", 1) for kv in envs.split(",") if kv))
        if os.path.exists(out):
            os.remove(out)
        t0 = time.perf_counter()
        with open(out, "wb") as fo:
            subprocess.run([os.path.join(ROOT, "fade_amd", "fade"), "annotate", "-t", "16", "-w", str(cfg["window"]), "-b", bam, fa], stdout=fo, stderr=subprocess.DEVNULL, env=env, check=True)
        res[label].append(time.perf_counter() - t0)
for label, ts in res.items():
    s = sorted(ts)
    print("%-12s min %.3f s (%.2f M reads/s)  median %.3f s (%.2f M reads/s)  all: %s" % (label, s[0], n / s[0] / 1e6, s[len(s) // 2], n / s[len(s) // 2] / 1e6, " ".join("%.3f" % x for x in ts)), flush=True)
