"""Score-pass time of the two sweeps of sw_pk_kernel: C2 as generated (pure A,C,G,T queries -> single-perm sweep) and
the same batch with one N planted in every read (-> general sweep).  Run on the GPU box: python tools/special_path_rate.py"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fade_amd
from fade_amd import synth
cfg = synth.config("C2")
g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
b = synth.make_reads(g, 1_000_000, 100, **cfg)
ctx = fade_amd.Context(device=0, max_batch_reads=1 << 20)
ctx.genome_upload(g.names, g.ascii_contigs())
def rate(batch, tag):
    ctx.annotate_upload(0, batch)
    for _ in range(3): ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
    t = time.perf_counter()
    for _ in range(10): ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
    dt = (time.perf_counter() - t) / 10
    p = ctx.last_profile(0)
    print(tag, "%.3f ms/step" % (dt * 1e3), "score pass %.3f ms" % p["forward_ms"], flush=True)
rate(b, "pure ACGT")
b2 = dict(b)
seq = b["seq_packed"].copy()
off = b["seq_off"][:-1].astype(np.int64)
seq[off + 10] |= 0x0f  # base 21 of every read becomes N (code 15)
b2["seq_packed"] = seq
rate(b2, "one N per read")
