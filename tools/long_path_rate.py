"""Rates at the edges of the wave kernels' ranges (level 2, resident batches, device time of the forward kernels):
long windows (-w large: the window streams through LDS in chunks), long reads (one alignment per wavefront up to 4,096
bases), and level-1 calls of long pairs as a whole (beyond 4,096 bases or 65,000 columns: a thread per alignment).
GPU box: python tools/long_path_rate.py  -> gpurun_out/long_path_rate.json"""
import json
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tools"))
import fade_amd  # noqa: E402
import synthgen as sg  # noqa: E402
from fade_amd import synth  # noqa: E402
from helpers import make_pairs  # noqa: E402

out = {}
cfg = synth.config("C2")
g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
ctx = fade_amd.Context(device=0)
ctx.genome_upload(g.names, g.ascii_contigs())
b = sg.make_reads(g, 400_000, 5, cfg)
pin = ctx.pinned_batch(sg.with_bounds(b))
for w in (100, 1000, 3900, 10000, 15900):  # windows of ~320, 2,100, 7,900, 20,100, 31,900 columns
    for rep in range(3):
        ctx.annotate_upload(0, pin)
        ctx.annotate_run(0, cfg["floor_len"], w)
        ctx.annotate_results(0)
    p = ctx.last_profile(0)
    out["window_%d" % w] = dict(columns=p["cells"] / max(p["alignments"], 1) / cfg["read_len"], alignments=p["alignments"], score_pass_ms=p["forward_ms"],
                                gcups=p["cells"] / (p["forward_ms"] * 1e-3) / 1e9, after_ms=p["traceback_ms"], candidates=p["candidates"])
    print("-w %5d  %6.0f columns  %6d alignments  score pass %8.3f ms  %7.0f GCUPS  (pass 2 + tracebacks %.3f ms, %d candidates)" % (
        w, out["window_%d" % w]["columns"], p["alignments"], p["forward_ms"], out["window_%d" % w]["gcups"], p["traceback_ms"], p["candidates"]), flush=True)
# long reads (level 2, resident batch): every re-aligned read is on the long list — one alignment per wavefront
for rl, w in ((700, 300), (1000, 300), (2000, 300), (4000, 300)):
    c3 = dict(cfg, read_len=rl, window=w, insert_mu=rl + 300)
    b3 = sg.make_reads(g, 100_000, 9, c3)
    pin3 = ctx.pinned_batch(sg.with_bounds(b3))
    for rep in range(2):
        ctx.annotate_upload(0, pin3)
        ctx.annotate_run(0, cfg["floor_len"], w)
        ctx.annotate_results(0)
    p = ctx.last_profile(0)
    out["long_reads_%d" % rl] = dict(alignments=p["alignments"], forward_ms=p["forward_ms"], gcups=p["cells"] / (p["forward_ms"] * 1e-3) / 1e9, after_ms=p["traceback_ms"])
    print("%4d-base reads, -w %d: %6d alignments, forward %8.3f ms  %7.0f GCUPS  (tracebacks %.3f ms)" % (rl, w, p["alignments"], p["forward_ms"],
          out["long_reads_%d" % rl]["gcups"], p["traceback_ms"]), flush=True)
rng = np.random.default_rng(1)
for n, lq, lr in ((1024, 600, 900), (1024, 1000, 1600), (256, 150, 36000), (64, 4500, 6000)):  # (the last: beyond 4,096 bases, a thread per alignment)
    qs, rs = make_pairs(rng, n, lq_range=(lq, lq), lr_range=(lr, lr), kinds=("related", "random"))
    q = [x.tobytes() for x in qs]
    r = [x.tobytes() for x in rs]
    c2 = fade_amd.Context(device=0, max_ref_len=40000)
    c2.sw_batch(q, r)
    t = time.perf_counter()
    c2.sw_batch(q, r)
    dt = time.perf_counter() - t
    c2.close()
    out["level1_%dx%d" % (lq, lr)] = dict(n=n, ms=dt * 1e3, gcups=n * lq * lr / dt / 1e9)
    print("level 1 (whole call: upload, one alignment per wavefront, tracebacks, results): %d pairs of %d x %d: %.1f ms, %.2f GCUPS" % (n, lq, lr, dt * 1e3, n * lq * lr / dt / 1e9), flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "long_path_rate.json"), "w"), indent=1)
