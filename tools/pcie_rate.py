#!/usr/bin/env python3
"""PCIe-inclusive rate of the level-2 path: every step uploads the batch from host memory, runs, and copies
rs / alignments / stats back (what a host driver pays per batch).  bench.py's `value` excludes this."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fade_amd  # noqa: E402
from fade_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = synth.config("C2")
g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
b = synth.make_reads(g, n, 100, **cfg)
ctx = fade_amd.Context(device=0)
ctx.genome_upload(g.names, g.ascii_contigs())
K = 10
res = {}
for tag, bb in (("pageable", b), ("pinned", ctx.pinned_copy(b)), ("pinned_compact", ctx.pinned_copy(ctx.compact_sequences(b)))):
    for _ in range(2):
        ctx.annotate(bb, cfg["floor_len"], cfg["window"])
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.annotate_upload(0, bb)
        ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
        rs, aln, st = ctx.annotate_collect(0, copy=False)
    dt = (time.perf_counter() - t0) / K
    res[tag] = dict(ms_per_step=dt * 1e3, reads_per_s=n / dt)
up = sum(b[k].nbytes for k in ("tid", "pos", "flag", "has_sa", "l_seq", "cigar_off", "cigar_ops", "seq_off", "seq_packed"))
up_compact = up - b["seq_packed"].nbytes + ctx.compact_sequences(b)["seq_packed"].nbytes
down = rs.nbytes + aln.nbytes
out = dict(reads=n, h2d_bytes=up, h2d_bytes_compact=up_compact, d2h_bytes=down, note="upload + run + collect every step, one slot, no overlap", **res)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pcie_rate.json"), "w"), indent=1)

# phase breakdown with the pinned batch
bb = ctx.pinned_copy(ctx.compact_sequences(b))
import numpy as np
ph = {"upload": [], "run": [], "collect": []}
for _ in range(6):
    t0 = time.perf_counter(); ctx.annotate_upload(0, bb); ctx.sync(); t1 = time.perf_counter()
    ctx.annotate_run(0, cfg["floor_len"], cfg["window"]); ctx.sync(); t2 = time.perf_counter()
    ctx.annotate_collect(0, copy=False); t3 = time.perf_counter()
    ph["upload"].append(t1 - t0); ph["run"].append(t2 - t1); ph["collect"].append(t3 - t2)
print(json.dumps({k: 1e3 * float(np.median(v)) for k, v in ph.items()}))

# two slots, two host threads (how the `fade` driver and bench.py drive a device): upload / run / collect of one batch
# overlap the other batch's
import threading
bbs = [ctx.pinned_copy(ctx.compact_sequences(synth.make_reads(g, n, 100 + k, **cfg))) for k in range(2)]


def loop(slot, k):
    for _ in range(k):
        ctx.annotate_upload(slot, bbs[slot])
        ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
        ctx.annotate_collect(slot, copy=False)


for k in (2, 10):
    th = [threading.Thread(target=loop, args=(s_, k)) for s_ in range(2)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = (time.perf_counter() - t0) / (2 * k)
print(json.dumps({"two_slots_pinned_compact": dict(ms_per_step=dt * 1e3, reads_per_s=n / dt)}))
