"""Device BGZF compression rate on a BAM payload (pinned source), per phase: python tools/bgzf_rate.py [MB]"""
import ctypes as C
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fade_amd  # noqa: E402
import synthgen as sg  # noqa: E402
from fade_amd import synth  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.config("C2")
cfg["contig_len"] = 2_000_000
g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], 42)
b = sg.make_reads(g, 1_000_000, 5, cfg)
w = sg.BamWriter("/tmp/bgzf_rate.bam", g)
w.write(b, 0)
w.close()
payload = gzip.decompress(open("/tmp/bgzf_rate.bam", "rb").read())
payload = (payload * (mb * 1024 * 1024 // len(payload) + 1))[:mb * 1024 * 1024]
ctx = fade_amd.Context(device=0)
L = ctx._L
ptr = C.c_void_p()
assert L.fadehip_host_alloc(ctx._h, len(payload), C.byref(ptr)) == 0
C.memmove(ptr, payload, len(payload))
res = {}
for rep in range(4):
    t0 = time.perf_counter()
    assert L.fadehip_bgzf_deflate_submit(ctx._h, 0, ptr, len(payload)) == 0
    t1 = time.perf_counter()
    o, n = C.c_void_p(), C.c_size_t()
    assert L.fadehip_bgzf_deflate_wait(ctx._h, 0, C.byref(o), C.byref(n)) == 0
    t2 = time.perf_counter()
    res["rep%d" % rep] = dict(submit_ms=(t1 - t0) * 1e3, wait_ms=(t2 - t1) * 1e3, GBps=len(payload) / (t2 - t0) / 1e9, ratio=n.value / len(payload))
out = C.string_at(o.value, n.value)
eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
assert gzip.decompress(out + eof) == payload
res["bytes"] = len(payload)
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bgzf_rate.json"), "w"), indent=1)
