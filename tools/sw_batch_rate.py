"""Level-1 throughput (fadehip_sw_batch: ASCII pairs in, {score, position, CIGAR} out, H2D / D2H included).
GPU box: python tools/sw_batch_rate.py"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import fade_amd  # noqa: E402
from helpers import concat, make_pairs  # noqa: E402

rng = np.random.default_rng(1)
ctx = fade_amd.Context(device=0)
for n in (1, 100, 10_000, 200_000):
    qs, rs = make_pairs(rng, n, lq_range=(150, 150), lr_range=(300, 345), kinds=("planted", "related", "random"))
    qc, qo = concat(qs)
    rc, ro = concat(rs)
    ctx.sw_batch_packed(qc, qo, rc, ro)
    k = 5 if n >= 10_000 else 50
    t = time.perf_counter()
    for _ in range(k):
        ctx.sw_batch_packed(qc, qo, rc, ro)
    dt = (time.perf_counter() - t) / k
    print("%7d pairs: %8.3f ms per call, %10.0f pairs/s" % (n, dt * 1e3, n / dt), flush=True)
