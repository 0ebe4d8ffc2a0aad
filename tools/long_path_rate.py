"""Throughput of the thread-per-alignment kernel for queries longer than 512 bases (level 1).  GPU box: python tools/long_path_rate.py"""
import sys, time, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import fade_amd
from helpers import make_pairs
rng = np.random.default_rng(1)
ctx = fade_amd.Context(device=0)
for n, lq, lr in ((64, 600, 900), (1024, 600, 900), (4096, 600, 900), (1024, 1000, 1600)):
    qs, rs = make_pairs(rng, n, lq_range=(lq, lq), lr_range=(lr, lr), kinds=("related", "random"))
    q = [x.tobytes() for x in qs]; r = [x.tobytes() for x in rs]
    ctx.sw_batch(q, r)
    t = time.perf_counter(); ctx.sw_batch(q, r); dt = time.perf_counter() - t
    print(n, lq, lr, "%.1f ms" % (dt * 1e3), "%.2f GCUPS" % (n * lq * lr / dt / 1e9), flush=True)
