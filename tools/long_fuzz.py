"""One-off differential run for reads beyond 512 bases (one alignment per wavefront, the six row classes of
fadehip_kernels.hpp sw_forward64_kernel, and the thread-per-alignment kernel past 4,096 bases), GPU vs the oracle: lengths
drawn over 513 .. 4,300 bases against references of 1 .. 7,000, every adversarial family of tests/helpers.make_pairs, short
pairs mixed into every batch.      GPU box: python tools/long_fuzz.py [seconds] [pairs_per_batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fade_amd  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_sw as S  # noqa: E402
from helpers import make_pairs  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    ctx = fade_amd.Context(device=0)
    t0, total, seed, cells = time.time(), 0, 500, 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        # lengths: uniform, or piled up against the class borders (768 / 1,024 / 1,536 / 2,048 / 3,072 / 4,096)
        if seed % 2:
            qs, rs = make_pairs(rng, n_pairs, lq_range=(513, 4300), lr_range=(1, 7000))
        else:
            qs, rs = [], []
            for _ in range(n_pairs // 7):
                edge = int(rng.choice([768, 1024, 1536, 2048, 3072, 4096])) + int(rng.integers(-2, 3))
                q, r = make_pairs(rng, 7, lq_range=(edge, edge), lr_range=(1, 2 * edge))
                qs += q
                rs += r
        q2, r2 = make_pairs(rng, 60, lq_range=(1, 512), lr_range=(1, 1500))
        S._compare(ctx, oracle, qs + q2, rs + r2)
        total += len(qs) + len(q2)
        cells += sum(len(a) * len(b) for a, b in zip(qs, rs))
        seed += 1
        print("seed %d ok, %d cases (%.1f G cells in long pairs), %.0f s" % (seed - 1, total, cells / 1e9, time.time() - t0), flush=True)
    print("long-read fuzz: %d batches, %d cases, %.1f G cells in pairs beyond 512 bases, 0 mismatches" % (seed - 500, total, cells / 1e9))


if __name__ == "__main__":
    main()
