"""One-off extended differential run (not part of the suite): many more seeds of tests/test_gpu_fuzz.py's record
generator and of tests/helpers.make_pairs than the suite affords, GPU vs the oracle.
GPU box: python tools/extended_fuzz.py [n_seeds] [pairs_per_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fade_amd  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import test_gpu_sw as S  # noqa: E402
from helpers import make_pairs  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    ctx = fade_amd.Context(device=0)
    t0 = time.time()
    total = 0
    for seed in range(100, 100 + n_seeds):
        rng = np.random.default_rng(seed)
        floor_len = int(rng.choice([0, 3, 5, 5, 12]))
        window = int(rng.choice([30, 100, 100, 300, 700]))
        F.test_fuzz_records.__wrapped__(ctx, oracle, seed, floor_len, window) if hasattr(F.test_fuzz_records, "__wrapped__") \
            else F.test_fuzz_records(ctx, oracle, seed, floor_len, window)
        qs, rs = make_pairs(rng, n_pairs, lq_range=(1, 512), lr_range=(1, 1500))
        S._compare(ctx, oracle, qs, rs)
        total += 3000 + n_pairs
        print("seed %d ok (floor %d, window %d), %d cases, %.0f s" % (seed, floor_len, window, total, time.time() - t0), flush=True)
    print("extended fuzz: %d seeds, %d cases, 0 mismatches" % (n_seeds, total))


if __name__ == "__main__":
    main()
