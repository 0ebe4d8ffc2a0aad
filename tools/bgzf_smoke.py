"""One small device BGZF compression, checked with gzip (development aid)."""
import gzip, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fade_amd
EOF_MARK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
ctx = fade_amd.Context(device=0)
rng = np.random.default_rng(1)
for n in (1000, 70000, 1_000_000):
    data = (b"ACGTTGCA" * 50 + rng.integers(0, 256, 100, dtype=np.uint8).tobytes()) * (n // 500 + 1)
    data = data[:n]
    print("compressing", n, flush=True)
    try:
        out = ctx.bgzf_deflate(data)
        print("  ->", len(out), "ok" if gzip.decompress(out + EOF_MARK) == data else "MISMATCH", flush=True)
    except Exception as e:
        print("  error:", e, flush=True)
