"""One-off: damaged BGZF members through the device inflater (fade_amd/csrc/bgzf_inflate.hpp).  Every outcome must be either
an error or the original bytes (the CRC sees to the rest) — never a hang, never other bytes — and the context must go on
working.  Damage: bit flips, overwritten and zeroed spans, spans copied from elsewhere in the stream, truncation, damage aimed
at the first bytes of a member's DEFLATE stream (block type, HLIT / HDIST / HCLEN, the code-length code).
    GPU box: python tools/inflate_corrupt_fuzz.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fade_amd  # noqa: E402
from test_gpu_inflate import member, payloads  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    ctx = fade_amd.Context(device=0)
    ps = payloads()
    names = ["text", "random", "period3", "skewed", "long_codes", "run_then_noise", "low_entropy"]
    rng = np.random.default_rng(99)
    t0, n, n_err, n_same, slowest = time.time(), 0, 0, 0, 0.0
    while time.time() - t0 < budget:
        pick = [names[int(k)] for k in rng.integers(0, len(names), int(rng.integers(1, 5)))]
        level = int(rng.choice([1, 6, 9]))
        ms = [member(ps[k], level) for k in pick]
        good = b"".join(ms)
        want = b"".join(ps[k] for k in pick)
        s = bytearray(good)
        starts = np.cumsum([0] + [len(m) for m in ms[:-1]])
        kind = int(rng.integers(0, 6))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            at, m = int(rng.integers(0, len(s))), int(rng.integers(1, 300))
            s[at:at + m] = rng.integers(0, 256, len(s[at:at + m]), dtype=np.uint8).tobytes()
        elif kind == 2:
            at, m = int(rng.integers(0, len(s))), int(rng.integers(1, 3000))
            s[at:at + m] = bytes(len(s[at:at + m]))
        elif kind == 3:
            at, frm, m = int(rng.integers(0, len(s))), int(rng.integers(0, len(s))), int(rng.integers(1, 2000))
            piece = bytes(s[frm:frm + m])[:len(s) - at]
            s[at:at + len(piece)] = piece
        elif kind == 4:
            s = s[:int(rng.integers(1, len(s)))]
        else:  # the head of one member's DEFLATE stream
            at = int(starts[int(rng.integers(0, len(starts)))]) + 18 + int(rng.integers(0, 12))
            s[at] = int(rng.integers(0, 256))
            if rng.random() < 0.5:
                s[at + 1] = int(rng.integers(0, 256))
        t = time.time()
        try:
            got = ctx.bgzf_inflate(bytes(s), out_cap=len(want) + 4 * 65536).tobytes()
            assert got == want, "damage of kind %d inflated to other bytes without an error (case %d)" % (kind, n)
            n_same += 1
        except fade_amd.FadeHipError:
            n_err += 1
        slowest = max(slowest, time.time() - t)
        n += 1
        if n % 500 == 0:
            assert ctx.bgzf_inflate(good).tobytes() == want
            print("%d damaged streams: %d errors, %d unharmed, slowest call %.1f ms (%.0f s)" % (n, n_err, n_same, slowest * 1e3, time.time() - t0), flush=True)
    assert ctx.bgzf_inflate(good).tobytes() == want
    print("inflate corruption fuzz: %d damaged streams: %d reported as errors, %d inflated to the original bytes, 0 to other bytes, slowest call %.1f ms" % (n, n_err, n_same, slowest * 1e3))


if __name__ == "__main__":
    main()
