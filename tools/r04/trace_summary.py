"""FADE_TRACE lines of e2e_quick.json, per variant: when the first front call starts, how long the first five calls take
(buffers, streams and staging memory are made in them), the steady state per call, and what is left after the last read."""
import json
import re
import sys

r = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/e2e_quick.json"))
for label, v in r.items():
    ev = []
    for l in v["timing"]:
        m = re.search(r"\[trace\] (\w+)\s+([\d.]+) ms\s+\+\s*([\d.]+) ms", l)
        if m:
            ev.append((m.group(1), float(m.group(2)), float(m.group(3))))
    if not ev:
        continue
    by = {}
    for n, a, d in ev:
        by.setdefault(n, []).append((a, d))
    fr, bk, wr = by["front"], by["back"], by["fwrite"]
    n = len(fr)
    end = max(a + d for _, a, d in ev)
    k0 = min(6, n - 1)
    steady = (fr[-3][0] - fr[k0][0]) / max(1, n - 3 - k0)
    print("%-8s wall %.3f s | create waited for until %.0f ms, upload ends %.0f, first front %.0f, front[%d] at %.0f ms (%.0f ms for %d calls) | "
          "%d calls, steady %.2f ms per call | last read ends %.0f, end %.0f ms | slow calls: %s" % (
              label, v["seconds"], by["upload"][0][0] + by["upload"][0][1], by["upload"][-1][0] + by["upload"][-1][1], fr[0][0], k0, fr[k0][0],
              fr[k0][0] - fr[0][0], k0, n, steady, by["fread"][-1][0] + by["fread"][-1][1], end,
              " ".join("%s@%.0f+%.0f" % (nm, a, d) for nm, a, d in ev if d > 6 and nm not in ("fasta", "upload"))))
