"""A window of a rocprofv3 --kernel-trace --memory-copy-trace run of `fade annotate -b` as a timeline: every kernel and copy
with start, end, duration and queue / stream, from `at` (fraction of the run) for `ms` milliseconds; and per kernel the sum of
durations over the whole run.   python tools/r04/stream_timeline.py <dir> [at=0.5] [ms=6]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
at = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ms = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
ev = []
rows = list(csv.DictReader(open(kf)))
cols = rows[0].keys()
qcol = "Queue_Id" if "Queue_Id" in cols else None
scol = "Stream_Id" if "Stream_Id" in cols else None
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1]
    if "sw_pk_kernel" in name:
        short = "score" if ", 1," in name or ", 1>" in name else "pass2"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short[:22], "q%s/s%s" % (r.get(qcol, "?") if qcol else "?", r.get(scol, "?") if scol else "?")))
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?").replace("MEMORY_COPY_", "")[:16], "s%s" % r.get("Stream_Id", "?")))
ev.sort()
t_first, t_last = ev[0][0], max(e[1] for e in ev)
print("run: %.1f ms of device activity, %d events" % ((t_last - t_first) / 1e6, len(ev)))
tot = collections.defaultdict(lambda: [0, 0])
for e in ev:
    tot[e[2]][0] += e[1] - e[0]
    tot[e[2]][1] += 1
for k, (t, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-24s %8.2f ms  %5d x %8.1f us" % (k, t / 1e6, n, t / n / 1e3))
w0 = t_first + at * (t_last - t_first)
print("window from %.1f ms for %.1f ms:" % ((w0 - t_first) / 1e6, ms))
for e in ev:
    if w0 <= e[0] < w0 + ms * 1e6:
        print("%9.1f %9.1f %8.1f  %-24s %s" % ((e[0] - w0) / 1e3, (e[1] - w0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
