"""Condense a rocprofv3 --kernel-trace --memory-copy-trace run (csv) into a per-stream timeline of the last N ms."""
import csv
import glob
import sys

d = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
kf = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
mf = glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kf)):
    name = r["Kernel_Name"]
    short = ("score" if "kernel<" in name and ", 1>" in name else "pass2" if "sw_pk_kernel" in name else
             name.split("(")[0].split("::")[-1][:14])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", short, r.get("Stream_Id", "?")))
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", "H2D" if "HOST_TO_DEV" in r["Direction"] else "D2H", r.get("Stream_Id", "?")))
ev.sort()
sc = [e for e in ev if e[3] == "score"]
print("score passes: %d, mean dur %.1f us, mean start-to-start %.1f us (last 20)" % (
    len(sc), sum(e[1] - e[0] for e in sc) / len(sc) / 1e3, (sc[-1][0] - sc[-21][0]) / 20 / 1e3 if len(sc) > 21 else 0))
t_end = ev[-1][1]
off = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0
sel = [e for e in ev if t_end - off - last_ms * 1e6 < e[0] <= t_end - off]
t0 = sel[0][0]
for e in sel:
    if e[1] - e[0] < 15_000 and e[3] not in ("score",):
        continue
    print("%9.1f %9.1f %7.1f %s %-14s s%s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3], e[4]))
