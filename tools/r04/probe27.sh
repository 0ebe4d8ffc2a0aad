#!/bin/bash
# round 4, GPU call 29: the box's CPU quota (16 of 256 hardware threads) against the number of inflating threads
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default= > /dev/null 2>&1
python3 - <<'PY'
import os, subprocess, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def stat():
    d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    return int(d["nr_throttled"]), int(d["throttled_usec"])
for t in (16, 15, 14, 13, 12, 24, 16, 14):
    a = stat()
    best = []
    for rep in range(4):
        t0 = time.perf_counter()
        with open("/tmp/o.bam", "wb") as fo:
            subprocess.run([R + "/fade_amd/fade", "annotate", "-t", str(t), "-w", "100", "-b", "/tmp/e2eq.bam", "/tmp/e2eq.fa"], stdout=fo, stderr=subprocess.DEVNULL, check=True)
        best.append(time.perf_counter() - t0)
    b = stat()
    print("-t %2d: %s s (best %.3f = %.2f M reads/s) | throttled %d times, %.1f ms" % (t, " ".join("%.3f" % x for x in best), min(best), 10 / min(best), b[0] - a[0], (b[1] - a[1]) / 1e3), flush=True)
PY
