"""Wall time of a process that makes n HIP streams (HSA queues), uses each once and leaves — what a queue costs a short run
from start to finish (creation, and the kernel's teardown at exit).  GPU box."""
import os
import subprocess
import sys
import time

exe = sys.argv[1]
for args in (["0"], ["1"], ["2"], ["3"], ["4"], ["5"], ["6"], ["1", "100"], ["1", "300"]):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        subprocess.run([exe] + args, check=True)
        ts.append(time.perf_counter() - t0)
    print("streams %s%s: best %.1f ms, median %.1f ms" % (args[0], (" + %s MB registered" % args[1]) if len(args) > 1 else "", min(ts) * 1e3, sorted(ts)[2] * 1e3), flush=True)
