"""Start-to-finish wall time of a minimal HIP process (one stream, one kernel, _exit) under runtime settings.  GPU box."""
import os
import subprocess
import sys
import time

exe = sys.argv[1]
envs = [
    {}, {"HSA_ENABLE_INTERRUPT": "0"}, {"HSA_ENABLE_SDMA": "0"}, {"ROCR_VISIBLE_DEVICES": "0"}, {"GPU_MAX_HW_QUEUES": "1"},
    {"HIP_VISIBLE_DEVICES": "0"}, {"AMD_DIRECT_DISPATCH": "0"}, {"HSA_NO_SCRATCH_RECLAIM": "1"}, {"HSA_ENABLE_INTERRUPT": "0", "HSA_ENABLE_SDMA": "0"}, {},
]
for env in envs:
    for args in (["1"], ["3"]):
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            subprocess.run([exe] + args, check=True, env=dict(os.environ, **env))
            ts.append(time.perf_counter() - t0)
        print("%-50s streams %s: best %.1f ms, median %.1f ms" % (env, args[0], min(ts) * 1e3, sorted(ts)[3] * 1e3), flush=True)
