"""Where a streamed step's time goes: host time inside upload / run / results per batch, and the streamed rate for 1..4
slots in flight (C2, 1 M reads per batch, records without an S op left out).  Run on the GPU box."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import fade_amd
from fade_amd import synth

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = synth.config(cfgname)
if os.environ.get("PROBE_READ_LEN"):  # other row classes: e.g. 210 -> R = 14, 300 -> R = 20
    cfg["read_len"] = int(os.environ["PROBE_READ_LEN"])
    cfg["insert_mu"] = max(cfg["insert_mu"], cfg["read_len"] + 200)
g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
ctx = fade_amd.Context(device=0)
ctx.genome_upload(g.names, g.ascii_contigs())
pinned = []
for k in range(nb):
    b = synth.make_reads(g, n, 100 * (k + 1), **cfg)
    sub, _ = ctx.clipped_only(b)
    pinned.append(ctx.pinned_batch(sub))
import os
out = {}
SLOTS = [int(x) for x in os.environ.get("PROBE_SLOTS", "1,2,3,4").split(",")]
MODES = [m == "resident" for m in os.environ.get("PROBE_MODES", "streamed,resident").split(",")]
REPS = int(os.environ.get("PROBE_REPS", "60"))
for n_slots in SLOTS:
    for resident in MODES:
        t_up = t_run = t_res = 0.0
        busy = [False] * n_slots
        reps = REPS
        for slot in range(n_slots):
            ctx.annotate_upload(slot, pinned[slot % nb])
        for phase in range(2):  # warm-up, then timed
            t_up = t_run = t_res = 0.0
            t0 = time.perf_counter()
            prefetch = os.environ.get("PROBE_PREFETCH", "1") == "1"
            for seq in range(reps):
                slot = seq % n_slots
                if busy[slot]:
                    a = time.perf_counter(); ctx.annotate_results(slot); t_res += time.perf_counter() - a
                if not resident and (not prefetch or seq < n_slots):
                    a = time.perf_counter(); ctx.annotate_upload(slot, pinned[seq % nb]); t_up += time.perf_counter() - a
                a = time.perf_counter(); ctx.annotate_run(slot, cfg["floor_len"], cfg["window"]); t_run += time.perf_counter() - a
                busy[slot] = True
                if not resident and prefetch and seq + n_slots < reps:  # the slot's next batch goes up beside this run
                    a = time.perf_counter(); ctx.annotate_upload(slot, pinned[(seq + n_slots) % nb]); t_up += time.perf_counter() - a
            for slot in range(n_slots):
                if busy[slot]:
                    a = time.perf_counter(); ctx.annotate_results(slot); t_res += time.perf_counter() - a
                    busy[slot] = False
            dt = time.perf_counter() - t0
        prof = ctx.last_profile(0)
        out["slots%d_%s" % (n_slots, "resident" if resident else "streamed")] = dict(
            score_ms_last=prof["forward_ms"], after_ms_last=prof["traceback_ms"], gate_ms_last=prof["gate_ms"],
            ms_per_batch=dt / reps * 1e3, reads_per_s=n * reps / dt, host_ms_upload=t_up / reps * 1e3,
            host_ms_run=t_run / reps * 1e3, host_ms_results=t_res / reps * 1e3)
print(json.dumps(out, indent=1))
