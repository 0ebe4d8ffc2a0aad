"""A/B of library builds / settings on ONE box: per variant (label=lib[:ENV=VAL,...]) a fresh process runs solo batches
(score-pass, gate and tail times from HIP events) and a two-slot streamed loop with every record sent, twice, alternating.
  python tools/kernel_ab.py C2 new=fade_amd/libfadehip.so old=fade_amd/libfadehip_r02.so ckpt=fade_amd/libfadehip.so:FADEHIP_CKPT=1
Writes gpurun_out/kernel_ab_<config>.json."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cfgname):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import fade_amd
    from fade_amd import synth
    import synthgen as sg
    cfg = synth.config(cfgname)
    if os.environ.get("PROBE_READ_LEN"):
        cfg["read_len"] = int(os.environ["PROBE_READ_LEN"])
        cfg["insert_mu"] = max(cfg["insert_mu"], cfg["read_len"] + 200)
    g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    ctx = fade_amd.Context(device=0)
    ctx.genome_upload(g.names, g.ascii_contigs())
    nb = 4
    pinned = []
    for k in range(nb):
        b = sg.make_reads(g, 1_000_000, 100 * (k + 1), cfg)
        if os.environ.get("PROBE_PREFILTER"):
            sub, _ = ctx.clipped_only(b)
            pinned.append(ctx.pinned_batch(sub))
        else:
            pinned.append(ctx.pinned_batch(sg.with_bounds(b)))
    solo = []
    for k in range(14):
        ctx.annotate_upload(0, pinned[k % nb])
        ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
        ctx.annotate_results(0)
        if k >= 2:
            solo.append(ctx.last_profile(0))
    out = dict(score_ms_mean=float(np.mean([p["forward_ms"] for p in solo])), score_ms_min=float(np.min([p["forward_ms"] for p in solo])),
               gate_ms=float(np.mean([p["gate_ms"] for p in solo])), after_ms=float(np.mean([p["traceback_ms"] for p in solo])),
               total_ms=float(np.mean([p["total_ms"] for p in solo])), alignments=int(solo[-1]["alignments"]),
               candidates=int(solo[-1]["candidates"]), snapshot_bytes=int(solo[-1]["snapshot_bytes"]))
    n_slots, reps = 2, 120
    for phase in range(2):
        busy = [False] * n_slots
        t0 = time.perf_counter()
        for seq in range(reps):
            slot = seq % n_slots
            if busy[slot]:
                ctx.annotate_results(slot)
            if seq < n_slots:
                ctx.annotate_upload(slot, pinned[seq % nb])
            ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
            busy[slot] = True
            if seq + n_slots < reps:
                ctx.annotate_upload(slot, pinned[(seq + n_slots) % nb])
        for slot in range(n_slots):
            if busy[slot]:
                ctx.annotate_results(slot)
        dt = time.perf_counter() - t0
    out["streamed_reads_per_s"] = 1_000_000 * reps / dt
    out["streamed_ms_per_batch"] = dt / reps * 1e3
    print("RESULT " + json.dumps(out))
    ctx.close()


def main():
    cfgname = sys.argv[1]
    variants = []
    for a in sys.argv[2:]:
        label, rest = a.split("=", 1)
        parts = rest.split(":")
        env = dict(kv.split("=", 1) for kv in parts[1].split(",")) if len(parts) > 1 and parts[1] else {}
        variants.append((label, os.path.join(ROOT, parts[0]), env))
    res = {}
    for rnd in range(2):
        for label, lib, env in variants:
            e = dict(os.environ, FADEHIP_LIB=lib, KERNEL_AB_CHILD=cfgname, **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            line = [l for l in p.stdout.decode().splitlines() if l.startswith("RESULT ")]
            r = json.loads(line[0][7:]) if line else dict(error=p.stderr.decode()[-500:])
            res.setdefault(label, []).append(r)
            print(label, rnd, json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "kernel_ab_%s.json" % cfgname), "w"), indent=1)


if __name__ == "__main__":
    if os.environ.get("KERNEL_AB_CHILD"):
        child(os.environ["KERNEL_AB_CHILD"])
    else:
        main()
