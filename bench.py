#!/usr/bin/env python3
"""bench.py — annotate reads/s of the MI355X hot path on BASELINE.json's config 2
(10M x 150 bp PE reads, --window-size 100, default --min-length), synthetic data.

A step = the whole configuration once: 10 DISTINCT batches of 1 M reads each (10 M reads) streamed through the C ABI
the way the `fade` driver does — pinned batch block -> fadehip_annotate_upload (one hipMemcpyAsync) -> fadehip_annotate_run
(asynchronous: gate, score pass, selection, device-planned traced pass, traceback, D2H of the results) ->
fadehip_annotate_results — with several slots in flight, all driven by one host thread.  Records that anno.d:61-65 gives
rs = 0 outright (unmapped, no S op) are left out of the batches by the packing step, as the driver's reader threads do;
they are counted in `value` (they are reads the path has annotated) and in the device's read_count.
Each batch's upload is issued right after the previous run of its slot (the ABI's prefetching upload), so H2D, kernels and
D2H of different batches overlap.  `value` is that streamed, PCIe-inclusive rate; `value_resident` is the same path with
the batches already in HBM.

One process per GPU: `python bench.py --gpus N` spawns N ranks itself (before anything touches the GPU); under
torchrun it is one of the ranks.  Reads shard per rank with no data-path collective; the only collective is the final
all-reduce of the stats.d counters (RCCL).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# int VALU issue: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz (measured by fade_amd/csrc/bench/valu_peak.hip: one
# wave-instruction per 4 cycles per SIMD for v_pk_*, v_add/max, v_bfe, DPP moves alike)
VALU_PEAK_TLANE = 39.3
STAT_NAMES = ["read_count", "clipped", "sup", "art_sup", "art", "art_mate", "aln_l", "aln_r"]
BATCHES_PER_STEP = 10


def pmc_summary(config):
    """The committed rocprofv3 --pmc passes of this build and config (profiles/collect.sh + summarize_pmc.py), or None.
    Replayed, not measured in this run: the fields that come from it carry `source`."""
    path = os.path.join(ROOT, "profiles", "r02_%s_pmc_summary.json" % config)
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except (OSError, ValueError):
        return None, None


def cpu_baseline(genome, cfg, full_batches, reps=3):
    """The oracle's striped AVX2 restatement of the reference path (kind "port": the reference's own parasail/htslib
    build cannot exist in this image) timed on this host's cores over a bounded sample of the same workload, one SW call
    per qualifying clip as the reference does, median of `reps`.  Checker code: never on the product path."""
    from oracle import pyoracle as O
    # one GPU's share of the host: the box exposes the whole node's threads, a 1-GPU job is sized to its share
    cores = min(usable_cpus(), int(os.environ.get("FADE_BENCH_CPU_THREADS", "32")))
    G = O.GenomeHolder(genome.names, [a.tobytes() for a in genome.ascii_contigs()])
    n = sum(len(b["pos"]) for b in full_batches)
    times = []
    # a pass of at least ~2 s: the batches as often as that takes, judged from the first one
    t0 = time.perf_counter()
    O.annotate_batch_soa(G, full_batches[0], cfg["floor_len"], cfg["window"], threads=cores, want_am=False, params=O.default_params(striped=True))
    t_one = time.perf_counter() - t0
    rounds = max(1, int(np.ceil(2.0 / max(t_one * len(full_batches), 1e-3))))
    n *= rounds
    for _ in range(reps):
        t0 = time.perf_counter()
        for _r in range(rounds):
            for b in full_batches:
                O.annotate_batch_soa(G, b, cfg["floor_len"], cfg["window"], threads=cores, want_am=False,
                                     params=O.default_params(striped=True))
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return dict(value=n / dt, unit="reads/s", cores=cores, kind="port",
                sample="%d reads of the same synthetic workload, median of %d passes (%.2f s wall = %.0f core-seconds each; striped "
                       "AVX2 int16 SW+trace oracle, one call per qualifying clip, on %d threads)" % (n, reps, dt, dt * cores, cores))


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the container's cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2: "<quota|max> <period>"
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                n = min(n, max(1, -(-q // period)))
        except (OSError, ValueError):
            pass
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: N fresh ranks, started before this process has touched the GPU."""
    port = int(os.environ.get("MASTER_PORT", "29533"))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch-reads", type=int, default=1_000_000, help="reads per batch; a step is %d batches" % BATCHES_PER_STEP)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--slots", type=int, default=2, help="batches in flight per GPU")
    ap.add_argument("--full-batches", action="store_true", help="send every record (no anno.d:61-65 filter in the packing step)")
    ap.add_argument("--all-sent", action="store_true", help="also measure the secondary figure with every record sent "
                    "(value_all_records_sent: nothing left to the packing step, PCIe-bound)")
    ap.add_argument("--no-all-sent", action="store_true", help=argparse.SUPPRESS)  # (the default now; accepted for old command lines)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0:
        if args.gpus > 1:
            spawn_ranks(args)  # does not return
        world = 1
    elif world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    import fade_amd
    from fade_amd import synth

    # FADE_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 control flow on a 1-GPU box
    backend = os.environ.get("FADE_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")

    cfg = synth.config(args.config)
    genome = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    n_slots = max(1, min(int(args.slots), fade_amd._lib.NUM_SLOTS))
    ctx = fade_amd.Context(device=local, max_batch_reads=max(args.batch_reads, 1 << 20))
    ctx.genome_upload(genome.names, genome.ascii_contigs())
    # per-GPU record range: each rank owns its own shard of the reads (seed 100 (k + 1) + rank)
    full, pinned, pinned_all = [], [], []
    t_gen = time.perf_counter()
    for k in range(BATCHES_PER_STEP):
        b = synth.make_reads(genome, args.batch_reads, 100 * (k + 1) + rank, **cfg)
        b.pop("_truth", None)
        if args.full_batches:
            sub = dict(b)
        else:
            sub, _ = ctx.clipped_only(b)  # what the driver's reader threads do while they pack a batch
        pinned.append(ctx.pinned_batch(sub))
        if args.all_sent and not args.full_batches and k < 4:
            allrec = dict(b)  # every record, for the secondary figure: nothing left to the packing step
            allrec["ref_span_bound"] = sub["ref_span_bound"]
            pinned_all.append(ctx.pinned_batch(allrec))
        full.append(b if (rank == 0 and not args.no_cpu and world == 1) else None)
    t_gen = time.perf_counter() - t_gen
    floor_len, window = cfg["floor_len"], cfg["window"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n_steps, resident, profs=None, pinned=pinned):
        """n_steps x BATCHES_PER_STEP batches through the slots; returns the summed stats.d counters."""
        stats = np.zeros(8, np.int64)
        busy = [False] * n_slots
        seq = 0

        done = [0]

        def finish(slot):
            _, _, st = ctx.annotate_results(slot)
            np.add(stats, st, out=stats)
            done[0] += 1
            if profs is not None and done[0] % 4 == 1:  # HIP-event times of every fourth launch of the timed region
                profs.append(ctx.last_profile(slot))
            busy[slot] = False

        total = n_steps * BATCHES_PER_STEP
        for _ in range(n_steps):
            for k in range(BATCHES_PER_STEP):
                slot = seq % n_slots
                if busy[slot]:
                    finish(slot)
                if not resident and seq < n_slots:
                    ctx.annotate_upload(slot, pinned[k % len(pinned)])  # the first batch of each slot; the later ones were prefetched
                ctx.annotate_run(slot, floor_len, window)
                busy[slot] = True
                # the batch of this slot's NEXT run goes up now, beside the run just enqueued (upload never waits for it)
                if not resident and seq + n_slots < total:
                    ctx.annotate_upload(slot, pinned[(seq + n_slots) % BATCHES_PER_STEP % len(pinned)])
                seq += 1
        for slot in range(n_slots):
            if busy[(seq + slot) % n_slots]:
                finish((seq + slot) % n_slots)
        return stats

    # ---- the measured metric: streamed steps
    run_steps(args.warmup, False)
    barrier()
    profs = []
    t0 = time.perf_counter()
    stats = run_steps(args.steps, False, profs)
    barrier()
    dt = time.perf_counter() - t0
    # ---- the same path with the inputs already in HBM (the slots keep the batches they were last handed)
    res_steps = max(1, min(args.steps, 5))
    run_steps(1, True)
    barrier()
    t0 = time.perf_counter()
    run_steps(res_steps, True)
    barrier()
    dt_res = time.perf_counter() - t0
    # ---- secondary: every record sent (the packing step leaves nothing out): 10x the bytes over PCIe, 10x the records
    # through upload's validation pass and the gate
    dt_all = None
    if pinned_all:
        run_steps(1, False, pinned=pinned_all)
        barrier()
        t0 = time.perf_counter()
        run_steps(2, False, pinned=pinned_all)
        barrier()
        dt_all = time.perf_counter() - t0
    # ---- the dominant kernel alone on the device (one slot, serial): its HIP-event time without a neighbour
    solo = []
    for k in range(3):
        ctx.annotate_upload(0, pinned[k])
        ctx.annotate_run(0, floor_len, window)
        ctx.annotate_results(0)
        solo.append(ctx.last_profile(0))

    t = torch.tensor([dt, dt_res], dtype=torch.float64, device=dev)
    st = torch.tensor(stats, dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)  # RCCL: the final stats reduction
    dt_max, dt_res_max = (float(x) for x in t.tolist())
    reads_per_step = args.batch_reads * BATCHES_PER_STEP
    total_reads = reads_per_step * world * args.steps

    if rank == 0:
        R = fade_amd._lib.row_class(cfg["read_len"])
        kernel = "sw_pk_kernel<%d,1> (score pass)" % R
        fwd = float(np.mean([p["forward_ms"] for p in profs]))        # every fourth launch of the timed region, slots sharing the device
        fwd_solo = float(np.mean([p["forward_ms"] for p in solo[1:]]))  # launches that had the device to themselves
        units = float(np.mean([p["alignments"] for p in profs]))
        alg = float(np.mean([p["algorithmic_bytes"] for p in profs]))  # packed query + window + 16 + 64 per unit
        snap = float(np.mean([p["snapshot_bytes"] for p in profs]))
        cells = float(np.mean([p["cells"] for p in profs]))
        achieved = alg / (fwd * 1e-3) / 1e9
        workload = "%s: %d x %d bp PE reads per GPU per step in %d distinct batches, -w %d, --min-length %d, p_softclip %.2f" % (
            args.config, reads_per_step, cfg["read_len"], BATCHES_PER_STEP, window, floor_len, cfg["p_sc"])
        pmc, pmc_path = pmc_summary(args.config)
        pk = None
        if pmc:
            for name, k in pmc["kernels"].items():
                if "sw_pk_kernel<%d, 1>" % R in name:
                    pk = k
        out = {
            "metric": "annotate reads/sec at 1/2/4/8 MI355X; rs/am tag bit-exact vs ref",
            "value": total_reads / dt_max,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "parity": "bit-exact vs the restated reference semantics (oracle/; unpinned: the reference has no tests or fixtures and cannot be built here)",
            "value_is": "streamed: pinned host batch -> upload (1 hipMemcpyAsync) -> run -> results on the host, %d slots in flight, one host thread" % n_slots,
            "value_resident": reads_per_step * world * res_steps / dt_res_max,
            "value_all_records_sent": None if dt_all is None else reads_per_step * 2 / dt_all,  # this rank's rate, per GPU
            "config": {"workload": workload, "batches_per_step": BATCHES_PER_STEP,
                       "records_sent_per_step": int(sum(p.n for p in pinned)),
                       "upload_bytes_per_step": int(sum(p.nbytes for p in pinned)),
                       "alignments_per_batch": units, "dp_cells_per_batch": cells, "slots_in_flight": n_slots,
                       "records_left_out": "unmapped or no S op (anno.d:61-65: rs = 0); counted in value and read_count" if not args.full_batches else "none"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pk["hbm_bytes_per_launch"] if pk else None,
                         "traffic_source": pmc_path if pk else None,
                         "algorithmic_bytes_per_launch": alg, "units_per_launch": units,
                         "bytes_per_unit": alg / max(units, 1.0),
                         "kernel_ms": fwd, "kernel_ms_alone_on_device": fwd_solo,
                         "frac_with_snapshots": (alg + snap) / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "snapshot_bytes_per_launch": snap,
                         "gcups": cells / (fwd_solo * 1e-3) / 1e9},
            # the kernel is integer-VALU issue bound, not HBM bound; instruction count from the committed PMC pass
            "roofline_valu": None if not pk else {
                "bound": "valu_int_issue", "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s",
                "achieved": pk["SQ_INSTS_VALU"] * 64 / (fwd_solo * 1e-3) / 1e12,
                "frac": pk["SQ_INSTS_VALU"] * 64 / (fwd_solo * 1e-3) / 1e12 / VALU_PEAK_TLANE,
                "valu_busy_frac_pmc": pk["valu_busy_frac"], "source": pmc_path + " (committed profile, not measured in this run)"},
            "kernels_ms": {"gate": float(np.mean([p["gate_ms"] for p in solo[1:]])), "score_pass": fwd_solo,
                           "after_score_pass": float(np.mean([p["traceback_ms"] for p in solo[1:]])),
                           "whole_run": float(np.mean([p["total_ms"] for p in solo[1:]])), "measured": "one batch alone on the device"},
            "stats": {k: int(v) for k, v in zip(STAT_NAMES, st.tolist())},
            "setup_s": {"synthetic_batches": t_gen},
        }
        assert out["stats"]["read_count"] == total_reads, (out["stats"], total_reads)
        if not args.no_cpu and world == 1:
            cb = cpu_baseline(genome, cfg, [b for b in full if b is not None])
            cb["gpu_over_cpu"] = out["value"] / cb["value"]
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
