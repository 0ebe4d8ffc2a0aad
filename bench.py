#!/usr/bin/env python3
"""bench.py — annotate reads/s of the MI355X hot path on BASELINE.json's config 2
(10M x 150 bp PE reads, --window-size 100, default --min-length), synthetic data.

One process per GPU (torch.distributed / RCCL when --gpus > 1); reads shard per rank with no
data-path collective; the only collective is the final all-reduce of the stats.d counters.
A step = one pass of the device annotate path (gate -> forward SW with trace -> traceback + artifact
gates -> stats) over one batch of reads already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_v12_pmc_summary.json")


def pmc_traffic(workload, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/collect.sh), already corrected (FETCH_SIZE x2, calibrated on known byte counts).
    None when the summary does not cover this workload."""
    try:
        s = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None, None
    if not workload.startswith(s.get("workload", "\0")):
        return None, None
    for name, k in s["kernels"].items():
        if kernel_prefix in name:
            return k["hbm_bytes_per_launch"], k
    return None, None


def cpu_baseline(genome, cfg, n_sample, seed):
    """The oracle's striped AVX2 restatement of the reference path (kind "port": the reference's own
    parasail/htslib build cannot exist in this image) timed on this host's cores over a bounded sample
    of the same workload, one SW call per qualifying clip as the reference does.  Checker code: never on
    the product path."""
    from fade_amd import synth
    from oracle import pyoracle as O
    # one GPU's share of the host: the box exposes the whole node's threads, a 1-GPU job is sized to 16 cores
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("FADE_BENCH_CPU_THREADS", "32")))
    b = synth.make_reads(genome, n_sample, seed, **cfg)
    G = O.GenomeHolder(genome.names, [a.tobytes() for a in genome.ascii_contigs()])
    t0 = time.perf_counter()
    rs, _ = O.annotate_batch_soa(G, b, cfg["floor_len"], cfg["window"], threads=cores, want_am=False,
                                 params=O.default_params(striped=True))
    dt = time.perf_counter() - t0
    return dict(value=n_sample / dt, unit="reads/s", cores=cores, kind="port",
                sample="%d reads of the same synthetic workload (%.2f s wall = %.0f core-seconds, striped "
                       "AVX2 int16 SW+trace oracle on %d threads)" % (n_sample, dt, dt * cores, cores)), b, rs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch-reads", type=int, default=1_000_000, help="reads per step per GPU")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU-baseline sample (0: scale with cores)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--slots", type=int, default=2, help="device slots in flight (1 = strictly serial steps)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import fade_amd
    from fade_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FADE_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 control flow on a 1-GPU box
    backend = os.environ.get("FADE_BENCH_BACKEND", "nccl")
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")

    cfg = synth.config(args.config)
    genome = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    # per-GPU record range: each rank owns its own shard of the reads (seed 100 + rank).
    # Two batches are resident, one per device slot; steps alternate between them and the two slots are driven
    # by two host threads, the way the `fade` driver double-buffers: the latency-bound tail of one batch
    # (traceback of the longest alignments) overlaps the next batch's scoring pass.
    n_slots = max(1, min(int(args.slots), fade_amd._lib.NUM_SLOTS))
    batches = [synth.make_reads(genome, args.batch_reads, 100 * (k + 1) + rank, **cfg) for k in range(n_slots)]
    batch = batches[0]
    ctx = fade_amd.Context(device=local, max_batch_reads=max(args.batch_reads, 1 << 20))
    ctx.genome_upload(genome.names, genome.ascii_contigs())
    for k in range(n_slots):
        ctx.annotate_upload(k, batches[k])  # inputs resident in HBM before the timed region

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import threading

    def run_steps(slot, n_steps, prof_out):
        for _ in range(n_steps):
            ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
            if prof_out is not None and len(prof_out) < 4:  # HIP-event times of a few steps
                prof_out.append(ctx.last_profile(slot))

    def run_all(n_steps, prof_out):
        if n_slots == 1:
            run_steps(0, n_steps, prof_out)
            return
        # the slot threads draw steps from one counter, so an odd or small step count still keeps every slot busy to the end
        left = [n_steps]
        lock = threading.Lock()

        def worker(slot):
            while True:
                with lock:
                    if left[0] == 0:
                        return
                    left[0] -= 1
                run_steps(slot, 1, prof_out if slot == 0 else None)

        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_slots)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()

    run_all(args.warmup, None)
    barrier()
    profs = []
    t0 = time.perf_counter()
    run_all(args.steps, profs)
    barrier()
    dt = time.perf_counter() - t0
    fwd_ms = [p["forward_ms"] for p in profs]
    tb_ms = [p["traceback_ms"] for p in profs]
    gate_ms = [p["gate_ms"] for p in profs]
    prof = ctx.last_profile(0)
    rs, aln, stats = ctx.annotate_collect(0)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    for k in range(1, n_slots):
        stats = stats + ctx.annotate_collect(k)[2]
    st = torch.tensor(stats, dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)  # RCCL: the final stats reduction
    dt_max = float(t.item())
    total_reads = args.batch_reads * world * args.steps

    if rank == 0:
        fwd = float(np.mean(fwd_ms))
        achieved = prof["algorithmic_bytes"] / (fwd * 1e-3) / 1e9
        workload = "%s: %d x %d bp PE reads per GPU per step, -w %d, --min-length %d, p_softclip %.2f" % (
            args.config, args.batch_reads, cfg["read_len"], cfg["window"], cfg["floor_len"], cfg["p_sc"])
        traffic, pmc = pmc_traffic(workload, "sw_pk_kernel<10, 1>")
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
            "config": {"workload": workload,
                       "alignments_per_step": int(prof["alignments"]), "dp_cells_per_step": int(prof["cells"]),
                       "slots_in_flight": n_slots},
            "roofline": {"bound": "hbm", "kernel": "sw_pk_kernel<10,1> (score pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(prof["algorithmic_bytes"]),
                         "kernel_ms": fwd, "gcups": prof["cells"] / (fwd * 1e-3) / 1e9},
            # the kernel is integer-VALU issue bound, not HBM bound; PMC view of the same launch
            "roofline_valu": None if pmc is None else {
                "bound": "valu_int_issue", "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s",
                "achieved": pmc["SQ_INSTS_VALU"] * 64 / (fwd * 1e-3) / 1e12,
                "frac": pmc["SQ_INSTS_VALU"] * 64 / (fwd * 1e-3) / 1e12 / VALU_PEAK_TLANE,
                "valu_busy_frac_pmc": pmc["valu_busy_frac"], "source": "profiles/r01_v12_pmc_summary.json"},
            "kernels_ms": {"gate": float(np.mean(gate_ms)), "sw_forward": fwd, "traceback": float(np.mean(tb_ms))},
            "stats": {k: int(v) for k, v in zip(
                ["read_count", "clipped", "sup", "art_sup", "art", "art_mate", "aln_l", "aln_r"], st.tolist())},
        }
        if not args.no_cpu and world == 1:
            cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("FADE_BENCH_CPU_THREADS", "32")))
            n_sample = args.cpu_sample if args.cpu_sample > 0 else min(4_000_000, 125000 * cores)
            cb, sb, srs = cpu_baseline(genome, cfg, n_sample, 1000 + rank)
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
