#!/usr/bin/env python3
"""bench.py — annotate reads/s of the MI355X hot path on BASELINE.json's config 2
(10M x 150 bp PE reads, --window-size 100, default --min-length), synthetic data.

A step = the whole configuration once: 10 DISTINCT batches of 1 M reads each (10 M reads) streamed through the C ABI
the way the `fade` driver does — pinned batch block -> fadehip_annotate_upload (one hipMemcpyAsync) -> fadehip_annotate_run
(asynchronous: gate, score pass, selection, device-planned traced pass, traceback, D2H of the results) ->
fadehip_annotate_results — with several slots in flight, all driven by one host thread.

`value` counts only work done inside the clock: EVERY record of the 10 M is sent to the device in every step and the
device's gate kernel does anno.d:61-65 (unmapped / no S op -> rs = 0) for all of them.  What the packing step (outside
the clock: it is the reader's job, htslib's in the reference) does is what a BAM decoder does anyway: it lays the records'
fixed fields and CIGARs out as the batch block, and copies the packed bases only of records that have a soft clip (the
only ones anno.d:61 lets through), passing the block's bounds along (ABI 3).  Nothing is filtered out.
`value_prefiltered` is round 2's figure (records without an S op left out of the batches by the packing step, counted via
n_skipped); `value_resident` the all-records path with the batches already in HBM.  The `e2e` block is the whole program
on a 10 M-read BAM file: `fade annotate -b` (GPU) and tools/cpu_annotate (the same reader / writer around the CPU oracle)
on the same host threads, wall time of the process.

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
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Integer / packed-16 VALU issue ceiling: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-instr/s.  It is the rate
# fade_amd/csrc/bench/valu_peak.hip measures for the opcodes the sweep is made of (v_pk_*16, v_perm, DPP moves, int32
# add / max: one wave-instruction per ~4.4 cycles per SIMD with 8 waves, profiles/r01_valu_peak.txt).  Plain f32 VALU
# issues at twice that on this chip (2.6 cycles measured, 78.6 T nominal "SIMD-32"); `frac_of_f32_issue_peak` prices the
# kernel against that ceiling too, although no integer or packed-16 opcode reaches it.
VALU_PEAK_TLANE = 39.3
VALU_F32_PEAK_TLANE = 78.6
STAT_NAMES = ["read_count", "clipped", "sup", "art_sup", "art", "art_mate", "aln_l", "aln_r"]
BATCHES_PER_STEP = 10
ROUND = "r03"


def pmc_summary(config):
    """The committed rocprofv3 --pmc passes of this build and config (profiles/collect.sh + summarize_pmc.py), or None.
    Replayed, not measured in this run: the fields that come from it carry `source`."""
    for rnd in (ROUND, "r02"):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (rnd, config))
        try:
            return json.load(open(path)), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
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
                work="every record of the sample goes through annotateTask: the anno.d:61-65 check for all of them, one SW call per "
                     "qualifying clip — the same work `value` times on the device (the packing of the batch blocks is outside both clocks)",
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


def e2e_legs(bam, fa, n_reads, cfg, threads, tmp):
    """The whole program on a BAM file, GPU driver and CPU comparator on the same threads: wall time of each process
    (start-up, FASTA load and genome upload, BGZF inflate, annotate, tags, BGZF deflate, exit), best of 2 runs each."""
    fade = os.path.join(ROOT, "fade_amd", "fade")
    cpu = os.path.join(ROOT, "tools", "cpu_annotate")
    res = {}

    def run(tag, exe, reps, env=None):
        best = None
        out = os.path.join(tmp, "bench_e2e.%s.bam" % tag)
        for _ in range(reps):
            if os.path.exists(out):
                os.remove(out)  # (truncating the previous output is not part of the run)
            t1 = time.perf_counter()
            with open(out, "wb") as fo:
                p = subprocess.run([exe, "annotate", "--timing", "-t", str(threads), "-w", str(cfg["window"]), "--min-length",
                                    str(cfg["floor_len"]), "-b", bam, fa], stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
            dt = time.perf_counter() - t1
            if p.returncode != 0:
                return dict(error=p.stderr.decode(errors="replace")[-400:])
            r = dict(seconds=dt, reads_per_s=n_reads / dt, out_bytes=os.path.getsize(out),
                     timing=[l for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("[timing]")][:8])
            if best is None or r["seconds"] < best["seconds"]:
                best = r
        os.remove(out)
        return best

    # default: the file path on the device (framing, annotate, tags, deflate as kernels), BGZF inflate on the host pool
    res["gpu"] = run("gpu", fade, 2)
    res["gpu_device_inflate"] = run("gpu_di", fade, 2, {"FADE_BAM_INFLATE": "device"})  # ... inflate on the device too
    res["gpu_host_pipeline"] = run("gpu_hp", fade, 1, {"FADE_BAM_DEVICE": "0"})  # round 2's pipeline (+ device deflate)
    res["cpu"] = run("cpu", cpu, 1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch-reads", type=int, default=1_000_000, help="reads per batch; a step is %d batches" % BATCHES_PER_STEP)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (fade annotate / tools/cpu_annotate on a BAM file)")
    ap.add_argument("--e2e-reads", type=int, default=10_000_000)
    ap.add_argument("--slots", type=int, default=2, help="batches in flight per GPU")
    ap.add_argument("--synth", default="native", choices=["native", "numpy"], help="generator of the synthetic reads "
                    "(tools/synthgen.cpp on all host threads, or fade_amd/synth.py: the same laws, another random stream)")
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
    if cfg.get("p_clip_indel"):
        args.synth = "numpy"  # (C6's indels in planted clips exist in fade_amd/synth.py only)
    t_setup = {}
    t0 = time.perf_counter()
    if args.synth == "native":
        import synthgen as sg
        genome = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"], kind=cfg.get("genome_kind", "uniform"))
    else:
        genome = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"], kind=cfg.get("genome_kind", "uniform"))
    t_setup["genome"] = time.perf_counter() - t0
    n_slots = max(1, min(int(args.slots), fade_amd._lib.NUM_SLOTS))
    ctx = fade_amd.Context(device=local, max_batch_reads=max(args.batch_reads, 1 << 20))
    ctx.genome_upload(genome.names, genome.ascii_contigs())
    do_cpu = rank == 0 and not args.no_cpu and world == 1
    do_e2e = rank == 0 and not args.no_e2e and world == 1 and args.synth == "native"
    tmp = os.environ.get("TMPDIR", "/tmp")
    bam_path, fa_path = os.path.join(tmp, "bench_e2e.bam"), os.path.join(tmp, "bench_e2e.fa")
    bam_writer = None
    if do_e2e:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s"])
        genome.write_fasta(fa_path)
        bam_writer = sg.BamWriter(bam_path, genome)
    # per-GPU record range: each rank owns its own shard of the reads (stream 100 (k + 1) + rank)
    full, pinned, pinned_pre = [], [], []
    t_gen = t_pack = t_bam = 0.0
    e2e_written = 0
    for k in range(BATCHES_PER_STEP):
        t0 = time.perf_counter()
        if args.synth == "native":
            b = sg.make_reads(genome, args.batch_reads, 100 * (k + 1) + rank, cfg)
        else:
            b = synth.make_reads(genome, args.batch_reads, 100 * (k + 1) + rank, **cfg)
            b.pop("_truth", None)
        t1 = time.perf_counter()
        # what a packing thread hands over: every record, bases only where the device can align, the block's bounds
        allrec = sg.with_bounds(b) if args.synth == "native" else ctx.with_bounds(b)
        pinned.append(ctx.pinned_batch(allrec))
        sub, _ = ctx.clipped_only(b)  # round 2's form: the records anno.d:61-65 settles left out by the packing step
        pinned_pre.append(ctx.pinned_batch(sub))
        t2 = time.perf_counter()
        if bam_writer is not None and e2e_written < args.e2e_reads:
            m = min(len(b["pos"]), args.e2e_reads - e2e_written)
            bam_writer.write(b if m == len(b["pos"]) else synth.take(b, np.arange(m)), e2e_written // 2)
            e2e_written += m
        t3 = time.perf_counter()
        t_gen, t_pack, t_bam = t_gen + t1 - t0, t_pack + t2 - t1, t_bam + t3 - t2
        full.append(b if (do_cpu and k < 2) else None)  # the CPU leg's bounded sample: 2 M reads
    if bam_writer is not None:
        t0 = time.perf_counter()
        bam_writer.close()
        t_bam += time.perf_counter() - t0
    t_setup.update(synthetic_batches=t_gen, pack_pinned_blocks=t_pack, e2e_bam=t_bam)
    floor_len, window = cfg["floor_len"], cfg["window"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n_steps, resident, profs=None, pinned=pinned, n_slots=n_slots):
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

    # ---- the measured metric: streamed steps, every record sent
    run_steps(args.warmup, False)
    barrier()
    profs = []
    t0 = time.perf_counter()
    stats = run_steps(args.steps, False, profs)
    barrier()
    dt = time.perf_counter() - t0
    # ---- the same path with the inputs already in HBM (the slots keep the batches they were last handed)
    res_steps = max(1, min(args.steps, 20))
    run_steps(1, True)
    barrier()
    t0 = time.perf_counter()
    run_steps(res_steps, True)
    barrier()
    dt_res = time.perf_counter() - t0
    # ---- round 2's figure: records without an S op left out by the packing step (a tenth of the records cross PCIe)
    run_steps(1, False, pinned=pinned_pre)
    barrier()
    t0 = time.perf_counter()
    run_steps(res_steps, False, pinned=pinned_pre)
    barrier()
    dt_pre = time.perf_counter() - t0
    # ---- the dominant kernel alone on the device (one slot, serial): the HIP-event times the roofline is priced with
    solo = []
    for k in range(2 * BATCHES_PER_STEP + 2):
        ctx.annotate_upload(0, pinned[k % BATCHES_PER_STEP])
        ctx.annotate_run(0, floor_len, window)
        ctx.annotate_results(0)
        if k >= 2:
            solo.append(ctx.last_profile(0))

    t = torch.tensor([dt, dt_res, dt_pre], dtype=torch.float64, device=dev)
    st = torch.tensor(stats, dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)  # RCCL: the final stats reduction
    dt_max, dt_res_max, dt_pre_max = (float(x) for x in t.tolist())
    reads_per_step = args.batch_reads * BATCHES_PER_STEP
    total_reads = reads_per_step * world * args.steps

    if rank == 0:
        R = fade_amd._lib.row_class(cfg["read_len"])
        kernel = "sw_pk_kernel<%d,1> (score pass)" % R
        fwd_streamed = float(np.mean([p["forward_ms"] for p in profs]))  # every fourth launch of the timed region, slots sharing the device
        fwd = float(np.mean([p["forward_ms"] for p in solo]))            # launches that had the device to themselves
        units = float(np.mean([p["alignments"] for p in solo]))
        alg = float(np.mean([p["algorithmic_bytes"] for p in solo]))  # packed query + window + 16 + 64 per unit
        snap = float(np.mean([p["snapshot_bytes"] for p in profs]))
        cells = float(np.mean([p["cells"] for p in solo]))
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
            "value_is": "streamed, every record sent: pinned host batch of all %d records -> upload (1 hipMemcpyAsync) -> run (the device's gate does "
                        "anno.d:61-65 for every record) -> results on the host, %d slots in flight, one host thread; the packing of the "
                        "batch blocks (the BAM decoder's job) is outside the clock, as is the reference's htslib decode" % (reads_per_step, n_slots),
            "value_resident": reads_per_step * world * res_steps / dt_res_max,
            "value_prefiltered": reads_per_step * world * res_steps / dt_pre_max,
            "value_prefiltered_is": "round 2's headline: the packing step leaves records without an S op out (n_skipped); not an all-work rate",
            "config": {"workload": workload, "batches_per_step": BATCHES_PER_STEP,
                       "records_sent_per_step": int(sum(p.n for p in pinned)),
                       "upload_bytes_per_step": int(sum(p.nbytes for p in pinned)),
                       "records_with_bases_per_step": int(sum(p.c.n_with_seq for p in pinned)),
                       "alignments_per_batch": units, "dp_cells_per_batch": cells, "slots_in_flight": n_slots,
                       "records_left_out": "none"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pk["hbm_bytes_per_launch"] if pk else None,
                         "traffic_source": pmc_path if pk else None,
                         "algorithmic_bytes_per_launch": alg, "units_per_launch": units,
                         "bytes_per_unit": alg / max(units, 1.0),
                         "kernel_ms": fwd, "kernel_ms_is": "HIP events around the kernel on its stream, launches that had the device to themselves "
                                                           "(one slot, serial; %d launches after the timed region)" % len(solo),
                         "kernel_ms_streamed": fwd_streamed,
                         "snapshot_bytes_per_launch": snap,
                         "gcups": cells / (fwd * 1e-3) / 1e9},
            # the kernel is integer-VALU issue bound, not HBM bound; instruction count from the committed PMC pass
            "roofline_valu": None if not pk else {
                "bound": "valu_int_issue", "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s",
                "peak_is": "256 CU x 4 SIMD x 16 lanes x 2.4 GHz: the measured issue rate of the integer / packed-16 opcodes the sweep consists of "
                           "(profiles/r01_valu_peak.txt), NOT the f32 rate (%.1f T: twice as fast, but no int or packed-16 opcode issues at it)" % VALU_F32_PEAK_TLANE,
                "achieved": pk["SQ_INSTS_VALU"] * 64 / (fwd * 1e-3) / 1e12,
                "frac": pk["SQ_INSTS_VALU"] * 64 / (fwd * 1e-3) / 1e12 / VALU_PEAK_TLANE,
                "frac_of_f32_issue_peak": pk["SQ_INSTS_VALU"] * 64 / (fwd * 1e-3) / 1e12 / VALU_F32_PEAK_TLANE,
                "valu_busy_frac_pmc": pk["valu_busy_frac"], "source": pmc_path + " (committed profile, not measured in this run)"},
            "kernels_ms": {"gate": float(np.mean([p["gate_ms"] for p in solo])), "score_pass": fwd,
                           "after_score_pass": float(np.mean([p["traceback_ms"] for p in solo])),
                           "whole_run": float(np.mean([p["total_ms"] for p in solo])), "measured": "one batch alone on the device"},
            "stats": {k: int(v) for k, v in zip(STAT_NAMES, st.tolist())},
            "setup_s": t_setup,
        }
        assert out["stats"]["read_count"] == total_reads, (out["stats"], total_reads)
        # beside the metric: the forward kernels' rates at the edges of the hot path (long windows, long reads), HIP-event times
        if world == 1 and args.synth == "native" and not args.no_e2e:
            try:
                ex = {}
                bw = sg.make_reads(genome, 400_000, 77, cfg)
                pw = ctx.pinned_batch(sg.with_bounds(bw))
                for _ in range(2):
                    ctx.annotate_upload(0, pw)
                    ctx.annotate_run(0, floor_len, 10000)
                    ctx.annotate_results(0)
                p = ctx.last_profile(0)
                ex["windows_of_20100_columns_gcups"] = p["cells"] / (p["forward_ms"] * 1e-3) / 1e9
                cl = dict(cfg, read_len=1000, window=300, insert_mu=1300)
                bl = sg.make_reads(genome, 50_000, 78, cl)
                pl = ctx.pinned_batch(sg.with_bounds(bl))
                for _ in range(2):
                    ctx.annotate_upload(0, pl)
                    ctx.annotate_run(0, floor_len, 300)
                    ctx.annotate_results(0)
                p = ctx.last_profile(0)
                ex["reads_of_1000_bases_gcups"] = p["cells"] / (p["forward_ms"] * 1e-3) / 1e9
                ex["what"] = ("forward kernels alone (HIP events), resident batches: -w 10000 on the workload's reads (the window streams through LDS in "
                              "chunks on the packed wave kernels); 1000-base reads at -w 300 (one alignment per wavefront, sw_forward64_kernel)")
                out["extras"] = ex
            except Exception as exc:  # (never at the expense of the metric's line)
                out["extras"] = {"error": repr(exc)[:200]}
    ctx.close()
    if rank == 0:
        if do_e2e:
            try:
                e = e2e_legs(bam_path, fa_path, e2e_written, cfg, usable_cpus(), tmp)
            except Exception as exc:  # (a missing binary, a full disk: never at the expense of the metric's line)
                e = {"gpu": {"error": repr(exc)[:300]}}
            g, c = e.get("gpu") or {}, e.get("cpu") or {}
            out["e2e"] = {"what": "`fade annotate -b` BAM -> BAM on a %d-read file of this workload, wall time of the whole process, %d host threads.  gpu = the "
                                  "default: the file path on the device (record framing, annotate, tags, BGZF deflate as kernels; BGZF inflate on the host pool); "
                                  "gpu_device_inflate = FADE_BAM_INFLATE=device (inflate as a kernel too: only compressed bytes cross PCIe); gpu_host_pipeline = "
                                  "FADE_BAM_DEVICE=0 (records handled by the host pool, level-2 batches to the device, deflate on the device); "
                                  "cpu = tools/cpu_annotate: the same reader / writer / codec around the CPU oracle" % (e2e_written, usable_cpus()),
                          "gpu_reads_per_s": g.get("reads_per_s"), "cpu_reads_per_s": c.get("reads_per_s"),
                          "gpu_over_cpu": (g["reads_per_s"] / c["reads_per_s"]) if g.get("reads_per_s") and c.get("reads_per_s") else None,
                          "gpu_device_inflate_reads_per_s": (e.get("gpu_device_inflate") or {}).get("reads_per_s"),
                          "gpu_host_pipeline_reads_per_s": (e.get("gpu_host_pipeline") or {}).get("reads_per_s"),
                          "gpu": g, "gpu_device_inflate": e.get("gpu_device_inflate"), "gpu_host_pipeline": e.get("gpu_host_pipeline"), "cpu": c}
            for pth in (bam_path, fa_path):
                if os.path.exists(pth):
                    os.remove(pth)
        else:
            out["e2e"] = None
        if do_cpu:
            cb = cpu_baseline(genome, cfg, [b for b in full if b is not None])
            cb["gpu_over_cpu"] = out["value"] / cb["value"]
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
