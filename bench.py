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
ROUND = "r04"


def pmc_summary(config):
    """The committed rocprofv3 --pmc passes of this build and config (profiles/collect.sh + summarize_pmc.py), or None.
    Replayed, not measured in this run: the fields that come from it carry `source`."""
    for rnd in (ROUND, "r03", "r02"):
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


def e2e_run(exe, bam, fa, n_reads, cfg, threads, out, reps, env=None):
    """One leg: `exe annotate -b bam fa > out`, wall time of the whole process, best of `reps` runs."""
    best = None
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
        r = dict(seconds=dt, reads_per_s=n_reads / dt, out_bytes=os.path.getsize(out), best_of=reps,
                 timing=[l for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("[timing]")][:8])
        if best is None or r["seconds"] < best["seconds"]:
            best = r
    if os.path.exists(out):
        os.remove(out)
    return best


def e2e_legs(bam, fa, n_reads, cfg, threads, tmp, which=("gpu", "gpu_device_inflate", "gpu_host_pipeline", "cpu", "cpu_zlib")):
    """The whole program on a BAM file, GPU driver and CPU comparator on the same threads: wall time of each process
    (start-up, FASTA load and genome upload, BGZF inflate, annotate, tags, BGZF deflate, exit), best of 3 runs for EVERY leg
    (the boxes differ by +-8 % from minute to minute) but cpu_zlib (one run)."""
    fade = os.path.join(ROOT, "fade_amd", "fade")
    cpu = os.path.join(ROOT, "tools", "cpu_annotate")
    legs = {
        # default: the file path on the device (framing, annotate, tags, deflate as kernels), BGZF inflate on the host pool
        "gpu": (fade, None),
        "gpu_device_inflate": (fade, {"FADE_BAM_INFLATE": "device"}),  # ... inflate on the device too
        "gpu_host_pipeline": (fade, {"FADE_BAM_DEVICE": "0"}),        # round 2's pipeline (+ device deflate)
        "cpu": (cpu, None),
        # the comparator with the codec reference FADE links (htslib's defaults: zlib's inflate, zlib level 6): one run
        "cpu_zlib": (cpu, {"FADE_BGZF_CODEC": "zlib"}),
    }
    return {k: e2e_run(legs[k][0], bam, fa, n_reads, cfg, threads, os.path.join(tmp, "bench_e2e.%s.bam" % k), 1 if k == "cpu_zlib" else 3, legs[k][1]) for k in which}


def bgzf_chunks(path, chunk_payload):
    """The BGZF members of a file grouped into runs whose payloads add up to at most chunk_payload bytes: [(bytes, isize sum)]."""
    raw = np.fromfile(path, dtype=np.uint8)
    out, at, lo, acc = [], 0, 0, 0
    n = len(raw)
    while at + 28 <= n:
        bsize = int(raw[at + 16]) + (int(raw[at + 17]) << 8) + 1
        isz = int(raw[at + bsize - 4]) | (int(raw[at + bsize - 3]) << 8) | (int(raw[at + bsize - 2]) << 16) | (int(raw[at + bsize - 1]) << 24)
        if acc and acc + isz > chunk_payload:
            out.append((raw[lo:at], acc))
            lo, acc = at, 0
        acc += isz
        at += bsize
    if at > lo:
        out.append((raw[lo:at], acc))
    return out


def records_path_rate(ctx, bam, names, cfg, n_reads, barrier, chunk_mb=32, passes=3):
    """value_from_records: the path from BAM RECORD BYTES.  The workload's BAM payload (what htslib's inflate hands to
    bam_read1 — record after record, block_size first), already inflated and in pinned host memory, goes through
    fadehip_bam_front_raw: H2D, record framing, the packing of the batch arrays, the gate (anno.d:61-65 decided on the
    device for every record — no host code looks at a record), score pass, pass 2, tags, the rewritten records left on the
    device (FADEHIP_BAM_NO_OUTPUT: no codec).  front and back on a thread each, as the `fade` driver runs them."""
    import ctypes as C
    import threading
    L = ctx._L
    chunks = []
    hdr = None
    for members, isz in bgzf_chunks(bam, chunk_mb << 20):
        pay = ctx.bgzf_inflate(members, out_cap=isz + 64)  # (setup: the device's inflater, checked against ISIZE / CRC32)
        assert len(pay) == isz
        if hdr is None:
            l_text = int(np.frombuffer(pay[4:8].tobytes(), "<i4")[0])
            at = 8 + l_text
            n_ref = int(np.frombuffer(pay[at:at + 4].tobytes(), "<i4")[0])
            at += 4
            for _ in range(n_ref):
                at += 4 + int(np.frombuffer(pay[at:at + 4].tobytes(), "<i4")[0]) + 4
            hdr = at
        ptr = C.c_void_p()
        ctx._chk(L.fadehip_host_alloc(ctx._h, max(len(pay), 64), C.byref(ptr)))
        C.memmove(ptr, pay.ctypes.data, len(pay))
        chunks.append((ptr, len(pay)))
    payload_bytes = sum(n for _, n in chunks)
    times, totals = [], None
    for it in range(passes + 1):
        st = ctx.bam_stream(names, floor_len=cfg["floor_len"], window=cfg["window"], first_record=hdr, no_output=True)
        err = []
        go = threading.Semaphore(0)

        def backs():
            try:
                for _ in chunks:
                    go.acquire()
                    st.back()
            except Exception as exc:  # noqa: BLE001
                err.append(exc)

        th = threading.Thread(target=backs)
        barrier()
        t0 = time.perf_counter()
        th.start()
        for j, (ptr, n) in enumerate(chunks):
            st.front_raw_ptr(ptr, n, last=(j == len(chunks) - 1))
            go.release()
        th.join()
        barrier()
        dt = time.perf_counter() - t0
        if err:
            raise err[0]
        totals = st.totals()
        st.close()
        if it:  # (the first pass sizes the stream's buffers)
            times.append(dt)
    for ptr, _ in chunks:
        L.fadehip_host_free(ctx._h, ptr)
    assert totals[1] == n_reads, (totals, n_reads)
    return dict(seconds=times, payload_bytes=payload_bytes, calls=len(chunks), stats=totals[0], records=totals[1])


def cpu_from_records(sample_bam, fa, cfg, threads, reps=3):
    """cpu_baseline on value_from_records' definition: tools/cpu_annotate --records-only — the sample is read, inflated and
    framed outside the clock; inside it every record goes from its BAM bytes (a bam1_t as htslib hands it over) through
    annotateTask of the oracle (gate, reverse complement, FASTA window, one striped SW call per qualifying clip, gates, tag
    strings) on all threads; nothing is compressed or written.  Median of `reps`."""
    exe = os.path.join(ROOT, "tools", "cpu_annotate")
    rs = []
    for _ in range(reps):
        p = subprocess.run([exe, "annotate", "--records-only", "-t", str(threads), "-w", str(cfg["window"]), "--min-length", str(cfg["floor_len"]), "-b",
                            sample_bam, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode != 0:
            return dict(error=p.stderr.decode(errors="replace")[-300:])
        rs.append(json.loads(p.stdout.decode().strip().splitlines()[-1]))
    rs.sort(key=lambda r: r["seconds"])
    r = rs[len(rs) // 2]
    return dict(value=r["records"] / r["seconds"], unit="reads/s", cores=threads, kind="port",
                sample="%d reads of the same workload as BAM records in memory, median of %d passes (%.2f s each); the oracle's annotateTask per record "
                       "incl. the tag strings, %d threads; reading / inflating / framing the sample is outside the clock, nothing is written" % (
                           r["records"], reps, r["seconds"], threads))


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
    ap.add_argument("--e2e-big-reads", type=int, default=30_000_000, help="reads of the second, larger end-to-end file (0: skip that leg)")
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
    # end to end: every rank has a BAM file of ITS shard of the reads ("sharded per GPU"); rank 0 alone
    # also runs the CPU comparator and the larger file
    do_e2e = not args.no_e2e and args.synth == "native"
    tmp = os.environ.get("TMPDIR", "/tmp")
    sfx = "" if world == 1 else ".%d" % rank
    bam_path, fa_path = os.path.join(tmp, "bench_e2e%s.bam" % sfx), os.path.join(tmp, "bench_e2e%s.fa" % sfx)
    big_path, sample_path = os.path.join(tmp, "bench_e2e_big.bam"), os.path.join(tmp, "bench_e2e_sample.bam")
    do_big = do_e2e and world == 1 and args.e2e_big_reads > 0
    bam_writer = big_writer = sample_writer = None
    if do_e2e:
        if rank == 0:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s"])
        genome.write_fasta(fa_path)
        bam_writer = sg.BamWriter(bam_path, genome)
        if do_big:
            big_writer = sg.BamWriter(big_path, genome)
        if do_cpu:
            sample_writer = sg.BamWriter(sample_path, genome)
    # per-GPU record range: each rank owns its own shard of the reads (stream 100 (k + 1) + rank)
    full, pinned, pinned_pre = [], [], []
    t_gen = t_pack = t_bam = 0.0
    e2e_written = big_reps = 0
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
        if big_writer is not None:  # the larger file: the workload's batches over and over under new read names
            big_reps = max(1, -(-args.e2e_big_reads // (args.batch_reads * BATCHES_PER_STEP)))
            for rep in range(big_reps):
                big_writer.write(b, (rep * args.batch_reads * BATCHES_PER_STEP + k * args.batch_reads) // 2)
        if sample_writer is not None and k < 2:  # the CPU legs' bounded sample: 2 M reads
            sample_writer.write(b, k * args.batch_reads // 2)
        t3 = time.perf_counter()
        t_gen, t_pack, t_bam = t_gen + t1 - t0, t_pack + t2 - t1, t_bam + t3 - t2
        full.append(b if (do_cpu and k < 2) else None)  # the CPU leg's bounded sample: 2 M reads
    if bam_writer is not None:
        t0 = time.perf_counter()
        for w_ in (bam_writer, big_writer, sample_writer):
            if w_ is not None:
                w_.close()
        t_bam += time.perf_counter() - t0
    big_reads = big_reps * args.batch_reads * BATCHES_PER_STEP if do_big else 0
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
        # reads of up to 152 bases in the 160-row class run on eight-lane groups x 19 rows since round 4 (FADEHIP_SCORE_G8=0: sixteen x 10)
        g8 = R == 10 and cfg["read_len"] <= 152 and os.environ.get("FADEHIP_SCORE_G8", "1") != "0"
        kernel = "sw_pk_kernel<19,1,LG=8> (score pass: 8 lanes x 19 rows per alignment pair, 16 alignments per wavefront)" if g8 else "sw_pk_kernel<%d,1> (score pass)" % R
        kpat = "sw_pk_kernel<19, 1" if g8 else "sw_pk_kernel<%d, 1" % R
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
                if kpat in name:
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
                                                           "(one slot, serial, each launch behind a host wait and an upload: %d launches after the timed region).  The "
                                                           "launch state of the three figures: kernel_ms = cold-ish solo launches (clock ramp and the CU-masked stream's "
                                                           "hand-over included; the noisiest), the rocprofv3 1-slot trace of profiles/ = the same launches back to back "
                                                           "(4-5 %% shorter), kernel_ms_streamed = inside the timed two-slot loop (tails of two passes overlap)" % len(solo),
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
    # ---- the path from BAM record bytes (no host code looks at a record): every rank on its shard's file
    rec = None
    if do_e2e:
        try:
            rec = records_path_rate(ctx, bam_path, genome.names, cfg, e2e_written, barrier)
        except Exception as exc:  # (never at the expense of the metric's line)
            rec = {"error": repr(exc)[:300]}
        tr = torch.tensor([max(rec["seconds"]) if "seconds" in rec else -1.0, min(rec["seconds"]) if "seconds" in rec else -1.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        if rank == 0 and "seconds" in rec and float(tr[0]) > 0:
            out["value_from_records"] = e2e_written * world / float(tr[1] if world == 1 else tr[0])
            out["value_from_records_is"] = (
                "starts from BAM record bytes: the %d-read workload's inflated BAM payload (%d bytes per rank: what htslib's inflate hands to bam_read1), "
                "pinned, in %d calls of <= 32 MB -> fadehip_bam_front_raw (H2D, record framing, packing of the batch arrays, the gate of anno.d:61-65 for "
                "every record, score pass, pass 2, tag sizes) -> fadehip_bam_back under FADEHIP_BAM_NO_OUTPUT (the records rewritten with their tags, "
                "left on the device; no codec); front and back on a thread each; nothing about a record is decided on the host (tools/synthgen's "
                "sg_compact, which packs `value`'s blocks, is not involved); best of %d passes%s; all passes: %s s" % (
                    e2e_written, rec["payload_bytes"], rec["calls"], len(rec["seconds"]), "" if world == 1 else " (slowest rank)",
                    ", ".join("%.3f" % x for x in rec["seconds"])))
            assert rec["stats"][0] == e2e_written
        elif rank == 0:
            out["value_from_records"] = None
            out["value_from_records_is"] = rec.get("error")
    ctx.close()
    # ---- end to end with N ranks: every rank runs `fade annotate -b` on ITS shard's file on ITS device, a share of the host
    # threads each; first rank 0 alone (the others wait), then all together: what the node gives N processes at once
    e2e_ranks = None
    if do_e2e and world > 1:
        fade = os.path.join(ROOT, "fade_amd", "fade")
        thr = max(1, usable_cpus() // world)
        env = {"FADE_DEVICE_MAP": str(local)}
        alone = None
        barrier()
        if rank == 0:
            alone = e2e_run(fade, bam_path, fa_path, e2e_written, cfg, thr, os.path.join(tmp, "bench_e2e%s.out.bam" % sfx), 2, env)
        barrier()
        mine = e2e_run(fade, bam_path, fa_path, e2e_written, cfg, thr, os.path.join(tmp, "bench_e2e%s.out.bam" % sfx), 1, env)
        barrier()
        secs = torch.tensor([mine.get("seconds", -1.0)], dtype=torch.float64, device=dev)
        allsecs = [torch.zeros_like(secs) for _ in range(world)]
        dist.all_gather(allsecs, secs)
        per = [float(x[0]) for x in allsecs]
        if rank == 0:
            ok = all(x > 0 for x in per) and alone and alone.get("seconds")
            agg = e2e_written * world / max(per) if ok else None
            e2e_ranks = {"what": "`fade annotate -b` on every rank's own %d-read shard file (BASELINE config 4: sharded per GPU; what `fade annotate --gpus N "
                                 "--out-shards` runs per lane), device = the rank's, %d host threads each; `alone`: rank 0 with the node to itself; `together`: all "
                                 "%d ranks at once, aggregate = all reads / the slowest rank's wall time" % (e2e_written, thr, world),
                         "alone_reads_per_s": alone.get("reads_per_s") if alone else None, "per_rank_seconds": per, "aggregate_reads_per_s": agg,
                         "e2e_weak_scaling": (agg / (world * alone["reads_per_s"])) if ok else None, "alone": alone, "rank0_together": mine}
        for pth in (bam_path, fa_path):
            if os.path.exists(pth):
                os.remove(pth)
    if rank == 0:
        if do_e2e and world > 1:
            out["e2e"] = e2e_ranks
        elif do_e2e:
            try:
                e = e2e_legs(bam_path, fa_path, e2e_written, cfg, usable_cpus(), tmp)
                if do_big:
                    e["big"] = e2e_legs(big_path, fa_path, big_reads, cfg, usable_cpus(), tmp, which=("gpu", "cpu"))
            except Exception as exc:  # (a missing binary, a full disk: never at the expense of the metric's line)
                e = {"gpu": {"error": repr(exc)[:300]}}
            g, c = e.get("gpu") or {}, e.get("cpu") or {}
            out["e2e"] = {"what": "`fade annotate -b` BAM -> BAM on a %d-read file of this workload, wall time of the whole process, %d host threads.  gpu = the "
                                  "default: the file path on the device (record framing, annotate, tags, BGZF deflate as kernels; BGZF inflate on the host pool); "
                                  "gpu_device_inflate = FADE_BAM_INFLATE=device (inflate as a kernel too: only compressed bytes cross PCIe); gpu_host_pipeline = "
                                  "FADE_BAM_DEVICE=0 (records handled by the host pool, level-2 batches to the device, deflate on the device); "
                                  "cpu = tools/cpu_annotate: the same reader / writer / codec around the CPU oracle — this build's own fast inflate / "
                                  "deflate / CRC (several times zlib's speed), so a STRONGER comparator than reference FADE's htslib + zlib -6 would be.  "
                                  "cpu_zlib = the same comparator with FADE_BGZF_CODEC=zlib (zlib's inflate, zlib level-6 deflate: htslib's defaults, "
                                  "what the reference program links), one run.  Every other leg is the best of 3 runs" % (e2e_written, usable_cpus()),
                          "gpu_reads_per_s": g.get("reads_per_s"), "cpu_reads_per_s": c.get("reads_per_s"),
                          "gpu_over_cpu": (g["reads_per_s"] / c["reads_per_s"]) if g.get("reads_per_s") and c.get("reads_per_s") else None,
                          "gpu_device_inflate_reads_per_s": (e.get("gpu_device_inflate") or {}).get("reads_per_s"),
                          "gpu_host_pipeline_reads_per_s": (e.get("gpu_host_pipeline") or {}).get("reads_per_s"),
                          "cpu_zlib_reads_per_s": (e.get("cpu_zlib") or {}).get("reads_per_s"),
                          "gpu_over_cpu_zlib": (g["reads_per_s"] / e["cpu_zlib"]["reads_per_s"]) if g.get("reads_per_s") and (e.get("cpu_zlib") or {}).get("reads_per_s") else None,
                          "gpu": g, "gpu_device_inflate": e.get("gpu_device_inflate"), "gpu_host_pipeline": e.get("gpu_host_pipeline"), "cpu": c,
                          "cpu_zlib": e.get("cpu_zlib")}
            if e.get("big"):
                bg, bc = e["big"].get("gpu") or {}, e["big"].get("cpu") or {}
                out["e2e"]["big"] = {"reads": big_reads, "what": "the same two legs (gpu default, cpu) on a %d-read file: the workload's batches %d times over under new read names"
                                                                  % (big_reads, big_reps),
                                     "gpu_reads_per_s": bg.get("reads_per_s"), "cpu_reads_per_s": bc.get("reads_per_s"),
                                     "gpu_over_cpu": (bg["reads_per_s"] / bc["reads_per_s"]) if bg.get("reads_per_s") and bc.get("reads_per_s") else None,
                                     "gpu": bg, "cpu": bc}
            for pth in (bam_path, fa_path, big_path):
                if os.path.exists(pth):
                    os.remove(pth)
        else:
            out["e2e"] = None
        if do_cpu:
            cb = cpu_baseline(genome, cfg, [b for b in full if b is not None])
            cb["gpu_over_cpu"] = out["value"] / cb["value"]
            out["cpu_baseline"] = cb
            if do_e2e and os.path.exists(sample_path):
                genome.write_fasta(fa_path)
                cr = cpu_from_records(sample_path, fa_path, cfg, cb["cores"])
                if cr.get("value") and out.get("value_from_records"):
                    cr["gpu_over_cpu"] = out["value_from_records"] / cr["value"]
                out["cpu_baseline_from_records"] = cr
                for pth in (sample_path, fa_path):
                    if os.path.exists(pth):
                        os.remove(pth)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
