"""Which library call creates the large host mappings a finished run still holds (GPU box): counts the anonymous
mappings >= 64 MB in /proc/self/smaps after each phase of one batch through the ABI."""
import gc
import os
import sys

LEAN = os.environ.get("PROBE_LEAN") == "1"  # what `fade annotate` does: one slot, no CU-masked stream
if LEAN:
    os.environ["FADEHIP_TAIL_CUS"] = "0"
SLOTS = (0,) if LEAN else (0, 1)

sys.path.insert(0, ".")
import fade_amd
from fade_amd import synth


def big(tag):
    rows, cur = [], None
    for line in open("/proc/self/smaps"):
        f = line.split()
        if "-" in f[0] and ":" not in f[0]:
            cur = [0, 0, line.strip()]
            rows.append(cur)
        elif f[0] == "Size:":
            cur[0] = int(f[1])
        elif f[0] == "Rss:":
            cur[1] = int(f[1])
    rows = [r for r in rows if r[0] >= 64 << 10 and len(r[2].split()) < 6]
    print("%-28s %2d anonymous mappings >= 64 MB, %6d MB mapped, %6d MB resident: %s" % (
        tag, len(rows), sum(r[0] for r in rows) >> 10, sum(r[1] for r in rows) >> 10, sorted(r[0] >> 10 for r in rows)), flush=True)


cfg = synth.config("C2")
g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
b = synth.make_reads(g, 1_000_000, 100, **cfg)
contigs = g.ascii_contigs()
big("inputs made")
ctx = fade_amd.Context(device=0)
big("context created")
ctx.genome_upload(g.names, contigs)
big("genome uploaded")
sub, _ = ctx.clipped_only(b)
pb = ctx.pinned_batch(sub)
big("pinned batch")
for slot in SLOTS:
    ctx.annotate_upload(slot, pb)
    big("slot %d uploaded" % slot)
    ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
    ctx.annotate_results(slot)
    big("slot %d run + results" % slot)
for rep in range(3):
    for slot in SLOTS:
        ctx.annotate_upload(slot, pb)
        ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
    for slot in SLOTS:
        ctx.annotate_results(slot)
big("six more runs")
del pb
ctx.close() if hasattr(ctx, "close") else None
del ctx
gc.collect()
big("context destroyed")
