#!/bin/bash
# Where the BGZF compressor's time goes on a BAM payload (GPU box host cores): builds the payload from synthetic reads,
# then runs the selftest's speed mode with per-stage timing.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out; : > gpurun_out/deflate_where.txt
make -C tools -s sam2bam
python3 - <<'PY'
import sys, subprocess, gzip
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from fade_amd import synth
import e2e_cli
cfg = synth.config("C2"); cfg["contig_len"] = 2_000_000
g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
b = synth.make_reads(g, 200_000, 5, **cfg)
e2e_cli.write_sam("/tmp/p.sam", b, g)
with open("/tmp/p.bam", "wb") as fo:
    subprocess.check_call(["tools/sam2bam", "/tmp/p.sam"], stdout=fo)
open("/tmp/p.payload", "wb").write(gzip.decompress(open("/tmp/p.bam", "rb").read()))
PY
g++ -O2 -std=c++17 -DFADE_DEFLATE_TIMING -o /tmp/deflate_speed fade_amd/csrc/host/selftest/deflate_selftest.cpp -lz
/tmp/deflate_speed /tmp/p.payload | tail -12 | tee -a $R/gpurun_out/deflate_where.txt
# DEFLATE_SKIPS="48,7 64,15": the same with other skip rules (after,cap) for efforts 1 and 2
for v in ${DEFLATE_SKIPS:-}; do
  (echo "== skip rule $v"; DEFLATE_SPEED_ONLY=1 DEFLATE_SKIP=$v /tmp/deflate_speed /tmp/p.payload | grep -B1 "effort [12]") | tee -a $R/gpurun_out/deflate_where.txt
done
# ... and with the layout hints the BGZF writer gives for BAM records (bases + qualities marked as free of repeats)
(echo "== with layout hints"; DEFLATE_HINTS=2 DEFLATE_SPEED_ONLY=1 /tmp/deflate_speed /tmp/p.payload | grep -B1 "effort [12]") | tee -a $R/gpurun_out/deflate_where.txt
