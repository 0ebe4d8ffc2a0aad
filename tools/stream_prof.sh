#!/bin/bash
# GPU box: the device file path under its own clocks and under rocprofv3 (kernel trace).  Needs /tmp/e2eq.bam from tools/e2e_quick.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FADEHIP_BAM_PROF=1 FADE_BAM_CHUNK_MB=${1:-128} $R/fade_amd/fade annotate --timing -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa 2> $R/gpurun_out/stream_prof.err > /tmp/o.bam
grep "timing\|fadehip bam" $R/gpurun_out/stream_prof.err | cut -c1-500
FADE_FAST_EXIT=0 FADE_BAM_CHUNK_MB=${1:-128} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stream -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o2.bam 2> $R/gpurun_out/stream_rocprof.err
tail -3 $R/gpurun_out/stream_rocprof.err
f=$(find /tmp/prof_stream -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/stream_kernel_stats.csv && cut -c1-160 "$f" | head -30
