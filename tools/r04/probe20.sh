#!/bin/bash
# round 4, GPU call 21: fixed costs of a 10 M-read file run: one compressor stream instead of two, call sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 timeout -k 10 600 python $R/tools/e2e_quick.py 10000000 default= bs1=FADEHIP_BAM_BACK_STREAMS=1 c16=FADE_BAM_CHUNK_MB=16 bs1c16=FADEHIP_BAM_BACK_STREAMS=1,FADE_BAM_CHUNK_MB=16 default2= > $R/gpurun_out/trace_e2e2.log 2>&1
grep -v "\[trace\]" $R/gpurun_out/trace_e2e2.log | cut -c1-330
