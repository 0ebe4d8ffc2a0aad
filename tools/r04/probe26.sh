#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default= nommap=FADE_BAM_MMAP=0 default2= nommap2=FADE_BAM_MMAP=0 > $R/gpurun_out/trace_e2e7.log 2>&1
grep -v "\[trace\]\|since process" $R/gpurun_out/trace_e2e7.log | cut -c1-330
python $R/tools/r04/trace_summary.py $R/gpurun_out/e2e_quick.json
