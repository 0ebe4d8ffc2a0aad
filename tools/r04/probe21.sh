#!/bin/bash
# round 4, GPU call 22: where the first calls of the file path spend their time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 FADEHIP_BAM_TRACE=1 timeout -k 10 600 python $R/tools/e2e_quick.py 10000000 default= > $R/gpurun_out/trace_e2e3.log 2>&1
grep "fadehip trace" $R/gpurun_out/trace_e2e3.log
python $R/tools/r04/trace_summary.py $R/gpurun_out/e2e_quick.json
