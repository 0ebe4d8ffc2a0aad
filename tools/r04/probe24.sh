#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 FADEHIP_BAM_TRACE=1 FADEHIP_BAM_TRACE_FINE=1 timeout -k 10 600 python $R/tools/e2e_quick.py 10000000 default= > $R/gpurun_out/trace_e2e5.log 2>&1
grep "fadehip trace\] front 0\|prepare" $R/gpurun_out/trace_e2e5.log | cut -c1-200
python $R/tools/r04/trace_summary.py $R/gpurun_out/e2e_quick.json
