#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python $R/tools/r04/ab_runs.py 10000000 7 two= one=FADEHIP_BAM_BACK_STREAMS=1 noprep=FADE_BAM_PREPARE=0 | tee $R/gpurun_out/ab_back_streams.txt
timeout -k 10 500 python $R/tools/r04/ab_runs.py 30000000 4 two= one=FADEHIP_BAM_BACK_STREAMS=1 | tee -a $R/gpurun_out/ab_back_streams.txt
