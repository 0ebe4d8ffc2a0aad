#!/bin/bash
# round 4, GPU call 19: the file path with the device split between the record kernels (j CUs of every XCD) and the compressor
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python $R/tools/e2e_quick.py 10000000 default= s4=FADEHIP_BAM_SPLIT=4 s6=FADEHIP_BAM_SPLIT=6 s8=FADEHIP_BAM_SPLIT=8 s12=FADEHIP_BAM_SPLIT=12 s16=FADEHIP_BAM_SPLIT=16 default2= 2>&1 | grep -v "^    \[timing\] since" | cut -c1-420
