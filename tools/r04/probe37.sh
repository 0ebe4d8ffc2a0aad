#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python $R/tools/r04/ab_runs.py 30000000 3 all= s12=FADE_BAM_DEVICE_SHARE=12 s8=FADE_BAM_DEVICE_SHARE=8 c28=FADEHIP_BGZF_CUS=28 | tee $R/gpurun_out/ab_share2.txt
