#!/bin/bash
# round 4: the compressor on fewer CUs of every XCD (the record kernels anywhere), alone and with a share of the inflating on the device
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python $R/tools/r04/ab_runs.py 30000000 3 all= c28=FADEHIP_BGZF_CUS=28 c24=FADEHIP_BGZF_CUS=24 c28s6=FADEHIP_BGZF_CUS=28,FADE_BAM_DEVICE_SHARE=6 c24s4=FADEHIP_BGZF_CUS=24,FADE_BAM_DEVICE_SHARE=4 s6=FADE_BAM_DEVICE_SHARE=6 | tee $R/gpurun_out/ab_bgzf_cus.txt
