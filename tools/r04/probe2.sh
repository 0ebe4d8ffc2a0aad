#!/bin/bash
# round 4, GPU call 2: the two-call front half, the back half with look-ahead, side-by-side pwrites
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04b.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04b.log
tail -30 gpurun_out/pytest_r04b.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
python $R/tools/e2e_quick.py 10000000 default= prof=FADEHIP_BAM_PROF=1: onestream=FADEHIP_BAM_BACK_STREAMS=1: seqwrite=FADE_BAM_WRITERS=0: devinf=FADE_BAM_INFLATE=device: chunk64=FADE_BAM_CHUNK_MB=64: chunk16=FADE_BAM_CHUNK_MB=16: > $R/gpurun_out/e2e_quick_r04b.log 2>&1
cat $R/gpurun_out/e2e_quick_r04b.log | cut -c1-700
