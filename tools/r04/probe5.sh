#!/bin/bash
# round 4, GPU call 5: the compressor with phase A on per-wave segments; G8 at 2 / 3 waves
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bgzf.py tests/test_gpu_bam_stream.py -m gpu -q -x > gpurun_out/pytest_r04e.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04e.log
tail -15 gpurun_out/pytest_r04e.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=$g timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04e_g$g.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04e_g$g.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04e_g$g.log | tail -4
done
timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default=FADEHIP_BAM_PROF=1: > $R/gpurun_out/e2e_quick_r04e.log 2>&1
cat $R/gpurun_out/e2e_quick_r04e.log | cut -c1-700
timeout -k 10 400 python $R/tools/r04/score_ab.py C2 1000000 2>&1 | cut -c1-400
