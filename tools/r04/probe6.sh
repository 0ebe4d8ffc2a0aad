#!/bin/bash
# round 4, GPU call 6: compressor with bigger tables + seams; G8 default; whole suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04f.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04f.log
tail -15 gpurun_out/pytest_r04f.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=$g timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04f_g$g.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04f_g$g.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04f_g$g.log | tail -2
done
timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default=FADEHIP_BAM_PROF=1: devinf=FADE_BAM_INFLATE=device: > $R/gpurun_out/e2e_quick_r04f.log 2>&1
cat $R/gpurun_out/e2e_quick_r04f.log | cut -c1-700
