#!/bin/bash
# round 4, GPU call 17: the whole suite on the last build; the compressor's rates without the profiling clocks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04n.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04n.log
tail -6 gpurun_out/pytest_r04n.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
FADEHIP_BGZF_GEOM=$g timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04n_g$g.log 2>&1
cp $R/gpurun_out/bgzf_rate.json $R/gpurun_out/r04_bgzf_rate_g$g.json
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04n_g$g.log | tail -2
done
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=32 timeout -k 10 300 python $R/tools/bgzf_rate.py 256 2>&1 | grep "fadehip bgzf\] [0-9]" | tail -1 | cut -c1-400 | tee $R/gpurun_out/r04_bgzf_phases_g32.txt
