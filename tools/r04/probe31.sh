#!/bin/bash
# round 4: compressor experiments: codec tests, clocks per phase, rate without the profiling clocks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bgzf.py -m gpu -q -x > gpurun_out/pytest_r04s.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/pytest_r04s.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=32 timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04s_g32.log 2>&1
grep "fadehip bgzf\] [0-9ABCD]" $R/gpurun_out/bgzf_rate_r04s_g32.log | tail -5 | cut -c1-400
FADEHIP_BGZF_GEOM=32 timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04s_g32b.log 2>&1
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04s_g32b.log | tail -2
cp $R/gpurun_out/bgzf_rate.json $R/gpurun_out/r04_bgzf_rate_g32.json
