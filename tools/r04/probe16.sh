#!/bin/bash
# round 4, GPU call 16: code lengths by a whole wave
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bgzf.py tests/test_gpu_bam_stream.py tests/test_gpu_cli.py -m gpu -q -x > gpurun_out/pytest_r04m.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04m.log
tail -5 gpurun_out/pytest_r04m.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=32 timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04m_g32.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04m_g32.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04m_g32.log | tail -2
timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default= > $R/gpurun_out/e2e_quick_r04m.log 2>&1
cat $R/gpurun_out/e2e_quick_r04m.log | cut -c1-600
