#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bgzf.py tests/test_gpu_inflate.py tests/test_gpu_bam_stream.py tests/test_gpu_cli.py -m gpu -q -x > gpurun_out/pytest_r04t.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/pytest_r04t.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
FADEHIP_BGZF_GEOM=$g timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04u_g$g.log 2>&1
cp $R/gpurun_out/bgzf_rate.json $R/gpurun_out/r04_bgzf_rate_g$g.json
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04u_g$g.log | tail -2
done
