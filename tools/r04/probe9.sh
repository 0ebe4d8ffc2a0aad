#!/bin/bash
# round 4, GPU call 9: the compressor with six waves per workgroup (room for the front half's kernels beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bgzf.py tests/test_gpu_bam_stream.py -m gpu -q -x > gpurun_out/pytest_r04i.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04i.log
tail -5 gpurun_out/pytest_r04i.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=32 timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04i_g32.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04i_g32.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04i_g32.log | tail -2
timeout -k 10 400 python $R/tools/e2e_quick.py 30000000 default=FADEHIP_BAM_PROF=1: > $R/gpurun_out/e2e_quick_r04i30.log 2>&1
cat $R/gpurun_out/e2e_quick_r04i30.log | cut -c1-700
timeout -k 10 400 python $R/tools/e2e_quick.py 10000000 default=FADEHIP_BAM_PROF=1: > $R/gpurun_out/e2e_quick_r04i.log 2>&1
cat $R/gpurun_out/e2e_quick_r04i.log | cut -c1-700
FADE_FAST_EXIT=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -o t -- $R/fade_amd/fade annotate --timing -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o.bam 2> $R/gpurun_out/tl_run.err
grep "timing\] total" $R/gpurun_out/tl_run.err | cut -c1-400
python $R/tools/r04/stream_timeline.py /tmp/prof_tl 0.5 4 > $R/gpurun_out/stream_timeline_r04i.txt 2>&1
head -80 $R/gpurun_out/stream_timeline_r04i.txt
