#!/bin/bash
# round 4, GPU call 7: run-skipping hash inserts; blocking sync; the device's share of the inflating; timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bgzf.py tests/test_gpu_bam_stream.py tests/test_gpu_cli.py -m gpu -q -x > gpurun_out/pytest_r04g.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04g.log
tail -15 gpurun_out/pytest_r04g.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=$g timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04g_g$g.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04g_g$g.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04g_g$g.log | tail -2
done
timeout -k 10 400 python $R/tools/e2e_quick.py 10000000 default=FADEHIP_BAM_PROF=1: spin=FADEHIP_BLOCKING_SYNC=0: share4=FADE_BAM_DEVICE_SHARE=4: share3=FADE_BAM_DEVICE_SHARE=3: share2=FADE_BAM_DEVICE_SHARE=2: devinf=FADE_BAM_INFLATE=device: > $R/gpurun_out/e2e_quick_r04g.log 2>&1
cat $R/gpurun_out/e2e_quick_r04g.log | cut -c1-700
FADE_FAST_EXIT=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -o t -- $R/fade_amd/fade annotate --timing -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o.bam 2> $R/gpurun_out/tl_run.err
grep "timing\] total" $R/gpurun_out/tl_run.err | cut -c1-400
python $R/tools/r04/stream_timeline.py /tmp/prof_tl 0.5 5 > $R/gpurun_out/stream_timeline_r04g.txt 2>&1
head -90 $R/gpurun_out/stream_timeline_r04g.txt
