#!/bin/bash
# round 4, GPU call 24: the file path's start-up rearranged (device brought up beside the FASTA and the first reads; prepare)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bam_stream.py tests/test_gpu_cli.py -q -x > gpurun_out/pytest_r04p.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_r04p.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 FADEHIP_BAM_TRACE=1 timeout -k 10 600 python $R/tools/e2e_quick.py 10000000 default= noprep=FADE_BAM_PREPARE=0 hostmalloc=FADEHIP_PIN_HOSTMALLOC=1 default2= > $R/gpurun_out/trace_e2e4.log 2>&1
grep -v "\[trace\]\|fadehip trace\] [fb]" $R/gpurun_out/trace_e2e4.log | cut -c1-330
python $R/tools/r04/trace_summary.py $R/gpurun_out/e2e_quick.json
