#!/bin/bash
# round 4, GPU call 26: the whole suite on the rearranged start-up (prepare, early reader, mapped input), e2e at 10 M and 30 M
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04q.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04q.log
tail -5 gpurun_out/pytest_r04q.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default= nommap=FADE_BAM_MMAP=0 default2= nommap2=FADE_BAM_MMAP=0 > $R/gpurun_out/trace_e2e6.log 2>&1
grep -v "\[trace\]\|since process" $R/gpurun_out/trace_e2e6.log | cut -c1-330
python $R/tools/r04/trace_summary.py $R/gpurun_out/e2e_quick.json
