#!/bin/bash
# round 4, GPU call 3: per-set mutexes in the front half; score-pass variants A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_bam_stream.py tests/test_gpu_cli.py tests/test_gpu_sw.py -m gpu -q -x > gpurun_out/pytest_r04c.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04c.log
tail -4 gpurun_out/pytest_r04c.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
python $R/tools/e2e_quick.py 10000000 default=FADEHIP_BAM_PROF=1: onestream=FADEHIP_BAM_BACK_STREAMS=1: devinf=FADE_BAM_INFLATE=device: chunk16=FADE_BAM_CHUNK_MB=16: > $R/gpurun_out/e2e_quick_r04c.log 2>&1
cat $R/gpurun_out/e2e_quick_r04c.log | cut -c1-700
python $R/tools/r04/score_ab.py C2 1000000 2>&1 | cut -c1-400
