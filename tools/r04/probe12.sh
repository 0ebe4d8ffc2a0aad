#!/bin/bash
# round 4, GPU call 12: eight-lane kernels for more read lengths; whole suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04j.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04j.log
tail -12 gpurun_out/pytest_r04j.log | cut -c1-400
PROBE_READ_LEN=101 PROBE_SLOTS=2 PROBE_MODES=streamed timeout -k 10 200 python tools/stream_probe.py C2 1000000 4 2>&1 | grep -v "^$" | tail -12
PROBE_READ_LEN=101 FADEHIP_SCORE_G8=0 PROBE_SLOTS=2 PROBE_MODES=streamed timeout -k 10 200 python tools/stream_probe.py C2 1000000 4 2>&1 | grep -v "^$" | tail -12
