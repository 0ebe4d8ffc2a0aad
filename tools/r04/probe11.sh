#!/bin/bash
# round 4, GPU call 11: the file path's evidence (profiles/collect_stream.sh r04) and the stream fuzz against both checkers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 330 python tools/stream_fuzz.py 300 4000 > gpurun_out/r04_stream_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r04_stream_fuzz.txt
bash profiles/collect_stream.sh r04 > gpurun_out/collect_stream_r04.log 2>&1; echo "collect_stream rc=$?"; tail -12 gpurun_out/collect_stream_r04.log | cut -c1-300
