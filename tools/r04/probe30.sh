#!/bin/bash
# round 4, GPU call 34: the differential runs on the last build (start-up rearranged, staging memory registered)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
make -C tools -s
timeout -k 10 420 python tools/stream_fuzz.py 330 9000 > gpurun_out/r04_stream_fuzz_b.txt 2>&1; echo "stream fuzz rc=$?"; tail -1 gpurun_out/r04_stream_fuzz_b.txt
timeout -k 10 260 python tools/r04/g8_fuzz.py 180 > gpurun_out/r04_g8_fuzz_b.txt 2>&1; echo "g8 fuzz rc=$?"; tail -1 gpurun_out/r04_g8_fuzz_b.txt
timeout -k 10 260 python tools/extended_fuzz.py 150 20000 > gpurun_out/r04_extended_fuzz_b.txt 2>&1; echo "extended fuzz rc=$?"; tail -1 gpurun_out/r04_extended_fuzz_b.txt
timeout -k 10 160 python tools/rules_fuzz.py 60 > gpurun_out/r04_rules_fuzz_b.txt 2>&1; echo "rules fuzz rc=$?"; tail -1 gpurun_out/r04_rules_fuzz_b.txt
