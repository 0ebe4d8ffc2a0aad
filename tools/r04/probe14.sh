#!/bin/bash
# round 4, GPU call 14: the whole suite on the last build; differential runs; the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04l.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04l.log
tail -6 gpurun_out/pytest_r04l.log | cut -c1-300
timeout -k 10 260 python tools/r04/g8_fuzz.py 200 > gpurun_out/r04_g8_fuzz.txt 2> gpurun_out/r04_g8_fuzz.err; echo "g8 fuzz rc=$?"; tail -1 gpurun_out/r04_g8_fuzz.txt; grep -c "eight-lane groups" gpurun_out/r04_g8_fuzz.err
timeout -k 10 330 python tools/extended_fuzz.py 150 8000 > gpurun_out/r04_extended_fuzz.txt 2>&1; echo "extended fuzz rc=$?"; tail -1 gpurun_out/r04_extended_fuzz.txt
timeout -k 10 200 python tools/rules_fuzz.py 60 > gpurun_out/r04_rules_fuzz.txt 2>&1; echo "rules fuzz rc=$?"; tail -1 gpurun_out/r04_rules_fuzz.txt
