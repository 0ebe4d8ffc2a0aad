#!/bin/bash
# Collects the rocprof evidence for bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats           (per-kernel time, no counters)
#   2. --pmc FETCH_SIZE   / 3. --pmc WRITE_SIZE   (separate passes; TCC slots do not fit both)
#   4. --pmc SQ_* issue counters
# plus the same two TCC passes over bench/store_calib (known byte counts in the kernel's access shapes).
# Outputs land under gpurun_out/prof_$TAG; copy the summaries into profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --slots 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $R/fade_amd/csrc/bench/store_calib > $OUT/calib_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $R/fade_amd/csrc/bench/store_calib > $OUT/calib_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -40
