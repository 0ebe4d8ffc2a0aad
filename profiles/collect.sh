#!/bin/bash
# Collects the rocprof evidence for bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats on the DEFAULT bench command of the config (two slots in flight, streamed): the per-kernel
#      average durations bench.py's roofline.kernel_ms must agree with
#   2. --kernel-trace --stats with one slot (every kernel alone on the device)
#   3. --pmc FETCH_SIZE   / 4. --pmc WRITE_SIZE   (separate passes; TCC slots do not fit both), one slot
#   5. --pmc SQ_* issue counters, one slot
# plus the same two TCC passes over bench/store_calib (known byte counts in the kernel's access shapes).
# Outputs land under gpurun_out/prof_$TAG; profiles/summarize_pmc.py $TAG $CONFIG copies the summaries into profiles/.
# usage: profiles/collect.sh r02_C2 C2
TAG=${1:-r04_C2}
CFG=${2:-C2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH2="python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu --no-e2e"
BENCH1="python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu --no-e2e --slots 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- $BENCH2 > $OUT/stats2.log 2>&1 || exit 1
grep '^{' $OUT/stats2.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH1 > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH1 > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH1 > $OUT/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH1 > $OUT/sq.log 2>&1 || exit 1
if [ ! -d $R/gpurun_out/prof_calib ]; then
  mkdir -p $R/gpurun_out/prof_calib
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_calib/calib_fetch -- $R/fade_amd/csrc/bench/store_calib > $R/gpurun_out/prof_calib/calib_fetch.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_calib/calib_write -- $R/fade_amd/csrc/bench/store_calib > $R/gpurun_out/prof_calib/calib_write.log 2>&1 || exit 1
fi
find $OUT -name "*.csv" | head -40
