#!/bin/bash
# GPU box: evidence for the file path on the device.  Writes gpurun_out/r03_stream_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# 1. end to end, the forms side by side (makes /tmp/e2eq.bam, 10 M reads)
python3 $R/tools/e2e_quick.py 10000000 default= device_inflate=FADE_BAM_INFLATE=device host_pipeline=FADE_BAM_DEVICE=0 cpu= > $O/r03_e2e_stream.log 2>&1 || exit 1
cp $R/gpurun_out/e2e_quick.json $O/r03_e2e_stream.json
echo e2e done
# 2. kernel traces: device inflate (every kernel of the path), host inflate (the default)
for mode in device host; do
  FADE_FAST_EXIT=0 FADE_BAM_INFLATE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o_$mode.bam 2> $O/r03_stream_rocprof_$mode.err || exit 1
  cp $(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1) $O/r03_stream_kernel_stats_${mode}_inflate.csv
done
# 3. counters of the codec kernels (separate pass, no other trace domains): instruction mix and HBM bytes
FADE_FAST_EXIT=0 FADE_BAM_INFLATE=device rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_pmc -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o_pmc.bam 2> $O/r03_stream_pmc.err || exit 1
cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) $O/r03_stream_pmc_sq.csv
# (a FETCH_SIZE / WRITE_SIZE pass over this command aborted inside rocprofv3 and sat there: not collected)
echo counters done
# 4. steady state: 30 M reads
python3 $R/tools/e2e_quick.py 30000000 default= cpu= > $O/r03_e2e_stream_30M.log 2>&1 || exit 1
cp $R/gpurun_out/e2e_quick.json $O/r03_e2e_stream_30M.json
grep -v "^    \[timing\] pool\|BAM reader" $O/r03_e2e_stream.log $O/r03_e2e_stream_30M.log | cut -c1-300
