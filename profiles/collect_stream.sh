#!/bin/bash
# GPU box: evidence for the file path on the device.  Writes gpurun_out/${TAG}_stream_*.   usage: profiles/collect_stream.sh [r04]
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# 1. end to end, the forms side by side (makes /tmp/e2eq.bam, 10 M reads)
python3 $R/tools/e2e_quick.py 10000000 default= device_inflate=FADE_BAM_INFLATE=device host_pipeline=FADE_BAM_DEVICE=0 cpu= > $O/${TAG}_e2e_stream.log 2>&1 || exit 1
cp $R/gpurun_out/e2e_quick.json $O/${TAG}_e2e_stream.json
echo e2e done
# 2. kernel traces: device inflate (every kernel of the path), host inflate (the default)
for mode in device host; do
  FADE_FAST_EXIT=0 FADE_BAM_INFLATE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o_$mode.bam 2> $O/${TAG}_stream_rocprof_$mode.err || exit 1
  cp $(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1) $O/${TAG}_stream_kernel_stats_${mode}_inflate.csv
done
# 3. counters of the codec kernels (separate pass, no other trace domains): instruction mix and HBM bytes
FADE_FAST_EXIT=0 FADE_BAM_INFLATE=device rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_pmc -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o_pmc.bam 2> $O/${TAG}_stream_pmc.err || exit 1
cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) $O/${TAG}_stream_pmc_sq.csv
echo counters done
# 4. steady state: 30 M reads
python3 $R/tools/e2e_quick.py 30000000 default= cpu= > $O/${TAG}_e2e_stream_30M.log 2>&1 || exit 1
cp $R/gpurun_out/e2e_quick.json $O/${TAG}_e2e_stream_30M.json
grep -v "^    \[timing\] pool\|BAM reader" $O/${TAG}_e2e_stream.log $O/${TAG}_e2e_stream_30M.log | cut -c1-300
# 5. HBM bytes of the path's kernels (round 3: this pass aborted inside rocprofv3; bounded here, and last, so that a repeat costs nothing else)
for c in FETCH_SIZE WRITE_SIZE; do
  FADE_FAST_EXIT=0 timeout -k 10 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o_$c.bam 2> $O/${TAG}_stream_pmc_$c.err || { echo "$c pass failed or timed out"; exit 1; }
  cp $(find /tmp/prof_$c -name "*counter_collection.csv" | head -1) $O/${TAG}_stream_pmc_$c.csv
done
echo hbm counters done
