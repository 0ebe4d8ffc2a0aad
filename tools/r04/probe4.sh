#!/bin/bash
# round 4, GPU call 4: timeline of the file path
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/tools/e2e_quick.py 10000000 default= > $R/gpurun_out/e2e_quick_r04d.log 2>&1
grep -v "^    \[timing\] since" $R/gpurun_out/e2e_quick_r04d.log | cut -c1-500
FADE_FAST_EXIT=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -o t -- $R/fade_amd/fade annotate --timing -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o.bam 2> $R/gpurun_out/tl_run.err
grep "timing\] total" $R/gpurun_out/tl_run.err | cut -c1-400
python $R/tools/r04/stream_timeline.py /tmp/prof_tl 0.5 7 > $R/gpurun_out/stream_timeline_r04d.txt 2>&1
head -130 $R/gpurun_out/stream_timeline_r04d.txt
