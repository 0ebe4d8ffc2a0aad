#!/bin/bash
# round 4, GPU call 23: the device's first 80 ms of a file run (kernels and copies), with the library's own call trace beside it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/tools/e2e_quick.py 3000000 default= > /dev/null 2>&1
rm -rf /tmp/tls
FADE_FAST_EXIT=0 FADEHIP_BAM_TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tls -o s -- $R/fade_amd/fade annotate --timing -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o.bam 2> $R/gpurun_out/tls.err
grep "trace\] front 0\|timing" $R/gpurun_out/tls.err | cut -c1-200
python $R/tools/r04/stream_timeline.py /tmp/tls 0.0 100 > $R/gpurun_out/tls_start.txt 2>&1
sed -n 38,260p $R/gpurun_out/tls_start.txt | cut -c1-130
