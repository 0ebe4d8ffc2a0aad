#!/bin/bash
# round 4, GPU call 13: the compressor's bit stream staged in LDS
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bgzf.py tests/test_gpu_bam_stream.py -m gpu -q -x > gpurun_out/pytest_r04k.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04k.log
tail -5 gpurun_out/pytest_r04k.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=32 timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04k_g32.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04k_g32.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04k_g32.log | tail -2
for c in FETCH_SIZE WRITE_SIZE; do
FADEHIP_BGZF_GEOM=32 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/bgzf_rate.py 128 > /tmp/pmc_$c.log 2>&1
f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee -a $R/gpurun_out/bgzf_pmc_r04k.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "bgzf" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k in acc:
    print(k[:48], {c: (v, n[(k, c)], "KiB per launch %.0f" % (v / n[(k, c)])) for c, v in acc[k].items()})
PY
done
timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default= > $R/gpurun_out/e2e_quick_r04k.log 2>&1
cat $R/gpurun_out/e2e_quick_r04k.log | cut -c1-600
