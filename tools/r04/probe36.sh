#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python bench.py --no-e2e --steps 10 --warmup 2 > gpurun_out/bench_r04_e.json 2> gpurun_out/bench_r04_e.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r04_e.json") if l.startswith("{")][0])
print("value %.4g  records %s" % (r["value"], r.get("value_from_records")))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/tools/e2e_quick.py 10000000 default= > /dev/null 2>&1
rm -rf /tmp/kst; FADE_FAST_EXIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o s -- $R/fade_amd/fade annotate -t 16 -w 100 -b /tmp/e2eq.bam /tmp/e2eq.fa > /tmp/o.bam 2> /dev/null
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "frame" in r["Name"]:
        print("%-40s calls %s avg %.1f us min %.1f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
