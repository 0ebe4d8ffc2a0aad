#!/bin/bash
# round 4, GPU call 10: the committed profiles of the bench line (C2) and the bench lines of C2 / C3 / C5
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/collect.sh r04_C2 C2 > gpurun_out/collect_r04_C2.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/collect_r04_C2.log
timeout -k 10 500 python bench.py --steps 50 --warmup 5 > gpurun_out/r04_C2_bench.json 2> gpurun_out/r04_C2_bench.err; echo "bench C2 rc=$?"
timeout -k 10 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu --no-e2e > gpurun_out/r04_C3_bench.json 2>> gpurun_out/r04_C2_bench.err; echo "bench C3 rc=$?"
timeout -k 10 300 python bench.py --config C5 --steps 20 --warmup 3 --no-cpu --no-e2e > gpurun_out/r04_C5_bench.json 2>> gpurun_out/r04_C2_bench.err; echo "bench C5 rc=$?"
timeout -k 10 300 python bench.py --config C6 --steps 20 --warmup 3 --no-cpu --no-e2e > gpurun_out/r04_C6_bench.json 2>> gpurun_out/r04_C2_bench.err; echo "bench C6 rc=$?"
python - <<'PY'
import json
for c in ("C2","C3","C5","C6"):
    try:
        d=json.loads([l for l in open('gpurun_out/r04_%s_bench.json'%c) if l.startswith('{')][-1])
        print(c, "%.3g"%d['value'], "%.3g"%d['value_resident'], d['roofline']['kernel'][:28], "kernel_ms %.3f streamed %.3f frac %.4f"%(d['roofline']['kernel_ms'], d['roofline']['kernel_ms_streamed'], d['roofline']['frac']), "from_records", d.get('value_from_records'))
    except Exception as e:
        print(c, "failed", e)
PY
