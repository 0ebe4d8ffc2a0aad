#!/bin/bash
# round 4, GPU call 15: the default bench command as the driver runs it (timed), and smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SECONDS=0; python bench.py > gpurun_out/bench_default_r04.json 2> gpurun_out/bench_default_r04.err; echo "bench rc=$? wall ${SECONDS}s"
true
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default_r04.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','steps','warmup','ms_per_step','value_from_records')})
print(d['roofline']['traffic'], d['roofline_valu'] and {k:d['roofline_valu'][k] for k in ('achieved','frac','valu_busy_frac_pmc')})
e=d['e2e']; print({k:e.get(k) for k in ('gpu_reads_per_s','cpu_reads_per_s','gpu_over_cpu')}, {k:e['big'].get(k) for k in ('gpu_reads_per_s','cpu_reads_per_s','gpu_over_cpu')})
print(d['setup_s'])
PY
python -c "import __graft_entry__ as g; g.smoke()"
