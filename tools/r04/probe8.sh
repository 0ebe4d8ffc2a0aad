#!/bin/bash
# round 4, GPU call 8: whole suite; compressor rates (roles for 0xff00 blocks, segments for 0x7f00); e2e; the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04h.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04h.log
tail -8 gpurun_out/pytest_r04h.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=$g timeout -k 10 300 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_r04h_g$g.log 2>&1
grep "fadehip bgzf\] [0-9]" $R/gpurun_out/bgzf_rate_r04h_g$g.log | tail -1 | cut -c1-400
grep "GBps\|ratio" $R/gpurun_out/bgzf_rate_r04h_g$g.log | tail -2
done
timeout -k 10 400 python $R/tools/e2e_quick.py 10000000 default=FADEHIP_BAM_PROF=1: share5=FADE_BAM_DEVICE_SHARE=5: devinf=FADE_BAM_INFLATE=device: > $R/gpurun_out/e2e_quick_r04h.log 2>&1
cat $R/gpurun_out/e2e_quick_r04h.log | cut -c1-700
cd $R && timeout -k 10 600 python bench.py > gpurun_out/bench_r04h.json 2> gpurun_out/bench_r04h.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_r04h.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r04h.json') if l.startswith('{')][-1])
e=d.get('e2e') or {}
print({k:d.get(k) for k in ('value','value_resident','value_from_records','ms_per_step')})
print('roofline', {k:d['roofline'][k] for k in ('kernel','kernel_ms','kernel_ms_streamed','frac','traffic')})
print('cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline_from_records') or {}))
print('e2e', {k:e.get(k) for k in ('gpu_reads_per_s','cpu_reads_per_s','gpu_over_cpu','gpu_device_inflate_reads_per_s','gpu_host_pipeline_reads_per_s')})
print('big', {k:(e.get('big') or {}).get(k) for k in ('reads','gpu_reads_per_s','cpu_reads_per_s','gpu_over_cpu')})
PY
