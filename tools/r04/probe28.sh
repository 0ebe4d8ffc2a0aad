#!/bin/bash
# round 4, GPU call 31: the default bench line (with the cpu_zlib leg), bench / stream / cli tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
S=$SECONDS
timeout -k 10 900 python bench.py > gpurun_out/bench_r04_b.json 2> gpurun_out/bench_r04_b.err; echo "bench rc=$? in $((SECONDS-S)) s"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r04_b.json") if l.startswith("{")][0])
e = r["e2e"]
print("value %.4g  ms/step %.3f  frac %.5f  kernel_ms %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms"]))
print("records %.4g  cpu_rec %s" % (r["value_from_records"], r["cpu_baseline_from_records"]["value"]))
print("e2e gpu %.3g cpu %.3g x%.2f | cpu_zlib %.3g x%.1f | devinfl %.3g hostpipe %.3g" % (e["gpu_reads_per_s"], e["cpu_reads_per_s"], e["gpu_over_cpu"], e["cpu_zlib_reads_per_s"], e["gpu_over_cpu_zlib"], e["gpu_device_inflate_reads_per_s"], e["gpu_host_pipeline_reads_per_s"]))
b = e["big"]
print("big gpu %.3g cpu %.3g x%.2f" % (b["gpu_reads_per_s"], b["cpu_reads_per_s"], b["gpu_over_cpu"]))
print("cpu_baseline", r["cpu_baseline"]["value"], r["cpu_baseline"]["gpu_over_cpu"])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_bam_stream.py tests/test_gpu_cli.py -q -x > gpurun_out/pytest_r04r.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_r04r.log | cut -c1-300
