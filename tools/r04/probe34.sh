#!/bin/bash
# round 4: the whole suite, the default bench line, a stream fuzz on the last build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04v.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04v.log
tail -4 gpurun_out/pytest_r04v.log | cut -c1-300
timeout -k 10 300 python bench.py > gpurun_out/bench_r04_c.json 2> gpurun_out/bench_r04_c.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r04_c.json") if l.startswith("{")][0])
e = r["e2e"]
print("value %.4g  ms/step %.3f  frac %.5f  kernel_ms %.4f traffic %s" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms"], r["roofline"].get("traffic")))
print("records %.4g  cpu_rec %s" % (r["value_from_records"], r["cpu_baseline_from_records"]["value"]))
print("e2e gpu %.3g cpu %.3g x%.2f | cpu_zlib %.3g x%.1f | devinfl %.3g hostpipe %.3g" % (e["gpu_reads_per_s"], e["cpu_reads_per_s"], e["gpu_over_cpu"], e["cpu_zlib_reads_per_s"], e["gpu_over_cpu_zlib"], e["gpu_device_inflate_reads_per_s"], e["gpu_host_pipeline_reads_per_s"]))
b = e["big"]
print("big gpu %.3g cpu %.3g x%.2f" % (b["gpu_reads_per_s"], b["cpu_reads_per_s"], b["gpu_over_cpu"]))
PY
export TMPDIR=/tmp
make -C tools -s
timeout -k 10 300 python tools/stream_fuzz.py 240 15000 > gpurun_out/r04_stream_fuzz_c.txt 2>&1; echo "stream fuzz rc=$?"; tail -1 gpurun_out/r04_stream_fuzz_c.txt
