#!/bin/bash
# round 4, GPU call 1: the whole -m gpu suite on the new build; the compressor's phase clocks and LDS counters; e2e baseline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r04a.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04a.log
tail -5 gpurun_out/pytest_r04a.log
cd /tmp && export TMPDIR=/tmp
FADEHIP_BGZF_PROF=1 FADEHIP_BGZF_GEOM=32 python $R/tools/bgzf_rate.py 256 > $R/gpurun_out/bgzf_rate_g32.log 2>&1
grep "fadehip bgzf" $R/gpurun_out/bgzf_rate_g32.log | tail -3
grep GBps $R/gpurun_out/bgzf_rate_g32.log | tail -2
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $pass | cut -d' ' -f1)
  FADEHIP_BGZF_GEOM=32 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/bgzf_rate.py 128 > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$pass" <<'PY' >> $R/gpurun_out/bgzf_pmc_r04a.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "bgzf" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    print(k[:60], {c: (v, n[(k, c)]) for c, v in acc[k].items()})
PY
done
cat $R/gpurun_out/bgzf_pmc_r04a.txt | cut -c1-400
python $R/tools/e2e_quick.py 10000000 default= devinf=FADE_BAM_INFLATE=device: > $R/gpurun_out/e2e_quick_r04a.log 2>&1
cat $R/gpurun_out/e2e_quick_r04a.log | cut -c1-600
