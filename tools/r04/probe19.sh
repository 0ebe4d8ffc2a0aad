#!/bin/bash
# round 4, GPU call 20: the stages of one 10 M-read file run as a timeline (FADE_TRACE)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FADE_TRACE=1 timeout -k 10 600 python $R/tools/e2e_quick.py 10000000 default= > $R/gpurun_out/trace_e2e.log 2>&1
python - <<'PY'
import json, os
r = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "e2e_quick.json")))
t = r["default"]["timing"]
print(len(t))
PY
grep -c trace $R/gpurun_out/trace_e2e.log
