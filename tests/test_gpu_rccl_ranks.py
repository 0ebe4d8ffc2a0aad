"""fadehip_stats_allreduce_rank (include/fadehip.h): the one collective of the path — SURVEY §8(e): one ncclAllReduce(int64,
sum) over the stats.d counters — with one PROCESS per rank, as the lanes of `fade annotate --gpus N` and a D host would
call it.  This box has one GPU: two ranks on it is all that can be tried; what RCCL says to that is recorded, not assumed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_PROG = r"""
import ctypes as C, json, sys
sys.path.insert(0, %(root)r)
import numpy as np
import fade_amd
rank, n, idp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ctx = fade_amd.Context(device=0)
c = (np.arange(8, dtype=np.int64) + 1) * (10 ** rank) + rank
rc = ctx._L.fadehip_stats_allreduce_rank(ctx._h, rank, n, idp.encode(), c.ctypes.data, 8)
msg = ctx._L.fadehip_last_error(ctx._h).decode() if rc else ""
print(json.dumps(dict(rank=rank, rc=rc, msg=msg, out=[int(x) for x in c])))
ctx.close()
"""


def _ranks(n, tmp_path, timeout=240):
    idp = str(tmp_path / ("ncclid_%d" % n))
    prog = RANK_PROG % dict(root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", prog, str(r), str(n), idp], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(n)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank did not come back: the collective hangs with %d ranks on one device" % n)
        assert p.returncode == 0, e.decode()[-1500:]
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    return outs


def test_one_rank_is_the_identity(tmp_path):
    (r,) = _ranks(1, tmp_path)
    assert r["rc"] == 0 and r["out"] == [k + 1 for k in range(8)]


def test_two_ranks_two_processes_on_the_one_device(tmp_path):
    """Either RCCL forms the communicator (then every rank must hold the column sums), or it refuses two ranks on one device
    (then EVERY rank must come back with FADEHIP_E_RCCL and a message — no rank may hang, none may return a wrong sum).
    Which of the two happened is written to gpurun_out/rccl_two_ranks.json; DESIGN.md §0(e) quotes it."""
    outs = _ranks(2, tmp_path)
    want = [(k + 1) * 1 + (k + 1) * 10 + 1 for k in range(8)]
    ok = [o["rc"] == 0 for o in outs]
    assert all(ok) or not any(ok), outs  # the ranks agree on what happened
    if all(ok):
        path = "communicator formed: two ranks on one device summed over RCCL"
        for o in outs:
            assert o["out"] == want, outs
    else:
        path = "refused"
        for o in outs:
            assert o["rc"] == -8 and ("ncclCommInitRank" in o["msg"] or "ncclAllReduce" in o["msg"]), outs  # FADEHIP_E_RCCL
            assert o["out"] == [(k + 1) * (10 ** o["rank"]) + o["rank"] for k in range(8)]  # the counters are left as they were
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_two_ranks.json"), "w") as f:
        json.dump(dict(path=path, ranks=outs), f, indent=1)
