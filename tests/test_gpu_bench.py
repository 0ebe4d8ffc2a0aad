"""bench.py's own launch logic on the GPU box: `--gpus N` outside torchrun must start N fresh ranks (before anything has
touched the GPU), report n_gpus = N and reduce the stats over all of them.  Two ranks share the box's one GPU here, so
the collective runs over gloo (FADE_BENCH_BACKEND); on a multi-GPU node the same code path uses RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()  # ONE json line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_ranks_and_reduces_over_them():
    small = ["--steps", "2", "--warmup", "1", "--batch-reads", "20000", "--no-cpu", "--no-e2e"]
    one = _bench(["--gpus", "1"] + small)
    two = _bench(["--gpus", "2"] + small, env={"FADE_BENCH_BACKEND": "gloo", "MASTER_PORT": "29611"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    per_step = 20000 * 10
    assert one["stats"]["read_count"] == 2 * per_step and two["stats"]["read_count"] == 2 * 2 * per_step
    assert two["stats"]["art"] > one["stats"]["art"] > 0  # rank 1 annotates its own shard (other seeds)
    for r in (one, two):
        assert r["roofline"]["kernel"].startswith("sw_pk_kernel<19,1,LG=8>") and 0 < r["roofline"]["frac"] < 0.05
        assert abs(r["roofline"]["bytes_per_unit"] - 316) < 10  # 75 + 161 + 16 + 64
        assert r["value"] > 0 and r["value_resident"] > 0 and r["cpu_baseline"] is None


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120, env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert p.returncode == 2 and b"WORLD_SIZE" in p.stderr


def test_bench_under_torchrun_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: the ranks come from torchrun, bench.py
    must not spawn again.  Two ranks on the one GPU (gloo for the collective)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch-reads", "20000", "--no-cpu", "--no-e2e"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, FADE_BENCH_BACKEND="gloo"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["stats"]["read_count"] == 2 * 20000 * 10 and r["scaling"] == "weak"


def test_bench_line_carries_the_record_path_and_the_end_to_end_legs():
    """One rank, small: value_from_records (the path from BAM record bytes, nothing decided on the host) beside value, the CPU
    baseline on the same definition, the e2e block with every leg best-of-3 and the larger file's legs."""
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "1", "--batch-reads", "40000", "--e2e-reads", "400000", "--e2e-big-reads", "800000"])
    assert r["value_from_records"] > 0 and "fadehip_bam_front_raw" in r["value_from_records_is"]
    c = r["cpu_baseline_from_records"]
    assert c["value"] > 0 and c["kind"] == "port" and c["cores"] >= 1 and c["gpu_over_cpu"] > 0
    e = r["e2e"]
    assert e["gpu_reads_per_s"] > 0 and e["cpu_reads_per_s"] > 0 and e["gpu"]["best_of"] == 3 and e["cpu"]["best_of"] == 3
    assert 0 < e["cpu_zlib_reads_per_s"] < e["cpu_reads_per_s"] and e["gpu_over_cpu_zlib"] > e["gpu_over_cpu"]  # (zlib -6 is the slower codec)
    assert e["big"]["reads"] == 800000 and e["big"]["gpu_reads_per_s"] > 0 and e["big"]["cpu_reads_per_s"] > 0
    assert "kernel_ms_is" in r["roofline"] and r["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_run_the_end_to_end_leg_on_their_shards():
    """`--gpus 2` with the e2e leg: every rank annotates ITS shard's BAM file (`fade annotate -b`, its own process and device
    context), rank 0 first alone, then both at once; the line reports per-rank times, the aggregate and e2e_weak_scaling."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--batch-reads", "40000", "--e2e-reads", "400000", "--no-cpu"],
               env={"FADE_BENCH_BACKEND": "gloo", "MASTER_PORT": "29677"})
    e = r["e2e"]
    assert r["n_gpus"] == 2 and len(e["per_rank_seconds"]) == 2 and all(x > 0 for x in e["per_rank_seconds"])
    assert e["alone_reads_per_s"] > 0 and e["aggregate_reads_per_s"] > 0 and 0 < e["e2e_weak_scaling"] < 3
    assert r["value_from_records"] > 0
