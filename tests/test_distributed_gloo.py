"""The N>1 path on CPU: world_size-2 gloo.  Reads shard per rank with no data-path collective;
the only collective is the final sum of the stats.d counters (bench.py does the same over RCCL)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fade_amd import shard, synth
    from oracle import pyoracle as O
    cfg, g, b = synth.make_config("C2", 4000, contig_len=150_000)
    lo, hi = shard.record_range(len(b["pos"]), rank, world)
    mine = synth.take(b, np.arange(lo, hi))
    G = O.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
    # the oracle stands in for the device path here (no GPU in this test): what is under test is
    # the sharding and the reduction, which are the same code bench.py runs
    rs, _ = O.annotate_batch_soa(G, mine, cfg["floor_len"], cfg["window"], threads=2, want_am=False)
    local = shard.stats_from_rs(rs)
    total = shard.allreduce_stats(local, device="cpu")
    if rank == 0:
        rs_all, _ = O.annotate_batch_soa(G, b, cfg["floor_len"], cfg["window"], threads=2, want_am=False)
        np.save(out, np.stack([total, shard.stats_from_rs(rs_all)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_stats_reduction(tmp_path):
    out = str(tmp_path / "stats.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    total, expect = np.load(out)
    assert total[0] == 4000
    assert np.array_equal(total, expect)


def test_record_ranges_partition():
    sys.path.insert(0, ROOT)
    from fade_amd import shard
    for n in (0, 1, 7, 4000, 10_000_001):
        for w in (1, 2, 3, 8):
            r = [shard.record_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
