"""Multi-GPU sharding of the annotate path.

Every record is independent (source/anno.d:44-46 is a pure map), so ranks own contiguous record
ranges and exchange nothing on the data path.  The one collective is the sum of the stats.d:45-54
counters at the end: torch.distributed all_reduce, which is RCCL over xGMI with the "nccl" backend
(gloo in the CPU tests)."""
import numpy as np

STAT_NAMES = ["read_count", "clipped", "sup", "art_sup", "art", "art_mate", "aln_l", "aln_r"]


def record_range(n_records, rank, world):
    """Contiguous [lo, hi) of the records rank owns; sizes differ by at most one."""
    base, extra = divmod(n_records, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def stats_from_rs(rs):
    """stats.d:45-54 Stats.parse over an array of rs bytes (host-side twin of the device stats kernel)."""
    v = np.asarray(rs, dtype=np.int64)
    sc, al, ar, ml, mr, sup = v & 1, (v >> 1) & 1, (v >> 2) & 1, (v >> 3) & 1, (v >> 4) & 1, (v >> 5) & 1
    return np.array([len(v), sc.sum(), sup.sum(), ((al | ar) & sup).sum(), (al | ar).sum(),
                     ((al & ml) | (ar & mr)).sum(), al.sum(), ar.sum()], dtype=np.int64)


def allreduce_stats(local, device="cuda"):
    """Sum the 8 counters over all ranks (the only collective of the path)."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(local, dtype=np.int64), device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
