"""Host-side sharding of a read batch across the GPUs of one node.

Reads are independent units (the reference runs one KminmersIterator per read on a thread pool,
src/main.rs:65-79), so the path partitions with no data-path collective: rank r takes a contiguous range of
reads balanced by cumulative bases, runs the same kernels on it, and the only cross-GPU traffic is an
all-reduce of the count vector.  Concatenating the ranks' outputs in rank order reproduces the unsharded output.
"""
import numpy as np

COUNT_FIELDS = ("n_reads", "n_bases", "n_minimizers", "n_kminmers")


def shard_bounds(read_off, world_size):
    """-> int64[world_size+1] read indices; shard r = reads [b[r], b[r+1]).  Balanced by cumulative bases:
    boundary r is the first read whose start offset reaches r/world of the total."""
    read_off = np.asarray(read_off, dtype=np.uint64)
    n_reads = len(read_off) - 1
    first, total = int(read_off[0]), int(read_off[-1]) - int(read_off[0])
    b = np.zeros(world_size + 1, dtype=np.int64)
    for r in range(1, world_size):
        target = first + (total * r) // world_size
        b[r] = int(np.searchsorted(read_off[: n_reads + 1], np.uint64(target), side="left"))
    b[world_size] = n_reads
    return np.maximum.accumulate(np.minimum(b, n_reads))


def local_shard(bases, read_off, world_size, rank):
    """-> (bases view, rebased read_off, first_read) of this rank's shard."""
    b = shard_bounds(read_off, world_size)
    r0, r1 = int(b[rank]), int(b[rank + 1])
    off = np.asarray(read_off[r0: r1 + 1], dtype=np.uint64)
    lo, hi = int(off[0]), int(off[-1])
    return bases[lo:hi], off - np.uint64(lo), r0


def allreduce_counts(counts, dist=None, device=None):
    """Sum the per-rank count dict over all ranks (RCCL when tensors live on the GPU, gloo on CPU)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: int(counts[k]) for k in COUNT_FIELDS}
    import torch

    t = torch.tensor([int(counts[k]) for k in COUNT_FIELDS], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(COUNT_FIELDS, t.tolist())}
