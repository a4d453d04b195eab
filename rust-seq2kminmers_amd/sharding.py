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


# ---- downstream k-min-mer counting across GPUs: the one place this path has a real exchange step ----
class EngineCountOps:
    """partition / count on the GPU through the C ABI (s2k_partition_device / s2k_count_device); tensors are torch CUDA tensors."""

    def __init__(self, eng, device):
        self.eng, self.device = eng, device

    def partition(self, keys, n_parts):
        import torch

        out = torch.empty_like(keys)
        off = torch.zeros(n_parts + 1, dtype=torch.int64, device=self.device)
        torch.cuda.synchronize(self.device)
        self.eng.partition_device(keys.data_ptr() if keys.numel() else 0, keys.numel(), n_parts, out.data_ptr() if keys.numel() else 0, off.data_ptr())
        return out, off.cpu()

    def count(self, keys, want_pairs=False):
        import torch

        n = keys.numel()
        torch.cuda.synchronize(self.device)
        if not want_pairs:
            return self.eng.count_device(keys.data_ptr() if n else 0, n), None, None
        k = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        c = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        nd = self.eng.count_device(keys.data_ptr() if n else 0, n, k.data_ptr(), c.data_ptr(), max(n, 1))
        return nd, k[:nd], c[:nd]


def count_kminmers(keys, ops, dist=None, collectives_on_device=True, want_pairs=False):
    """Distinct k-min-mer hashes and their multiplicities over ALL ranks.  `keys`: this rank's k-min-mer hashes (int64 view of
    the u64 values).  One rank: count locally.  Several: every rank splits its keys by hash prefix into world_size groups,
    one all-to-all (RCCL when the tensors live on the GPU) moves group p to rank p, which then owns every occurrence of its
    range of the hash space and counts it; the totals are summed with an all-reduce.
    Returns dict(n_keys, n_distinct, n_distinct_local, keys, counts) -- keys/counts are this rank's range (want_pairs)."""
    import torch

    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    if world == 1:
        nd, k, c = ops.count(keys, want_pairs)
        return {"n_keys": int(keys.numel()), "n_distinct": nd, "n_distinct_local": nd, "keys": k, "counts": c}
    grouped, off = ops.partition(keys, world)
    send = (off[1:] - off[:-1]).to(torch.int64)
    cdev = keys.device if collectives_on_device else torch.device("cpu")
    send_c = send.to(cdev)
    recv_c = torch.empty_like(send_c)
    dist.all_to_all_single(recv_c, send_c)  # how many keys come from each rank
    recv = recv_c.cpu()
    src = grouped if collectives_on_device else grouped.cpu()
    got = torch.empty(int(recv.sum()), dtype=keys.dtype, device=cdev)
    dist.all_to_all_single(got, src, output_split_sizes=[int(x) for x in recv], input_split_sizes=[int(x) for x in send])
    got = got.to(keys.device)
    nd, k, c = ops.count(got, want_pairs)
    t = torch.tensor([int(keys.numel()), nd], dtype=torch.int64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {"n_keys": int(t[0]), "n_distinct": int(t[1]), "n_distinct_local": nd, "keys": k, "counts": c, "n_received": int(got.numel())}
