"""Downstream k-min-mer counting (the consumer hinted at in src/lib.rs:256-257).  Oracle = a dictionary
over the oracle's k-min-mer hashes.  GPU: s2k_count_device / s2k_partition_device through the C ABI; multi-GPU logic
(partition by hash prefix -> all-to-all -> local count) rehearsed with two gloo ranks -- on the CPU with numpy stand-ins
for the two device ops, and on one GPU with the real ones."""
import importlib.util
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from s2k_loader import import_package

pkg = import_package()
spec = importlib.util.spec_from_file_location("s2k_sharding", os.path.join(ROOT, "rust-seq2kminmers_amd", "sharding.py"))
sharding = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sharding)


class NumpyCountOps:
    """test stand-in for the two device ops (same contract as sharding.EngineCountOps), CPU tensors"""

    def partition(self, keys, n_parts):
        import torch

        k = keys.numpy().view(np.uint64)
        part = ((k.astype(object) * n_parts) >> 64).astype(np.int64) if len(k) else np.zeros(0, dtype=np.int64)
        order = np.argsort(part, kind="stable")
        off = np.zeros(n_parts + 1, dtype=np.int64)
        np.cumsum(np.bincount(part, minlength=n_parts), out=off[1:])
        return torch.from_numpy(k[order].view(np.int64).copy()), torch.from_numpy(off)

    def count(self, keys, want_pairs=False):
        import torch

        u, c = np.unique(keys.numpy().view(np.uint64), return_counts=True)
        return len(u), torch.from_numpy(u.view(np.int64).copy()), torch.from_numpy(c.astype(np.int32))


def _worker(rank, world, port, gpu, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    allk = rng.integers(0, 2 ** 64, size=60000, dtype=np.uint64)
    allk[::7] = allk[3]                                   # a heavy hitter
    allk[1::11] = rng.integers(0, 50, size=len(allk[1::11]), dtype=np.uint64)  # tiny values: all in part 0
    allk[5] = np.uint64(2 ** 64 - 1)                      # the table's EMPTY sentinel is a legal key
    mine = allk[rank::world].copy()
    if gpu:
        dev = torch.device("cuda", 0)
        eng = pkg.Engine(0)
        ops = sharding.EngineCountOps(eng, dev)
        keys = torch.from_numpy(mine.view(np.int64)).to(dev)
    else:
        ops = NumpyCountOps()
        keys = torch.from_numpy(mine.view(np.int64))
    r = sharding.count_kminmers(keys, ops, dist, collectives_on_device=False, want_pairs=True)
    k = r["keys"].cpu().numpy().view(np.uint64)
    c = r["counts"].cpu().numpy()
    lo = [(int(x) * world) >> 64 for x in k]
    assert all(p == rank for p in lo), "a rank must own exactly its range of the hash space"
    q.put((rank, r["n_keys"], r["n_distinct"], dict(zip(k.tolist(), c.tolist()))))
    dist.destroy_process_group()


def _run_two_ranks(gpu):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, gpu, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    allk = rng.integers(0, 2 ** 64, size=60000, dtype=np.uint64)
    allk[::7] = allk[3]
    allk[1::11] = rng.integers(0, 50, size=len(allk[1::11]), dtype=np.uint64)
    allk[5] = np.uint64(2 ** 64 - 1)
    u, c = np.unique(allk, return_counts=True)
    want = dict(zip(u.tolist(), c.tolist()))
    merged = {}
    for _, n_keys, n_distinct, d in res:
        assert n_keys == len(allk) and n_distinct == len(want)
        assert not (set(d) & set(merged))  # ranges are disjoint
        merged.update(d)
    assert merged == want


def test_exchange_by_hash_prefix_two_ranks_cpu():
    _run_two_ranks(False)


@pytest.mark.gpu
def test_exchange_by_hash_prefix_two_ranks_one_gpu():
    _run_two_ranks(True)


@pytest.mark.gpu
def test_count_device_matches_dictionary(oracle, ecoli):
    import torch

    dev = torch.device("cuda", 0)
    eng = pkg.Engine(0)
    ops = sharding.EngineCountOps(eng, dev)
    # k-min-mers of a read set with repeats: the same genome window sequenced many times (what a counting consumer sees)
    rng = np.random.default_rng(9)
    reads = [bytes(ecoli[a: a + 3000]) for a in rng.integers(0, len(ecoli) - 3000, size=400)]
    got = eng.extract_reads(reads, 21, 3, 0.05, pkg.HashMode.Hpc)
    ref = oracle.batch(*pkg.pack_reads(reads), 21, 3, 0.05, 1)
    assert (got["hash"] == ref["hash"]).all()
    want = {}
    for h in ref["hash"].tolist():
        want[h] = want.get(h, 0) + 1
    assert len(want) < len(ref["hash"])  # duplicates exist
    keys = torch.from_numpy(got["hash"].view(np.int64)).to(dev)
    r = sharding.count_kminmers(keys, ops, want_pairs=True)
    assert r["n_distinct"] == len(want) and r["n_keys"] == len(ref["hash"])
    assert dict(zip(r["keys"].cpu().numpy().view(np.uint64).tolist(), r["counts"].cpu().numpy().tolist())) == want
    # sizes only, capacity too small, empty input, partition contract
    assert eng.count_device(keys.data_ptr(), keys.numel()) == len(want)
    k2 = torch.zeros(4, dtype=torch.int64, device=dev)
    c2 = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(pkg.S2kError) as e:
        eng.count_device(keys.data_ptr(), keys.numel(), k2.data_ptr(), c2.data_ptr(), 4)
    assert e.value.status == 7
    assert eng.count_device(0, 0) == 0
    for parts in (1, 3, 8, 64):
        g, off = ops.partition(keys, parts)
        off = off.numpy()
        assert off[0] == 0 and off[-1] == keys.numel() and (np.diff(off) >= 0).all()
        gk = g.cpu().numpy().view(np.uint64)
        assert sorted(gk.tolist()) == sorted(got["hash"].tolist())
        for p in range(parts):
            seg = gk[off[p]: off[p + 1]]
            assert all(((int(x) * parts) >> 64) == p for x in seg[:200])
