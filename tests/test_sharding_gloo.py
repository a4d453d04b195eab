"""N>1 path on CPU: world_size-2 gloo.  Each rank takes its shard (sharding.py), the oracle stands in for the
GPU kernels (test infrastructure only), counts are all-reduced, and the concatenation of the shards' outputs in
rank order must equal the unsharded output."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_reads():
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.integers(0, 6000, size=60)] + [0, 1, 31, 40000, 9216]
    reads = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)] for n in lens]
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    return np.concatenate(reads), off


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import s2k_oracle as so
    from s2k_loader import import_package

    pkg = import_package()
    import importlib.util

    spec = importlib.util.spec_from_file_location("s2k_sharding", os.path.join(ROOT, "rust-seq2kminmers_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases, off = _make_reads()
    lb, loff, r0 = sh.local_shard(bases, off, world, rank)
    o = so.get()
    res = o.batch(lb, loff, 31, 5, 0.02, so.HPC)
    mn = o.batch_minimizers(lb, loff, 31, 0.02, so.HPC)
    local = {"n_reads": len(loff) - 1, "n_bases": len(lb), "n_minimizers": mn["n"], "n_kminmers": res["n"]}
    tot = sh.allreduce_counts(local, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (r0, res["km_off"].tolist(), res["hash"].tolist(), res["start"].tolist(), res["end"].tolist()))
    if rank == 0:
        q.put((tot, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_properties():
    import importlib.util

    spec = importlib.util.spec_from_file_location("s2k_sharding", os.path.join(ROOT, "rust-seq2kminmers_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    _, off = _make_reads()
    for w in (1, 2, 3, 4, 8):
        b = sh.shard_bounds(off, w)
        assert b[0] == 0 and b[-1] == len(off) - 1 and (np.diff(b) >= 0).all()
        sizes = [int(off[b[i + 1]]) - int(off[b[i]]) for i in range(w)]
        assert sum(sizes) == int(off[-1])
        assert max(sizes) - min(sizes) <= 2 * 40000  # balanced by bases up to one (longest) read
    b = sh.shard_bounds(np.array([0], dtype=np.uint64), 4)  # no reads at all
    assert (b == 0).all()


def test_world2_gloo_shards_reproduce_unsharded_output(oracle):
    from oracle import s2k_oracle as so

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tot, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bases, off = _make_reads()
    ref = oracle.batch(bases, off, 31, 5, 0.02, so.HPC)
    mn = oracle.batch_minimizers(bases, off, 31, 0.02, so.HPC)
    assert tot == {"n_reads": len(off) - 1, "n_bases": len(bases), "n_minimizers": mn["n"], "n_kminmers": ref["n"]}
    gathered.sort(key=lambda g: g[0])
    hashes, starts, ends, km_off = [], [], [], [0]
    for (r0, ko, h, s, e) in gathered:
        assert r0 == len(km_off) - 1
        base = km_off[-1]
        km_off += [base + x for x in ko[1:]]
        hashes += h
        starts += s
        ends += e
    assert km_off == ref["km_off"].tolist()
    assert hashes == ref["hash"].tolist() and starts == ref["start"].tolist() and ends == ref["end"].tolist()
