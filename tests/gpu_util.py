import numpy as np

from oracle import s2k_oracle as so
from s2k_loader import import_package

pkg = import_package()
OMODE = {pkg.HashMode.Regular: so.REGULAR, pkg.HashMode.Hpc: so.HPC, pkg.HashMode.Simd: so.SIMD, pkg.HashMode.HpcSimd: so.HPCSIMD}
FIELDS = ("km_off", "hash", "start", "end", "rev")


def compare(eng, oracle, reads, l, k, d, mode, force_serial=False, expect_path=None, minimizers=True, tag=""):
    bases, off = pkg.pack_reads(reads)
    got = eng.extract(bases, off, l, k, d, mode, want_minimizers=minimizers, force_serial=force_serial)
    ref = oracle.batch(bases, off, l, k, d, OMODE[mode])
    ctx = (tag, int(mode), l, k, d, force_serial)
    assert got["n"] == ref["n"], ("n_kminmers", ctx, got["n"], ref["n"])
    for f in FIELDS:
        if not (got[f] == ref[f]).all():
            bad = int(np.nonzero(got[f] != ref[f])[0][0])
            raise AssertionError(("mismatch", f, ctx, bad, int(got[f][bad]), int(ref[f][bad])))
    if minimizers:
        rm = oracle.batch_minimizers(bases, off, l, d, OMODE[mode])
        assert got["n_minimizers"] == rm["n"], ("n_minimizers", ctx, got["n_minimizers"], rm["n"])
        for a, b in (("mn_off", "mn_off"), ("mn_j", "j"), ("mn_jend", "jend"), ("mn_hash", "hash")):
            if not (got[a] == rm[b]).all():
                bad = int(np.nonzero(got[a] != rm[b])[0][0])
                raise AssertionError(("mismatch", a, ctx, bad, int(got[a][bad]), int(rm[b][bad])))
    c = got["counts"]
    assert c["n_kminmers"] == ref["n"] and c["n_reads"] == len(reads) and c["n_bases"] == len(bases)
    assert c["xor_hash"] == (int(np.bitwise_xor.reduce(ref["hash"])) if ref["n"] else 0)
    assert c["hash_bound"] == oracle.hash_bound(d)
    if expect_path is not None:
        # 0 = tiled kernel, fused single pass; 2 = tiled kernel + k-min-mer kernel (k > 32, or a tile with > 30 read starts):
        # a caller that expects "tiled" accepts both, "fused" / "two-kernel" / "serial" (1) are exact
        ok = {0: (0, 2), "desc": (0,), "legacy": (2,), 1: (1,), 2: (2,)}[expect_path]
        assert c["path"] in ok, ("path", ctx, c["path"])
    return got


def rand_read(rng, n, hp=0.0, alphabet=b"ACGT", odd=0.0):
    """n bases; hp = probability mass of homopolymer extension; odd = fraction of non-ACGT bytes."""
    if n == 0:
        return b""
    a = np.frombuffer(alphabet, dtype=np.uint8)
    s = a[rng.integers(0, len(a), size=n)]
    if hp > 0:
        reps = rng.choice([1, 2, 3, 4, 8, 30], size=n, p=[1 - hp, hp * 0.5, hp * 0.25, hp * 0.15, hp * 0.07, hp * 0.03])
        s = np.repeat(s, reps)[:n]
    if odd > 0:
        m = rng.random(n) < odd
        s = s.copy()
        s[m] = np.frombuffer(b"NNNnacgtXR*-", dtype=np.uint8)[rng.integers(0, 12, size=int(m.sum()))]
    return s.tobytes()


M64 = (1 << 64) - 1


def folds_np(km_off, hash_, start, end, rev):
    """Order-sensitive whole-run folds of an output (numpy, mod 2^64): SUM x[g] (2 g + 1) over the global item index g,
    and SUM km_off[r] (2 r + 1) over r = 0 .. n_reads -- the host-side twin of oracle s2k_oracle_synth_checksums' fold_*."""
    n = len(hash_)
    with np.errstate(over="ignore"):
        w = (np.arange(n, dtype=np.uint64) << np.uint64(1)) + np.uint64(1)
        wr = (np.arange(len(km_off), dtype=np.uint64) << np.uint64(1)) + np.uint64(1)
        cnt = np.diff(np.asarray(km_off, dtype=np.uint64))
        return {
            "fold_hash": int((np.asarray(hash_, dtype=np.uint64) * w).sum(dtype=np.uint64)),
            "fold_start": int((np.asarray(start).astype(np.uint64) * w).sum(dtype=np.uint64)),
            "fold_end": int((np.asarray(end).astype(np.uint64) * w).sum(dtype=np.uint64)),
            "fold_rev": int((np.asarray(rev).astype(np.uint64) * w).sum(dtype=np.uint64)),
            "fold_km_off": int((np.asarray(km_off, dtype=np.uint64) * wr).sum(dtype=np.uint64)),
            "fold_count": int((cnt * wr[:-1]).sum(dtype=np.uint64)),
        }


def folds_torch(t, n, n_reads):
    """The same folds of device-resident outputs (dict of torch tensors as the tests allocate them: int64 hash / km_off,
    int32 start / end, uint8 rev); int64 arithmetic wraps mod 2^64 like the oracle's uint64."""
    import torch

    dev = t["hash"].device
    out = {}
    chunk = 1 << 26
    acc = {f: 0 for f in ("hash", "start", "end", "rev")}
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        w = torch.arange(a, b, dtype=torch.int64, device=dev) * 2 + 1
        for f in acc:
            acc[f] = (acc[f] + int((t[f][a:b].to(torch.int64) * w).sum().item())) & M64
    for f in acc:
        out["fold_" + f] = acc[f]
    wr = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * 2 + 1
    ko = t["km_off"][: n_reads + 1]
    out["fold_km_off"] = int((ko * wr).sum().item()) & M64
    out["fold_count"] = int(((ko[1:] - ko[:-1]) * wr[:-1]).sum().item()) & M64
    return out


FOLD_FIELDS = ("fold_hash", "fold_start", "fold_end", "fold_rev", "fold_km_off", "fold_count")


def sample_reads_compare(eng_out, n, oracle, seed, first_base, off_of, read_ids, l, k, d, omode, threads=8, gen=None):
    """>= 1 % element-wise check of a full-size run: the reads `read_ids` (ascending) are regenerated on the host, run
    through the oracle, and compared field by field with the slices [km_off[r], km_off[r+1]) of the device output.
    eng_out: dict of torch tensors; off_of(r) -> (stream offset, length) of read r; gen(r, length) -> the bases of read r
    (default: the slice of the uniform synthetic stream)."""
    import torch

    read_ids = np.asarray(read_ids, dtype=np.int64)
    lens = np.array([off_of(int(r))[1] for r in read_ids], dtype=np.uint64)
    soff = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    bases = np.empty(int(soff[-1]), dtype=np.uint8)
    for i, r in enumerate(read_ids):
        a, ln = off_of(int(r))
        bases[int(soff[i]): int(soff[i + 1])] = gen(int(r), ln) if gen else oracle.synth_bases(seed, first_base + a, ln)
    ref = oracle.batch(bases, soff, l, k, d, omode, threads=threads)
    dev = eng_out["km_off"].device
    idx = torch.from_numpy(read_ids).to(dev)
    ko = eng_out["km_off"]
    s0, s1 = ko[idx], ko[idx + 1]
    cnt = (s1 - s0).cpu().numpy().astype(np.uint64)
    rcnt = np.diff(ref["km_off"]).astype(np.uint64)
    assert (cnt == rcnt).all(), ("per-read k-min-mer counts differ", int(read_ids[np.nonzero(cnt != rcnt)[0][0]]))
    tot = int(cnt.sum())
    assert tot == ref["n"]
    if tot == 0:
        return 0
    lens_t = (s1 - s0)
    excl = torch.cumsum(lens_t, 0) - lens_t
    g = torch.repeat_interleave(s0 - excl, lens_t) + torch.arange(tot, dtype=torch.int64, device=dev)
    assert int(g.max().item()) < n
    for f, dt in (("hash", np.uint64), ("start", np.uint32), ("end", np.uint32), ("rev", np.uint8)):
        got = eng_out[f][g].cpu().numpy().view(dt)
        if not (got == ref[f]).all():
            bad = int(np.nonzero(got != ref[f])[0][0])
            rr = int(np.searchsorted(ref["km_off"], bad, side="right") - 1)
            raise AssertionError(("sampled read differs", f, "read", int(read_ids[rr]), "item", bad - int(ref["km_off"][rr]), int(got[bad]), int(ref[f][bad])))
    return tot


def spread_sample(n_reads, frac, rng, must=()):
    """ascending read indices: a `frac` random sample across the whole stream plus the first / last reads and `must`"""
    m = max(1, int(n_reads * frac))
    ids = set(rng.choice(n_reads, size=min(m, n_reads), replace=False).tolist())
    ids.update(r for r in (0, 1, n_reads - 2, n_reads - 1) if 0 <= r < n_reads)
    ids.update(int(r) for r in must if 0 <= int(r) < n_reads)
    return np.array(sorted(ids), dtype=np.int64)
