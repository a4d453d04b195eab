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
        assert c["path"] == expect_path, ("path", ctx, c["path"])
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
