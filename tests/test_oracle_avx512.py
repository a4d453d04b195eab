"""The AVX-512 CPU baseline (oracle/s2k_oracle_avx512.c) must reproduce the scalar oracle's Simd / HpcSimd
result semantics exactly (SURVEY.md 8a traps i, ii, v, vi), since its throughput is quoted beside the GPU's."""
import random

import numpy as np
import pytest

from oracle import s2k_oracle as so


@pytest.fixture(scope="module")
def avx():
    try:
        a = so.OracleAvx512()
    except RuntimeError as e:
        pytest.skip(str(e))
    if not a.supported():
        pytest.skip("host CPU lacks AVX-512 F/BW/VL/VBMI2")
    return a


def _rand(rng, n):
    alpha = b"ACGT" * 8 + b"NnacgtXQ*d"
    out = bytearray()
    while len(out) < n:
        out += bytes([rng.choice(alpha)]) * rng.choice([1, 1, 1, 1, 2, 3, 7, 70])
    return bytes(out[:n])


def test_avx512_matches_scalar_simd_semantics(avx, oracle, ecoli):
    rng = random.Random(5)
    seqs = [ecoli[:20000], ecoli[3000:3100]] + [_rand(rng, n) for n in (32, 33, 46, 47, 48, 62, 63, 64, 65, 100, 1000, 5000, 31 + 16 * 7 - 1)]
    for s in seqs:
        for l in (1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 15, 16, 17, 25, 28, 31):
            for d in (0.01, 0.3, 1.0):
                b = oracle.hash_bound(d)
                for hpc, mode in ((0, so.SIMD), (1, so.HPCSIMD)):
                    rj, rje, rh = oracle.minimizers(s, l, b, mode)
                    for variant in (0, 1):  # the doubling scan; the reference's rolling 16-lane scan (same results, two algorithms)
                        j, je, h = avx.minimizers(s, l, b, hpc, variant=variant)
                        assert len(j) == len(rj), (len(s), l, d, hpc, variant)
                        assert (j == rj).all() and (je == rje).all() and (h == rh).all(), (len(s), l, d, hpc, variant)


def test_avx512_batch_count(avx, oracle):
    rng = random.Random(6)
    reads = [_rand(rng, rng.choice([0, 10, 31, 32, 500, 4000])) for _ in range(50)]
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    for hpc, mode in ((0, so.SIMD), (1, so.HPCSIMD)):
        ref = oracle.batch(bases, off, 31, 3, 0.05, mode, count_only=True)["n"]
        for variant in (0, 1):
            assert avx.batch_count(bases, off, 31, 3, 0.05, hpc, variant=variant) == ref
            assert avx.batch_count(bases, off, 31, 3, 0.05, hpc, threads=4, variant=variant) == ref
