"""Pins the CPU oracle (oracle/s2k_oracle.c) against every known-answer vector the reference
holds for the seq -> k-min-mer path: G1 tests/main.rs:41-57, G2
src/old/nthash_hpc.rs.orig:68-77, G3 tests/main.rs:76-78, G4 tests/main.rs:82-89 -- and against the
derived checkpoints the surveyor computed independently."""
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLD
from gpu_util import FOLD_FIELDS, folds_np
from oracle import s2k_oracle as so

KAT = json.load(open(os.path.join(GOLD, "ref_kat.json")))
MODES = {"regular": so.REGULAR, "hpc": so.HPC, "simd": so.SIMD, "hpcsimd": so.HPCSIMD}


def test_hash_bound(oracle):
    # src/lib.rs:91 ; values from SURVEY.md section 0.1
    assert oracle.hash_bound(0.01) == 42949672
    assert oracle.hash_bound(0.001) == 4294967
    assert oracle.hash_bound(1e-4) == 429496
    assert oracle.hash_bound(1.0) == 0xFFFFFFFF
    assert oracle.hash_bound(7.5) == 0xFFFFFFFF  # saturating cast
    assert oracle.hash_bound(-1.0) == 0
    assert oracle.hash_bound(float("nan")) == 0
    # f32 re-derivation (src/nthash_avx512_32.rs:46-48) agrees at the BASELINE densities
    assert oracle.hash_bound_simd(42949672) == 42949672
    assert oracle.hash_bound_simd(4294967) == 4294967


def test_seeds_and_mix(oracle):
    L = oracle.lib
    assert L.s2k_oracle_seed_h(ord("A")) == 0x95C60474 and L.s2k_oracle_seed_h(ord("C")) == 0x62A02B4C
    assert L.s2k_oracle_seed_h(ord("G")) == 0x82572324 and L.s2k_oracle_seed_h(ord("T")) == 0x4BE24456
    assert L.s2k_oracle_seed_h(ord("N")) == 0 and L.s2k_oracle_seed_h(ord("a")) == 1 and L.s2k_oracle_seed_h(0) == 1
    assert L.s2k_oracle_seed_rc(ord("A")) == 0x4BE24456 and L.s2k_oracle_seed_rc(ord("T")) == 0x95C60474
    assert L.s2k_oracle_seed_rc(ord("C")) == 0x82572324 and L.s2k_oracle_seed_rc(ord("G")) == 0x62A02B4C
    x = 0x12345678
    y = x ^ (x << 13)
    y ^= y >> 7
    y ^= (y << 17) & 0xFFFFFFFFFFFFFFFF
    assert oracle.mix32(x) == y & 0xFFFFFFFFFFFFFFFF


def test_G1_regular_kat(oracle, ecoli):
    g = KAT["G1"]
    km = oracle.kminmers(ecoli, g["l"], g["k"], g["density"], so.REGULAR)
    assert [int(x) for x in km["hash"]] == g["hashes32"]
    # first three full tuples, SURVEY.md 8c
    assert (int(km["start"][0]), int(km["end"][0]), int(km["rev"][0])) == (2341, 9477, 1)
    assert (int(km["start"][1]), int(km["end"][1]), int(km["rev"][1])) == (4611, 14536, 0)
    assert (int(km["start"][2]), int(km["end"][2]), int(km["rev"][2])) == (6460, 17829, 1)


def test_G2_hpc_u64_kat(oracle):
    g = KAT["G2"]
    bound64 = int(g["density"] * float(2**64 - 1))
    j, h = oracle.hpc_literal_u64(g["seq"].encode(), g["l"], bound64)
    assert [[int(a), int(b)] for a, b in zip(j, h)] == g["pos_hash64"]
    # the l-mer at p=41 passes the bound but is the dropped last HPC l-mer
    assert 41 not in [int(a) for a in j]


def test_G3_hpc_equivalence(oracle, ecoli):
    a, b, c = oracle.hpc(ecoli, 0), oracle.hpc(ecoli, 1), oracle.hpc(ecoli, 2)
    assert a[0] == b[0] == c[0]
    assert len(a[0]) == KAT["G3"]["runs"]
    assert (b[1] == c[1]).all()


def test_hpc_string_semantics(oracle):
    assert oracle.hpc(b"AACCCGT", 0)[0] == b"ACGT"
    s, p = oracle.hpc(b"AACCCGT", 1)
    assert s == b"ACGT" and list(p) == [0, 2, 5, 6]
    # encode_rle collapses only ACTGactgNn (src/hpc.rs:14); hpc collapses anything (src/hpc.rs:34)
    assert oracle.hpc(b"AAXXC", 0)[0] == b"AXC"
    assert oracle.hpc(b"AAXXC", 1)[0] == b"AXXC"
    assert oracle.hpc(b"AAXXC", 2)[0] == b"AXC"
    assert oracle.hpc(b"", 0)[0] == b"#"  # src/hpc.rs:39-40 pushes prev_char unconditionally
    assert oracle.hpc(b"", 2)[0] == b""


def test_G4_mode_grid(oracle, ecoli):
    g = KAT["G4"]
    for l in g["l"]:
        for k in g["k"]:
            r = oracle.kminmers(ecoli, l, k, g["density"], so.REGULAR)["hash"]
            s = oracle.kminmers(ecoli, l, k, g["density"], so.SIMD)["hash"]
            h = oracle.kminmers(ecoli, l, k, g["density"], so.HPC)["hash"]
            hs = oracle.kminmers(ecoli, l, k, g["density"], so.HPCSIMD)["hash"]
            assert len(r) == len(s) and (r == s).all(), (l, k)
            assert len(h) == len(hs) and (h == hs).all(), (l, k)
            if (l, k) == (31, 5):
                assert (len(h), len(r)) == (1471, 1942)


def test_survey_checkpoints_config1(oracle, ecoli):
    """BASELINE config 1 numbers computed independently by the surveyor."""
    b = oracle.hash_bound(0.01)
    j, je, h = oracle.minimizers(ecoli, 31, b, so.REGULAR)
    km = oracle.kminmers(ecoli, 31, 10, 0.01, so.REGULAR)
    assert (len(j), len(km["hash"])) == (1946, 1937)
    assert int(np.bitwise_xor.reduce(km["hash"])) == 0xE3846A0B9F6CE945
    assert (int(km["start"].sum()), int(km["end"].sum()), int(km["rev"].sum())) == (95451161, 96405826, 994)
    assert (int(j[0]), int(je[0]), int(h[0])) == (8, 38, 31493461)
    assert (int(j[-1]), int(je[-1]), int(h[-1])) == (99894, 99924, 18118674)
    assert (int(km["hash"][0]), int(km["start"][0]), int(km["end"][0]), int(km["rev"][0])) == (10804563991471863687, 8, 321, 1)
    assert (int(km["hash"][-1]), int(km["start"][-1]), int(km["end"][-1]), int(km["rev"][-1])) == (5911554579291328986, 99601, 99924, 0)
    j, je, h = oracle.minimizers(ecoli, 31, b, so.HPC)
    km = oracle.kminmers(ecoli, 31, 10, 0.01, so.HPC)
    assert (len(j), len(km["hash"])) == (1475, 1466)
    assert int(np.bitwise_xor.reduce(km["hash"])) == 0xC926709C04238712
    assert (int(km["start"].sum()), int(km["end"].sum()), int(km["rev"].sum())) == (73649678, 74603761, 764)
    assert (int(j[0]), int(je[0]), int(h[0])) == (31, 73, 9594429)
    assert (int(j[-1]), int(je[-1]), int(h[-1])) == (99826, 99866, 23023354)
    assert (int(km["hash"][0]), int(km["start"][0]), int(km["end"][0]), int(km["rev"][0])) == (9743344244064261373, 31, 1109, 0)
    assert (int(km["hash"][-1]), int(km["start"][-1]), int(km["end"][-1]), int(km["rev"][-1])) == (3174290890741566874, 99660, 99866, 1)


def test_committed_checkpoints_reproduce(oracle, ecoli):
    ck = json.load(open(os.path.join(GOLD, "ecoli_checkpoints.json")))
    for c in ck["configs"]:
        km = oracle.kminmers(ecoli, c["l"], c["k"], c["density"], MODES[c["mode"]])
        assert len(km["hash"]) == c["n_kminmers"]
        if c["n_kminmers"]:
            assert int(np.bitwise_xor.reduce(km["hash"])) == c["xor_hash"]
            assert int(km["start"].sum()) == c["sum_start"] and int(km["end"].sum()) == c["sum_end"]


def test_committed_cases_reproduce(oracle):
    cases = json.load(open(os.path.join(GOLD, "derived_cases.json")))["cases"]
    assert len(cases) >= 80
    for c in cases:
        seq = bytes.fromhex(c["seq_hex"])
        km = oracle.kminmers(seq, c["l"], c["k"], c["density"], MODES[c["mode"]])
        for f in ("hash", "start", "end", "rev"):
            assert [int(x) for x in km[f]] == c["kminmers"][f], (c["name"], c["mode"], f)


def _rand_hp(rng, n):
    out = bytearray()
    alpha = b"ACGT" * 6 + b"NnacgtX*"
    while len(out) < n:
        out += bytes([rng.choice(alpha)]) * rng.choice([1, 1, 1, 2, 2, 3, 5, 9, 40, 300])
    return bytes(out[:n])


def test_hpc_literal_state_machine_equals_closed_form(oracle):
    """The ring-buffer state machine (src/nthash_hpc.rs:115-283, transliterated) and the closed form
    (runs; drop last l-mer; end = st[p+l]-1) agree on homopolymer-rich reads with N/lowercase/other."""
    rng = random.Random(7)
    for it in range(400):
        n = rng.choice([1, 2, 5, 9, 33, 64, 200, 1000])
        l = rng.choice([1, 2, 4, 7, 15, 31, 32, 40])
        s = _rand_hp(rng, n)
        bound = rng.choice([0xFFFFFFFF, oracle.hash_bound(0.3), oracle.hash_bound(0.02)])
        if len(s) <= l:
            continue
        a = oracle.hpc_literal(s, l, bound)
        b = oracle.minimizers(s, l, bound, so.HPC)
        assert all((x == y).all() and len(x) == len(y) for x, y in zip(a, b)), (it, n, l)


def test_regular_rolling_equals_definition(oracle):
    rng = random.Random(11)
    for it in range(100):
        n = rng.choice([10, 50, 333, 2000])
        l = rng.choice([1, 3, 10, 31, 32, 33, 63])
        s = _rand_hp(rng, n)
        if n <= l:
            continue
        j, je, h = oracle.minimizers(s, l, 0xFFFFFFFF, so.REGULAR)
        ref = oracle.nthash32_all(s, l)
        assert len(h) == n - l + 1 and (h == ref).all()
        assert (j == np.arange(n - l + 1)).all() and (je == j + l - 1).all()


def test_kminmer_rolling_equals_closed_form(oracle):
    """src/lib.rs:238-249 (rolling) vs src/lib.rs:275-288 (closed form) incl. k >= 64."""
    rng = np.random.default_rng(5)
    for k in (1, 2, 5, 10, 31, 64, 70):
        mh = rng.integers(0, 2**32, size=200, dtype=np.uint64).astype(np.uint32)
        hr, rr = oracle.kminmer_hashes_rolling(mh, k)
        mx = [oracle.mix32(int(x)) for x in mh]
        rot = lambda x, r: ((x << (r % 64)) | (x >> ((64 - r % 64) % 64))) & (2**64 - 1) if r % 64 else x
        for c in range(len(mh) - k + 1):
            F = R = 0
            for i in range(k):
                F ^= rot(mx[c + i], k - 1 - i)
                R ^= rot(mx[c + i], i)
            assert int(hr[c]) == min(F, R) and int(rr[c]) == (R < F)


def test_batch_matches_per_read(oracle):
    rng = random.Random(3)
    reads = [_rand_hp(rng, rng.choice([0, 5, 31, 32, 33, 100, 2000, 5000])) for _ in range(40)]
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    for mode in (so.REGULAR, so.HPC):
        for threads in (1, 3):
            res = oracle.batch(bases, off, 31, 3, 0.05, mode, threads=threads)
            for r, s in enumerate(reads):
                km = oracle.kminmers(s, 31, 3, 0.05, mode)
                a, b = int(res["km_off"][r]), int(res["km_off"][r + 1])
                assert b - a == len(km["hash"])
                assert (res["hash"][a:b] == km["hash"]).all() and (res["start"][a:b] == km["start"]).all()
                assert (res["end"][a:b] == km["end"]).all() and (res["rev"][a:b] == km["rev"]).all()
            assert oracle.batch_count_timed(bases, off, 31, 3, 0.05, mode, threads=threads) == res["n"]


def test_synth_is_position_keyed(oracle):
    a = oracle.synth_bases(1, 0, 1000)
    b = oracle.synth_bases(1, 137, 500)
    assert (a[137:637] == b).all()
    assert set(a.tobytes()) <= set(b"ACGT")
    cnt = np.bincount(oracle.synth_bases(9, 0, 200000), minlength=256)
    assert all(abs(cnt[c] - 50000) < 1500 for c in b"ACGT")


def test_synth_checksums_equal_materialised_batch(oracle):
    """the read-by-read checksum driver used by the full-size GPU parity test agrees with the batch API"""
    n_reads, L = 37, 3001
    bases = oracle.synth_bases(5, 0, n_reads * L)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    for mode in (0, 1):
        ref = oracle.batch(bases, off, 31, 10, 0.01, mode, threads=2)
        cs = oracle.synth_checksums(5, n_reads, L, 31, 10, 0.01, mode, threads=3)
        assert cs["n_kminmers"] == ref["n"] > 0
        assert cs["xor_hash"] == int(np.bitwise_xor.reduce(ref["hash"]))
        assert cs["sum_start"] == int(ref["start"].astype(np.uint64).sum())
        assert cs["sum_end"] == int(ref["end"].astype(np.uint64).sum())
        assert cs["n_rev"] == int(ref["rev"].sum())
        assert cs["n_minimizers"] == int(oracle.batch_minimizers(bases, off, 31, 0.01, mode)["n"])
        f = folds_np(ref["km_off"], ref["hash"], ref["start"], ref["end"], ref["rev"])
        assert {x: cs[x] for x in FOLD_FIELDS} == f


def test_synth_checksums_ragged_equal_materialised_batch(oracle):
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 5000, size=41)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    bases = oracle.synth_bases(9, 0, int(off[-1]))
    for mode in (0, 1):
        ref = oracle.batch(bases, off, 31, 10, 0.01, mode, threads=2)
        cs = oracle.synth_checksums_off(9, off, 31, 10, 0.01, mode, threads=3)
        assert cs["n_kminmers"] == ref["n"] > 0 and cs["xor_hash"] == int(np.bitwise_xor.reduce(ref["hash"]))
        assert cs["sum_start"] == int(ref["start"].astype(np.uint64).sum()) and cs["sum_end"] == int(ref["end"].astype(np.uint64).sum())
        assert cs["n_rev"] == int(ref["rev"].sum())
