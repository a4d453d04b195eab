"""GPU tier: the HIP path, called through the C ABI, against the CPU oracle -- bit-exact hashes,
k-min-mer tuples and original-space positions -- plus size-independent properties at
larger sizes.  Mirrors tests/main.rs of the reference: KATs first, then cross-checks over an (l,k) grid."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD
from gpu_util import FIELDS, FOLD_FIELDS, OMODE, compare, folds_torch, pkg, rand_read, sample_reads_compare, spread_sample

pytestmark = pytest.mark.gpu
HM = pkg.HashMode
SCALAR = (HM.Regular, HM.Hpc)


@pytest.fixture(scope="module")
def eng():
    e = pkg.Engine(0)
    yield e
    e.close()


def test_G1_reference_kat(eng, ecoli):
    """tests/main.rs:41-57,60-73 through the reference-shaped iterator."""
    kat = json.load(open(os.path.join(GOLD, "ref_kat.json")))["G1"]
    it = pkg.KminmersIterator.new(ecoli, kat["l"], kat["k"], kat["density"], HM.Regular, engine=eng)
    items = list(it)
    assert [x.get_hash() for x in items] == kat["hashes32"]
    assert [x.offset for x in items] == list(range(15))
    assert (items[0].start, items[0].end, items[0].rev) == (2341, 9477, True)


def test_ecoli_checkpoints(eng, oracle, ecoli):
    ck = json.load(open(os.path.join(GOLD, "ecoli_checkpoints.json")))
    for c in ck["configs"]:
        mode = HM.Regular if c["mode"] == "regular" else HM.Hpc
        for serial in (False, True):
            got = compare(eng, oracle, [ecoli], c["l"], c["k"], c["density"], mode, force_serial=serial, tag="ecoli")
            assert got["n"] == c["n_kminmers"] and got["n_minimizers"] == c["n_minimizers"]
            assert got["counts"]["xor_hash"] == c["xor_hash"]
            assert int(got["start"].astype(np.uint64).sum()) == c["sum_start"]
            assert int(got["end"].astype(np.uint64).sum()) == c["sum_end"]


def test_G4_mode_grid(eng, oracle, ecoli):
    """tests/main.rs:82-89: Regular == Simd and Hpc == HpcSimd (hash-only equality) over the (l,k) grid."""
    for l in (5, 7, 11, 17, 25, 31):
        for k in (2, 5, 8):
            r = {m: eng.extract_reads([ecoli], l, k, 0.01, m) for m in HM}
            assert (r[HM.Regular]["hash"] == r[HM.Simd]["hash"]).all() and r[HM.Regular]["n"] == r[HM.Simd]["n"]
            assert (r[HM.Hpc]["hash"] == r[HM.HpcSimd]["hash"]).all() and r[HM.Hpc]["n"] == r[HM.HpcSimd]["n"]
            for m in HM:
                ref = oracle.kminmers(ecoli, l, k, 0.01, OMODE[m])
                assert (r[m]["hash"] == ref["hash"]).all() and (r[m]["start"] == ref["start"]).all()
                assert (r[m]["end"] == ref["end"]).all() and (r[m]["rev"] == ref["rev"]).all()


def test_committed_cases_single_and_batched(eng, oracle):
    cases = json.load(open(os.path.join(GOLD, "derived_cases.json")))["cases"]
    groups = {}
    for c in cases:
        mode = {"regular": HM.Regular, "hpc": HM.Hpc, "simd": HM.Simd, "hpcsimd": HM.HpcSimd}[c["mode"]]
        seq = bytes.fromhex(c["seq_hex"])
        got = eng.extract_reads([seq], c["l"], c["k"], c["density"], mode, want_minimizers=True)
        for f in ("hash", "start", "end", "rev"):
            assert [int(x) for x in got[f]] == c["kminmers"][f], (c["name"], c["mode"], f)
        assert [int(x) for x in got["mn_j"]] == c["minimizers"]["j"], (c["name"], c["mode"])
        assert [int(x) for x in got["mn_jend"]] == c["minimizers"]["jend"], (c["name"], c["mode"])
        assert [int(x) for x in got["mn_hash"]] == c["minimizers"]["hash"], (c["name"], c["mode"])
        groups.setdefault((mode, c["l"], c["k"], c["density"]), []).append(seq)
    # the same reads concatenated into batches (reads of very different lengths share tiles)
    for (mode, l, k, d), reads in groups.items():
        compare(eng, oracle, reads, l, k, d, mode, tag="batched-cases")
        if mode in SCALAR:
            compare(eng, oracle, reads, l, k, d, mode, force_serial=True, expect_path=1, tag="batched-cases-serial")


def test_ragged_batches_all_tile_alignments(eng, oracle):
    """Read boundaries at every interesting offset relative to the 9216-base tiles and the 144-base lane
    chunks; empty reads; reads of length l, l+1; reads spanning several tiles."""
    rng = np.random.default_rng(42)
    T = 9216
    lens = [0, 1, 30, 31, 32, 33, 143, 144, 145, 0, 0, 5000, T - 5200 - 589, 1, T, T + 1, T - 1, 31, 2 * T + 7, 3, 17, T - 31,
            T - 30, 31, 62, 0, 20000, 288, 289, 9000, 100, 40000, 15, 16, 144 * 3]
    for hp, odd in ((0.0, 0.0), (0.3, 0.02)):
        reads = [rand_read(rng, n, hp=hp, odd=odd) for n in lens]
        for mode in SCALAR:
            compare(eng, oracle, reads, 31, 10, 0.01, mode, expect_path=0, tag="ragged")
            compare(eng, oracle, reads, 31, 3, 0.05, mode, expect_path=0, tag="ragged")
            compare(eng, oracle, reads, 31, 10, 0.01, mode, force_serial=True, expect_path=1, tag="ragged-serial")


def test_l_grid_dynamic_and_static_paths(eng, oracle):
    rng = np.random.default_rng(7)
    reads = [rand_read(rng, int(n), hp=0.25, odd=0.01) for n in rng.integers(0, 30000, size=24)]
    for l in (1, 2, 4, 5, 7, 10, 11, 12, 15, 17, 21, 25, 28, 31, 32, 33, 47, 64):  # 5, 7, 10, 11, 12, 15, 17, 21, 25, 28, 31 have unrolled instantiations
        for mode in SCALAR:
            compare(eng, oracle, reads, l, 5, 0.02, mode, expect_path=0, tag="lgrid")
    for l in (65, 100, 255):  # beyond the tiled kernel: serial path
        for mode in SCALAR:
            compare(eng, oracle, reads[:8], l, 3, 0.05, mode, expect_path=1, tag="lgrid-long")


def test_k_range(eng, oracle):
    rng = np.random.default_rng(8)
    reads = [rand_read(rng, int(n), hp=0.2) for n in (50000, 20, 9000, 12345, 70000)]
    for k in (1, 2, 10, 31, 64, 65, 66, 100, 300):
        for mode in SCALAR:
            compare(eng, oracle, reads, 31, k, 0.02, mode, tag="krange")


def test_density_extremes_and_pool_retry(eng, oracle):
    rng = np.random.default_rng(9)
    reads = [rand_read(rng, int(n), hp=0.2) for n in (30000, 100, 9216, 18432, 5)]
    for d in (0.0, 1e-6, 0.3, 0.9, 1.0, 2.0):
        for mode in SCALAR:
            compare(eng, oracle, reads, 31, 4, d, mode, tag="density")
    # low-complexity read: every position of a long poly-A-free repeat passes -> far more minimizers
    # than the density predicts, the record pool must grow transparently
    rep = (b"ACGTTGCA" * 6000)
    h = oracle.nthash32_all(rep[:200], 31)
    d = (int(h.min()) + 0.5) / 4294967295.0
    for mode in SCALAR:
        compare(eng, oracle, [rep, rand_read(rng, 20000)], 31, 5, d, mode, tag="low-complexity")


def test_long_homopolymers_exceed_any_fixed_halo(eng, oracle):
    """Runs of 300 .. 30000 equal bases: the l run heads after a tile lie arbitrarily far to the right."""
    rng = np.random.default_rng(10)
    parts = []
    for run in (300, 3000, 9216, 9217, 30000, 700):
        parts.append(rand_read(rng, int(rng.integers(20, 4000)), hp=0.2))
        parts.append(bytes([rng.choice(list(b"ACGT"))]) * run)
    parts.append(rand_read(rng, 5000, hp=0.2))
    one = b"".join(parts)
    reads = [one, b"A" * 50000, rand_read(rng, 3000), b"C" * 9216 + b"G" * 9216 + rand_read(rng, 100), one[::-1]]
    for mode in SCALAR:
        compare(eng, oracle, reads, 31, 5, 0.05, mode, expect_path=0, tag="homopolymers")
        compare(eng, oracle, reads, 7, 2, 0.5, mode, expect_path=0, tag="homopolymers")


def test_odd_bytes(eng, oracle):
    rng = np.random.default_rng(11)
    base = rand_read(rng, 30000, hp=0.2, odd=0.05)
    compare(eng, oracle, [base, base.lower(), base[:100]], 31, 5, 0.05, HM.Regular, expect_path=0, tag="odd")
    compare(eng, oracle, [base, base.lower(), base[:100]], 31, 5, 0.05, HM.Hpc, expect_path=0, tag="odd")
    # bytes >= 0x80 are ordinary "other" bytes (seed 1) in both modes: the tiled kernels take them (round 1 handed an
    # Hpc call with such a byte to the serial kernels)
    hi = bytearray(base)
    for i in range(0, len(hi), 997):
        hi[i] = 0x80 | (hi[i] & 0x7F)
    hi[5000:5040] = bytes([0xC1]) * 40
    reads = [bytes(hi), rand_read(rng, 12000)]
    compare(eng, oracle, reads, 31, 5, 0.05, HM.Regular, expect_path=0, tag="hibit")
    compare(eng, oracle, reads, 31, 5, 0.05, HM.Hpc, expect_path=0, tag="hibit")
    compare(eng, oracle, reads, 31, 5, 0.05, HM.HpcSimd, expect_path=0, tag="hibit")


def test_one_high_byte_does_not_change_the_path(eng, oracle):
    """Round 1 sent a whole Hpc call to the read-serial kernels when a single byte >= 0x80 appeared anywhere in it (the
    staged bytes carried read-start marks in bit 7).  Now such a byte is an ordinary 'other' byte: the call stays on the
    tiled kernels (counts.path == 0) and agrees with the oracle and with the serial kernels."""
    rng = np.random.default_rng(77)
    reads = [rand_read(rng, int(n), hp=0.2) for n in rng.integers(5000, 40000, size=300)]
    dirty = bytearray(reads[137])
    dirty[len(dirty) // 2] = 0xC1
    reads[137] = bytes(dirty)
    for mode in (HM.Hpc, HM.HpcSimd):
        a = compare(eng, oracle, reads, 31, 10, 0.01, mode, expect_path=0, tag="one-high-byte")
        b = compare(eng, oracle, reads, 31, 10, 0.01, mode, force_serial=True, expect_path=1, tag="one-high-byte-serial")
        assert a["n"] == b["n"] and (a["hash"] == b["hash"]).all()


def test_kminmer_kernel_when_its_guess_of_the_record_count_is_wrong(oracle):
    """The k-min-mer kernel fetches a tile's records with its first round trip, as many as the tiles of the context's LAST call held on average
    (+ 2.5 sigma; Desc::spec_n): a call of another shape on the same context meets tiles with more records than were fetched (they fetch again) and
    tiles with far fewer.  Sparse -> dense -> sparse on one context, k-min-mers only (the lane-serial kernel), every tuple against the oracle."""
    rng = np.random.default_rng(606)
    reads = [rand_read(rng, int(n)) for n in rng.integers(4000, 40000, size=60)]  # ~1.3 Mbp: > 64 tiles, the hint is updated
    own = pkg.Engine(0)
    try:
        for mode in (HM.Regular, HM.Hpc):
            for d in (0.002, 0.019, 0.01, 0.03, 0.001):  # ~37, ~350, ~184, ~550, ~18 minimizers per Regular tile
                compare(own, oracle, reads, 31, 10, d, mode, expect_path="desc", minimizers=False, tag="spec_n")
    finally:
        own.close()


def test_every_packing_routine_rebuilds_the_same_stream(oracle, tmp_path):
    """The 2-bit packing of host bases has three implementations (scalar, AVX2 + pext, AVX-512; S2K_PACK_ISA picks one, the default is the best the
    CPU has -- read once per process, hence the children): each must hand the device the caller's bytes exactly, exceptions (N runs, lower case,
    bytes >= 0x80, a zero byte) included, also when a group of 64 / 32 bases straddles them and when the stream does not end on a group."""
    import subprocess
    import sys

    from conftest import ROOT

    rng = np.random.default_rng(177)
    n = 9_000_003  # (> 8 MiB: packed; not a multiple of 64)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    for a in rng.integers(0, n - 100, size=400):
        s[a:a + int(rng.integers(1, 70))] = ord("N")
    odd = rng.integers(0, n, size=3000)
    s[odd] = np.frombuffer(b"nacgtXR*-\xc1\x00", dtype=np.uint8)[rng.integers(0, 11, size=len(odd))]
    lens = []
    while sum(lens) < n:
        lens.append(int(rng.integers(500, 40000)))
    off = np.minimum(np.concatenate([[0], np.cumsum(lens)]), n).astype(np.uint64)
    ref = oracle.batch(s, off, 31, 10, 0.01, 1, threads=4)
    np.save(tmp_path / "s.npy", s)
    np.save(tmp_path / "off.npy", off)
    child = ("import sys, os, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
             "from gpu_util import pkg\n"
             "s = np.load(%r); off = np.load(%r)\n"
             "eng = pkg.Engine(0); got = eng.extract(s, off, 31, 10, 0.01, pkg.HashMode.Hpc)\n"
             "np.savez(sys.argv[1], **{f: got[f] for f in ('hash', 'start', 'end', 'rev', 'km_off')})\n") % (
                 ROOT, ROOT, str(tmp_path / "s.npy"), str(tmp_path / "off.npy"))
    for isa in ("0", "1", "2"):
        out = str(tmp_path / ("isa%s.npz" % isa))
        r = subprocess.run([sys.executable, "-c", child, out], env=dict(os.environ, S2K_PACK_ISA=isa), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (isa, r.stderr[-2000:])
        got = np.load(out)
        assert len(got["hash"]) == ref["n"], isa
        for f in FIELDS:
            assert (got[f] == ref[f]).all(), (isa, f)


def test_packed_host_transfer_is_exact(eng, oracle):
    """s2k_extract sends the bases 2-bit packed over PCIe (copy threads pack while they fill the pinned ring; every byte that
    is not A/C/G/T travels in an exception list; a chunk that is mostly not DNA text goes as it is).  The stream rebuilt on
    the device must be the caller's bytes exactly: results equal the oracle's and the unpacked path's, on a batch that spans
    two 64 MiB chunks and holds N runs, odd bytes, bytes >= 0x80 and a lower-case stretch longer than a packing slice."""
    rng = np.random.default_rng(91)
    n = (64 << 20) + 5_000_000
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    for a in rng.integers(0, n - 100, size=3000):
        s[a:a + int(rng.integers(1, 40))] = ord("N")
    odd = rng.integers(0, n, size=20000)
    s[odd] = np.frombuffer(b"nacgtXR*-\xc1\x00", dtype=np.uint8)[rng.integers(0, 11, size=len(odd))]
    s[(64 << 20) + 100_000:(64 << 20) + 4_600_000] |= 0x20  # lower case: more exceptions than a slice's list holds
    lens = []
    while sum(lens) < n:
        lens.append(int(rng.integers(2000, 60000)))
    off = np.minimum(np.concatenate([[0], np.cumsum(lens)]), n).astype(np.uint64)
    ref = oracle.batch(s, off, 31, 10, 0.01, 0, threads=4)
    for pack2 in (True, False):
        got = eng.extract(s, off, 31, 10, 0.01, HM.Regular, pack2=pack2)
        assert got["counts"]["path"] == 0 and got["n"] == ref["n"], pack2
        for f in FIELDS:
            assert (got[f] == ref[f]).all(), (pack2, f)
    refh = oracle.batch(s[: 20_000_000], off[: int(np.searchsorted(off, 20_000_000))], 31, 10, 0.01, 1, threads=4)
    nr = len(off[: int(np.searchsorted(off, 20_000_000))]) - 1
    goth = eng.extract(s, off[: nr + 1], 31, 10, 0.01, HM.Hpc)
    assert goth["n"] == refh["n"] and (goth["hash"] == refh["hash"]).all() and (goth["end"] == refh["end"]).all()


def test_host_call_is_pipelined_in_sub_batches(eng, oracle):
    """s2k_extract cuts a call into sub-batches of whole reads and overlaps H2D / kernels / D2H of consecutive ones
    (s2k_set_host_batch; default 2^29 bases).  With a tiny sub-batch size a small call runs through dozens of them -- empty
    reads, reads shorter than l, a read longer than a sub-batch, minimizer triples, a result that outgrows the first host
    allocation -- and must return exactly what one sub-batch returns and what the oracle says."""
    rng = np.random.default_rng(2024)
    reads = [rand_read(rng, int(n), hp=0.25, odd=0.01) for n in rng.integers(0, 9000, size=400)]
    reads[7] = b""
    reads[8] = b"ACGT"
    reads[100] = rand_read(rng, 150_000, hp=0.2)  # longer than a sub-batch
    reads.append(b"")
    try:
        for mode in (HM.Hpc, HM.Regular):
            for d in (0.01, 0.5):
                eng.set_host_batch(0)
                one = compare(eng, oracle, reads, 15, 4, d, mode, expect_path=0, tag="one-batch")
                eng.set_host_batch(40_000)
                many = compare(eng, oracle, reads, 15, 4, d, mode, expect_path=0, tag="sub-batches")
                assert one["n"] == many["n"] and one["counts"]["xor_hash"] == many["counts"]["xor_hash"]
                assert (one["km_off"] == many["km_off"]).all()
    finally:
        eng.set_host_batch(0)


def test_simd_result_semantics(eng, oracle):
    """SURVEY.md 8a traps (i)-(vi): strict '<', f32 bound, kept last l-mer, start-of-run end, low-nibble
    seeds, dropped final 16-block when #l-mers % 16 == 0."""
    rng = np.random.default_rng(12)
    reads = [rand_read(rng, int(n), hp=0.3, odd=0.05) for n in (0, 10, 31, 46, 47, 62, 63, 64, 1000, 20000, 31 + 16 * 9 - 1)]
    for mode in (HM.Simd, HM.HpcSimd):
        for l in (4, 9, 16, 31):
            for d in (0.02, 0.33, 1.0):
                compare(eng, oracle, reads, l, 3, d, mode, expect_path=0, tag="simd")  # both on the tiled kernel
                compare(eng, oracle, reads, l, 3, d, mode, force_serial=True, expect_path=1, tag="simd-serial")


def test_simd_tail_rule_across_tiles(eng, oracle):
    """Simd semantics on the tiled kernel: reads whose l-mer count is a multiple of 16 lose their last 16 l-mers
    (src/nthash_avx512_32.rs:134-138); make such reads end before, on and after tile boundaries (9216 bases)."""
    rng = np.random.default_rng(21)
    for l in (12, 31, 17):
        lens = [l - 1 + 16 * int(m) for m in rng.integers(1, 700, size=120)]          # every read triggers the rule (or is < 32 l-mers)
        lens += [int(x) for x in rng.integers(0, 12000, size=60)]                      # ordinary neighbours
        lens += [9216 - (l - 1 + 16 * 40) % 9216, l - 1 + 16 * 40, 9216, l - 1 + 16 * 576]  # ends aligned with tile edges
        order = rng.permutation(len(lens))
        reads = [rand_read(rng, lens[i], odd=0.02) for i in order]
        for d in (0.01, 0.5):
            compare(eng, oracle, reads, l, 4, d, HM.Simd, expect_path=0, tag="simd-tail")
            compare(eng, oracle, reads, l, 4, d, HM.Simd, force_serial=True, expect_path=1, tag="simd-tail-serial")


def test_hpcsimd_tail_rule_across_tiles(eng, oracle):
    """HpcSimd on the tiled kernel: the tail rule applies to the number of RUNS of the whole read (told from tile to tile,
    DESIGN.md 3.1a), the end position is the start of the last run, the last HPC l-mer is kept.  Reads are built from runs so
    that the run count is controlled: R = l - 1 + 16 m triggers the rule; lengths up to several tiles."""
    rng = np.random.default_rng(22)

    def from_runs(nruns, stretch=0.0):
        if nruns == 0:
            return b""
        letters = rng.integers(0, 4, size=nruns)
        letters[1:] = (letters[:-1] + 1 + rng.integers(0, 3, size=nruns - 1)) % 4  # adjacent runs differ
        reps = rng.choice([1, 2, 3, 5, 9], size=nruns, p=[0.6, 0.2, 0.1, 0.07, 0.03])
        if stretch:
            m = rng.random(nruns) < stretch
            reps[m] = rng.integers(40, 400, size=int(m.sum()))
        return np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[letters], reps).tobytes()

    for l in (12, 31, 7):
        nruns = [l - 1 + 16 * int(m) for m in rng.integers(1, 900, size=90)] + [int(x) for x in rng.integers(0, 9000, size=50)]
        nruns += [l - 1, l, l + 1, l + 15, l + 16, l + 30, l + 31, l + 32, l - 1 + 32, l - 1 + 48]
        order = rng.permutation(len(nruns))
        reads = [from_runs(nruns[i], stretch=0.002 if i % 5 == 0 else 0.0) for i in order]
        reads += [b"A" * 500, b"ACGT" * 8, bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.arange(l) % 4])]  # one run; no runs; exactly l bases
        # empty / shorter-than-l reads right after ordinary ones: their (void) windows must not eat the predecessor's last l-mer
        reads += [from_runs(40), b"", from_runs(50), b"A", from_runs(60), b"ACG"[:min(3, l - 1)], from_runs(38), b"", b"", from_runs(33)]
        for d in (0.01, 0.5):
            compare(eng, oracle, reads, l, 4, d, HM.HpcSimd, expect_path=0, tag="hpcsimd-tail")
            compare(eng, oracle, reads, l, 4, d, HM.HpcSimd, force_serial=True, expect_path=1, tag="hpcsimd-tail-serial")


def test_status_codes(eng):
    lib = eng.lib
    b, off = pkg.pack_reads([b"ACGT" * 100])
    for (l, k, mode, want) in ((0, 5, 0, 2), (256, 5, 0, 2), (31, 0, 0, 3), (31, 5000, 0, 3), (32, 5, 2, 2), (31, 5, 7, 1)):
        with pytest.raises(pkg.S2kError) as e:
            eng.extract(b, off, l, k, 0.1, mode)
        assert e.value.status == want, (l, k, mode)
    with pytest.raises(pkg.S2kError) as e:
        eng.extract(b, np.array([0, 300, 200], dtype=np.uint64), 31, 5, 0.1, 0)
    assert e.value.status == 1
    assert eng.extract(b, off, 31, 5, 0.1, 0)["n"] >= 0  # the context stays usable after errors


def test_device_read_table_is_validated_on_the_device(eng, oracle):
    """s2k_extract_device trusts nothing about a read table that lives in HBM: a table that does not start at 0, is not
    non-decreasing or does not end at n_bases comes back as S2K_ERR_INVALID_ARG (not as an out-of-bounds device read),
    synchronously and through s2k_sync; the standalone HPC op checks the same; the context stays usable."""
    import torch

    rng = np.random.default_rng(77)
    reads = [rand_read(rng, 20000, hp=0.2) for _ in range(40)]
    bases, off = pkg.pack_reads(reads)
    dev = torch.device("cuda", 0)
    d_b = torch.from_numpy(bases).to(dev)
    n_reads, n_bases = len(reads), len(bases)
    cap = 40000
    t = {"km_off": torch.zeros(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.zeros(cap, dtype=torch.int64, device=dev),
         "start": torch.zeros(cap, dtype=torch.int32, device=dev), "end": torch.zeros(cap, dtype=torch.int32, device=dev),
         "rev": torch.zeros(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    good = off.astype(np.int64)
    bad_tables = []
    b = good.copy(); b[0] = 5; bad_tables.append(b)                      # does not start at 0
    b = good.copy(); b[7], b[8] = good[8], good[7]; bad_tables.append(b)  # not monotone
    b = good.copy(); b[-1] += 100; bad_tables.append(b)                   # ends past n_bases
    b = good.copy(); b[20:] = 1 << 45; bad_tables.append(b)               # garbage far outside the stream
    b = np.zeros_like(good); bad_tables.append(b)                         # all zero: ends before n_bases
    for mode in (HM.Regular, HM.Hpc, HM.HpcSimd):
        for flags in (0, pkg.FLAG_FORCE_SERIAL):
            for b in bad_tables:
                d_o = torch.from_numpy(b).to(dev)
                torch.cuda.synchronize()
                with pytest.raises(pkg.S2kError) as e:
                    eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, 0.01, int(mode), o, flags=flags)
                assert e.value.status == 1, (int(mode), flags)
                eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, 0.01, int(mode), o, sync=False, flags=flags)
                with pytest.raises(pkg.S2kError) as e:
                    eng.sync()
                assert e.value.status == 1
    # standalone HPC op
    d_o = torch.from_numpy(bad_tables[1]).to(dev)
    d_ho = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with pytest.raises(pkg.S2kError) as e:
        eng.hpc_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, d_ho.data_ptr(), 0, 0, 0)
    assert e.value.status == 1
    # the good table still works on the same context, bit for bit
    d_o = torch.from_numpy(good).to(dev)
    torch.cuda.synchronize()
    c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, 0.01, int(HM.Hpc), o)
    ref = oracle.batch(bases, off, 31, 10, 0.01, OMODE[HM.Hpc])
    assert c["n_kminmers"] == ref["n"] and (t["hash"][: ref["n"]].cpu().numpy().view(np.uint64) == ref["hash"]).all()
    eng.trim()  # s2k_trim: buffers grow again on demand
    c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, 0.01, int(HM.Regular), o)
    assert c["n_kminmers"] == oracle.batch(bases, off, 31, 10, 0.01, OMODE[HM.Regular])["n"]


def test_device_api_capacity_and_async(eng, oracle):
    import torch

    rng = np.random.default_rng(13)
    reads = [rand_read(rng, 10000) for _ in range(50)]
    bases, off = pkg.pack_reads(reads)
    dev = torch.device("cuda", 0)
    d_b = torch.from_numpy(bases).to(dev)
    d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
    ref = oracle.batch(bases, off, 31, 10, 0.01, 0)
    nk = ref["n"]

    def mk(cap):
        t = {"km_off": torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev), "hash": torch.zeros(max(cap, 1), dtype=torch.int64, device=dev),
             "start": torch.zeros(max(cap, 1), dtype=torch.int32, device=dev), "end": torch.zeros(max(cap, 1), dtype=torch.int32, device=dev),
             "rev": torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)}
        o = pkg.DeviceOut()
        o.km_capacity = cap
        o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
        return t, o

    torch.cuda.synchronize()
    t, o = mk(nk // 2)
    with pytest.raises(pkg.S2kError) as e:
        eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), 31, 10, 0.01, 0, o)
    assert e.value.status == 7  # S2K_ERR_CAPACITY
    # entries below the capacity are still correct, nothing was written past it
    assert (t["hash"].cpu().numpy().view(np.uint64)[: nk // 2] == ref["hash"][: nk // 2]).all()
    t, o = mk(nk)
    eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), 31, 10, 0.01, 0, o, sync=False)
    c = eng.sync()
    assert c["n_kminmers"] == nk and c["path"] == 0
    assert (t["hash"].cpu().numpy().view(np.uint64) == ref["hash"]).all()
    assert (t["km_off"].cpu().numpy().view(np.uint64) == ref["km_off"]).all()
    assert (t["start"].cpu().numpy().view(np.uint32) == ref["start"]).all()
    assert (t["rev"].cpu().numpy() == ref["rev"]).all()
    # misaligned base pointers stay on the tiled kernels (the library realigns the stream with one device copy), same answer
    for shift in (1, 7, 15):
        d_b2 = torch.zeros(len(bases) + 16, dtype=torch.uint8, device=dev)
        d_b2[shift:shift + len(bases)] = d_b
        torch.cuda.synchronize()
        for m in (0, 1):
            refm = ref if m == 0 else oracle.batch(bases, off, 31, 10, 0.01, 1)
            t, o = mk(max(refm["n"], 1))
            c = eng.extract_device(d_b2.data_ptr() + shift, d_o.data_ptr(), len(reads), len(bases), 31, 10, 0.01, m, o)
            assert c["path"] == 0 and c["n_kminmers"] == refm["n"]
            assert (t["hash"][:refm["n"]].cpu().numpy().view(np.uint64) == refm["hash"]).all()
            assert (t["end"][:refm["n"]].cpu().numpy().view(np.uint32) == refm["end"]).all()


def test_synth_generator_matches_oracle(eng, oracle):
    import torch

    dev = torch.device("cuda", 0)
    for (seed, first, n) in ((1, 0, 100000), (5, 12345, 70001), (9, 31, 17), (2, 16, 64)):
        d = torch.zeros(n + 32, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.synth_bases_device(seed, first, n, d.data_ptr())
        eng.lib.s2k_sync(eng.ctx, None)
        got = d.cpu().numpy()
        assert (got[:n] == oracle.synth_bases(seed, first, n)).all()
        assert (got[n:] == 0).all()


def test_standalone_hpc(eng, oracle, ecoli):
    """tests/main.rs:76-78 equivalent for the GPU op."""
    s, p = pkg.encode_rle_simd(ecoli, engine=eng)
    rs, rp = oracle.hpc(ecoli, 2)
    assert s == rs and (p == rp).all() and len(s) == 72873
    assert pkg.hpc(b"AACCCGTTTTT", engine=eng) == b"ACGT"
    assert pkg.hpc(ecoli, engine=eng) == oracle.hpc(ecoli, 0)[0]


def test_standalone_hpc_batches(eng, oracle):
    """s2k_hpc_device on ragged batches (segment-parallel path): reads that start inside a homopolymer, empty and
    one-base reads, runs longer than a 4096-byte segment, boundaries on segment edges; also the one-thread-per-read
    path taken for an unaligned base pointer."""
    import torch

    rng = np.random.default_rng(41)
    dev = torch.device("cuda", 0)
    lens = [0, 1, 2, 4096, 4095, 4097, 1, 0, 8192, 30000, 12288 - 5, 5, 0, 70000] + [int(x) for x in rng.integers(0, 9000, size=150)]
    reads = [rand_read(rng, n, hp=0.4) for n in lens]
    reads[9] = b"A" * 30000                      # one run spanning several segments
    reads[13] = reads[13][:100] + b"C" * 9000 + reads[13][9100:]
    reads[20] = b"T" * len(reads[20])            # neighbours of equal letters: read starts are forced run heads
    reads[21] = b"T" * max(len(reads[21]), 3)
    reads[22] = b"T" + reads[22]
    bases, off = pkg.pack_reads(reads)
    exp_s, exp_p, exp_off = [], [], [0]
    for r in reads:
        cs, cp = oracle.hpc(r, 2) if len(r) else (b"", np.zeros(0, dtype=np.uint64))
        exp_s.append(cs)
        exp_p.append(np.asarray(cp, dtype=np.uint32))
        exp_off.append(exp_off[-1] + len(cs))
    exp_s, exp_p = b"".join(exp_s), np.concatenate(exp_p)
    for shift in (0, 1):  # 1: unaligned base pointer -> read-serial kernels
        d_b = torch.zeros(len(bases) + 32, dtype=torch.uint8, device=dev)
        d_b[shift:shift + len(bases)] = torch.from_numpy(bases).to(dev)
        d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
        d_ho = torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev)
        d_h = torch.zeros(len(bases) + 1, dtype=torch.uint8, device=dev)
        d_p = torch.zeros(len(bases) + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        n = eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), len(bases), d_ho.data_ptr(), d_h.data_ptr(), d_p.data_ptr(), len(bases))
        assert n == len(exp_s)
        assert list(d_ho.cpu().numpy()) == exp_off
        assert d_h[:n].cpu().numpy().tobytes() == exp_s
        assert (d_p[:n].cpu().numpy().view(np.uint32) == exp_p).all()
        # sizes only (no output arrays), then a capacity that is too small
        assert eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), len(bases), d_ho.data_ptr(), 0, 0, 0) == n
        d_h.zero_()
        with pytest.raises(pkg.S2kError) as e:
            eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), len(bases), d_ho.data_ptr(), d_h.data_ptr(), d_p.data_ptr(), n // 2)
        assert e.value.status == 7
        assert d_h[:n // 2].cpu().numpy().tobytes() == exp_s[:n // 2] and int(d_h[n // 2:].max().item()) == 0
        # output arrays that start at any byte / word: the compressed bytes leave the kernel as aligned dwords with a byte head and tail
        for o_shift in (1, 2, 3):
            d_h2 = torch.zeros(len(bases) + 8, dtype=torch.uint8, device=dev)
            d_p2 = torch.zeros(len(bases) + 8, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            assert eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), len(bases), d_ho.data_ptr(), d_h2.data_ptr() + o_shift,
                                  d_p2.data_ptr() + 4 * o_shift, len(bases)) == n
            assert d_h2[o_shift:o_shift + n].cpu().numpy().tobytes() == exp_s and int(d_h2[:o_shift].max().item()) == 0
            assert int(d_h2[o_shift + n:].max().item()) == 0
            assert (d_p2[o_shift:o_shift + n].cpu().numpy().view(np.uint32) == exp_p).all()
    # hundreds of reads inside one 4096-byte segment (more read starts than the block has threads), empty ones among them
    lens = [int(x) for x in rng.integers(0, 9, size=6000)] + [5000] + [int(x) for x in rng.integers(0, 4, size=3000)]
    reads = [rand_read(rng, n, hp=0.5) for n in lens]
    bases, off = pkg.pack_reads(reads)
    exp = [oracle.hpc(r, 2) if len(r) else (b"", np.zeros(0, dtype=np.uint64)) for r in reads]
    exp_s = b"".join(e[0] for e in exp)
    exp_p = np.concatenate([np.asarray(e[1], dtype=np.uint32) for e in exp])
    exp_off = np.concatenate([[0], np.cumsum([len(e[0]) for e in exp])])
    d_b = torch.zeros(len(bases) + 32, dtype=torch.uint8, device=dev)
    d_b[:len(bases)] = torch.from_numpy(bases).to(dev)
    d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_ho = torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev)
    d_h = torch.zeros(len(bases) + 1, dtype=torch.uint8, device=dev)
    d_p = torch.zeros(len(bases) + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    n = eng.hpc_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), d_ho.data_ptr(), d_h.data_ptr(), d_p.data_ptr(), len(bases))
    assert n == len(exp_s) and (d_ho.cpu().numpy() == exp_off).all()
    assert d_h[:n].cpu().numpy().tobytes() == exp_s and (d_p[:n].cpu().numpy().view(np.uint32) == exp_p).all()


def test_encode_rle_flavour(eng, oracle, ecoli):
    """encode_rle (src/hpc.rs:7-25): only runs of ACTGactgNn collapse (src/hpc.rs:14), every other repeated character stays;
    hpc / encode_rle_simd collapse any repeated byte.  GPU ops vs the oracle's restatement of the scalar functions
    (which = 1 encode_rle, 0 hpc, 2 encode_rle_simd); the empty-string quirk (both scalar functions flush their '#'
    sentinel, src/hpc.rs:22-24,39-40) lives in the facades; input that contains '#' is refused."""
    import torch

    rng = np.random.default_rng(43)
    s, p = pkg.encode_rle(ecoli, engine=eng)  # tests/main.rs:76: encode_rle(s).0 == hpc(s) on the ACGT-only fixture
    rs, rp = oracle.hpc(ecoli, 1)
    assert s == rs and (p == rp).all() and len(s) == 72873 and pkg.hpc(ecoli, engine=eng) == s
    odd = bytearray(rand_read(rng, 40000, hp=0.4, odd=0.0))
    for i in rng.integers(0, len(odd) - 8, size=600):  # runs of characters outside the alphabet, and of n / lowercase
        c = int(rng.choice(np.frombuffer(b"XX**--RRnnNNaacc..\x00\x00\xc1\xc1", dtype=np.uint8)))
        odd[i:i + int(rng.integers(1, 7))] = bytes([c]) * int(rng.integers(1, 7))
    odd = bytes(odd)
    for which, fn in ((1, pkg.encode_rle), (2, pkg.encode_rle_simd)):
        gs, gp = fn(odd, engine=eng)
        es, ep = oracle.hpc(odd, which)
        assert gs == es and (gp == ep).all(), which
    assert pkg.hpc(odd, engine=eng) == oracle.hpc(odd, 0)[0]
    assert len(pkg.encode_rle(odd, engine=eng)[0]) > len(pkg.encode_rle_simd(odd, engine=eng)[0])  # "XXXX" stays in one, collapses in the other
    assert pkg.encode_rle(b"AAXXXaaNNnn--A", engine=eng)[0] == b"AXXXaNn--A"
    # the sentinel quirks
    assert pkg.hpc(b"", engine=eng) == oracle.hpc(b"", 0)[0] == b"#"
    es, ep = pkg.encode_rle(b"", engine=eng)
    assert es == oracle.hpc(b"", 1)[0] == b"#" and list(ep) == [0]
    for fn in (pkg.hpc, pkg.encode_rle):
        with pytest.raises(ValueError):
            fn(b"AC#GT", engine=eng)
    # batches through s2k_hpc_device_ex: aligned (segment-parallel) and unaligned (read-serial) base pointers
    dev = torch.device("cuda", 0)
    lens = [0, 1, 5, 4096, 4097, 9000, 0, 30000] + [int(x) for x in rng.integers(0, 6000, size=80)]
    reads = [rand_read(rng, n, hp=0.4, odd=0.1) for n in lens]
    reads[3] = b"X" * 4096                      # never collapses
    reads[5] = b"n" * 9000                      # collapses to one character
    reads[7] = b"A" * 10000 + b"-" * 10000 + b"N" * 10000
    bases, off = pkg.pack_reads(reads)
    exp = [oracle.hpc(r, 1) if len(r) else (b"", np.zeros(0, dtype=np.uint64)) for r in reads]
    exp_s = b"".join(e[0] for e in exp)
    exp_p = np.concatenate([np.asarray(e[1], dtype=np.uint32) for e in exp])
    exp_off = np.concatenate([[0], np.cumsum([len(e[0]) for e in exp])])
    for shift in (0, 1):
        d_b = torch.zeros(len(bases) + 32, dtype=torch.uint8, device=dev)
        d_b[shift:shift + len(bases)] = torch.from_numpy(bases).to(dev)
        d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
        d_ho = torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev)
        d_h = torch.zeros(len(bases) + 1, dtype=torch.uint8, device=dev)
        d_p = torch.zeros(len(bases) + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        n = eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), len(bases), d_ho.data_ptr(), d_h.data_ptr(), d_p.data_ptr(),
                           len(bases), rle=True)
        assert n == len(exp_s) and (d_ho.cpu().numpy() == exp_off).all()
        assert d_h[:n].cpu().numpy().tobytes() == exp_s and (d_p[:n].cpu().numpy().view(np.uint32) == exp_p).all()


def test_synthetic_config2_sample_properties(eng, oracle):
    """BASELINE config 2 shape (10 kbp uniform reads, l=31 k=10 d=0.01) at a size the oracle finishes in
    seconds: full comparison, and tiled == serial."""
    n_reads, L = 400, 10000
    bases = oracle.synth_bases(1, 0, n_reads * L)
    off = (np.arange(n_reads + 1, dtype=np.uint64) * L)
    for mode in SCALAR:
        a = eng.extract(bases, off, 31, 10, 0.01, mode)
        b = eng.extract(bases, off, 31, 10, 0.01, mode, force_serial=True)
        ref = oracle.batch(bases, off, 31, 10, 0.01, OMODE[mode], threads=8)
        for f in ("km_off", "hash", "start", "end", "rev"):
            assert (a[f] == ref[f]).all() and (b[f] == ref[f]).all(), (mode, f)
        assert a["counts"]["path"] == 0 and b["counts"]["path"] == 1
        # properties: sorted starts within a read, start <= end, offsets are dense
        for r in range(0, n_reads, 37):
            s, e = int(a["km_off"][r]), int(a["km_off"][r + 1])
            assert (np.diff(a["start"][s:e].astype(np.int64)) > 0).all()
            assert (a["end"][s:e] > a["start"][s:e]).all() and int(a["end"][s:e].max(initial=0)) < L


def _xor_reduce(t):
    import torch

    while t.numel() > 1:
        h = t.numel() // 2
        r = t[:h] ^ t[h:2 * h]
        t = torch.cat((r, t[2 * h:])) if t.numel() & 1 else r
    return int(t.item()) & 0xFFFFFFFFFFFFFFFF if t.numel() else 0


def test_full_size_config2_whole_run_checksums(eng, oracle):
    """BASELINE config 2 at its full size -- 1 000 000 x 10 kbp = 10 Gbp, l=31 k=10 d=0.01, all four HashModes (the two
    scalar ones and the result semantics of Simd / HpcSimd): every k-min-mer the GPU wrote is folded into count /
    XOR(hash) / SUM(start) / SUM(end) / #rev and compared with the oracle run over the same 10 Gbp on the host cores
    (reads generated on the fly on both sides)."""
    import os
    import torch

    n_reads, L, seed = 1_000_000, 10_000, 1
    dev = torch.device("cuda", 0)
    d_b = torch.empty(n_reads * L + 64, dtype=torch.uint8, device=dev)
    d_o = (torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L)
    torch.cuda.synchronize()
    eng.synth_bases_device(seed, 0, n_reads * L, d_b.data_ptr())
    cap = int(n_reads * L * 0.0215)
    t = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
         "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
         "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    threads = max(1, min(os.cpu_count() or 1, 64))
    for mode in (HM.Regular, HM.Hpc, HM.Simd, HM.HpcSimd):
        torch.cuda.synchronize()
        c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, 31, 10, 0.01, int(mode), o)
        ref = oracle.synth_checksums(seed, n_reads, L, 31, 10, 0.01, OMODE[mode], threads=threads)
        n = c["n_kminmers"]
        assert c["path"] == 0 and c["n_bases"] == n_reads * L
        assert (n, c["n_minimizers"], c["xor_hash"]) == (ref["n_kminmers"], ref["n_minimizers"], ref["xor_hash"]), (int(mode), c, ref)
        assert int(t["km_off"][-1].item()) == n and bool((t["km_off"][1:] >= t["km_off"][:-1]).all())
        assert _xor_reduce(t["hash"][:n]) == ref["xor_hash"]
        assert int(t["start"][:n].to(torch.int64).sum().item()) == ref["sum_start"]
        assert int(t["end"][:n].to(torch.int64).sum().item()) == ref["sum_end"]
        assert int(t["rev"][:n].to(torch.int64).sum().item()) == ref["n_rev"]
        # order-sensitive: position-weighted folds over the global item index, and over km_off
        assert folds_torch(t, n, n_reads) == {x: ref[x] for x in FOLD_FIELDS}, int(mode)
        # >= 1 % of the reads element by element, drawn across the whole stream (the last tile and tiles
        # beyond the 3 x 3072 statically dealt ones included)
        ids = spread_sample(n_reads, 0.0105, np.random.default_rng(1000 + int(mode)), must=(921, 922, 9215, 9216, n_reads // 2))
        assert len(ids) >= n_reads // 100
        assert sample_reads_compare(t, n, oracle, seed, 0, lambda r: (r * L, L), ids, 31, 10, 0.01, OMODE[mode], threads=threads) > 0


def test_mid_size_every_kernel_instantiation_whole_run_checksums(eng, oracle):
    """200 Mbp (21 700 tiles: more than the 3 x 3072 a launch deals statically, so the cursors of the dynamic deal are in play)
    through every instantiation of the tiled kernel -- the unrolled l = 5, 7, 10, 11, 12, 15, 17, 21, 25, 28, 31 and the run-time-l one (l = 23, 64) --
    in all four modes, whole-run checksums against the oracle."""
    import os
    import torch

    n_reads, L, seed = 20_000, 10_000, 7
    dev = torch.device("cuda", 0)
    d_b = torch.empty(n_reads * L + 64, dtype=torch.uint8, device=dev)
    d_o = (torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L)
    torch.cuda.synchronize()
    eng.synth_bases_device(seed, 0, n_reads * L, d_b.data_ptr())
    cap = int(n_reads * L * 0.11)
    t = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
         "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
         "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    threads = max(1, min(os.cpu_count() or 1, 64))
    for l, d in ((5, 0.02), (7, 0.03), (10, 0.03), (11, 0.02), (12, 0.05), (15, 0.01), (17, 0.02), (21, 0.02), (23, 0.01), (25, 0.01), (28, 0.02), (31, 0.003), (64, 0.01)):
        for mode in (HM.Regular, HM.Hpc, HM.Simd, HM.HpcSimd):
            if l > 31 and mode in (HM.Simd, HM.HpcSimd):
                continue  # the reference's SIMD iterators stop at l = 31
            torch.cuda.synchronize()
            c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, l, 6, d, int(mode), o)
            ref = oracle.synth_checksums(seed, n_reads, L, l, 6, d, OMODE[mode], threads=threads)
            n = c["n_kminmers"]
            assert c["path"] == 0
            assert (n, c["n_minimizers"], c["xor_hash"]) == (ref["n_kminmers"], ref["n_minimizers"], ref["xor_hash"]), (l, int(mode), c, ref)
            assert int(t["start"][:n].to(torch.int64).sum().item()) == ref["sum_start"], (l, int(mode))
            assert int(t["end"][:n].to(torch.int64).sum().item()) == ref["sum_end"], (l, int(mode))
            assert int(t["rev"][:n].to(torch.int64).sum().item()) == ref["n_rev"], (l, int(mode))
            assert folds_torch(t, n, n_reads) == {x: ref[x] for x in FOLD_FIELDS}, (l, int(mode))


def test_full_size_config3_shard_whole_run_checksums(eng, oracle):
    """BASELINE config 3 at the size ONE GPU holds when 8 share the 200 Gbp: ~25 Gbp of ONT-like reads (lengths
    lognormal, mean 20 kbp, sigma 0.5, clipped to [1 k, 200 k]), l=31 k=10 d=0.01, both scalar modes; every k-min-mer
    folded into count / XOR / SUM(start) / SUM(end) / #rev and compared with the oracle run on the host cores."""
    import os
    import torch

    rng = np.random.default_rng(303)
    lens = _lognormal_lengths(rng, 1_250_000, 20000, 0.5, 1000, 200000)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    n_reads, n_bases = len(lens), int(off[-1])
    dev = torch.device("cuda", 0)
    d_b = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
    d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    eng.synth_bases_device(3, 0, n_bases, d_b.data_ptr())
    cap = int(n_bases * 0.0215)
    t = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
         "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
         "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    threads = max(1, min(os.cpu_count() or 1, 64))
    for mode in SCALAR:
        torch.cuda.synchronize()
        c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, 0.01, int(mode), o)
        ref = oracle.synth_checksums_off(3, off, 31, 10, 0.01, OMODE[mode], threads=threads)
        n = c["n_kminmers"]
        assert c["path"] == 0 and c["n_bases"] == n_bases and n_bases > 24_000_000_000
        assert (n, c["n_minimizers"], c["xor_hash"]) == (ref["n_kminmers"], ref["n_minimizers"], ref["xor_hash"]), (int(mode), c, ref)
        assert _xor_reduce(t["hash"][:n]) == ref["xor_hash"]
        assert int(t["start"][:n].to(torch.int64).sum().item()) == ref["sum_start"]
        assert int(t["end"][:n].to(torch.int64).sum().item()) == ref["sum_end"]
        assert int(t["rev"][:n].to(torch.int64).sum().item()) == ref["n_rev"]
        assert folds_torch(t, n, n_reads) == {x: ref[x] for x in FOLD_FIELDS}, int(mode)
        ids = spread_sample(n_reads, 0.0105, np.random.default_rng(3000 + int(mode)))
        assert len(ids) >= n_reads // 100
        assert sample_reads_compare(t, n, oracle, 3, 0, lambda r: (int(off[r]), int(lens[r])), ids, 31, 10, 0.01, OMODE[mode],
                                    threads=threads) > 0
    del d_b, t
    torch.cuda.empty_cache()


def _lognormal_lengths(rng, n, mean, sigma, lo, hi):
    mu = np.log(mean) - sigma * sigma / 2
    return np.clip(rng.lognormal(mu, sigma, size=n), lo, hi).astype(np.int64)


def test_baseline_config3_ont_like_lengths(eng, oracle):
    """BASELINE config 3 shape at oracle-checkable size: ONT-like reads, lengths lognormal(mean 20 kbp, sigma 0.5)
    clipped to [1 k, 200 k]; tiles are cut independently of the ragged read lengths."""
    rng = np.random.default_rng(31)
    lens = _lognormal_lengths(rng, 160, 20000, 0.5, 1000, 200000)
    reads = [rand_read(rng, int(n), hp=0.15) for n in lens]
    for mode in SCALAR:
        compare(eng, oracle, reads, 31, 10, 0.01, mode, expect_path="desc", tag="C3-ont")


def test_baseline_config4_hifi_like_hpc_backmap(eng, oracle):
    """BASELINE config 4: HiFi-like reads N(15 k, 2 k), homopolymer-enriched (geometric run lengths, 0.1 % of runs
    stretched to 20..3000 bases so that l+1 run starts overflow any fixed look-ahead), Hpc mode, every start/end
    checked against the oracle."""
    rng = np.random.default_rng(32)
    reads = []
    for _ in range(1500):  # 22 Mbp, every tuple compared
        n = int(max(2000, rng.normal(15000, 2000)))
        runs = rng.geometric(0.5, size=n)  # mean run length 2
        stretch = rng.random(n) < 0.001
        runs[stretch] = rng.integers(20, 3000, size=int(stretch.sum()))
        letters = rng.integers(0, 4, size=n)
        letters[1:] = (letters[:-1] + 1 + rng.integers(0, 3, size=n - 1)) % 4  # consecutive runs differ... mostly
        s = np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[letters], runs)[:n]
        reads.append(s.tobytes())
    got = compare(eng, oracle, reads, 31, 10, 0.01, HM.Hpc, expect_path="desc", tag="C4-hifi")
    assert got["n"] > 0 and int((got["end"] - got["start"]).max()) > 500  # long runs really widen the spans
    compare(eng, oracle, reads, 31, 10, 0.01, HM.Regular, expect_path="desc", tag="C4-hifi-regular")


def test_full_size_config4_hifi_like_whole_run_checksums(eng, oracle):
    """BASELINE configs[3] AT SIZE and device-resident: 1 000 000 HiFi-like reads (~15 Gbp; lengths
    ~N(15 k, 2 k), homopolymer runs of geometric length with mean 2, ~0.1 % of the runs 20..2999 bases long, generated in HBM
    by s2k_synth_hifi_device -- the oracle's twin, tests/test_hifi_generator.py), l=31 k=10 d=0.01, the two modes that compress
    homopolymers (Hpc, HpcSimd) and Regular: whole-run counts / XOR / sums / order-sensitive folds against the oracle on the host
    cores, and >= 1 % of the reads field by field (start / end are the original-space back-map of src/nthash_hpc.rs:280-281)."""
    import os
    import torch

    n_reads, seed = 1_000_000, 3
    lens = eng.synth_hifi_lengths(seed, 0, n_reads)
    assert np.array_equal(lens, oracle.hifi_lengths(seed, 0, n_reads))
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    n_bases = int(off[-1])
    assert n_bases > 14_900_000_000
    dev = torch.device("cuda", 0)
    d_b = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
    d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    eng.synth_hifi_device(seed, 0, n_reads, d_o.data_ptr(), d_b.data_ptr())
    cap = int(n_bases * 0.0135)
    t = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
         "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
         "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    threads = max(1, min(os.cpu_count() or 1, 64))
    for mode in (HM.Hpc, HM.HpcSimd, HM.Regular):
        torch.cuda.synchronize()
        c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, 0.01, int(mode), o)
        ref = oracle.hifi_checksums(seed, off, 31, 10, 0.01, OMODE[mode], threads=threads)
        n = c["n_kminmers"]
        assert c["path"] == 0 and c["n_bases"] == n_bases
        assert (n, c["n_minimizers"], c["xor_hash"]) == (ref["n_kminmers"], ref["n_minimizers"], ref["xor_hash"]), (int(mode), c, ref)
        assert int(t["km_off"][-1].item()) == n and bool((t["km_off"][1:] >= t["km_off"][:-1]).all())
        assert _xor_reduce(t["hash"][:n]) == ref["xor_hash"]
        assert int(t["start"][:n].to(torch.int64).sum().item()) == ref["sum_start"]
        assert int(t["end"][:n].to(torch.int64).sum().item()) == ref["sum_end"]
        assert int(t["rev"][:n].to(torch.int64).sum().item()) == ref["n_rev"]
        assert folds_torch(t, n, n_reads) == {x: ref[x] for x in FOLD_FIELDS}, int(mode)
        ids = spread_sample(n_reads, 0.0105, np.random.default_rng(4000 + int(mode)))
        assert len(ids) >= n_reads // 100
        assert sample_reads_compare(t, n, oracle, seed, 0, lambda r: (int(off[r]), int(lens[r])), ids, 31, 10, 0.01, OMODE[mode],
                                    threads=threads, gen=lambda r, ln: oracle.hifi_read(seed, r, ln)) > 0
        if mode == HM.Hpc:  # the long runs really widen the spans the back-map has to cover
            assert int((t["end"][:n] - t["start"][:n]).max().item()) > 2000
    del d_b, t
    torch.cuda.empty_cache()


def test_baseline_config5_mbp_contigs_sparse_density(eng, oracle):
    """BASELINE config 5: chromosome-scale 1 Mbp contigs, d = 0.001 (bound 4 294 967): one read spans >100 tiles,
    most waves emit almost nothing, k-min-mer windows span ~5 kbp = several tiles."""
    rng = np.random.default_rng(33)
    reads = [rand_read(rng, 1_000_000, hp=0.2) for _ in range(6)] + [rand_read(rng, 1_000_000)]
    for mode in SCALAR:
        got = compare(eng, oracle, reads, 31, 10, 0.001, mode, expect_path="desc", tag="C5-contigs")
        assert got["counts"]["hash_bound"] == 4294967


def test_full_size_config5_whole_run_checksums(eng, oracle):
    """BASELINE config 5 at its full size: 1000 contigs of 1 Mbp, d = 0.001, both scalar modes, whole-run checksums."""
    import os
    import torch

    n_reads, L, seed = 1000, 1_000_000, 5
    dev = torch.device("cuda", 0)
    d_b = torch.empty(n_reads * L + 64, dtype=torch.uint8, device=dev)
    d_o = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    eng.synth_bases_device(seed, 0, n_reads * L, d_b.data_ptr())
    cap = int(n_reads * L * 0.003) + 4096
    t = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
         "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
         "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    threads = max(1, min(os.cpu_count() or 1, 64))
    for mode in SCALAR:
        torch.cuda.synchronize()
        c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, 31, 10, 0.001, int(mode), o)
        ref = oracle.synth_checksums(seed, n_reads, L, 31, 10, 0.001, OMODE[mode], threads=threads)
        n = c["n_kminmers"]
        assert c["path"] == 0 and c["hash_bound"] == 4294967
        assert (n, c["n_minimizers"], c["xor_hash"]) == (ref["n_kminmers"], ref["n_minimizers"], ref["xor_hash"]), (int(mode), c, ref)
        assert _xor_reduce(t["hash"][:n]) == ref["xor_hash"]
        assert int(t["start"][:n].to(torch.int64).sum().item()) == ref["sum_start"]
        assert int(t["end"][:n].to(torch.int64).sum().item()) == ref["sum_end"]
        assert int(t["rev"][:n].to(torch.int64).sum().item()) == ref["n_rev"]
        assert folds_torch(t, n, n_reads) == {x: ref[x] for x in FOLD_FIELDS}, int(mode)
        ids = spread_sample(n_reads, 0.02, np.random.default_rng(5000 + int(mode)))
        assert sample_reads_compare(t, n, oracle, seed, 0, lambda r: (r * L, L), ids, 31, 10, 0.001, OMODE[mode], threads=threads) > 0


def test_descriptor_and_legacy_paths_agree(eng, oracle):
    """The default path (8-byte tile-relative records, tile words, scan, k-min-mer kernel without per-read tables; counts.path 0)
    and the legacy one (16-byte records with the read index, per-read scans; path 2, S2K_FLAG_LEGACY_PATH) against the oracle and
    against each other, with minimizer triples, on inputs that make the descriptor path work: reads that span many tiles, reads
    that start exactly at a tile boundary, tiles without a minimizer, empty reads, k = 1 and k = 32."""
    rng = np.random.default_rng(4242)
    T = 9216
    reads = [rand_read(rng, 3 * T, hp=0.1), rand_read(rng, T - 7), rand_read(rng, 7), b"", rand_read(rng, 2 * T), b"", b"",
             rand_read(rng, 5 * T + 1, hp=0.3), b"A" * 20000, rand_read(rng, 40), rand_read(rng, 31), rand_read(rng, 32), rand_read(rng, 4 * T - 72)]
    reads += [rand_read(rng, int(n), hp=0.15) for n in rng.integers(2000, 60000, size=60)]
    for mode in (HM.Regular, HM.Hpc, HM.Simd, HM.HpcSimd):
        for (l, k, d) in ((31, 10, 0.01), (31, 1, 0.02), (21, 32, 0.05), (12, 5, 0.001), (31, 10, 0.0002)):
            a = compare(eng, oracle, reads, l, k, d, mode, expect_path="desc", tag="desc-path")
            bases, off = pkg.pack_reads(reads)
            b = eng.extract(bases, off, l, k, d, mode, want_minimizers=True, legacy=True)
            assert b["counts"]["path"] == 2
            for f in FIELDS + ("mn_off", "mn_j", "mn_jend", "mn_hash"):
                assert (a[f] == b[f]).all(), (int(mode), l, k, d, f)


def test_descriptor_path_falls_back_when_it_must(eng, oracle):
    """What the descriptor path does not handle sends the whole call to the legacy path, transparently (counts.path 2): a tile with
    more than 30 read starts (reads shorter than ~300 bases), k > 32, and an l-mer whose span does not fit 18 bits (a homopolymer of
    more than 262 143 bases inside one Hpc l-mer)."""
    rng = np.random.default_rng(99)
    short = [rand_read(rng, int(n), hp=0.1) for n in rng.integers(40, 260, size=4000)]
    compare(eng, oracle, short, 15, 3, 0.1, HM.Hpc, expect_path="legacy", tag="short-reads")
    compare(eng, oracle, short, 15, 3, 0.1, HM.Regular, expect_path="legacy", tag="short-reads")
    longish = [rand_read(rng, 30000, hp=0.1) for _ in range(8)]
    compare(eng, oracle, longish, 21, 33, 0.05, HM.Hpc, expect_path="legacy", tag="k33")
    compare(eng, oracle, longish, 21, 32, 0.05, HM.Hpc, expect_path="desc", tag="k32")
    giant = rand_read(rng, 5000) + b"G" * 300000 + rand_read(rng, 5000)
    got = compare(eng, oracle, [giant, rand_read(rng, 20000)], 12, 3, 0.5, HM.Hpc, expect_path="legacy", tag="giant-homopolymer")
    assert int((got["end"].astype(np.int64) - got["start"].astype(np.int64)).max()) > 300000
    compare(eng, oracle, [giant, rand_read(rng, 20000)], 12, 3, 0.5, HM.Regular, expect_path="desc", tag="giant-homopolymer-regular")


def test_chunked_pipeline_equals_single_launch(oracle):
    """The descriptor path cuts a call into chunks of tiles (minimizer kernel of chunk c+1 beside the scan + k-min-mer kernel of chunk
    c on a second stream).  S2K_DESC_CHUNKS = 1 (no second stream), 3 and 13 must give the same bytes as the default on a batch
    whose reads straddle every chunk boundary; run in child processes because the variable is read when a context is created."""
    import subprocess
    import sys

    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from s2k_loader import import_package
pkg = import_package()
eng = pkg.Engine(0)
rng = np.random.default_rng(7)
lens = rng.integers(3000, 90000, size=12000)
off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(off[-1]))]
bases = np.repeat(bases, rng.choice([1, 1, 1, 2, 5], size=len(bases)))[: int(off[-1])].copy()
h = hashlib.sha256()
for mode in (0, 1):
    r = eng.extract(bases, off, 31, 10, 0.01, mode, want_minimizers=True)
    assert r["counts"]["path"] == 0
    for f in ("km_off", "hash", "start", "end", "rev", "mn_off", "mn_j", "mn_jend", "mn_hash"):
        h.update(r[f].tobytes())
    print(mode, r["n"], r["n_minimizers"])
# low-complexity sequence: tiles with far more minimizers than their slab holds take space from the overflow region, which the chunks
# share (one cursor for the whole call: the k-min-mer kernel of a chunk reads its records while the next chunk's are written)
lens = rng.integers(20000, 400000, size=12)
off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(off[-1]))]
bases = np.repeat(bases, rng.choice([1, 1, 3, 60, 700], size=len(bases)))[: int(off[-1])].copy()
for mode, k, d in ((0, 1, 0.1), (0, 10, 0.05), (1, 3, 0.5)):
    r = eng.extract(bases, off, 15, k, d, mode, want_minimizers=True)
    assert r["counts"]["path"] == 0
    for f in ("km_off", "hash", "start", "end", "rev", "mn_off", "mn_j", "mn_jend", "mn_hash"):
        h.update(r[f].tobytes())
    print("lowc", mode, k, r["n"], r["n_minimizers"])
print(h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for chunks in ("", "1", "3", "13"):
        env = dict(os.environ)
        env.pop("S2K_DESC_CHUNKS", None)
        if chunks:
            env["S2K_DESC_CHUNKS"] = chunks
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[chunks] = p.stdout.strip().splitlines()
    assert outs[""] == outs["1"] == outs["3"] == outs["13"], outs
    assert int(outs[""][0].split()[1]) > 1_000_000  # ~500 Mbp: 54 000 tiles, several chunks by default


def test_hpcsimd_run_counts_by_lookback(oracle):
    """HpcSimd on the tiled kernel without a second pass over the bases: the tail rule needs the run count of the whole read, and
    the tiles tell each other (a word per tile, published after its compaction; the tile in which a read ends looks back), also
    across the chunks of the two-stream pipeline.  (a) reads whose run count triggers the rule (R = l - 1 + 16 m), several tiles
    long, across every chunk boundary; (b) reads whose last 40 run heads span 25 kbp across a chunk boundary (the end of the read is
    within the rule's reach of a tile but 2.7 tiles away in bytes); (c) = (a) with S2K_FULL_RUNS=1: the fall-back that counts
    the runs of every read in a pre-pass (taken when a look-back gives up).  No case may need a re-run."""
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from s2k_loader import import_package
from gpu_util import compare
from oracle import s2k_oracle
pkg = import_package()
eng = pkg.Engine(0)
oracle = s2k_oracle.get()
rng = np.random.default_rng(int(sys.argv[1]))
l = 31
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
def from_runs(nruns, reps=None):
    letters = rng.integers(0, 4, size=nruns)
    letters[1:] = (letters[:-1] + 1 + rng.integers(0, 3, size=nruns - 1)) %% 4
    if reps is None:
        reps = rng.choice([1, 2, 3, 5, 9], size=nruns, p=[0.6, 0.2, 0.1, 0.07, 0.03])
    return np.repeat(ACGT[letters], reps).tobytes()
TILE = 9216
reads = []
if sys.argv[2] == "a":
    total = 0
    while total < 14 * 64 * TILE:  # > 13 chunks of 64 tiles
        m = int(rng.integers(200, 2500))
        r = from_runs(l - 1 + 16 * m if rng.random() < 0.7 else int(rng.integers(40, 30000)))
        reads.append(r); total += len(r)
else:
    # chunk boundaries of a forced 13-chunk call over exactly 13 * 64 tiles: tile 64 c
    total_tiles = 13 * 64
    pos = 0
    for c in range(1, 13):
        b = 64 * c * TILE
        fill = from_runs(l - 1 + 16 * 40)
        while pos + len(fill) + 30000 < b - 400:  # ordinary rule-triggering reads up to just before the boundary
            reads.append(fill); pos += len(fill); fill = from_runs(l - 1 + 16 * int(rng.integers(30, 300)))
        # one read: R = l - 1 + 16 m runs; its last 40 runs start ~100 bytes before the boundary, the first of them 25 kbp long
        m = 4
        R = l - 1 + 16 * m
        reps = rng.choice([1, 2, 3], size=R)
        head_len = int(reps[: R - 40].sum())
        start = b - 100 - head_len
        pad = start - pos
        assert pad > l + 2
        reads.append(from_runs(max(pad // 2, 40), reps=None)[:pad]); pos += len(reads[-1])
        if len(reads[-1]) < pad:
            reads.append(b"A" * (pad - len(reads[-1]))); pos += len(reads[-1])
        reps[R - 40] = 25000
        r = from_runs(R, reps=reps)
        reads.append(r); pos += len(r)
    tail = total_tiles * TILE - pos
    assert tail > 0
    while tail > 0:
        r = from_runs(l - 1 + 16 * 20)[:tail]
        reads.append(r); tail -= len(r)
for d in (0.5, 0.02):
    compare(eng, oracle, reads, l, 4, d, pkg.HashMode.HpcSimd, expect_path=0, tag="hpcsimd-chunked-" + sys.argv[2])
print("ok", len(reads), sum(map(len, reads)))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for case, seed, full in (("a", "5", False), ("b", "6", False), ("a", "5", True)):
        env = dict(os.environ)
        env["S2K_DESC_CHUNKS"] = "13"
        env["S2K_TRACE"] = "1"
        env.pop("S2K_FULL_RUNS", None)
        if full:
            env["S2K_FULL_RUNS"] = "1"
        p = subprocess.run([sys.executable, "-c", code, seed, case], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, (case, full, p.stdout[-1000:], p.stderr[-3000:])
        assert p.stdout.strip().startswith("ok"), p.stdout
        assert "re-run" not in p.stderr, (case, full, p.stderr[-2000:])


def test_hpcsimd_lookback_two_processes_one_gpu():
    """Two processes share the GPU while both run HpcSimd: neither minimizer kernel has all its waves resident, so a tile can wait
    for the word of a tile whose wave has not started -- the look-back must give up after a bounded wait and the call must be run
    again with the runs counted first, not hang and not answer wrongly.  Each process compares the look-back engine with one
    created under S2K_FULL_RUNS=1 (pre-pass) on the same input, byte for byte, several calls in a row."""
    import subprocess
    import sys

    code = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from s2k_loader import import_package
pkg = import_package()
rng = np.random.default_rng(int(sys.argv[1]))
lens = rng.integers(2000, 40000, size=14000)
off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(off[-1]))]
bases = np.repeat(bases, rng.choice([1, 1, 1, 2, 4], size=len(bases)))[: int(off[-1])].copy()
eng = pkg.Engine(0)
os.environ["S2K_FULL_RUNS"] = "1"
eng_pre = pkg.Engine(0)
del os.environ["S2K_FULL_RUNS"]
ref = eng_pre.extract(bases, off, 31, 10, 0.02, pkg.HashMode.HpcSimd, want_minimizers=True)
for it in range(6):
    got = eng.extract(bases, off, 31, 10, 0.02, pkg.HashMode.HpcSimd, want_minimizers=True)
    assert got["n"] == ref["n"] and got["n_minimizers"] == ref["n_minimizers"], (it, got["n"], ref["n"])
    for f in ("km_off", "hash", "start", "end", "rev", "mn_off", "mn_j", "mn_jend", "mn_hash"):
        assert (got[f] == ref[f]).all(), (it, f)
print("ok", ref["n"])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["S2K_TRACE"] = "1"
    env.pop("S2K_FULL_RUNS", None)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for seed in (41, 42)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-500:], se[-3000:])
        assert so.strip().startswith("ok"), so
    print("re-runs seen:", sum(se.count("re-run") for _, se in outs))


def test_minimizer_iterator_facades(eng, oracle, ecoli):
    """The crate's minimizer iterators (re-exported at src/lib.rs:6-13) as Python classes over S2K_FLAG_WANT_MINIMIZERS, against
    the oracle's minimizers in all four modes: NtHashHPCIterator yields (start, end, hash) (src/nthash_hpc.rs:193),
    NtHashSIMDIterator (pos, hash) (src/nthash_avx512_32.rs), NtHashHPCSIMDIterator (start, end, hash) (src/nthash_hpc_simd.rs:56-66);
    they take the u32 hash bound like the reference."""
    from oracle import s2k_oracle as so

    rng = np.random.default_rng(2024)
    seqs = [ecoli[:30000], rand_read(rng, 12345, hp=0.3, odd=0.01), rand_read(rng, 48 + 31, hp=0.0), b"ACGT" * 20 + b"A" * 500 + b"CGTA" * 30]
    for l, d in ((31, 0.01), (12, 0.05), (28, 0.1)):
        bound = oracle.hash_bound(d)
        assert pkg.hash_bound(pkg.load_library().s2k_density_for_bound(bound)) == bound
        for seq in seqs:
            for cls, omode, pair in ((pkg.NtHashHPCIterator, so.HPC, False), (pkg.NtHashHPCSIMDIterator, so.HPCSIMD, False),
                                     (pkg.NtHashSIMDIterator, so.SIMD, True), (pkg.RegularMinimizers, so.REGULAR, False)):
                j, je, h = oracle.minimizers(seq, l, bound, omode)
                it = cls.new(seq, l, bound, engine=eng)
                assert len(it) == len(h)
                got = list(it)
                exp = list(zip(map(int, j), map(int, h))) if pair else list(zip(map(int, j), map(int, je), map(int, h)))
                assert got == exp, (cls.__name__, l, d, len(seq))
    for b in (0, 1, 12345, 0x028F5C28, 0x7FFFFFFF, 0xFFFFFFFE, 0xFFFFFFFF):  # every bound is reachable through a density
        assert pkg.hash_bound(pkg.load_library().s2k_density_for_bound(b)) == b
    with pytest.raises(pkg.S2kError):
        pkg.NtHashHPCIterator(b"ACGT", 31, 1000, engine=eng)  # KSizeOutOfRange, src/nthash_hpc.rs:117-121
    # seq.len() == l is accepted by the reference's iterators (only l > len is an error): the scalar Hpc iterator yields nothing (its only
    # l-mer is the last one, src/nthash_hpc.rs:265-267), the SIMD ones yield the single l-mer when its hash is below the bound.  Expected
    # values from the oracle's plain hash functions (the oracle's minimizer functions follow KminmersIterator: len <= l -> nothing).
    for seq in (b"ACGTTGCAGTCATGCATGCAAGTCGATCGAC", b"ACGTACGTACGTACGTACGTACGTACGTACG", b"AAGTACGTACGTACGTACGTACGTACGTACG", b"ACGTACGTACGTACGTACGTACGTACGTAAA"):
        l = len(seq)
        a = np.frombuffer(seq, dtype=np.uint8)
        assert list(pkg.NtHashHPCIterator.new(seq, l, 0xFFFFFFFF, engine=eng)) == []
        h0 = int(oracle.nthash32_all(a, l)[0])
        for bound in (0xFFFFFFFF, h0 + 1, h0, 1):
            bs = oracle.hash_bound_simd(bound)  # (strict '<' against the bound as the SIMD path re-derives it, src/nthash_avx512_32.rs:47-48)
            assert list(pkg.NtHashSIMDIterator.new(seq, l, bound, engine=eng)) == ([(0, h0)] if h0 < bs else []), (seq, bound)
        hs, hp = oracle.hpc(a)
        exp = []
        if len(hs) == l:  # every base its own run: one HPC l-mer; end = start of the last run (src/nthash_hpc_simd.rs:64)
            hh = int(oracle.nthash32_all(hs, l)[0])
            exp = [(int(hp[0]), int(hp[l - 1]), hh)]
        assert list(pkg.NtHashHPCSIMDIterator.new(seq, l, 0xFFFFFFFF, engine=eng)) == exp, seq


def test_degenerate_batches(eng, oracle):
    """no reads at all; only empty reads; a single base; everything shorter than l"""
    for mode in SCALAR + (HM.Simd, HM.HpcSimd):
        r = eng.extract(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64), 31, 10, 0.01, mode, want_minimizers=True)
        assert r["n"] == 0 and r["n_minimizers"] == 0 and list(r["km_off"]) == [0]
        compare(eng, oracle, [b"", b"", b""], 31, 10, 0.01, mode, tag="all-empty")
        compare(eng, oracle, [b"A"], 31, 10, 0.5, mode, tag="one-base")
        compare(eng, oracle, [b"ACGT" * 7, b"", b"ACGTTGCA" * 3 + b"ACGTTGC"], 31, 1, 1.0, mode, tag="shorter-than-l")


def test_fuzz_random_batches(eng, oracle):
    """Randomised batches: read lengths clustered around l, around the 144-base lane chunks and around the 9216-base
    tiles, mixed with long reads; random l (static and dynamic instantiations), k, density, alphabet noise."""
    rng = np.random.default_rng(int(os.environ.get("S2K_FUZZ_SEED", 2026)))
    T = 9216
    for it in range(int(os.environ.get("S2K_FUZZ_ITERS", 120))):
        l = int(rng.choice([31, 31, 31, 5, 12, 15, 16, 20, 21, 31, 32, 40, 64]))
        k = int(rng.choice([1, 2, 3, 5, 10, 17]))
        d = float(rng.choice([0.003, 0.01, 0.02, 0.1, 0.5]))
        n_reads = int(rng.integers(1, 60))
        lens = []
        for _ in range(n_reads):
            kind = rng.integers(0, 6)
            if kind == 0:
                lens.append(int(max(0, l + rng.integers(-3, 4))))
            elif kind == 1:
                lens.append(int(144 * rng.integers(1, 5) + rng.integers(-2, 3)))
            elif kind == 2:
                lens.append(int(T * rng.integers(1, 3) + rng.integers(-40, 41)))
            elif kind == 3:
                lens.append(int(rng.integers(0, 400)))
            else:
                lens.append(int(rng.integers(1000, 30000)))
        hp = float(rng.choice([0.0, 0.2, 0.5]))
        odd = float(rng.choice([0.0, 0.0, 0.03]))
        reads = [rand_read(rng, n, hp=hp, odd=odd) for n in lens]
        for mode in SCALAR + ((HM.Simd, HM.HpcSimd) if l <= 31 else ()):  # Simd modes: l <= 31 (src/nthash_avx512_32.rs:33)
            compare(eng, oracle, reads, l, k, d, mode, expect_path=0, tag="fuzz%d" % it)


def test_fuzz_many_tiny_reads(eng, oracle):
    """hundreds of read starts per 9216-base tile (more than the per-tile boundary lists hold), all four modes:
    lengths around l, around the 16-block tail rule of the Simd modes, empty reads in between."""
    rng = np.random.default_rng(int(os.environ.get("S2K_FUZZ_SEED", 7)))
    for it in range(int(os.environ.get("S2K_FUZZ_ITERS", 24))):
        l = int(rng.choice([4, 7, 12, 15, 31]))
        k = int(rng.choice([1, 2, 4]))
        d = float(rng.choice([0.05, 0.3, 1.0]))
        n_reads = int(rng.integers(200, 1500))
        kinds = rng.integers(0, 5, size=n_reads)
        lens = np.where(kinds == 0, 0,
               np.where(kinds == 1, l + rng.integers(-2, 3, size=n_reads),
               np.where(kinds == 2, l - 1 + 16 * rng.integers(1, 6, size=n_reads) + rng.integers(-1, 2, size=n_reads),
               np.where(kinds == 3, rng.integers(1, 60, size=n_reads), rng.integers(60, 300, size=n_reads)))))
        hp = float(rng.choice([0.0, 0.3]))
        reads = [rand_read(rng, int(max(0, n)), hp=hp, odd=0.01) for n in lens]
        if it % 3 == 0:
            reads.insert(len(reads) // 2, rand_read(rng, 30000, hp=hp))  # a long read among them
        for mode in SCALAR + (HM.Simd, HM.HpcSimd):
            compare(eng, oracle, reads, l, k, d, mode, expect_path=0, tag="tiny%d" % it)


def test_device_api_streams_timing_and_minimizer_capacity(eng, oracle):
    """s2k_set_stream with a non-default torch stream, HIP-event timing totals, minimizer-triple capacity."""
    import torch

    rng = np.random.default_rng(14)
    reads = [rand_read(rng, 12000, hp=0.1) for _ in range(40)]
    bases, off = pkg.pack_reads(reads)
    dev = torch.device("cuda", 0)
    d_b = torch.from_numpy(bases).to(dev)
    d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
    ref = oracle.batch(bases, off, 31, 10, 0.01, 1)
    rm = oracle.batch_minimizers(bases, off, 31, 0.01, 1)
    nk, nm = ref["n"], rm["n"]

    def mk(kcap, mcap):
        t = {n: torch.zeros(max(c, 1), dtype=dt, device=dev) for n, c, dt in (
            ("hash", kcap, torch.int64), ("start", kcap, torch.int32), ("end", kcap, torch.int32), ("rev", kcap, torch.uint8),
            ("mn_j", mcap, torch.int32), ("mn_jend", mcap, torch.int32), ("mn_hash", mcap, torch.int32))}
        t["km_off"] = torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev)
        t["mn_off"] = torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev)
        o = pkg.DeviceOut()
        o.km_capacity, o.mn_capacity = kcap, mcap
        for f in ("km_off", "hash", "start", "end", "rev", "mn_off", "mn_j", "mn_jend", "mn_hash"):
            setattr(o, f, t[f].data_ptr())
        return t, o

    e2 = pkg.Engine(0)
    side = torch.cuda.Stream(device=dev)
    e2.set_stream(side.cuda_stream)
    torch.cuda.synchronize()
    e2.enable_timing(True)
    t, o = mk(nk, nm)
    for _ in range(3):
        e2.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), 31, 10, 0.01, 1, o, sync=False)
    c = e2.sync()
    ms_all, n_calls = e2.timing_total(0)
    ms_min, _ = e2.timing_total(1)
    assert n_calls == 3 and 0 < ms_min <= ms_all and e2.last_kernel_ms(1) > 0
    assert c["n_kminmers"] == nk and c["n_minimizers"] == nm
    side.synchronize()
    assert (t["hash"].cpu().numpy().view(np.uint64) == ref["hash"]).all()
    assert (t["mn_j"].cpu().numpy().view(np.uint32) == rm["j"]).all() and (t["mn_hash"].cpu().numpy().view(np.uint32) == rm["hash"]).all()
    assert (t["mn_off"].cpu().numpy().view(np.uint64) == rm["mn_off"]).all()
    # minimizer arrays too small -> S2K_ERR_CAPACITY, k-min-mer outputs still complete and correct
    t, o = mk(nk, nm // 3)
    with pytest.raises(pkg.S2kError) as e:
        e2.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), 31, 10, 0.01, 1, o)
    assert e.value.status == 7
    assert (t["hash"].cpu().numpy().view(np.uint64) == ref["hash"]).all()
    assert (t["mn_j"].cpu().numpy().view(np.uint32)[: nm // 3] == rm["j"][: nm // 3]).all()
    e2.close()


def test_contexts_are_independent_across_threads(oracle):
    """include/s2k.h: one s2k_ctx per (thread, device), contexts are independent -- the shape of the reference's
    parallel_fastx workers (src/main.rs:65-79), each with its own engine, running at the same time."""
    import threading

    rng = np.random.default_rng(51)
    jobs = []
    for i in range(4):
        reads = [rand_read(rng, int(n), hp=0.2) for n in rng.integers(0, 40000, size=30)]
        mode = (HM.Regular, HM.Hpc, HM.Simd, HM.HpcSimd)[i]
        bases, off = pkg.pack_reads(reads)
        jobs.append((bases, off, mode, oracle.batch(bases, off, 31, 6, 0.02, OMODE[mode])))
    errors = []

    def work(job):
        try:
            bases, off, mode, ref = job
            e = pkg.Engine(0)
            for _ in range(5):
                got = e.extract(bases, off, 31, 6, 0.02, mode)
                for f in FIELDS:
                    assert (got[f] == ref[f]).all(), (int(mode), f)
            e.close()
        except Exception as ex:  # noqa: BLE001 - reported to the main thread
            errors.append(repr(ex))

    th = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_fuzz_standalone_hpc_short():
    """tools/fuzz_hpc.py (random ragged batches through s2k_hpc_device / _ex against the oracle's hpc / encode_rle: both rules, aligned and
    unaligned base pointers, output arrays at odd offsets, thousands of reads in one segment, runs longer than a segment), 200 batches"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_hpc.py"), "77", "200"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "0 mismatches" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])


def test_two_chained_contexts_alternate(oracle):
    """s2k_chain_after: two contexts chained both ways and fed alternately with DIFFERENT batches (different sizes, both scalar modes, several chunks
    each) give what each batch gives alone; unlinking and destroying one context leaves the other usable."""
    import torch

    rng = np.random.default_rng(15)
    dev = torch.device("cuda", 0)
    batches = []
    for n_reads, rl in ((260, 30000), (90, 52000), (400, 9000)):
        reads = [rand_read(rng, int(rl + rng.integers(-500, 500)), hp=0.2) for _ in range(n_reads)]
        bases, off = pkg.pack_reads(reads)
        batches.append((bases, off, torch.from_numpy(bases).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev)))
    os.environ["S2K_DESC_CHUNKS"] = "3"  # (read when a context is created: every call below is a three-chunk pipeline)
    try:
        a, b = pkg.Engine(0), pkg.Engine(0)
    finally:
        del os.environ["S2K_DESC_CHUNKS"]
    a.chain_after(b)
    b.chain_after(a)
    with pytest.raises(pkg.S2kError):
        a.chain_after(a)

    def mk(n_reads, cap):
        t = {"km_off": torch.zeros(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.zeros(cap, dtype=torch.int64, device=dev),
             "start": torch.zeros(cap, dtype=torch.int32, device=dev), "end": torch.zeros(cap, dtype=torch.int32, device=dev),
             "rev": torch.zeros(cap, dtype=torch.uint8, device=dev)}
        o = pkg.DeviceOut()
        o.km_capacity = cap
        o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
        return t, o

    for mode in (1, 0, 3, 2):  # Hpc, Regular, HpcSimd (tile words + look-back), Simd
        refs = [oracle.batch(bb[0], bb[1], 31, 10, 0.01, OMODE[mode]) for bb in batches]
        pending = []  # (engine, tensors, batch index)
        for i in range(9):
            e = a if i % 2 == 0 else b
            j = i % 3
            bases, off, d_b, d_o = batches[j]
            t, o = mk(len(off) - 1, refs[j]["n"] + 8)
            e.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(off) - 1, len(bases), 31, 10, 0.01, mode, o, sync=False)
            # (a context finishes its own previous call before it takes the next one: that call's tensors may go now)
            pending = [p for p in pending if p[0] is not e]
            pending.append((e, t, j))
            if i >= 7:  # the last call of each context: checked in full
                c = e.sync()
                ref = refs[j]
                assert c["n_kminmers"] == ref["n"] and c["path"] == 0
                assert (t["hash"][:ref["n"]].cpu().numpy().view(np.uint64) == ref["hash"]).all()
                for f in ("start", "end"):
                    assert (t[f][:ref["n"]].cpu().numpy().view(np.uint32) == ref[f]).all(), f
                assert (t["rev"][:ref["n"]].cpu().numpy() == ref["rev"]).all() and (t["km_off"].cpu().numpy().view(np.uint64) == ref["km_off"]).all()
    a.sync()
    b.sync()
    b.chain_after(None)
    a.chain_after(None)
    a.close()
    bases, off, d_b, d_o = batches[0]
    ref0 = oracle.batch(bases, off, 31, 10, 0.01, OMODE[0])
    t, o = mk(len(off) - 1, ref0["n"] + 8)
    c = b.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(off) - 1, len(bases), 31, 10, 0.01, 0, o, sync=True)
    assert c["n_kminmers"] == ref0["n"]
    b.close()


def test_destroying_a_chained_to_context_unlinks_the_survivor(oracle):
    """s2k_destroy takes the destroyed context out of every chain (s2k_chain_after): the survivor, still 'chained' to it by the caller's
    book-keeping, runs its next calls unchained instead of following a dangling link."""
    import torch

    rng = np.random.default_rng(16)
    dev = torch.device("cuda", 0)
    reads = [rand_read(rng, int(20000 + rng.integers(-300, 300)), hp=0.2) for _ in range(200)]
    bases, off = pkg.pack_reads(reads)
    d_b, d_o = torch.from_numpy(bases).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev)
    a, b = pkg.Engine(0), pkg.Engine(0)
    a.chain_after(b)
    b.chain_after(a)
    ref = oracle.batch(bases, off, 31, 10, 0.01, OMODE[1])
    cap = ref["n"] + 8

    def run(e):
        t = {"km_off": torch.zeros(len(off), dtype=torch.int64, device=dev), "hash": torch.zeros(cap, dtype=torch.int64, device=dev),
             "start": torch.zeros(cap, dtype=torch.int32, device=dev), "end": torch.zeros(cap, dtype=torch.int32, device=dev),
             "rev": torch.zeros(cap, dtype=torch.uint8, device=dev)}
        o = pkg.DeviceOut()
        o.km_capacity = cap
        o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
        c = e.extract_device(d_b.data_ptr(), d_o.data_ptr(), len(off) - 1, len(bases), 31, 10, 0.01, 1, o, sync=True)
        assert c["n_kminmers"] == ref["n"]
        assert (t["hash"][:ref["n"]].cpu().numpy().view(np.uint64) == ref["hash"]).all()
        assert (t["km_off"].cpu().numpy().view(np.uint64) == ref["km_off"]).all()

    run(a)
    run(b)       # both have a recorded tiles_done event now
    dead = a.ctx  # (the handle's value only: it is compared with the registry of live contexts, never followed)
    a.close()     # destroyed FIRST, while b is still chained to it
    run(b)        # must not touch a's event
    # chaining to a context that is no longer alive is refused (round 6), and leaves b as it was
    assert b.lib.s2k_chain_after(b.ctx, dead) != 0  # S2K_ERR_INVALID_ARG
    run(b)
    b.close()
