"""The C++ facade (include/s2k.hpp) over the C ABI: compiles on CPU; runs on the GPU and must print exactly
the oracle's k-min-mers (tests/main.rs:60-73 style KAT loop)."""
import os
import subprocess

import pytest

from conftest import GOLD, ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
CSRC = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")


def _build(tmpdir):
    exe = os.path.join(tmpdir, "facade_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", CSRC, "-ls2k", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_facade_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_facade_matches_oracle(tmp_path, oracle, ecoli):
    from oracle import s2k_oracle as so

    exe = _build(str(tmp_path))
    seqf = tmp_path / "seq.txt"
    seqf.write_bytes(ecoli)
    for (l, k, d, mode, omode) in ((10, 5, 0.0001, 0, so.REGULAR), (31, 10, 0.01, 1, so.HPC), (31, 10, 0.01, 0, so.REGULAR)):
        out = subprocess.run([exe, str(seqf), str(l), str(k), repr(d), str(mode)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        got = [tuple(int(x) for x in line.split()) for line in out.stdout.splitlines()]
        ref = oracle.kminmers(ecoli, l, k, d, omode)
        exp = [(int(h), int(s), int(e), i, int(r)) for i, (h, s, e, r) in enumerate(zip(ref["hash"], ref["start"], ref["end"], ref["rev"]))]
        assert got == exp
    # the minimizer-triple iterators of the facade (NtHashHPCIterator / NtHashSIMDIterator / NtHashHPCSIMDIterator)
    out = subprocess.run([exe, str(seqf), "21", "4", "0.02", "1", "minimizers"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rows = [line.split() for line in out.stdout.splitlines() if line.startswith("M ")]
    bound = oracle.hash_bound(0.02)
    for mode, omode in ((1, so.HPC), (2, so.SIMD), (3, so.HPCSIMD)):
        got = [(int(r[2]), int(r[3]), int(r[4])) for r in rows if int(r[1]) == mode]
        j, je, h = oracle.minimizers(ecoli, 21, bound, omode)
        assert got == list(zip(map(int, j), map(int, je), map(int, h))) and len(got) > 100, mode
