"""FASTA/FASTQ ingest + batching (counterpart of parallel_fastx + src/main.rs:51-83).
The parser is host code (CPU tier); the file driver runs the GPU path and is checked against the oracle."""
import numpy as np
import pytest

from s2k_loader import import_package

pkg = import_package()


def _reads(rng, n, maxlen=3000):
    return [bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5 if i % 7 == 0 else 4, size=int(rng.integers(0, maxlen)))])
            for i in range(n)]


def _write_fasta(path, reads, width=0, crlf=False, trailing_newline=True):
    nl = "\r\n" if crlf else "\n"
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write((">read%d some description%s" % (i, nl)).encode())
            s = r.decode()
            if width:
                lines = [s[j:j + width] for j in range(0, len(s), width)] or [""]
            else:
                lines = [s]
            body = nl.join(lines)
            last = i == len(reads) - 1
            f.write((body + ("" if last and not trailing_newline else nl)).encode())


def _write_fastq(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))


def _collect(path, max_bases, max_reads=0):
    rd = pkg.FastxReader(str(path))
    out = []
    nbatches = 0
    while True:
        b = rd.next_batch(max_bases, max_reads)
        if b is None:
            break
        nbatches += 1
        bases, off = b
        for r in range(len(off) - 1):
            out.append(bases[int(off[r]):int(off[r + 1])].tobytes())
    rd.close()
    return out, nbatches


@pytest.mark.parametrize("width,crlf,trail", [(0, False, True), (60, False, True), (70, True, True), (0, False, False), (13, False, False)])
def test_fasta_parsing(tmp_path, width, crlf, trail):
    rng = np.random.default_rng(1)
    reads = _reads(rng, 57)
    p = tmp_path / "x.fa"
    _write_fasta(p, reads, width, crlf, trail)
    got, nb = _collect(p, 20000)
    assert got == reads and nb > 3
    got, nb = _collect(p, 1 << 30)
    assert got == reads and nb == 1
    got, nb = _collect(p, 1 << 30, max_reads=5)
    assert got == reads and nb == 12


def test_fastq_parsing_and_errors(tmp_path):
    rng = np.random.default_rng(2)
    reads = _reads(rng, 33)
    p = tmp_path / "x.fq"
    _write_fastq(p, reads)
    got, _ = _collect(p, 5000)
    assert got == reads
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r1\nACGT\nIIII\n")  # missing '+' line
    rd = pkg.FastxReader(str(bad))
    with pytest.raises(pkg.S2kError):
        rd.next_batch()
    with pytest.raises(pkg.S2kError):
        pkg.FastxReader(str(tmp_path / "does_not_exist.fa"))
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    assert _collect(empty, 100) == ([], 0)


@pytest.mark.gpu
def test_run_file_matches_oracle(tmp_path, oracle):
    from oracle import s2k_oracle as so

    rng = np.random.default_rng(3)
    reads = _reads(rng, 400, maxlen=30000)
    p = tmp_path / "reads.fa"
    _write_fasta(p, reads, width=80)
    bases, off = pkg.pack_reads(reads)
    eng = pkg.Engine(0)
    for mode, omode in ((pkg.HashMode.Regular, so.REGULAR), (pkg.HashMode.Hpc, so.HPC)):
        ref = oracle.batch(bases, off, 31, 5, 0.01, omode)
        mn = oracle.batch_minimizers(bases, off, 31, 0.01, omode)
        for batch in (1 << 30, 300000):  # one batch / many batches (read order and totals must not depend on batching)
            tot = eng.run_file(str(p), 31, 5, 0.01, mode, batch_bases=batch)
            assert tot["n_reads"] == len(reads) and tot["n_bases"] == len(bases)
            assert tot["n_kminmers"] == ref["n"] and tot["n_minimizers"] == mn["n"]
            assert tot["xor_hash"] == int(np.bitwise_xor.reduce(ref["hash"]))
    eng.close()


def _gpu_split(eng, text, fastq):
    """record splitting on the GPU; returns the list of reads"""
    import torch

    dev = torch.device("cuda", 0)
    t = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(dev) if len(text) else torch.zeros(16, dtype=torch.uint8, device=dev)
    st, nr, nb = eng.parse_fastx_device(t.data_ptr(), len(text), fastq)
    assert st == 7 or (nr == 0 and nb == 0)  # no output buffers given: sizes only
    d_b = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
    d_o = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    st, nr2, nb2 = eng.parse_fastx_device(t.data_ptr(), len(text), fastq, d_b.data_ptr(), nb, d_o.data_ptr(), nr + 1)
    assert (st, nr2, nb2) == (0, nr, nb)
    b, o = d_b.cpu().numpy(), d_o.cpu().numpy()
    assert int(o[-1]) == nb and (b[nb:] == 0).all()
    return [b[int(o[r]):int(o[r + 1])].tobytes() for r in range(nr)]


@pytest.mark.gpu
def test_gpu_record_splitting_equals_host_parser(tmp_path):
    """s2k_fastx_parse_device (records split in HBM) against the host parser and the reads that were written"""
    eng = pkg.Engine(0)
    rng = np.random.default_rng(11)
    reads = _reads(rng, 301, maxlen=9000) + [b"", b"A", b"ACGT" * 5000]
    for width, crlf, trail in ((0, False, True), (60, False, True), (70, True, True), (0, False, False), (13, False, False), (4096, False, True)):
        p = tmp_path / "x.fa"
        _write_fasta(p, reads, width, crlf, trail)
        got = _gpu_split(eng, p.read_bytes(), fastq=False)
        assert got == reads == _collect(p, 1 << 30)[0], (width, crlf, trail)
    p = tmp_path / "x.fq"
    _write_fastq(p, reads)
    text = p.read_bytes()
    assert _gpu_split(eng, text, fastq=True) == reads
    assert _gpu_split(eng, text + b"\n\n", fastq=True) == reads  # trailing blank lines
    # quality strings full of '@' and '>' must not be taken for headers
    with open(p, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+r%d\n%s\n" % (i, r, i, (b"@>+" * len(r))[:len(r)]))
    assert _gpu_split(eng, p.read_bytes(), fastq=True) == reads
    # blank lines between FASTA records, empty text, header-only text
    assert _gpu_split(eng, b">a\nAC\n\n>b\n\nGT\nT\n", fastq=False) == [b"AC", b"GTT"]
    assert _gpu_split(eng, b"", fastq=False) == []
    assert _gpu_split(eng, b">only a header", fastq=False) == [b""]
    # malformed FASTQ is reported, not mis-parsed
    import torch

    bad = b"@r1\nACGT\nIIII\n@r2\nAC\n+\nII\n"
    t = torch.from_numpy(np.frombuffer(bad, dtype=np.uint8).copy()).cuda()
    with pytest.raises(pkg.S2kError):
        eng.parse_fastx_device(t.data_ptr(), len(bad), True)
    eng.close()


@pytest.mark.gpu
def test_run_file_many_batches_fasta_and_fastq(tmp_path, oracle):
    """file mode with batch boundaries falling inside records (the driver must cut at record starts), FASTA with
    wrapped lines and one record much longer than a batch, FASTQ with '@' in the qualities"""
    from oracle import s2k_oracle as so

    rng = np.random.default_rng(17)
    reads = _reads(rng, 900, maxlen=12000) + [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=3_300_000)])] + _reads(rng, 50, 5000)
    bases, off = pkg.pack_reads(reads)
    ref = oracle.batch(bases, off, 31, 7, 0.01, so.HPC, threads=8)
    want = (len(reads), len(bases), ref["n"], int(np.bitwise_xor.reduce(ref["hash"])))
    eng = pkg.Engine(0)
    fa, fq = tmp_path / "r.fa", tmp_path / "r.fq"
    _write_fasta(fa, reads, width=61)
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, (b"@I" * len(r))[:len(r)]))
    for path in (fa, fq):
        for batch in (1 << 20, 1 << 30):
            tot = eng.run_file(str(path), 31, 7, 0.01, pkg.HashMode.Hpc, batch_bases=batch)
            assert (tot["n_reads"], tot["n_bases"], tot["n_kminmers"], tot["xor_hash"]) == want, (path.name, batch)
    # a file of several pinned chunks: FASTA text crosses PCIe 2-bit packed (headers, newlines, N runs and a lower-case
    # stretch travel as exceptions; the chunk that is mostly lower case goes as it is) and must arrive byte for byte
    big = []
    for i in range(12):
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(1_500_000, 3_500_000)))].copy()
        for q in rng.integers(0, len(a) - 200, size=40):
            a[q:q + int(rng.integers(1, 150))] = ord("N")
        if i in (5, 6):
            a |= 0x20
        big.append(a.tobytes())
    bb, bo = pkg.pack_reads(big)
    rb = oracle.batch(bb, bo, 31, 7, 0.01, so.HPC, threads=8)
    bf = tmp_path / "big.fa"
    _write_fasta(bf, big, width=70)
    for flags in (0, pkg.FLAG_NO_PACK2):
        tot = eng.run_file(str(bf), 31, 7, 0.01, pkg.HashMode.Hpc, flags=flags)
        assert (tot["n_reads"], tot["n_bases"], tot["n_kminmers"], tot["xor_hash"]) == (len(big), len(bb), rb["n"], int(np.bitwise_xor.reduce(rb["hash"]))), flags
    # low-complexity input: far more k-min-mers than the density-based capacity guess -> exact retry inside the driver
    lc = tmp_path / "lc.fa"
    reads2 = [b"ACGT" * 20000 for _ in range(40)]
    _write_fasta(lc, reads2)
    b2, o2 = pkg.pack_reads(reads2)
    for d in (0.01, 1.0):
        r2 = oracle.batch(b2, o2, 31, 7, d, so.REGULAR, threads=8)
        tot = eng.run_file(str(lc), 31, 7, d, pkg.HashMode.Regular)
        assert tot["n_kminmers"] == r2["n"] and tot["xor_hash"] == (int(np.bitwise_xor.reduce(r2["hash"])) if r2["n"] else 0)
    eng.close()


@pytest.mark.gpu
def test_cli_demo_and_file_mode(tmp_path, oracle):
    """src/main.rs:13-48 (demo) and :51-83 (file mode) through the C++ driver."""
    import os
    import subprocess

    from conftest import ROOT
    from oracle import s2k_oracle as so

    exe = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc", "s2k_main")
    demo = subprocess.run([exe], capture_output=True, text=True)
    assert demo.returncode == 0, demo.stderr
    seq = b"AACTGCACTGCACTGCACTGCACACTGCACTGCACTGCACTGCACACTGCACTGCACTGACTGCACTGCACTGCACTGCACTGCCTGC"
    for name, omode in (("Regular", so.REGULAR), ("Simd", so.SIMD), ("Hpc", so.HPC), ("HpcSimd", so.HPCSIMD)):
        ref = oracle.kminmers(seq, 28, 5, 0.1, omode)
        block = demo.stdout.split("mode: %s\n" % name)[1].split("mode: ")[0]
        lines = [x for x in block.splitlines() if x.startswith("kminmer:")]
        assert len(lines) == len(ref["hash"])
        for i, ln in enumerate(lines):
            assert "hash: %d, start: %d, end: %d, offset: %d, rev: %s" % (int(ref["hash"][i]), int(ref["start"][i]), int(ref["end"][i]), i,
                                                                           "true" if ref["rev"][i] else "false") in ln
    rng = np.random.default_rng(5)
    reads = _reads(rng, 100, maxlen=20000)
    p = tmp_path / "r.fq"
    _write_fastq(p, reads)
    bases, off = pkg.pack_reads(reads)
    ref = oracle.batch(bases, off, 31, 5, 0.01, so.REGULAR)
    out = subprocess.run([exe, str(p)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "kminmers: %d " % ref["n"] in out.stdout and "FASTA to kminmers in" in out.stdout
