"""Randomized parity run of the standalone homopolymer compression (s2k_hpc_device / _ex, csrc/s2k_hpc_seg.hip) against the oracle's
restatement of src/hpc.rs: python tools/fuzz_hpc.py <seed> <iterations>.  Every batch: ragged reads (empty ones, thousands of tiny ones inside
one 4096-byte segment, runs longer than a segment, reads that start inside a run of the read before), any-byte rule and encode_rle rule,
base pointer aligned (segment-parallel kernels) or not (read-serial kernels), output arrays at any byte / word offset, sizes-only call."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import pkg, rand_read
from oracle import s2k_oracle as so
seed, iters = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
eng = pkg.Engine(0); oracle = so.get(); dev = torch.device("cuda", 0)
bad = 0; t0 = time.time(); total_bases = 0
for it in range(iters):
    shape = int(rng.integers(0, 5))
    if shape == 0:
        lens = [int(x) for x in rng.integers(0, int(rng.choice([3, 9, 40])), size=int(rng.integers(100, 9000)))]
    elif shape == 1:
        lens = [int(rng.integers(20000, 300000)) for _ in range(int(rng.integers(1, 5)))]
    elif shape == 2:
        lens = [int(4096 * rng.integers(0, 4) + rng.integers(-3, 4)) for _ in range(int(rng.integers(1, 40)))]
        lens = [max(0, x) for x in lens]
    else:
        lens = [int(rng.integers(0, 12000)) for _ in range(int(rng.integers(1, 120)))]
    hp = float(rng.choice([0.0, 0.3, 0.6, 0.95])); odd = float(rng.choice([0.0, 0.0, 0.05, 0.4]))
    alphabet = b"ACGT" if rng.random() < 0.7 else (b"AC" if rng.random() < 0.5 else b"A")
    reads = [rand_read(rng, n, hp=hp, alphabet=alphabet, odd=odd) for n in lens]
    if rng.random() < 0.3 and reads:  # one very long run somewhere
        i = int(rng.integers(0, len(reads))); c = bytes([int(rng.choice(np.frombuffer(b"ATn-X", dtype=np.uint8)))])
        reads[i] = reads[i][:len(reads[i]) // 2] + c * int(rng.integers(4000, 20000)) + reads[i][len(reads[i]) // 2:]
    rle = bool(rng.random() < 0.5)
    which = 1 if rle else 2
    bases, off = pkg.pack_reads(reads)
    n_bases = len(bases); total_bases += n_bases
    exp = [oracle.hpc(r, which) if len(r) else (b"", np.zeros(0, dtype=np.uint64)) for r in reads]
    exp_s = b"".join(e[0] for e in exp)
    exp_p = np.concatenate([np.asarray(e[1], dtype=np.uint32) for e in exp]) if exp else np.zeros(0, dtype=np.uint32)
    exp_off = np.concatenate([[0], np.cumsum([len(e[0]) for e in exp])]).astype(np.int64)
    shift = int(rng.choice([0, 0, 0, 1, 5])); o_shift = int(rng.integers(0, 4))
    d_b = torch.zeros(n_bases + 64, dtype=torch.uint8, device=dev)
    if n_bases: d_b[shift:shift + n_bases] = torch.from_numpy(bases).to(dev)
    d_o = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_ho = torch.zeros(len(reads) + 1, dtype=torch.int64, device=dev)
    d_h = torch.zeros(n_bases + 16, dtype=torch.uint8, device=dev)
    d_p = torch.zeros(n_bases + 16, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    try:
        n = eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), n_bases, d_ho.data_ptr(), d_h.data_ptr() + o_shift, d_p.data_ptr() + 4 * o_shift,
                           n_bases + 8, rle=rle)
        assert n == len(exp_s), ("runs", n, len(exp_s))
        assert (d_ho.cpu().numpy() == exp_off).all(), "hpc_off"
        h = d_h.cpu().numpy()
        assert h[o_shift:o_shift + n].tobytes() == exp_s, "bytes"
        assert not h[:o_shift].any() and not h[o_shift + n:].any(), "bytes outside the range were written"
        pp = d_p.cpu().numpy().view(np.uint32)
        assert (pp[o_shift:o_shift + n] == exp_p).all(), "positions"
        assert not pp[:o_shift].any() and not pp[o_shift + n:].any(), "positions outside the range were written"
        assert eng.hpc_device(d_b.data_ptr() + shift, d_o.data_ptr(), len(reads), n_bases, d_ho.data_ptr(), 0, 0, 0, rle=rle) == n, "sizes only"
    except AssertionError as e:
        bad += 1
        print("MISMATCH it=%d shape=%d reads=%d bases=%d rle=%s shift=%d o_shift=%d: %s" % (it, shape, len(reads), n_bases, rle, shift, o_shift, e), flush=True)
        if bad > 5: break
    if it % 100 == 99:
        print("hpc fuzz %d / %d batches, %.1f Mbp, %d bad, %.0f s" % (it + 1, iters, total_bases / 1e6, bad, time.time() - t0), flush=True)
print("hpc fuzz seed %d: %d batches, %.1f Mbp, %d mismatches" % (seed, it + 1, total_bases / 1e6, bad), flush=True)
sys.exit(1 if bad else 0)
