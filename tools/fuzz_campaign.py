"""Longer randomized parity run than the test suite's (other seeds, more shapes): python tools/fuzz_campaign.py <seed> <iterations>.
Every batch goes through the C ABI in all four HashModes (tiled kernels, and the read-serial ones every 8th batch) and is
compared with the oracle tuple by tuple, minimizer triples included."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import pkg, compare, rand_read
from oracle import s2k_oracle as so
HM = pkg.HashMode
seed, iters = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed); T = 9216
# FUZZ_CHUNKS="1,2,3,7": one engine per forced chunk count of the two-stream pipeline (S2K_DESC_CHUNKS is read when a context is
# created), one of them drawn per batch; unset: one engine with the library's default
engines = []
for ck in [c for c in os.environ.get("FUZZ_CHUNKS", "").split(",") if c]:
    os.environ["S2K_DESC_CHUNKS"] = ck
    engines.append(pkg.Engine(0))
if not engines: engines.append(pkg.Engine(0))
oracle = so.get()
bad = 0; t0 = time.time()
for it in range(iters):
    l = int(rng.choice([31, 31, 31, 4, 5, 7, 10, 11, 12, 15, 16, 17, 20, 21, 25, 28, 31, 32, 33, 40, 63, 64, 65, 80]))
    k = int(rng.choice([1, 2, 3, 5, 10, 17, 40]))
    d = float(rng.choice([0.001, 0.003, 0.01, 0.02, 0.1, 0.5, 1.0]))
    shape = int(rng.integers(0, 5))
    eng = engines[int(rng.integers(0, len(engines)))]
    if len(engines) > 1 and rng.random() < 0.15: shape = 5
    lens = []
    if shape == 0:      # many tiny reads: > 30 read starts per tile (the per-hit read-table search)
        lens = [int(rng.integers(0, 3 * l + 8)) for _ in range(int(rng.integers(200, 3000)))]
    elif shape == 1:    # a few long reads
        lens = [int(rng.integers(20000, 400000)) for _ in range(int(rng.integers(1, 6)))]
    elif shape == 5:    # a batch of several hundred tiles: every forced chunk count really chunks (>= 64 tiles per chunk)
        lens = [int(rng.integers(500, 60000)) for _ in range(int(rng.integers(100, 400)))]
    else:
        for _ in range(int(rng.integers(1, 80))):
            kind = rng.integers(0, 6)
            if kind == 0: lens.append(int(max(0, l + rng.integers(-3, 4))))
            elif kind == 1: lens.append(int(144 * rng.integers(1, 5) + rng.integers(-2, 3)))
            elif kind == 2: lens.append(int(T * rng.integers(1, 3) + rng.integers(-40, 41)))
            elif kind == 3: lens.append(int(rng.integers(0, 400)))
            else: lens.append(int(rng.integers(1000, 30000)))
    hp = float(rng.choice([0.0, 0.2, 0.5, 0.9])); odd = float(rng.choice([0.0, 0.0, 0.03, 0.3]))
    alphabet = b"ACGT" if rng.random() < 0.8 else b"AC"
    if os.environ.get("FUZZ_LOWC"):  # low-complexity sequence and dense minimizers: tiles overflow their slabs, pools are re-sized
        hp = float(rng.choice([0.5, 0.9, 0.97])); alphabet = b"AC" if rng.random() < 0.7 else b"A"
        d = float(rng.choice([0.05, 0.1, 0.5, 1.0]))
    reads = [rand_read(rng, n, hp=hp, alphabet=alphabet, odd=odd) for n in lens]
    if rng.random() < 0.2:
        eng.set_host_batch(int(rng.integers(2000, 200000)))
    else:
        eng.set_host_batch(0)
    for mode in (HM.Regular, HM.Hpc, HM.Simd, HM.HpcSimd):
        if l > 31 and mode in (HM.Simd, HM.HpcSimd): continue  # the reference's SIMD iterators stop at l = 31
        for serial in ((False, True) if it % 8 == 0 else (False,)):
            try:
                compare(eng, oracle, reads, l, k, d, mode, force_serial=serial, tag="campaign")
            except AssertionError as e:
                bad += 1
                print("FAIL seed", seed, "iter", it, "mode", int(mode), "serial", serial, "l", l, "k", k, "d", d, "shape", shape, "hp", hp, "odd", odd, "n_reads", len(lens), str(e)[:300], flush=True)
    if it % 25 == 0: print("iter", it, "ok so far, failures", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done", iters, "iterations, failures", bad)
sys.exit(1 if bad else 0)
