"""Re-run one iteration of tests/test_gpu_parity.py::test_fuzz_random_batches and print where the minimizer records differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import pkg, OMODE, rand_read
from oracle import s2k_oracle as so
HM = pkg.HashMode
target = int(sys.argv[1]); want_mode = int(sys.argv[2]) if len(sys.argv) > 2 else None
rng = np.random.default_rng(2026); T = 9216
eng = pkg.Engine(0); oracle = so.get()
for it in range(target + 1):
    l = int(rng.choice([31, 31, 31, 5, 12, 15, 16, 20, 21, 31, 32, 40, 64])); k = int(rng.choice([1, 2, 3, 5, 10, 17]))
    d = float(rng.choice([0.003, 0.01, 0.02, 0.1, 0.5])); n_reads = int(rng.integers(1, 60)); lens = []
    for _ in range(n_reads):
        kind = rng.integers(0, 6)
        if kind == 0: lens.append(int(max(0, l + rng.integers(-3, 4))))
        elif kind == 1: lens.append(int(144 * rng.integers(1, 5) + rng.integers(-2, 3)))
        elif kind == 2: lens.append(int(T * rng.integers(1, 3) + rng.integers(-40, 41)))
        elif kind == 3: lens.append(int(rng.integers(0, 400)))
        else: lens.append(int(rng.integers(1000, 30000)))
    hp = float(rng.choice([0.0, 0.2, 0.5])); odd = float(rng.choice([0.0, 0.0, 0.03]))
    reads = [rand_read(rng, n, hp=hp, odd=odd) for n in lens]
print("iter", target, "l", l, "k", k, "d", d, "n_reads", n_reads, "hp", hp, "odd", odd, "lens", lens)
bases, off = pkg.pack_reads(reads)
for mode in (HM.Regular, HM.Hpc, HM.Simd, HM.HpcSimd):
    if want_mode is not None and int(mode) != want_mode: continue
    got = eng.extract(bases, off, l, k, d, mode, want_minimizers=True)
    rm = oracle.batch_minimizers(bases, off, l, d, OMODE[mode])
    print("mode", int(mode), "n_min gpu", got["n_minimizers"], "oracle", rm["n"])
    n = min(got["n_minimizers"], rm["n"])
    for a, b in (("mn_j", "j"), ("mn_jend", "jend"), ("mn_hash", "hash")):
        bad = np.nonzero(got[a][:n] != rm[b][:n])[0]
        print(" ", a, "mismatches", len(bad), bad[:12])
        for i in bad[:6]:
            r = int(np.searchsorted(rm["mn_off"], i, side="right") - 1)
            pos = int(off[r]) + int(rm["j"][i])
            print("    idx", int(i), "read", r, "j", int(rm["j"][i]), "stream pos", pos, "tile", pos // T, "in-tile", pos % T, "raw lane", (pos % T) // 144,
                  "gpu", int(got[a][i]), "ref", int(rm[b][i]))
