"""Scratch diagnostics on the GPU box: where do tiled / serial kernels diverge from the oracle?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import pkg, OMODE, rand_read
from oracle import s2k_oracle as so

o = so.get()
eng = pkg.Engine(0)
ecoli = open("tests/golden/ecoli.genome.100k.fa").read().split("\n")[1].encode()
rng = np.random.default_rng(3)


def diag(tag, reads, l, k, d, mode, serial):
    bases, off = pkg.pack_reads(reads)
    try:
        got = eng.extract(bases, off, l, k, d, mode, want_minimizers=True, force_serial=serial)
    except Exception as e:
        print(tag, "EXC", e)
        return
    rm = o.batch_minimizers(bases, off, l, d, OMODE[mode])
    rk = o.batch(bases, off, l, k, d, OMODE[mode])
    msg = [tag, "mode", int(mode), "l", l, "k", k, "serial" if serial else "tiled", "path", got["counts"]["path"],
           "nmin", got["n_minimizers"], rm["n"], "nkm", got["n"], rk["n"]]
    ok = got["n_minimizers"] == rm["n"] and got["n"] == rk["n"]
    n = min(got["n_minimizers"], rm["n"])
    for a, b in (("mn_j", "j"), ("mn_jend", "jend"), ("mn_hash", "hash")):
        neq = np.nonzero(got[a][:n] != rm[b][:n])[0]
        if len(neq):
            ok = False
            i = int(neq[0])
            msg += ["|", a, "first_bad", i, "of", len(neq), "got", int(got[a][i]), "ref", int(rm[b][i]),
                    "ctx_got", got["mn_j"][max(0, i - 2):i + 3].tolist(), "ctx_ref", rm["j"][max(0, i - 2):i + 3].tolist()]
    n = min(got["n"], rk["n"])
    for f in ("hash", "start", "end", "rev"):
        neq = np.nonzero(got[f][:n] != rk[f][:n])[0]
        if len(neq):
            ok = False
            i = int(neq[0])
            msg += ["|", f, "first_bad", i, "of", len(neq), "got", int(got[f][i]), "ref", int(rk[f][i])]
    if not (got["mn_off"] == rm["mn_off"]).all():
        ok = False
        msg += ["| mn_off differs"]
    print("OK  " if ok else "FAIL", *msg)


for serial in (True, False):
    for mode in (pkg.HashMode.Regular, pkg.HashMode.Hpc):
        diag("tiny", [b"ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGATCGATTAGCTAGCTAGGATCGATCGATCGGGATATCGCGATATTTAGC" * 3], 7, 2, 0.5, mode, serial)
        diag("ecoli10", [ecoli], 10, 5, 0.0001, mode, serial)
        diag("ecoli31", [ecoli], 31, 10, 0.01, mode, serial)
        diag("ecoli31d1", [ecoli[:3000]], 31, 2, 1.0, mode, serial)
        diag("rand2", [rand_read(rng, 20000), rand_read(rng, 500, hp=0.3), b"", rand_read(rng, 9216)], 31, 3, 0.05, mode, serial)
        diag("rand2l5", [rand_read(rng, 20000), rand_read(rng, 500, hp=0.3), b"", rand_read(rng, 9216)], 5, 3, 0.05, mode, serial)
for mode in (pkg.HashMode.Simd, pkg.HashMode.HpcSimd):
    diag("simd", [ecoli[:5000], rand_read(rng, 300, hp=0.3, odd=0.1)], 9, 2, 0.3, mode, True)
