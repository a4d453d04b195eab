#!/usr/bin/env python3
"""Generate tests/golden/*.json.

Two kinds of fixtures:
  * ref_kat.json   -- known-answer vectors copied (as data) from the reference's own tests:
        G1 tests/main.rs:41-57  (Regular, l=10 k=5 d=1e-4 on tests/ecoli.genome.100k.fa, H=u32)
        G2 src/old/nthash_hpc.rs.orig:68-77 (HPC iterator, H=u64, l=4 d=0.1)
    They are written verbatim here, not computed.
  * derived_cases.json / ecoli_checkpoints.json -- outputs of the CPU oracle (oracle/s2k_oracle.c)
    AFTER it reproduced G1..G4; they pin what the reference's tests do not (positions, rev, offsets,
    Hpc-mode u32 hashes, odd bytes, edge lengths).  SURVEY.md section 8c lists the same checkpoints
    computed independently by the surveyor; test_oracle_golden.py checks those numbers too.

Run from the repo root:  python oracle/make_golden.py
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import s2k_oracle as so  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MODES = {"regular": so.REGULAR, "hpc": so.HPC, "simd": so.SIMD, "hpcsimd": so.HPCSIMD}


def density_for_bound(b):
    """a density whose src/lib.rs:91 bound is exactly b"""
    d = (b + 0.5) / 4294967295.0
    assert so.get().hash_bound(d) == b
    return d


def case(o, name, seq, l, k, d, mode):
    km = o.kminmers(seq, l, k, d, MODES[mode])
    j, je, h = o.minimizers(seq, l, o.hash_bound(d), MODES[mode])
    return {
        "name": name, "seq_hex": bytes(seq).hex(), "l": l, "k": k, "density": d, "mode": mode,
        "minimizers": {"j": [int(x) for x in j], "jend": [int(x) for x in je], "hash": [int(x) for x in h]},
        "kminmers": {"hash": [int(x) for x in km["hash"]], "start": [int(x) for x in km["start"]],
                     "end": [int(x) for x in km["end"]], "rev": [int(x) for x in km["rev"]]},
    }


def rand_dna(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def hp_rich(rng, n_runs, max_run=6, alphabet=b"ACGT"):
    out = bytearray()
    prev = None
    for _ in range(n_runs):
        c = rng.choice(alphabet)
        while c == prev:
            c = rng.choice(alphabet)
        out += bytes([c]) * rng.randint(1, max_run)
        prev = c
    return bytes(out)


def main():
    o = so.get()
    os.makedirs(GOLD, exist_ok=True)
    ecoli = open(os.path.join(GOLD, "ecoli.genome.100k.fa")).read().split("\n")[1].encode()

    # ---- reference KATs (data copied from the reference's tests) -----------------------------
    kat = {
        "G1": {"source": "tests/main.rs:41-57,60-73", "input": "ecoli.genome.100k.fa line 2",
               "l": 10, "k": 5, "density": 0.0001, "mode": "regular",
               "hashes32": [143479479014703, 1415094313937202, 7085699921625713, 2731023262850893,
                            3529660833839258, 2520689800435504, 3515165585325381, 2855190423625803,
                            5122855536061684, 244022361441902, 2856446528761135, 906939906227534,
                            2115341643533671, 246274980452770, 159737436030657]},
        "G2": {"source": "src/old/nthash_hpc.rs.orig:68-77 (H=u64 build)",
               "seq": "ACTGCACATGATGAGTAGATGATGATGATGATGATATGATGATAT", "l": 4, "density": 0.1,
               "pos_hash64": [[0, 1693589515812555183], [6, 876319423165292601], [13, 771890730643629033],
                              [16, 826464090118103095], [33, 1245321008145464903], [34, 1193606442387228521]]},
        "G3": {"source": "tests/main.rs:76-78", "runs": 72873},
        "G4": {"source": "tests/main.rs:82-89", "l": [5, 7, 11, 17, 25, 31], "k": [2, 5, 8], "density": 0.01},
    }
    json.dump(kat, open(os.path.join(GOLD, "ref_kat.json"), "w"), indent=1)

    # ---- ecoli checkpoints (BASELINE config 1 and the G4 grid counts) -------------------------
    ck = {"source": "oracle after G1-G4; cross-checked against SURVEY.md 8c", "configs": []}
    for mode in ("regular", "hpc"):
        for (l, k, d) in [(31, 10, 0.01), (31, 10, 0.001), (10, 5, 0.0001), (31, 5, 0.01), (17, 8, 0.01), (5, 2, 0.01)]:
            km = o.kminmers(ecoli, l, k, d, MODES[mode])
            j, je, h = o.minimizers(ecoli, l, o.hash_bound(d), MODES[mode])
            n = len(km["hash"])
            ck["configs"].append({
                "mode": mode, "l": l, "k": k, "density": d,
                "n_minimizers": len(j), "n_kminmers": n,
                "xor_hash": int(np.bitwise_xor.reduce(km["hash"])) if n else 0,
                "sum_start": int(km["start"].sum()), "sum_end": int(km["end"].sum()), "n_rev": int(km["rev"].sum()),
                "first_minimizer": [int(j[0]), int(je[0]), int(h[0])] if len(j) else None,
                "last_minimizer": [int(j[-1]), int(je[-1]), int(h[-1])] if len(j) else None,
                "first_kminmer": [int(km["hash"][0]), int(km["start"][0]), int(km["end"][0]), int(km["rev"][0])] if n else None,
                "last_kminmer": [int(km["hash"][-1]), int(km["start"][-1]), int(km["end"][-1]), int(km["rev"][-1])] if n else None,
            })
    json.dump(ck, open(os.path.join(GOLD, "ecoli_checkpoints.json"), "w"), indent=1)

    # ---- short adversarial reads with full outputs ---------------------------------------------
    rng = random.Random(20261003)
    cases = []
    l, k = 7, 3
    d = 0.2
    for mode in ("regular", "hpc", "simd", "hpcsimd"):
        cases.append(case(o, "all_A_200", b"A" * 200, l, k, d, mode))
        cases.append(case(o, "len_eq_l", rand_dna(rng, l), l, k, d, mode))
        cases.append(case(o, "len_eq_l_plus_1", rand_dna(rng, l + 1), l, k, 1.0, mode))
        cases.append(case(o, "len_eq_l_plus_2", rand_dna(rng, l + 2), l, k, 1.0, mode))
        cases.append(case(o, "empty", b"", l, k, d, mode))
        cases.append(case(o, "R_eq_l", hp_rich(rng, l, 5), l, 1, 1.0, mode))
        cases.append(case(o, "R_eq_l_plus_1", hp_rich(rng, l + 1, 5), l, 1, 1.0, mode))
        cases.append(case(o, "R_eq_l_plus_2", hp_rich(rng, l + 2, 5), l, 1, 1.0, mode))
        cases.append(case(o, "long_runs_300", hp_rich(rng, 12, 2) + b"C" * 300 + hp_rich(rng, 40, 3) + b"T" * 700 + hp_rich(rng, 30, 3), l, k, 0.5, mode))
        cases.append(case(o, "with_N_lower_other", b"ACGTNNACGTacgtnnXYZ*ACGGGTTTAAACCCNNNNACGTTGCAQQQACGTACGTTTGACCA" * 3, l, k, 0.5, mode))
        cases.append(case(o, "density_1", rand_dna(rng, 300), l, k, 1.0, mode))
        cases.append(case(o, "density_0", rand_dna(rng, 300), l, k, 0.0, mode))
        cases.append(case(o, "hp_rich_600", hp_rich(rng, 300, 6), 11, 4, 0.1, mode))
        cases.append(case(o, "l31_k10_3000", rand_dna(rng, 3000), 31, 10, 0.01, mode))
        cases.append(case(o, "l31_k10_hp_4000", hp_rich(rng, 2000, 4), 31, 10, 0.02, mode))
        cases.append(case(o, "l1", rand_dna(rng, 100), 1, 2, 0.5, mode))
        cases.append(case(o, "l28_k5_demo", b"AACTGCACTGCACTGCACTGCACACTGCACTGCACTGCACTGCACACTGCACTGCACTGACTGCACTGCACTGCACTGCACTGCCTGC", 28, 5, 0.1, mode))
    # exactly k-1 / k minimizers, and a hash equal to the bound (scalar `<=` keeps it, SIMD `<` drops it)
    for mode in ("regular", "hpc"):
        s = rand_dna(rng, 400) if mode == "regular" else hp_rich(rng, 300, 4)
        j, je, h = o.minimizers(s, 9, 0xFFFFFFFF, MODES[mode])
        hs = sorted(set(int(x) for x in h))
        for want in (3, 4):
            b = hs[want - 1]
            cases.append(case(o, "exactly_%d_minimizers_bound_equal" % want, s, 9, 4, density_for_bound(b), mode))
        cases.append(case(o, "bound_equal_minus_1", s, 9, 4, density_for_bound(hs[3] - 1), mode))
    for mode in ("simd", "hpcsimd"):
        # sentinel % 16 == 0 tail-mask quirk (src/nthash_avx512_32.rs:134-138)
        for n_lmers in (16, 32, 48, 33):
            s = rand_dna(rng, n_lmers + 9 - 1) if mode == "simd" else None
            if s is None:
                while True:
                    s = hp_rich(rng, n_lmers + 9 - 1, 3)
                    if len(s) > 9:
                        break
            cases.append(case(o, "simd_tail_%d" % n_lmers, s, 9, 2, 1.0, mode))
    json.dump({"source": "oracle/make_golden.py (oracle outputs, pinned by G1-G4)", "cases": cases},
              open(os.path.join(GOLD, "derived_cases.json"), "w"), indent=None, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
