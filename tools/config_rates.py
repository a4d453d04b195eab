"""Device-resident rates on the other BASELINE.json configs (shapes only differ; l=31 k=10): C3 = one GPU's 25 Gbp shard
of ONT-like ragged reads, C5 = 1000 contigs of 1 Mbp at d = 0.001."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
pkg = import_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)

def run(tag, lens, density, seed, modes=(pkg.HashMode.Regular, pkg.HashMode.Hpc)):
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    n_reads, n_bases = len(lens), int(off[-1])
    d_b = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
    d_o = torch.from_numpy(off).to(dev)
    torch.cuda.synchronize()
    eng.synth_bases_device(seed, 0, n_bases, d_b.data_ptr())
    cap = int(n_bases * density * 2.4) + 1_000_000
    t = {k: torch.empty(n, dtype=dt, device=dev) for k, n, dt in (("km_off", n_reads + 1, torch.int64), ("hash", cap, torch.int64), ("start", cap, torch.int32), ("end", cap, torch.int32), ("rev", cap, torch.uint8))}
    o = pkg.DeviceOut(); o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    torch.cuda.synchronize()
    for mode in modes:
        eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, density, int(mode), o)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, 31, 10, density, int(mode), o)
            best = min(best, time.perf_counter() - t0)
        print("%s %-8s %.2f Gbp  %.1f Gbp/s  kminmers=%d path=%d" % (tag, mode.name, n_bases / 1e9, n_bases / best / 1e9, c["n_kminmers"], c["path"]), flush=True)

rng = np.random.default_rng(303)
mu = np.log(20000) - 0.5 * 0.5 / 2
run("C3-shard (ONT-like, ragged)", np.clip(rng.lognormal(mu, 0.5, size=1_250_000), 1000, 200000).astype(np.int64), 0.01, 3)
run("C5 (1 Mbp contigs, d=0.001)", np.full(1000, 1_000_000, dtype=np.int64), 0.001, 5)
run("C2 (1 M x 10 kbp) all four modes", np.full(1_000_000, 10_000, dtype=np.int64), 0.01, 1, modes=(pkg.HashMode.Regular, pkg.HashMode.Hpc, pkg.HashMode.Simd, pkg.HashMode.HpcSimd))
