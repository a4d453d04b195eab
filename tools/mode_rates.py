"""Device-resident rate of chosen HashModes on BASELINE configs[1] (1 M x 10 kbp, l=31 k=10 d=0.01): python tools/mode_rates.py hpc hpcsimd
(environment knobs such as S2K_DESC_CHUNKS apply; used for the HpcSimd pre-pass work)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
pkg = import_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
names = {"regular": pkg.HashMode.Regular, "hpc": pkg.HashMode.Hpc, "simd": pkg.HashMode.Simd, "hpcsimd": pkg.HashMode.HpcSimd}
L = [int(a[2:]) for a in sys.argv[1:] if a.startswith("l=")]
L = L[0] if L else 31  # l=25: a value without a compile-time instantiation takes the run-time-l hash loop
modes = [names[a] for a in sys.argv[1:] if a in names] or list(names.values())
n_reads, rl = 1_000_000, 10_000
n_bases = n_reads * rl
off = (np.arange(n_reads + 1, dtype=np.int64) * rl)
d_b = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
d_o = torch.from_numpy(off).to(dev)
torch.cuda.synchronize()
eng.synth_bases_device(1, 0, n_bases, d_b.data_ptr())
cap = int(n_bases * 0.01 * 2.4) + 1_000_000
t = {k: torch.empty(n, dtype=dt, device=dev) for k, n, dt in (("km_off", n_reads + 1, torch.int64), ("hash", cap, torch.int64), ("start", cap, torch.int32), ("end", cap, torch.int32), ("rev", cap, torch.uint8))}
o = pkg.DeviceOut(); o.km_capacity = cap
o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
torch.cuda.synchronize()
for mode in modes:
    eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, L, 10, 0.01, int(mode), o)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, L, 10, 0.01, int(mode), o)
        ts.append(time.perf_counter() - t0)
    print("%-8s l=%d chunks=%s  best %.3f ms  median %.3f ms  %.1f Gbp/s  kminmers=%d" % (mode.name, L, os.environ.get("S2K_DESC_CHUNKS", "default"), min(ts) * 1e3, sorted(ts)[2] * 1e3, n_bases / min(ts) / 1e9, c["n_kminmers"]), flush=True)
