"""kernel times of one HpcSimd extraction of 10 Gbp (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s2k_loader import import_package
pkg = import_package(); eng = pkg.Engine(0); dev = torch.device("cuda", 0)
n_reads, L = 1_000_000, 10_000
d_b = torch.empty(n_reads * L + 64, dtype=torch.uint8, device=dev)
d_o = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize(); eng.synth_bases_device(1, 0, n_reads * L, d_b.data_ptr())
cap = int(n_reads * L * 0.0215)
t = {k: torch.empty(n, dtype=dt, device=dev) for k, n, dt in (("km_off", n_reads + 1, torch.int64), ("hash", cap, torch.int64), ("start", cap, torch.int32), ("end", cap, torch.int32), ("rev", cap, torch.uint8))}
o = pkg.DeviceOut(); o.km_capacity = cap
o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
for _ in range(4):
    c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, 31, 10, 0.01, int(pkg.HashMode.HpcSimd), o)
print(c["n_kminmers"])
