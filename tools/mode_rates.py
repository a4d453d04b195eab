"""Device-resident rate of every HashMode (tiled vs read-serial kernels) on a config-2 shaped batch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
pkg = import_package()
eng = pkg.Engine(0)
n_reads, L = int(os.environ.get("S2K_READS", 200_000)), 10_000
dev = torch.device("cuda", 0)
d_b = torch.empty(n_reads * L + 64, dtype=torch.uint8, device=dev)
d_o = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize()
eng.synth_bases_device(1, 0, n_reads * L, d_b.data_ptr())
cap = int(n_reads * L * 0.025)
t = {k: torch.empty(n, dtype=dt, device=dev) for k, n, dt in (("km_off", n_reads + 1, torch.int64), ("hash", cap, torch.int64), ("start", cap, torch.int32), ("end", cap, torch.int32), ("rev", cap, torch.uint8))}
o = pkg.DeviceOut(); o.km_capacity = cap
o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
torch.cuda.synchronize()
for mode in pkg.HashMode:
    for flags in (0, 2):
        eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, 31, 10, 0.01, int(mode), o, flags=flags)
        t0 = time.perf_counter()
        c = eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, 31, 10, 0.01, int(mode), o, flags=flags)
        dt = time.perf_counter() - t0
        print("%-8s %-12s path=%d  %.1f Gbp/s  kminmers=%d" % (mode.name, "force_serial" if flags else "default", c["path"], n_reads * L / dt / 1e9, c["n_kminmers"]), flush=True)
# standalone homopolymer compression (s2k_hpc_device): compressed bytes + run starts of the whole batch
d_ho = torch.empty(n_reads + 1, dtype=torch.int64, device=dev)
d_h = torch.empty(n_reads * L, dtype=torch.uint8, device=dev)
d_p = torch.empty(n_reads * L, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    n = eng.hpc_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_reads * L, d_ho.data_ptr(), d_h.data_ptr(), d_p.data_ptr(), n_reads * L)
    dt = time.perf_counter() - t0
print("s2k_hpc_device (string + positions)  %.1f Gbp/s  runs=%d" % (n_reads * L / dt / 1e9, n), flush=True)
