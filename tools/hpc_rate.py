"""Device-resident rate of the standalone homopolymer compression (s2k_hpc_device / _ex) on 2 Gbp of the uniform synthetic stream:
python tools/hpc_rate.py [rle]   (under rocprofv3 --kernel-trace --stats it shows which kernel of the two-pass pipeline costs what)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
pkg = import_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
rle = "rle" in sys.argv[1:]
n_reads, rl = 200_000, 10_000
n_bases = n_reads * rl
d_o = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * rl
d_b = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.synth_bases_device(1, 0, n_bases, d_b.data_ptr())
cap = int(n_bases * 0.8) + 4096
t = {"off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hpc": torch.empty(cap, dtype=torch.uint8, device=dev), "pos": torch.empty(cap, dtype=torch.int32, device=dev)}
torch.cuda.synchronize()
ts = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runs = eng.hpc_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, n_bases, t["off"].data_ptr(), t["hpc"].data_ptr(), t["pos"].data_ptr(), cap, rle=rle)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("standalone hpc (%s): best %.3f ms, median %.3f ms -> %.1f Gbp/s; %d runs" % ("encode_rle rule" if rle else "any byte", min(ts) * 1e3, sorted(ts)[3] * 1e3, n_bases / min(ts) / 1e9, runs))
