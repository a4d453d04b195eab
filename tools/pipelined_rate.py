"""Two contexts (two caller streams, two sets of output arrays) fed alternately with the headline batch: the tail of one call (its last k-min-mer
kernel, the totals, the host's look at the counts) runs beside the first chunk of the next call of the other context.
python tools/pipelined_rate.py [hpc|regular] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
pkg = import_package()
mode = pkg.HashMode.Regular if "regular" in sys.argv[1:] else pkg.HashMode.Hpc
steps = 20
dev = torch.device("cuda", 0)
n_reads, rl = 1_000_000, 10_000
n_bases = n_reads * rl
d_off = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * rl
d_bases = torch.empty(n_bases + 256, dtype=torch.uint8, device=dev)
engs = [pkg.Engine(0) for _ in range(2)]
if "chain" in sys.argv[1:]:
    engs[0].chain_after(engs[1]); engs[1].chain_after(engs[0])
torch.cuda.synchronize()
engs[0].synth_bases_device(1, 0, n_bases, d_bases.data_ptr())
torch.cuda.synchronize()
cap = int(n_bases * 0.024) + 1_000_000
outs = []
for e in engs:
    t = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
         "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev), "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut(); o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    outs.append((t, o))
def run(n_eng, steps):
    for i in range(4): engs[i % n_eng].extract_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases, 31, 10, 0.01, mode, outs[i % n_eng][1], sync=False)
    for e in engs: e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): engs[i % n_eng].extract_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases, 31, 10, 0.01, mode, outs[i % n_eng][1], sync=False)
    c = [e.sync() for e in engs]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, c
for n_eng in (1, 2, 1, 2):
    dt, c = run(n_eng, steps)
    print("%d context(s): %.3f ms per step, %.1f Gbp/s, kminmers %s" % (n_eng, dt / steps * 1e3, n_bases * steps / dt / 1e9, [x["n_kminmers"] for x in c[:n_eng]]), flush=True)
same = all((outs[0][0][f] == outs[1][0][f]).all().item() for f in ("km_off",)) and (outs[0][0]["hash"][:c[0]["n_kminmers"]] == outs[1][0]["hash"][:c[0]["n_kminmers"]]).all().item()
print("outputs of the two contexts identical:", same)
