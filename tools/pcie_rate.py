"""PCIe-inclusive rate of the host-buffer API (never the headline value)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
import torch
pkg = import_package()
eng = pkg.Engine(0)
if os.environ.get('S2K_HOST_BATCH'):
    eng.set_host_batch(int(os.environ['S2K_HOST_BATCH']))
n_reads, rl = int(os.environ.get('S2K_PCIE_READS', 400_000)), 10_000
_d = torch.empty(n_reads * rl + 64, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
eng.synth_bases_device(1, 0, n_reads * rl, _d.data_ptr())  # the library's own generator; the input is then moved to pageable host memory
eng.lib.s2k_sync(eng.ctx, None)
bases = _d[: n_reads * rl].cpu().numpy().copy()
del _d
off = np.arange(n_reads + 1, dtype=np.uint64) * rl
import ctypes as C
def c_extract(mode):  # the C call alone (what a Rust/C++ caller pays), no numpy conversion of the result
    p = pkg.Params(31, 10, 0.01, int(mode), 0)
    res = pkg.Result()
    t0 = time.perf_counter()
    st = eng.lib.s2k_extract(eng.ctx, bases.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), n_reads, C.byref(p), C.byref(res))
    dt = time.perf_counter() - t0
    assert st == 0
    nk = int(res.n_kminmers)
    eng.lib.s2k_result_free(C.byref(res))
    return dt, nk
for mode in (pkg.HashMode.Regular, pkg.HashMode.Hpc):
    c_extract(mode)  # warm-up (allocations, pinned ring)
    dt, nk = min(c_extract(mode) for _ in range(3))
    print("s2k_extract host->host", mode.name, "%.1f Gbp/s" % (n_reads * rl / dt / 1e9), "kminmers", nk, flush=True)
if os.environ.get('S2K_PCIE_SKIP_FILE'):
    sys.exit(0)
# file mode
path = "/tmp/s2k_reads.fa"
with open(path, "wb") as f:
    for i in range(n_reads):
        f.write(b">r%d\n" % i)
        f.write(bases[i * rl:(i + 1) * rl].tobytes())
        f.write(b"\n")
eng.run_file(path, 31, 10, 0.01, pkg.HashMode.Regular)  # warm-up: page cache settles, pinned ring exists
for rep in range(2):
    for mode in (pkg.HashMode.Regular, pkg.HashMode.Hpc):
        tot = eng.run_file(path, 31, 10, 0.01, mode)
        print("s2k_run_file (FASTA, page cache)", mode.name, "%.2f Gbp/s" % (tot["n_bases"] / tot["seconds"] / 1e9), tot["n_kminmers"], flush=True)
os.remove(path)
