"""PCIe-inclusive rate of the host-buffer API (never the headline value)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2k_loader import import_package
from oracle import s2k_oracle as so
pkg = import_package()
eng = pkg.Engine(0)
n_reads, rl = 100_000, 10_000
bases = so.get().synth_bases(1, 0, n_reads * rl)
off = np.arange(n_reads + 1, dtype=np.uint64) * rl
for mode in (pkg.HashMode.Regular, pkg.HashMode.Hpc):
    eng.extract(bases, off, 31, 10, 0.01, mode)  # warm-up (allocations)
    t0 = time.perf_counter()
    r = eng.extract(bases, off, 31, 10, 0.01, mode)
    dt = time.perf_counter() - t0
    print("s2k_extract host->host", mode.name, "%.1f Gbp/s" % (n_reads * rl / dt / 1e9), "kminmers", r["n"])
# file mode
path = "/tmp/s2k_reads.fa"
with open(path, "wb") as f:
    for i in range(n_reads):
        f.write(b">r%d\n" % i)
        f.write(bases[i * rl:(i + 1) * rl].tobytes())
        f.write(b"\n")
for mode in (pkg.HashMode.Regular, pkg.HashMode.Hpc):
    eng.run_file(path, 31, 10, 0.01, mode)
    tot = eng.run_file(path, 31, 10, 0.01, mode)
    print("s2k_run_file (FASTA 1 Gbp, page cache)", mode.name, "%.2f Gbp/s" % (tot["n_bases"] / tot["seconds"] / 1e9), tot["n_kminmers"])
os.remove(path)
