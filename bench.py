#!/usr/bin/env python3
"""bench.py -- input bases/s (Gbp/s) of k-min-mer extraction, l=31 k=10 d=0.01, on MI355X.

Workloads (BASELINE.json configs):
  c2  (default, configs[1])  per GPU 1 000 000 synthetic uniform-random ACGT reads of 10 kbp (10 Gbp of ASCII), the
      reference's benches/bench.rs:19-31 convention.  Weak scaling: every rank owns its own 10 Gbp.
  ont (configs[2])           ONE batch of ragged reads -- lengths lognormal(mean 20 kbp, sigma 0.5) clipped to
      [1 k, 200 k], 1.25 M reads (~25 Gbp) per GPU, i.e. 10 M reads / ~200 Gbp on 8 GPUs -- cut into contiguous
      shards balanced by cumulative bases (sharding.shard_bounds); rank r runs the HIP path on its shard.
      value = all bases / slowest shard's time, so the load balance over ragged lengths is part of the number.
  hifi (configs[3])          per GPU 1 000 000 HiFi-like reads (~15 Gbp): lengths ~N(15 k, 2 k), homopolymer runs of geometric length
      with mean 2, ~0.1 % of them 20..2999 bases long, generated in HBM by s2k_synth_hifi_device; the default
      c2 run also reports its rate (and that of configs[4], 1 Mbp contigs at d = 0.001) under "other_configs".
Bases are generated in HBM by the library's splitmix64 generator (rank r generates exactly its part of the one
global stream).  A "step" is one pass of the whole hot path (s2k_extract_device: tile index + minimizer kernel + scans +
k-min-mer kernel) over the resident batch.  The K timed steps are measured twice, each time between a barrier + device synchronisation: through
ONE context (every call waits for the one before it: `one_context`, and the source of roofline.* and per_rank), and alternating between TWO
contexts chained with s2k_chain_after (two sets of output arrays, the same input) -- the double buffering of a loop over many batches, in which
the tail of a call runs beside the first chunk of the next: that is `value` (--contexts 1 makes `value` the one-context figure).  Reads shard
across GPUs with no data-path collective; the only cross-GPU traffic is an all-reduce of the count vector (RCCL).

Launch: `python bench.py --gpus N` starts N rank processes itself (fresh children, before anything touches the GPU);
under torchrun / torch.distributed.run (WORLD_SIZE set) it is one rank and --gpus must equal WORLD_SIZE.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes / HIP-event time of ALL kernels of a step, vs HBM peak; the dominant kernel's own
                  figure is kept beside it (kernel_*)
  cpu_baseline -- the CPU oracle (a port of the reference's scalar path) timed on this host, N=1 only
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c2", "ont", "hifi"], default="c2")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (default 1 000 000 for c2, 1 250 000 for ont)")
    ap.add_argument("--read-len", type=int, default=10_000, help="c2: read length")
    ap.add_argument("--l", type=int, default=31)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--mode", choices=["hpc", "regular"], default="hpc",
                    help="headline HashMode: hpc = the full fused path (HPC + ntHash + select + emit)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-side legs that follow the timed region: the CPU baseline and the PCIe-inclusive runs (profiling / A-B tools pass it)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the run of the other scalar HashMode")
    ap.add_argument("--cpu-sample-reads", type=int, default=120_000, help="reads of the same workload timed on the CPU")
    ap.add_argument("--verify-reads", type=int, default=2000, help="reads drawn across the whole stream and compared field by field with the oracle (outside the timed region)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: map every rank to GPU 0")
    ap.add_argument("--count", choices=["auto", "on", "off"], default="auto", nargs="?", const="on",
                    help="downstream k-min-mer count (hash table in HBM; N > 1: ONE all-to-all by hash prefix over RCCL), outside the headline. "
                         "auto = on when N > 1, so that every multi-GPU run exercises the one real exchange step of the path")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank owns its own batch (c2: 10 Gbp each; ont: 1.25 M reads each).  strong: --workload ont with a "
                         "FIXED total (--total-reads, default 10 M reads = ~200 Gbp = BASELINE configs[2]) cut over the N ranks")
    ap.add_argument("--total-reads", type=int, default=10_000_000, help="--scaling strong: reads of the whole job")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive legs (host->host s2k_extract, s2k_run_file) that N=1 runs after the timed region")
    ap.add_argument("--e2e-reads", type=int, default=400_000, help="reads (of --read-len bases) of the PCIe-inclusive legs: 4 Gbp by default")
    ap.add_argument("--legacy-path", action="store_true", help="legacy records (16 B with the read index + per-read scans) instead of the descriptor path")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the rates of BASELINE configs[3] / configs[4] and of the 'next' rows "
                                                                    "(standalone HPC, minimizer triples) that the default c2 run reports beside the headline (--no-other-mode skips them too)")
    ap.add_argument("--contexts", type=int, choices=[1, 2], default=2,
                    help="2 (default): the timed steps alternate between two library contexts (two caller streams, two sets of output arrays), so that the "
                         "tail of one call -- its last k-min-mer kernel, the totals, the host's look at the counts -- runs beside the first chunk of the "
                         "next: the double buffering of a production loop.  1: one context, every call waits for the one before it (also reported as "
                         "'one_context' when 2 is measured; --single-device rehearsals use 1)")
    ap.add_argument("--dump-shard", default=None, help="(tests) write this rank's outputs to <path>.rank<r>.npz")
    return ap.parse_args()


def spawn_ranks(n):
    """Start n rank processes (fresh interpreters; this parent never touches the GPU) and wait for them."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def bind_to_gpu_node(props):
    """Runs this rank on the CPUs of the NUMA node its GPU hangs off (what `numactl --cpunodebind` does for a deployed rank): on a two-socket host a
    1-GPU job's CPU share is a quota, not a placement, and the PCIe-inclusive legs moved between 50 and 140 Gbp/s with the socket the process happened
    to start on (profiles/r06_e2e_numa_bind.txt).  Returns the node, or None when sysfs does not name one (nothing is changed then)."""
    try:
        bus = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except (OSError, ValueError, AttributeError):
        return None


def usable_cpus():
    """CPUs this process may actually use: the affinity mask and the cgroup CPU quota both bound it (a GPU box hands a
    1-GPU job a share of the host, e.g. 16 of 256 hardware threads: timing 256 threads there measures the quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                n = min(n, max(1, int(math.ceil(int(q) / int(per)))))
            break
        except (OSError, ValueError):
            continue
    return n


class SensorSampler:
    """Shader clock and package power of one GPU during a timed region, from the amdgpu hwmon files (freq1_input in Hz, power1_input in uW):
    a thread that reads them every 10 ms.  The kernels of this path sustain ~2.0 GHz where the card idles at 2.4: the line says which it was."""

    def __init__(self, props):
        import glob

        self.freq = self.power = None
        want = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if want not in os.path.realpath(d):
                    continue
                for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                    f, pw = os.path.join(h, "freq1_input"), os.path.join(h, "power1_input")
                    if os.path.exists(f):
                        self.freq, self.power = f, (pw if os.path.exists(pw) else None)
            except OSError:
                continue
        self.mhz, self.watts, self._stop, self._thr = [], [], False, None
        self.active = False  # samples are kept only while a timed region is open (set around the timed steps, not the warm-ups)

    def _run(self):
        while not self._stop:
            if self.active:
                try:
                    f = int(open(self.freq).read()) / 1e6
                    w = int(open(self.power).read()) / 1e6 if self.power else None
                    if self.active:  # (the region may have closed while the files were read)
                        self.mhz.append(f)
                        if w is not None:
                            self.watts.append(w)
                except (OSError, ValueError):
                    pass
            time.sleep(0.005 if self.active else 0.001)  # (a timed region lasts ~0.1 s: ~20 samples; idle, the thread only looks at the flag)

    def start(self):
        if self.freq:
            import threading

            self.mhz, self.watts, self._stop = [], [], False
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()

    def stop(self):
        if self._thr:
            self._stop = True
            self._thr.join()
            self._thr = None
        m = sorted(x for x in self.mhz if x > 0)
        if not m:
            return None
        out = {"sclk_mhz_median": round(m[len(m) // 2]), "sclk_mhz_min": round(m[0]), "sclk_mhz_max": round(m[-1]), "samples": len(m)}
        if self.watts:
            out["power_w_mean"] = round(sum(self.watts) / len(self.watts))
        return out


def kernel_source_id():
    """sha256 over the device sources of the library (csrc/*.hip, *.h, Makefile): what a committed PMC measurement belongs to"""
    import glob
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def ont_lengths(n_reads, seed=2):
    """lognormal with mean 20 kbp, sigma 0.5, clipped to [1 k, 200 k]; identical on every rank."""
    import numpy as np

    sigma = 0.5
    mu = math.log(20000.0) - sigma * sigma / 2.0
    x = np.random.default_rng(seed).lognormal(mu, sigma, size=n_reads)
    return np.clip(x, 1000, 200000).astype(np.int64)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let bench.py spawn them)" % (args.gpus, world))

    if args.contexts == 2 and not args.single_device:
        # two contexts = four streams of the library beside torch's own: every stream gets a hardware queue of its own (the runtime's default is
        # four per process; two streams that share a queue serialise, and a wait in one holds up the other) -- read by the HIP runtime at start-up
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    from s2k_loader import import_package

    n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if not args.single_device and n_dev < args.gpus:
        sys.exit("bench.py: --gpus %d but only %d GPU(s) visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); "
                 "--single-device maps every rank to GPU 0 for rehearsals" % (args.gpus, n_dev))
    if args.scaling == "strong" and args.workload != "ont":
        sys.exit("bench.py: --scaling strong is defined for --workload ont (BASELINE configs[2]: one 200 Gbp batch cut over the GPUs)")
    pkg = import_package()
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: gloo need not resolve the host name (it may not resolve in a container)
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            # RCCL over xGMI for tensors in HBM, gloo for tensors in host memory: the extraction itself has no collective (reads are
            # sharded), so if RCCL cannot come up on a node the few control words (the barrier, the max over ranks of the time, the
            # count sums) travel over gloo instead and the line says so -- a broken fabric must not cost the measurement
            dist.init_process_group("cpu:gloo,cuda:nccl")
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the few collective words live
    torch.cuda.set_device(dev)
    numa_node = None if os.environ.get("S2K_BENCH_NO_NUMA") else bind_to_gpu_node(torch.cuda.get_device_properties(dev))
    collective = None
    if dist is not None:
        # proof that the process group really spans N ranks on N devices: every rank contributes (rank, device index, PCI bus id)
        props = torch.cuda.get_device_properties(dev)
        # the device's identity as the DRIVER reports it: uuid, else PCI domain:bus:device; never hash() (randomised per process) and never the
        # rank's own input (local_rank), which would make two ranks on one device look distinct
        dev_id = str(getattr(props, "uuid", "") or "")
        if not dev_id or set(dev_id) <= set("0-"):
            dev_id = "pci:%s:%s:%s" % (getattr(props, "pci_domain_id", "?"), getattr(props, "pci_bus_id", "?"), getattr(props, "pci_device_id", "?"))
        import hashlib

        dg = hashlib.sha256(dev_id.encode()).digest()  # (deterministic across processes; travels with the other words, over RCCL or gloo)
        words = [rank, local_rank, int.from_bytes(dg[:7], "little"), int.from_bytes(dg[7:14], "little")]
        rccl_error = None
        if red_dev.type == "cuda":
            try:
                me = torch.tensor(words, dtype=torch.int64, device=red_dev)
                allv = [torch.zeros_like(me) for _ in range(dist.get_world_size())]
                dist.all_gather(allv, me)
                torch.cuda.synchronize(dev)
                ok = 1
            except Exception as e:  # (e.g. two ranks mapped to one device, a fabric that is down)
                rccl_error, ok = "%s: %s" % (type(e).__name__, str(e).strip().splitlines()[-1][:200]), 0
            agree = torch.tensor([ok], dtype=torch.int32)  # every rank takes the same road from here
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 0:
                red_dev = torch.device("cpu")
                rccl_error = rccl_error or "RCCL failed on another rank"
        if red_dev.type == "cpu":
            me = torch.tensor(words, dtype=torch.int64)
            allv = [torch.zeros_like(me) for _ in range(dist.get_world_size())]
            dist.all_gather(allv, me)
        backend_text = args.backend if args.backend != "nccl" else ("nccl (RCCL)" if rccl_error is None else "gloo (RCCL unusable: %s)" % rccl_error)
        collective = {"backend": backend_text, "world_size_seen": dist.get_world_size(),
                      "ranks": [int(v[0]) for v in allv], "devices": [int(v[1]) for v in allv],
                      "device_ids": ["%014x%014x" % (int(v[2]), int(v[3])) for v in allv], "device_id_of": "sha256(uuid or pci domain:bus:device)[:14]",
                      "distinct_devices": len({(int(v[2]), int(v[3])) for v in allv}), "visible_devices": n_dev,
                      "device_name": props.name}
        assert collective["world_size_seen"] == args.gpus and sorted(collective["ranks"]) == list(range(args.gpus)), collective
        if not args.single_device:
            assert collective["distinct_devices"] == args.gpus, "ranks share a device: %r" % (collective,)
    eng = pkg.Engine(local_rank)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)  # one stream for torch events and the library's kernels

    import importlib.util

    spec = importlib.util.spec_from_file_location("s2k_sharding", os.path.join(ROOT, "rust-seq2kminmers_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)

    mode = pkg.HashMode.Hpc if args.mode == "hpc" else pkg.HashMode.Regular
    other = pkg.HashMode.Regular if args.mode == "hpc" else pkg.HashMode.Hpc

    # ---- this rank's shard, resident in HBM before the timed region ---------------------------------
    if args.workload == "c2":
        n_reads, rl = args.reads or 1_000_000, args.read_len
        n_bases = n_reads * rl
        first_base = rank * n_bases  # each rank owns a distinct part of the stream
        seed = 1
        d_off = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * rl
        host_off = None
        shard_info = {"reads_per_gpu": n_reads, "read_len": rl, "sharding": "reads, contiguous per rank"}
        wl_text = "%d x %d bp uniform-random ACGT reads per GPU (%.1f Gbp/GPU)" % (n_reads, rl, n_bases / 1e9)
    elif args.workload == "hifi":
        n_reads, seed = args.reads or 1_000_000, 3
        r_first = rank * n_reads  # each rank owns its own reads of the one numbered sequence of reads
        hl = eng.synth_hifi_lengths(seed, r_first, n_reads)
        host_off = np.concatenate(([0], np.cumsum(hl))).astype(np.uint64)
        n_bases, first_base = int(host_off[-1]), 0
        d_off = torch.from_numpy(host_off.astype(np.int64)).to(dev)
        shard_info = {"reads_per_gpu": n_reads, "sharding": "reads, contiguous per rank"}
        wl_text = "%d HiFi-like reads per GPU, lengths ~N(15k, 2k), geometric homopolymer runs of mean 2, ~0.1%% of the runs 20..2999 bases (%.1f Gbp/GPU)" % (
            n_reads, n_bases / 1e9)
    else:
        per_gpu = args.reads or 1_250_000
        total_reads = args.total_reads if args.scaling == "strong" else per_gpu * world
        if args.scaling == "strong": # the whole job is fixed; refuse what cannot fit one rank's HBM instead of dying in an allocation
            est = total_reads * 20000 // world
            need = est * (1.0 + 2.4 * args.density * 17 + 2.1 * args.density * 8) + (4 << 30)
            free_b, total_b = torch.cuda.mem_get_info(dev)
            if need > free_b:
                sys.exit("bench.py: --scaling strong: %.0f Gbp per rank needs ~%.0f GB of HBM, %.0f GB free: use more GPUs or --total-reads (%d reads fit)"
                         % (est / 1e9, need / 1e9, free_b / 1e9, int(total_reads * (free_b * 0.9) / need)))
        lens = ont_lengths(total_reads)
        goff = np.zeros(len(lens) + 1, dtype=np.uint64)
        np.cumsum(lens, out=goff[1:])
        b = sharding.shard_bounds(goff, world)
        r0, r1 = int(b[rank]), int(b[rank + 1])
        host_off = (goff[r0: r1 + 1] - goff[r0]).astype(np.uint64)
        n_reads, n_bases, first_base, seed = r1 - r0, int(host_off[-1]), int(goff[r0]), 2
        d_off = torch.from_numpy(host_off.astype(np.int64)).to(dev)
        shard_info = {"reads_total": int(len(lens)), "bases_total": int(goff[-1]), "reads_this_rank": n_reads, "bases_this_rank": n_bases,
                      "sharding": "contiguous read ranges balanced by cumulative bases (sharding.shard_bounds)"}
        wl_text = "%d ragged reads, lognormal(mean 20 kbp, sigma 0.5) clipped [1k,200k] (%.1f Gbp) in %d shard(s), %s scaling" % (
            len(lens), int(goff[-1]) / 1e9, world, args.scaling)
    d_bases = torch.empty(n_bases + 256, dtype=torch.uint8, device=dev)
    if args.workload == "hifi":
        eng.synth_hifi_device(seed, r_first, n_reads, d_off.data_ptr(), d_bases.data_ptr())
    else:
        eng.synth_bases_device(seed, first_base, n_bases, d_bases.data_ptr())
    cap = int(n_bases * (2.4 * args.density)) + 1_000_000
    outs = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
            "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
            "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (outs[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    torch.cuda.synchronize(dev)

    def step(m, sync=False):
        return eng.extract_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases, args.l, args.k, args.density, m, o, sync=sync,
                                  flags=pkg.FLAG_LEGACY_PATH if args.legacy_path else 0)

    def barrier():
        if dist is not None:
            torch.cuda.synchronize(dev)
            # the barrier: an all-reduce of one word where the collective words live (HBM over RCCL, or host memory over gloo); nobody leaves
            # it before everybody has entered it
            dist.all_reduce(torch.zeros(1, dtype=torch.int32, device=red_dev))
        torch.cuda.synchronize(dev)

    def timed(m, steps, warmup, strict=True, sensors=None):
        for _ in range(warmup):
            step(m)
        eng.sync()
        eng.enable_timing(True)  # HIP events around the kernels, on the stream the kernels are launched on
        barrier()
        if sensors is not None:
            sensors.active = True  # clock / power samples of the timed steps only
        t0 = time.perf_counter()
        for _ in range(steps):
            step(m)
        counts = eng.sync()
        if sensors is not None:
            sensors.active = False
        barrier()
        dt = time.perf_counter() - t0
        k_ms, k_n = eng.timing_total(1)
        km_ms, _ = eng.timing_total(2)
        all_ms, _ = eng.timing_total(0)
        eng.enable_timing(False)
        # (strict: the headline must not hide a second run of a call inside its time; the compatibility modes only report it --
        # HpcSimd re-runs a call with a run-count pre-pass when a look-back for run heads gives up, which happens when several
        # processes share one GPU and no minimizer kernel has all its waves resident, i.e. in --single-device rehearsals)
        assert k_n == steps or not strict, "internal re-runs (workspace growth) inside the timed region: %d event sets for %d steps" % (k_n, steps)
        timed.reruns = k_n - steps
        spread = None
        if dist is not None:
            t = torch.tensor([dt, -dt, all_ms / steps, -(all_ms / steps)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0].item())
            spread = {"wall_ms_per_step_max": round(float(t[0].item()) / steps * 1e3, 3), "wall_ms_per_step_min": round(-float(t[1].item()) / steps * 1e3, 3),
                      "kernels_ms_per_step_max": round(float(t[2].item()), 3), "kernels_ms_per_step_min": round(-float(t[3].item()), 3)}
        return dt, counts, k_ms / steps, km_ms / steps, all_ms / steps, spread

    sensors = SensorSampler(torch.cuda.get_device_properties(dev))
    sensors.start()  # shader clock and power during the headline's timed regions (one context, then two): the thread idles outside them
    dt, counts, min_ms, km_ms, pipe_ms, rank_spread = timed(mode, args.steps, args.warmup, sensors=sensors)
    assert counts["path"] == (2 if args.legacy_path else 0), "the tiled HIP kernels (descriptor path unless --legacy-path) must be the ones measured"
    # ---- the same K steps, alternating between two contexts: what a loop over many batches gets (double buffering) -------------
    dt_one, two_ctx = dt, None
    if args.contexts == 2 and not args.single_device:
        eng1 = eng2 = outs2 = None
        try:
            # two fresh contexts on their own (non-blocking) streams; `eng` above runs on torch's current stream, which is the legacy default
            # stream here -- beside it a second context was measured SLOWER than one context alone
            eng1 = pkg.Engine(local_rank)
            eng2 = pkg.Engine(local_rank)
            eng1.chain_after(eng2)  # the minimizer kernels of one context's call wait for those of the other's (s2k_chain_after): the persistent
            eng2.chain_after(eng1)  # kernels never compete, and the tail of a call runs beside the first chunk of the next
            outs2 = {kk: torch.empty_like(v) for kk, v in outs.items()}
            o2 = pkg.DeviceOut()
            o2.km_capacity = cap
            o2.km_off, o2.hash, o2.start, o2.end, o2.rev = (outs2[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
            torch.cuda.synchronize(dev)

            def step_i(i):
                e, oo = (eng1, o) if i % 2 == 0 else (eng2, o2)
                return e.extract_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases, args.l, args.k, args.density, mode, oo, sync=False,
                                        flags=pkg.FLAG_LEGACY_PATH if args.legacy_path else 0)

            for i in range(2 * ((max(args.warmup, 2) + 1) // 2)):  # untimed: both contexts have their workspace
                step_i(i)
            eng1.sync()
            eng2.sync()
            barrier()
            sensors.active = True
            t0 = time.perf_counter()
            for i in range(args.steps):
                step_i(i)
            c0 = eng1.sync()
            c1 = eng2.sync()
            sensors.active = False
            barrier()
            dt2c = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt2c], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2c = float(t[0].item())
            # both contexts computed the same thing (outside the timed region)
            nk0 = c0["n_kminmers"]
            assert c0["path"] == c1["path"] == counts["path"] and nk0 == c1["n_kminmers"] == counts["n_kminmers"] and c0["xor_hash"] == c1["xor_hash"]
            torch.cuda.synchronize(dev)
            for f in ("hash", "start", "end", "rev"):
                assert torch.equal(outs[f][:nk0], outs2[f][:nk0]), "the two contexts disagree on " + f
            assert torch.equal(outs["km_off"], outs2["km_off"])
            two_ctx = dt2c
        except pkg.S2kError as e:  # (no room for a second workspace: a shard sized to fill the device)
            two_ctx = None
            sys.stderr.write("bench.py: second context not measured (%s); the line reports one context\n" % e)
        finally:
            for e2 in (eng1, eng2):
                if e2 is not None:
                    e2.chain_after(None)
            for e2 in (eng1, eng2):
                if e2 is not None:
                    e2.close()
            del outs2
            torch.cuda.empty_cache()
        if two_ctx is not None:
            dt = two_ctx
    clocks = sensors.stop()
    # ---- verification outside the timed region: a read sample against the oracle ------------------
    verified = None
    if rank == 0 and args.verify_reads > 0:
        # a sample of reads drawn across the WHOLE stream (first and last reads, and tiles the dynamic deal hands out,
        # included), regenerated on the host, run through the oracle and compared field by field with the slices
        # [km_off[r], km_off[r+1]) of the device output
        from oracle import s2k_oracle as so

        orc = so.get()
        nv = min(args.verify_reads, n_reads)
        ids = set(np.random.default_rng(12345).choice(n_reads, size=nv, replace=False).tolist())
        ids.update(r for r in (0, 1, n_reads - 2, n_reads - 1) if 0 <= r < n_reads)
        ids = np.array(sorted(ids), dtype=np.int64)
        starts = ids.astype(np.uint64) * np.uint64(args.read_len) if host_off is None else host_off[ids]
        lens = np.full(len(ids), args.read_len, dtype=np.uint64) if host_off is None else host_off[ids + 1] - host_off[ids]
        voff = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
        hb = np.empty(int(voff[-1]), dtype=np.uint8)
        for i in range(len(ids)):
            hb[int(voff[i]): int(voff[i + 1])] = (orc.hifi_read(seed, r_first + int(ids[i]), int(lens[i])) if args.workload == "hifi"
                                                  else orc.synth_bases(seed, first_base + int(starts[i]), int(lens[i])))
        a0, n0 = int(starts[0]), int(min(lens[0], 4096))
        assert (d_bases[a0: a0 + n0].cpu().numpy() == hb[:n0]).all(), "device generator and oracle generator disagree"
        ref = orc.batch(hb, voff, args.l, args.k, args.density, so.HPC if mode == pkg.HashMode.Hpc else so.REGULAR, threads=4)
        tid = torch.from_numpy(ids).to(dev)
        s0, s1 = outs["km_off"][tid], outs["km_off"][tid + 1]
        ln = s1 - s0
        verified = bool((ln.cpu().numpy().astype(np.uint64) == np.diff(ref["km_off"])).all())
        if verified and ref["n"]:
            g = torch.repeat_interleave(s0 - (torch.cumsum(ln, 0) - ln), ln) + torch.arange(ref["n"], dtype=torch.int64, device=dev)
            verified = bool((outs["hash"][g].cpu().numpy().view(np.uint64) == ref["hash"]).all()
                            and (outs["start"][g].cpu().numpy().view(np.uint32) == ref["start"]).all()
                            and (outs["end"][g].cpu().numpy().view(np.uint32) == ref["end"]).all()
                            and (outs["rev"][g].cpu().numpy() == ref["rev"]).all())
        assert verified, "GPU output differs from the oracle on the verification sample"
        verified = {"ok": True, "reads": int(len(ids)), "kminmers": int(ref["n"])}
    if args.dump_shard:  # tests: concatenating the ranks' dumps must reproduce the unsharded run
        nk = counts["n_kminmers"]
        np.savez(args.dump_shard + ".rank%d.npz" % rank, km_off=outs["km_off"].cpu().numpy().view(np.uint64),
                 hash=outs["hash"][:nk].cpu().numpy().view(np.uint64), start=outs["start"][:nk].cpu().numpy().view(np.uint32),
                 end=outs["end"][:nk].cpu().numpy().view(np.uint32), rev=outs["rev"][:nk].cpu().numpy(), first_base=first_base, n_bases=n_bases)
    # the other scalar HashMode, for the record (not the headline)
    other_line = None
    if not args.no_other_mode:
        s2 = max(2, args.steps // 2)
        # (the same number of warm-up steps as the headline: this leg starts behind seconds of host-side verification, i.e. on an idle chip whose clocks
        # take a few steps to come up -- with one warm-up step the Regular kernel read 4.2-4.4 ms here against 3.9 ms in a process of its own)
        dt2, counts2, min_ms2, km_ms2, pipe_ms2, _ = timed(other, s2, max(1, args.warmup))
        alg2 = counts2["n_bases"] + 17 * counts2["n_kminmers"] + 16 * (n_reads + 1)
        tot2 = sharding.allreduce_counts(counts2, dist, red_dev)
        other_line = {"mode": "regular" if args.mode == "hpc" else "hpc", "value": round(tot2["n_bases"] * s2 / dt2 / 1e9, 2), "unit": "Gbp/s",
                      "kernel_ms": round(min_ms2, 3), "kminmer_kernel_ms": round(km_ms2, 3), "pipeline_ms": round(pipe_ms2, 3),
                      "roofline_frac": round(alg2 / (pipe_ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

        # the two compatibility modes (result semantics of the reference's AVX-512 iterators, src/nthash_avx512_32.rs /
        # src/nthash_hpc_simd.rs) on the same kernels: HpcSimd gets the run count of a read from the tiles themselves, no pre-pass
        compat = {}
        for nm, hm in (("simd", pkg.HashMode.Simd), ("hpcsimd", pkg.HashMode.HpcSimd)):
            s3 = max(2, args.steps // 4)
            dt3, counts3, _, _, pipe_ms3, _ = timed(hm, s3, max(1, args.warmup // 2), strict=False)
            assert counts3["path"] == (2 if args.legacy_path else 0)
            tot3 = sharding.allreduce_counts(counts3, dist, red_dev)
            compat[nm] = {"value": round(tot3["n_bases"] * s3 / dt3 / 1e9, 1), "pipeline_ms": round(pipe_ms3, 3), "reruns": int(timed.reruns)}
        other_line.update(compat)

    # ---- rates the headline does not show, N = 1, outside the timed region: the other BASELINE configs that fit one GPU and the
    #      "next" rows of SURVEY.md 8f (standalone HPC = src/hpc.rs, README.md:23 "HPC alone"; minimizer triples = the iterators of
    #      src/nthash_hpc.rs:193) -----------------------------------------------------------------------------------------------
    other_configs = standalone_hpc = minimizers_only = None
    if rank == 0 and world == 1 and args.workload == "c2" and not args.no_other_configs and not args.no_other_mode and not args.legacy_path:
        def rate(d_b, d_o, nr, nb, m, dens, out, steps=3, flags=0):
            def one():
                return eng.extract_device(d_b.data_ptr(), d_o.data_ptr(), nr, nb, args.l, args.k, dens, m, out, sync=False, flags=flags)
            one()
            eng.sync()
            eng.enable_timing(True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            c = eng.sync()
            torch.cuda.synchronize(dev)
            dtx = time.perf_counter() - t0
            all_ms, n_ev = eng.timing_total(0)
            k_ms, _ = eng.timing_total(1)
            eng.enable_timing(False)
            return {"value": round(nb * steps / dtx / 1e9, 1), "pipeline_ms": round(all_ms / steps, 3), "kminmers": c["n_kminmers"],
                    "frac": round((nb + 17 * c["n_kminmers"] + 16 * (nr + 1)) / (all_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                    "reruns": int(n_ev - steps)}

        def out_for(nr, cap):
            t = {"km_off": torch.empty(nr + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
                 "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
                 "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
            oo = pkg.DeviceOut()
            oo.km_capacity = cap
            oo.km_off, oo.hash, oo.start, oo.end, oo.rev = (t[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
            return t, oo

        other_configs = {}
        # BASELINE configs[3]: HiFi-like reads, HPC on (the one input on which the Hpc kernel's data-dependent parts matter:
        # run heads beyond the staged look-ahead, long spans in the back-map, src/nthash_hpc.rs:253-263,280-281)
        nr3 = 1_000_000
        hl = eng.synth_hifi_lengths(3, 0, nr3)
        hoff3 = np.concatenate(([0], np.cumsum(hl))).astype(np.int64)
        nb3 = int(hoff3[-1])
        d_o3 = torch.from_numpy(hoff3).to(dev)
        d_b3 = torch.empty(nb3 + 256, dtype=torch.uint8, device=dev)
        eng.synth_hifi_device(3, 0, nr3, d_o3.data_ptr(), d_b3.data_ptr())
        t3, o3 = out_for(nr3, int(nb3 * 0.0135) + 1_000_000)
        c3 = {"config": 3, "gbp": round(nb3 / 1e9, 2)}  # (README.md, 'Reading the bench line': what the workloads are)
        for nm, hm in (("hpc", pkg.HashMode.Hpc), ("hpcsimd", pkg.HashMode.HpcSimd), ("regular", pkg.HashMode.Regular)):
            c3[nm] = rate(d_b3, d_o3, nr3, nb3, hm, args.density, o3)
        other_configs["hifi"] = c3
        del d_b3, d_o3, t3, o3
        # BASELINE configs[4]: 1 Mbp contigs at d = 0.001 (sparse minimizers)
        nr4, rl4 = 10_000, 1_000_000
        d_o4 = torch.arange(0, nr4 + 1, dtype=torch.int64, device=dev) * rl4
        d_b4 = torch.empty(nr4 * rl4 + 256, dtype=torch.uint8, device=dev)
        eng.synth_bases_device(4, 0, nr4 * rl4, d_b4.data_ptr())
        t4, o4 = out_for(nr4, int(nr4 * rl4 * 0.0026) + 1_000_000)
        c4 = {"config": 4, "gbp": round(nr4 * rl4 / 1e9, 2)}
        for nm, hm in (("hpc", pkg.HashMode.Hpc), ("regular", pkg.HashMode.Regular)):
            c4[nm] = rate(d_b4, d_o4, nr4, nr4 * rl4, hm, 0.001, o4)
        other_configs["contigs"] = c4
        del d_b4, d_o4, t4, o4
        torch.cuda.empty_cache()

        # minimizer triples (start, end, hash32) beside the k-min-mers: S2K_FLAG_WANT_MINIMIZERS on the headline workload
        mcap = int(n_bases * 2.2 * args.density) + 1_000_000
        tm = {"mn_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "mn_j": torch.empty(mcap, dtype=torch.int32, device=dev),
              "mn_jend": torch.empty(mcap, dtype=torch.int32, device=dev), "mn_hash": torch.empty(mcap, dtype=torch.int32, device=dev)}
        o.mn_capacity = mcap
        o.mn_off, o.mn_j, o.mn_jend, o.mn_hash = (tm[x].data_ptr() for x in ("mn_off", "mn_j", "mn_jend", "mn_hash"))
        minimizers_only = {}
        for nm, hm in ((args.mode, mode), ("regular" if args.mode == "hpc" else "hpc", other)):
            minimizers_only[nm] = rate(d_bases, d_off, n_reads, n_bases, hm, args.density, o, flags=pkg.FLAG_WANT_MINIMIZERS)["value"]
        o.mn_capacity = 0
        o.mn_off = o.mn_j = o.mn_jend = o.mn_hash = None
        del tm

        # standalone homopolymer compression (src/hpc.rs:7-147: compressed bytes + run starts): 2 Gbp of the headline workload
        nrh = min(n_reads, 200_000)
        nbh = int(host_off[nrh]) if host_off is not None else nrh * args.read_len
        hcap = int(nbh * 0.80) + 4096
        th = {"off": torch.empty(nrh + 1, dtype=torch.int64, device=dev), "hpc": torch.empty(hcap, dtype=torch.uint8, device=dev),
              "pos": torch.empty(hcap, dtype=torch.int32, device=dev)}
        standalone_hpc = {"unit": "Gbp/s", "gbp": round(nbh / 1e9, 2)}
        for nm, rle in (("hpc", False), ("encode_rle", True)):  # src/hpc.rs:28-41,44-147 (any repeated byte collapses) / :7-25 (only ACTGactgNn)
            best, runs = None, 0
            for _ in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                runs = eng.hpc_device(d_bases.data_ptr(), d_off.data_ptr(), nrh, nbh, th["off"].data_ptr(), th["hpc"].data_ptr(), th["pos"].data_ptr(), hcap, rle=rle)
                torch.cuda.synchronize(dev)
                dth = time.perf_counter() - t0
                best = dth if best is None else min(best, dth)
            standalone_hpc[nm] = round(nbh / best / 1e9, 1)
            standalone_hpc[nm + "_gb_s"] = round((nbh + 5 * runs + 8 * (nrh + 1)) / best / 1e9, 1)  # algorithmic bytes: 1 read + 5 written per run
        del th
        torch.cuda.empty_cache()

    # whole-job counts: the only collective on this path (RCCL all-reduce of a few words)
    tot = sharding.allreduce_counts(counts, dist, red_dev)
    tot_bases, tot_min, tot_km = tot["n_bases"], tot["n_minimizers"], tot["n_kminmers"]
    balance = None
    if dist is not None:
        t = torch.tensor([float(n_bases)], dtype=torch.float64, device=red_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        balance = round(float(tmax.item()) / (tot_bases / world), 4)  # largest shard / mean shard, in bases

    # ---- downstream of the path, outside the headline: distinct k-min-mer hashes over all ranks -------------
    count_line = None
    if args.count == "on" or (args.count == "auto" and world > 1):
        ops = sharding.EngineCountOps(eng, dev)
        keys = outs["hash"][: counts["n_kminmers"]]
        sharding.count_kminmers(keys, ops, dist, collectives_on_device=(red_dev.type == "cuda"))  # warm-up (table allocation)
        # three passes, each timed on its own (max over ranks); the median is reported: in a process that has initialised RCCL
        # one call in a few takes ~0.7 s instead of 45 ms whatever it does (tools/debug/count_under_nccl.py) -- a background
        # thread of the process group, not this path
        tcs = []
        for _ in range(3):
            barrier()
            tc0 = time.perf_counter()
            cr = sharding.count_kminmers(keys, ops, dist, collectives_on_device=(red_dev.type == "cuda"))
            barrier()
            tc = time.perf_counter() - tc0
            if dist is not None:
                tt = torch.tensor([tc], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tc = float(tt.item())
            tcs.append(tc)
        tc = sorted(tcs)[1]
        # bytes: every key read once for the partition (N > 1: written and read once more, 8 B over xGMI), once for the insert,
        # plus one 12-byte table slot touched per insert and the 12-byte slots swept by the compaction (>= 2 slots per key)
        count_line = {"n_keys": cr["n_keys"], "n_distinct": cr["n_distinct"], "ms": round(tc * 1e3, 3), "ms_passes": [round(x * 1e3, 3) for x in tcs],
                      "keys_per_s": round(cr["n_keys"] / tc / 1e9, 3), "unit": "G keys/s",
                      "exchange": None if world == 1 else "all_to_all_single by hash prefix (%s), 8 B per key" % (collective or {}).get("backend", args.backend)}

    # ---- roofline (rank 0's launch): SURVEY.md 8d  B = N_bases*1 + N_kminmers*17 + 16*(N_reads+1) --------------------
    # each input byte once, each k-min-mer once (u64 hash + u32 start + u32 end + u8 rev), both offset tables;
    # divided by the HIP-event time of ALL kernels of the step (tile index, minimizer kernel, scans, k-min-mer kernel).
    alg_bytes = counts["n_bases"] + 17 * counts["n_kminmers"] + 16 * (n_reads + 1)
    achieved = alg_bytes / (pipe_ms * 1e-3) / 1e9
    # the dominant kernel on its own: it reads the bases and the read table once and writes 8 B per minimizer record (16 B on the legacy path)
    kern_bytes = counts["n_bases"] + (16 if args.legacy_path else 8) * counts["n_minimizers"] + 8 * (n_reads + 1)
    traffic, traffic_stale, issue = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and args.workload == "c2":
        try:
            tj = json.load(open(tpath))
            if tj.get("mode") == args.mode and tj.get("n_bases") == n_bases:
                # the PMC passes belong to ONE state of the kernels: the file carries the id of the sources it was measured on
                if tj.get("kernel_source_id") == kernel_source_id():
                    traffic = tj.get("hbm_bytes_per_step")
                    sq = tj.get("sq_per_step")
                    if sq:
                        # the limiter as a number: vector instructions issued per step against what the SIMDs could issue in the step's time at one
                        # wave64 instruction per 2 cycles (SIMD-32, MI355X_MICROARCH.md) and the shader clock sampled during THIS run
                        mhz = (clocks or {}).get("sclk_mhz_median") or tj.get("shader_mhz") or 2400
                        n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
                        issue = {"valu_insts_per_step": int(sq["SQ_INSTS_VALU"]), "all_insts_per_step": int(sq["all_insts"]), "shader_mhz": mhz,
                                 "frac_of_2cycle_issue": round(sq["SQ_INSTS_VALU"] * 2.0 / (n_simd * pipe_ms * 1e-3 * mhz * 1e6), 3),
                                 "lds_conflict_frac": round(sq["SQ_LDS_BANK_CONFLICT"] / max(sq["SQ_LDS_IDX_ACTIVE"], 1.0), 3)}
                else:
                    traffic_stale = {"measured_on": tj.get("kernel_source_id"), "now": kernel_source_id(), "stale_value": tj.get("hbm_bytes_per_step")}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_stale": traffic_stale,
                "algorithmic_bytes_per_step": int(alg_bytes), "bytes_per_base": round(alg_bytes / max(n_bases, 1), 4),
                "time_ms": round(pipe_ms, 3), "frac_at_value": round(alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": "tile_minimizer_kernel<%d,%s>" % (args.l, "hpc" if mode == pkg.HashMode.Hpc else "regular"),
                "kernel_ms": round(min_ms, 3), "kernel_bytes": int(kern_bytes),
                "kernel_frac": round(kern_bytes / (min_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kminmer_kernel_ms": round(km_ms, 3),
                "kminmer_exposed_ms": round(max(pipe_ms - min_ms, 0.0), 3)}

    # ---- PCIe-inclusive legs (SURVEY 8d "what is timed (2)"): never `value`, reported beside it; N = 1, outside the timed region -----
    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end and not args.no_cpu_baseline and args.workload == "c2":
        import ctypes as C

        ne = min(args.e2e_reads, n_reads)
        rl = args.read_len
        hb = d_bases[: ne * rl].cpu().numpy()  # pageable host memory, as a caller's buffer would be
        hoff = np.arange(ne + 1, dtype=np.uint64) * rl

        def c_extract(m):  # the C call alone (what a Rust / C++ caller pays): host buffers in, host SoA out
            p = pkg.Params(args.l, args.k, args.density, int(m), 0)
            res = pkg.Result()
            t0 = time.perf_counter()
            st = eng.lib.s2k_extract(eng.ctx, hb.ctypes.data_as(C.c_void_p), hoff.ctypes.data_as(C.c_void_p), ne, C.byref(p), C.byref(res))
            dt = time.perf_counter() - t0
            assert st == 0, st
            nk = int(res.n_kminmers)
            eng.lib.s2k_result_free(C.byref(res))
            return dt, nk

        legs = {}
        for m, name in ((mode, args.mode), (other, "regular" if args.mode == "hpc" else "hpc")):
            c_extract(m)  # warm-up: pinned rings, result pool
            dt_e, nk_e = min(c_extract(m) for _ in range(3))
            legs[name] = {"gbp_s": round(ne * rl / dt_e / 1e9, 1), "h2d": int(ne * rl // 4 + 8 * (ne + 1)), "d2h": int(17 * nk_e + 8 * (ne + 1))}
        file_leg = None
        try:
            path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "s2k_bench_%d.fa" % os.getpid())
            with open(path, "wb") as f:
                for i in range(ne):
                    f.write(b">r%d\n" % i)
                    f.write(hb[i * rl:(i + 1) * rl].tobytes())
                    f.write(b"\n")
            fbytes = os.path.getsize(path)
            eng.run_file(path, args.l, args.k, args.density, mode)  # warm-up: page cache settles
            best = min((eng.run_file(path, args.l, args.k, args.density, mode) for _ in range(2)), key=lambda r: r["seconds"])
            file_leg = round(best["n_bases"] / best["seconds"] / 1e9, 1)
            os.remove(path)
        except OSError as ex:  # no room for the file: the leg is skipped, the bench line stands
            file_leg = "n/a (%s)" % type(ex).__name__
        e2e = {"gbp": round(ne * rl / 1e9, 2), "extract_gbp_s": legs[args.mode]["gbp_s"], "extract_other_mode_gbp_s": legs["regular" if args.mode == "hpc" else "hpc"]["gbp_s"],
               "h2d_bytes": legs[args.mode]["h2d"], "d2h_bytes": legs[args.mode]["d2h"], "run_file_gbp_s": file_leg, "host_cpus": usable_cpus(), "numa_node": numa_node}
        del hb

    # ---- CPU baseline: the oracle = scalar port of the reference, on this host's cores --------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "c2":  # N=1 only (the host cores are shared by all ranks)
        from oracle import s2k_oracle as so

        orc = so.Oracle(native=True)  # -O3 -march=native, like the reference's -Ctarget-cpu=native (.cargo/config:2)
        ns = min(args.cpu_sample_reads, n_reads)
        rl = args.read_len
        hb = d_bases[: ns * rl].cpu().numpy()
        hoff = np.arange(ns + 1, dtype=np.uint64) * rl
        omode = so.HPC if mode == pkg.HashMode.Hpc else so.REGULAR
        ncpu, nphys, nhw = usable_cpus(), physical_cores(), os.cpu_count() or 1
        sample = ns * rl

        def clocked(fn, threads, target_s, rate1=None):
            reps = 1 if rate1 is None else max(1, min(64, int(math.ceil(target_s * rate1 * threads * 0.5 / sample))))
            n, sec = fn(threads, reps)
            return n, sample * reps / sec / 1e9, reps

        n1, r1, _ = clocked(lambda th, rp: orc.batch_count_clocked(hb, hoff, args.l, args.k, args.density, omode, threads=th, repeats=rp), 1, 0)
        n2, r2, reps2 = clocked(lambda th, rp: orc.batch_count_clocked(hb, hoff, args.l, args.k, args.density, omode, threads=th, repeats=rp),
                                ncpu, 3.0, r1 * 1e9)
        assert n1 == n2
        # the reference's own fast path is AVX-512 (HashMode::Simd / HpcSimd): two restatements of it, if the host can -- "doubling" (a log-step
        # scan of our own, round 2) and "reference_shape" (the reference's 16-lane rolling scan with the lane-15 carry and compress-stores,
        # src/nthash_avx512_32.rs:117-151,367-420, src/hpc.rs:74-115); README.md, 'Reading the bench line'
        avx = {}
        try:
            av = so.OracleAvx512()
            if av.supported():
                hp = 1 if mode == pkg.HashMode.Hpc else 0
                for key, variant in (("avx512", 0), ("avx512_reference_shape", 1)):
                    if variant and not av.has_reference_shape():
                        continue
                    fn = lambda th, rp, v=variant: av.batch_count_clocked(hb, hoff, args.l, args.k, args.density, hp, threads=th, repeats=rp, variant=v)  # noqa: E731
                    a1, ra1, _ = clocked(fn, 1, 0)
                    a2, ra2, _ = clocked(fn, ncpu, 3.0, ra1 * 1e9)
                    assert a1 == a2
                    avx[key] = {"mode": "HpcSimd" if hp else "Simd", "value": round(ra1, 3), "cores": 1, "all_cores": round(ra2, 2), "threads": ncpu}
            else:
                avx["avx512"] = "n/a (no AVX-512 F/BW/VL/VBMI2)"
        except Exception as e:  # the baseline must never break the bench line
            avx.setdefault("avx512", "n/a (%s)" % type(e).__name__)
        cpu = {"value": round(r1, 4), "unit": "Gbp/s", "cores": 1, "kind": "port",
               "sample": "first %d reads (%.2f Gbp) of the workload, count-only" % (ns, sample / 1e9),
               "all_cores": round(r2, 3), "threads": ncpu, "host_threads": nhw, "host_cores": nphys}
        cpu.update(avx)

    if rank == 0:
        value = tot_bases * args.steps / dt / 1e9
        line = {
            "metric": "input bases/s (Gbp/s) for k-min-mer extraction, l=%d k=%d d=%g" % (args.l, args.k, args.density),
            "value": round(value, 2), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "contexts": 2 if two_ctx is not None else 1,
            "one_context": {"value": round(tot_bases * args.steps / dt_one / 1e9, 2), "ms_per_step": round(dt_one / args.steps * 1e3, 3)},
            "config": dict({"workload": "%s, HashMode::%s, l=%d k=%d d=%g; inputs resident in HBM" % (
                wl_text, "Hpc" if mode == pkg.HashMode.Hpc else "Regular", args.l, args.k, args.density),
                "name": {"c2": "BASELINE configs[1]", "ont": "BASELINE configs[2]", "hifi": "BASELINE configs[3]"}[args.workload], "mode": args.mode,
                "largest_shard_over_mean": balance}, **shard_info),
            "counts": {"bases": tot_bases, "minimizers": tot_min, "kminmers": tot_km, "xor_hash_rank0": counts["xor_hash"]},
            "verified_vs_oracle": verified,
            "other_mode": other_line, "other_configs": other_configs, "standalone_hpc": standalone_hpc, "minimizers_only": minimizers_only,
            "end_to_end": e2e, "downstream_count": count_line, "collective": collective, "per_rank": rank_spread,
            "roofline": roofline, "issue": issue, "clocks": clocks, "cpu_baseline": cpu,
        }
        line = {k: v for k, v in line.items() if v is not None or k in ("vs_baseline",)}  # (absent legs leave no key behind)
        out = json.dumps(line, separators=(",", ":"))
        # the driver keeps the tail of stdout only: a line that outgrew 4 KB sheds its optional sections (never the measurement) and says so
        for drop in ("minimizers_only", "standalone_hpc", "other_configs", "end_to_end", "clocks", "issue"):
            if len(out) <= 4096:
                break
            line.pop(drop, None)
            line["truncated"] = line.get("truncated", []) + [drop]
            out = json.dumps(line, separators=(",", ":"))
        print(out, flush=True)
    if dist is not None:
        barrier()
        try:
            dist.destroy_process_group()
        except Exception:  # (a communicator that never came up: nothing to tear down)
            pass
    eng.close()


if __name__ == "__main__":
    main()
