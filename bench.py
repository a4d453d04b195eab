#!/usr/bin/env python3
"""bench.py -- input bases/s (Gbp/s) of k-min-mer extraction, l=31 k=10 d=0.01, on MI355X.

Workload (BASELINE.json configs[1]): per GPU, 1 000 000 synthetic uniform-random ACGT
reads of 10 kbp (10 Gbp of ASCII, generated in HBM by the library's splitmix64 generator -- the
reference's benches/bench.rs:19-31 convention).  A "step" is one pass of the whole hot path
(s2k_extract_device: minimizer kernel + scans + k-min-mer kernel) over that resident batch.  Reads
shard across GPUs with no data-path collective (weak scaling: every rank owns its own 10 Gbp); the
only cross-GPU traffic is an all-reduce of the count vector.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes of the dominant kernel / its HIP-event duration, vs HBM peak
  cpu_baseline -- the CPU oracle (a port of the reference's scalar path) timed on this host
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--l", type=int, default=31)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--mode", choices=["hpc", "regular"], default="hpc",
                    help="headline HashMode: hpc = the full fused path (HPC + ntHash + select + emit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=120_000, help="reads of the same workload timed on the CPU (1 thread)")
    ap.add_argument("--verify-reads", type=int, default=300)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: map every rank to GPU 0")
    args = ap.parse_args()

    import torch
    from s2k_loader import import_package

    pkg = import_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    eng = pkg.Engine(local_rank)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)  # one stream for torch events and the library's kernels

    n_reads, rl = args.reads, args.read_len
    n_bases = n_reads * rl
    mode = pkg.HashMode.Hpc if args.mode == "hpc" else pkg.HashMode.Regular
    other = pkg.HashMode.Regular if args.mode == "hpc" else pkg.HashMode.Hpc

    # ---- inputs resident in HBM before the timed region -----------------------------------------
    d_bases = torch.empty(n_bases + 256, dtype=torch.uint8, device=dev)
    d_off = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * rl
    eng.synth_bases_device(1, rank * n_bases, n_bases, d_bases.data_ptr())  # each rank owns a distinct shard
    cap = int(n_bases * (2.4 * args.density)) + 1_000_000
    outs = {"km_off": torch.empty(n_reads + 1, dtype=torch.int64, device=dev), "hash": torch.empty(cap, dtype=torch.int64, device=dev),
            "start": torch.empty(cap, dtype=torch.int32, device=dev), "end": torch.empty(cap, dtype=torch.int32, device=dev),
            "rev": torch.empty(cap, dtype=torch.uint8, device=dev)}
    o = pkg.DeviceOut()
    o.km_capacity = cap
    o.km_off, o.hash, o.start, o.end, o.rev = (outs[x].data_ptr() for x in ("km_off", "hash", "start", "end", "rev"))
    torch.cuda.synchronize(dev)

    def step(m, sync=False):
        return eng.extract_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases, args.l, args.k, args.density, m, o, sync=sync)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(m, steps, warmup):
        for _ in range(warmup):
            step(m)
        eng.sync()
        eng.enable_timing(True)  # HIP events around the kernels, on the same stream
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(m)
        counts = eng.sync()
        barrier()
        dt = time.perf_counter() - t0
        k_ms, k_n = eng.timing_total(1)
        km_ms, _ = eng.timing_total(2)
        all_ms, _ = eng.timing_total(0)
        eng.enable_timing(False)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, counts, k_ms / max(k_n, 1), km_ms / max(k_n, 1), all_ms / max(k_n, 1)

    dt, counts, min_ms, km_ms, pipe_ms = timed(mode, args.steps, args.warmup)
    assert counts["path"] == 0, "the tiled HIP kernels must be the ones measured"
    # ---- verification outside the timed region: a read sample against the oracle ------------------
    verified = None
    if rank == 0 and args.verify_reads > 0:
        from oracle import s2k_oracle as so

        orc = so.get()
        nv = min(args.verify_reads, n_reads)
        hb = d_bases[: nv * rl].cpu().numpy()
        assert (hb[: 4096] == orc.synth_bases(1, 0, 4096)).all()
        ref = orc.batch(hb, np.arange(nv + 1, dtype=np.uint64) * rl, args.l, args.k, args.density,
                        so.HPC if mode == pkg.HashMode.Hpc else so.REGULAR, threads=4)
        nk = ref["n"]
        verified = bool((outs["km_off"][: nv + 1].cpu().numpy().view(np.uint64) == ref["km_off"]).all()
                        and (outs["hash"][:nk].cpu().numpy().view(np.uint64) == ref["hash"]).all()
                        and (outs["start"][:nk].cpu().numpy().view(np.uint32) == ref["start"]).all()
                        and (outs["end"][:nk].cpu().numpy().view(np.uint32) == ref["end"]).all()
                        and (outs["rev"][:nk].cpu().numpy() == ref["rev"]).all())
        assert verified, "GPU output differs from the oracle on the verification sample"
    # the other scalar HashMode, for the record (not the headline)
    dt2, counts2, min_ms2, km_ms2, pipe_ms2 = timed(other, max(2, args.steps // 2), 1)

    # whole-job counts: the only collective on this path (RCCL all-reduce of a few words)
    import importlib.util

    spec = importlib.util.spec_from_file_location("s2k_sharding", os.path.join(ROOT, "rust-seq2kminmers_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    tot = sharding.allreduce_counts(counts, dist, dev)
    tot_bases, tot_min, tot_km = tot["n_bases"], tot["n_minimizers"], tot["n_kminmers"]

    # ---- roofline of the dominant kernel (tiled minimizer kernel), per launch ---------------------
    # B = N_bases*1 + N_kminmers*17 + 16*(N_reads+1)   (each input byte once, each k-min-mer once)
    alg_bytes = counts["n_bases"] + 17 * counts["n_kminmers"] + 16 * (n_reads + 1)
    achieved = alg_bytes / (min_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("mode") == args.mode and tj.get("n_bases") == n_bases:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "tile_minimizer_kernel<31,%s>" % ("hpc" if mode == pkg.HashMode.Hpc else "regular"),
                "kernel_ms": round(min_ms, 3), "kminmer_kernel_ms": round(km_ms, 3), "pipeline_ms": round(pipe_ms, 3),
                "algorithmic_bytes_per_launch": int(alg_bytes), "bytes_per_base": round(alg_bytes / n_bases, 4)}

    # ---- CPU baseline: the oracle = scalar port of the reference, on this host's cores --------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N=1 only (the host cores are shared by all ranks)
        from oracle import s2k_oracle as so

        orc = so.Oracle(native=True)  # -O3 -march=native, like the reference's -Ctarget-cpu=native (.cargo/config:2)
        ns = min(args.cpu_sample_reads, n_reads)
        hb = d_bases[: ns * rl].cpu().numpy()
        hoff = np.arange(ns + 1, dtype=np.uint64) * rl
        omode = so.HPC if mode == pkg.HashMode.Hpc else so.REGULAR
        t0 = time.perf_counter()
        n1 = orc.batch_count_timed(hb, hoff, args.l, args.k, args.density, omode, threads=1)
        t1 = time.perf_counter() - t0
        ncpu = os.cpu_count() or 1
        t0 = time.perf_counter()
        n2 = orc.batch_count_timed(hb, hoff, args.l, args.k, args.density, omode, threads=ncpu)
        t2 = time.perf_counter() - t0
        assert n1 == n2
        # the reference's own fast path is AVX-512 (HashMode::Simd / HpcSimd): time a restatement of it too, if the host can
        avx = None
        try:
            av = so.OracleAvx512()
            if av.supported():
                hp = 1 if mode == pkg.HashMode.Hpc else 0
                t0 = time.perf_counter()
                a1 = av.batch_count(hb, hoff, args.l, args.k, args.density, hp, threads=1)
                ta1 = time.perf_counter() - t0
                t0 = time.perf_counter()
                a2 = av.batch_count(hb, hoff, args.l, args.k, args.density, hp, threads=ncpu)
                ta2 = time.perf_counter() - t0
                assert a1 == a2
                avx = {"mode": "HpcSimd" if hp else "Simd", "value": round(ns * rl / ta1 / 1e9, 4), "cores": 1,
                       "all_cores": {"value": round(ns * rl / ta2 / 1e9, 4), "cores": ncpu},
                       "note": "oracle/s2k_oracle_avx512.c: restatement of the Simd-mode semantics (strict <, f32 bound), not the reference's code"}
            else:
                avx = "n/a (host CPU lacks AVX-512 F/BW/VL/VBMI2)"
        except Exception as e:  # the baseline must never break the bench line
            avx = "n/a (%s)" % type(e).__name__
        cpu = {"value": round(ns * rl / t1 / 1e9, 4), "unit": "Gbp/s", "cores": 1, "kind": "port",
               "sample": "first %d reads (%.2f Gbp) of the same synthetic workload, count-only iteration as in src/main.rs:65-76, "
                         "oracle/s2k_oracle.c built -O3 -march=native" % (ns, ns * rl / 1e9),
               "all_cores": {"value": round(ns * rl / t2 / 1e9, 4), "cores": ncpu},
               "avx512": avx,
               "reference_published": "README.md:23: scalar ~0.1-0.2 GB/s, AVX-512 ~1 GB/s per thread (ntHash only, unstated CPU)"}

    if rank == 0:
        value = tot_bases * args.steps / dt / 1e9
        line = {
            "metric": "input bases/s (Gbp/s) for k-min-mer extraction, l=%d k=%d d=%g" % (args.l, args.k, args.density),
            "value": round(value, 2), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%d x %d bp uniform-random ACGT reads per GPU (%.1f Gbp/GPU), HashMode::%s, l=%d k=%d d=%g; "
                                   "inputs resident in HBM" % (n_reads, rl, n_bases / 1e9, "Hpc" if mode == pkg.HashMode.Hpc else "Regular",
                                                               args.l, args.k, args.density),
                       "reads_per_gpu": n_reads, "read_len": rl, "mode": args.mode, "sharding": "reads, contiguous per rank"},
            "counts": {"bases": tot_bases, "minimizers": tot_min, "kminmers": tot_km, "xor_hash_rank0": counts["xor_hash"]},
            "verified_vs_oracle": verified,
            "other_mode": {"mode": "regular" if args.mode == "hpc" else "hpc",
                           "value": round(counts2["n_bases"] * world * max(2, args.steps // 2) / dt2 / 1e9, 2), "unit": "Gbp/s",
                           "kernel_ms": round(min_ms2, 3), "kminmer_kernel_ms": round(km_ms2, 3),
                           "roofline_frac": round((counts2["n_bases"] + 17 * counts2["n_kminmers"] + 16 * (n_reads + 1)) / (min_ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
