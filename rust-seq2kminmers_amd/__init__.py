"""rust-seq2kminmers_amd -- host-side mirror of the reference's seq -> k-min-mer interface over the
MI355X C ABI (include/s2k.h, built from csrc/ into csrc/libs2k.so).

Reference interface mirrored here (names, argument meaning, error behaviour):
    KminmersIterator::new(seq, l, k, density, mode)   src/lib.rs:89-131
    impl Iterator { type Item = KminmerHash }         src/lib.rs:179-270
    KminmerHash{hash,start,end,offset,rev}, ==/ord by hash only   src/kminmer.rs:128-135,181-204
    HashMode{Regular,Hpc,Simd,HpcSimd}                src/lib.rs:21-27
    hpc(), encode_rle_simd()                          src/hpc.rs:28-41, 44-147

There is no CPU fallback: importing works anywhere (so the CPU test tier can check the ABI), but
creating an Engine without a GPU raises.  The directory name carries a hyphen (as the task names it),
so load it with `import_package()` from s2k_loader.py or importlib.
"""
import ctypes as C
import enum
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libs2k.so")

ABI_VERSION = 2  # include/s2k.h S2K_ABI_VERSION this mirror was written against

ABI_SYMBOLS = [
    "s2k_trim", "s2k_density_for_bound", "s2k_abi_version", "s2k_device_count", "s2k_create", "s2k_destroy", "s2k_set_stream", "s2k_chain_after", "s2k_set_host_batch", "s2k_strerror",
    "s2k_last_error", "s2k_hash_bound", "s2k_extract", "s2k_result_free", "s2k_extract_device", "s2k_sync",
    "s2k_hpc_device", "s2k_hpc_device_ex", "s2k_count_device", "s2k_partition_device", "s2k_synth_bases_device", "s2k_synth_hifi_lengths", "s2k_synth_hifi_device", "s2k_last_kernel_ms", "s2k_enable_timing", "s2k_timing_total", "s2k_fastx_open", "s2k_fastx_next", "s2k_fastx_close", "s2k_run_file", "s2k_fastx_parse_device",
]


class HashMode(enum.IntEnum):  # src/lib.rs:21-27
    Regular = 0
    Hpc = 1
    Simd = 2
    HpcSimd = 3


FLAG_WANT_MINIMIZERS = 1
FLAG_FORCE_SERIAL = 2
FLAG_LEGACY_PATH = 8  # round-2 records (16 B with the read index, per-read scans) instead of the descriptor path
FLAG_NO_PACK2 = 4  # s2k_extract: bases as ASCII over PCIe instead of 2-bit packed + exception list
HPC_RLE_ALPHABET = 1  # s2k_hpc_device_ex: the run rule of encode_rle (src/hpc.rs:14)


class S2kError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("s2k status %d: %s" % (status, msg))
        self.status = status


class Params(C.Structure):
    _fields_ = [("l", C.c_uint32), ("k", C.c_uint32), ("density", C.c_double), ("mode", C.c_int32), ("flags", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_minimizers", C.c_uint64), ("n_kminmers", C.c_uint64),
                ("xor_hash", C.c_uint64), ("hash_bound", C.c_uint32), ("path", C.c_uint32)]

    def as_dict(self):
        return {f: int(getattr(self, f)) for f, _ in self._fields_}


_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


class Result(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_kminmers", C.c_uint64), ("km_off", _u64p), ("hash", _u64p), ("start", _u32p),
                ("end", _u32p), ("rev", _u8p), ("n_minimizers", C.c_uint64), ("mn_off", _u64p), ("mn_j", _u32p),
                ("mn_jend", _u32p), ("mn_hash", _u32p), ("counts", Counts), ("_owner", C.c_void_p)]


class DeviceOut(C.Structure):
    _fields_ = [("km_capacity", C.c_uint64), ("km_off", C.c_void_p), ("hash", C.c_void_p), ("start", C.c_void_p),
                ("end", C.c_void_p), ("rev", C.c_void_p), ("mn_capacity", C.c_uint64), ("mn_off", C.c_void_p),
                ("mn_j", C.c_void_p), ("mn_jend", C.c_void_p), ("mn_hash", C.c_void_p)]


_lib = None


def load_library(path=None):
    """dlopen the C-ABI library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("S2K_LIB") or LIB_PATH  # S2K_LIB: tests load the host-ASan build of the same library
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and the system ROCm has
    # another; whichever is mapped first serves both (same SONAME).  Import torch first so that device
    # memory and streams handed over by torch belong to the runtime this library talks to.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise ImportError("%s is missing: build it with `make -C %s` (or __graft_entry__.build())" % (p, os.path.dirname(p)))
    L = C.CDLL(p)
    L.s2k_abi_version.restype = C.c_int
    if L.s2k_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, this binding needs %d: rebuild it (make -C %s)" % (p, L.s2k_abi_version(), ABI_VERSION, os.path.dirname(p)))
    L.s2k_trim.argtypes = [C.c_void_p]
    L.s2k_device_count.restype = C.c_int
    L.s2k_create.restype = C.c_void_p
    L.s2k_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.s2k_destroy.argtypes = [C.c_void_p]
    L.s2k_destroy.restype = None
    L.s2k_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.s2k_chain_after.argtypes = [C.c_void_p, C.c_void_p]
    L.s2k_set_host_batch.argtypes = [C.c_void_p, C.c_uint64]
    L.s2k_set_host_batch.restype = C.c_int
    L.s2k_strerror.restype = C.c_char_p
    L.s2k_strerror.argtypes = [C.c_int]
    L.s2k_last_error.restype = C.c_char_p
    L.s2k_last_error.argtypes = [C.c_void_p]
    L.s2k_hash_bound.restype = C.c_uint32
    L.s2k_hash_bound.argtypes = [C.c_double]
    L.s2k_density_for_bound.restype = C.c_double
    L.s2k_density_for_bound.argtypes = [C.c_uint32]
    L.s2k_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Params), C.POINTER(Result)]
    L.s2k_result_free.argtypes = [C.POINTER(Result)]
    L.s2k_result_free.restype = None
    L.s2k_extract_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(Params),
                                     C.POINTER(DeviceOut), C.POINTER(Counts)]
    L.s2k_sync.argtypes = [C.c_void_p, C.POINTER(Counts)]
    L.s2k_hpc_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.s2k_hpc_device_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.s2k_count_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.s2k_partition_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    L.s2k_synth_bases_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.s2k_synth_hifi_lengths.restype = None
    L.s2k_synth_hifi_lengths.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.s2k_synth_hifi_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.s2k_last_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.s2k_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.s2k_timing_total.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.s2k_fastx_open.restype = C.c_void_p
    L.s2k_fastx_open.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.s2k_fastx_next.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.s2k_fastx_close.argtypes = [C.c_void_p]
    L.s2k_fastx_close.restype = None
    L.s2k_run_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Params), C.c_uint64, C.POINTER(Counts), C.POINTER(C.c_double)]
    L.s2k_fastx_parse_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    if path is None:
        _lib = L
    return L


def hash_bound(density):
    """(density * u32::MAX as f64) as u32 -- src/lib.rs:91"""
    return int(load_library().s2k_hash_bound(float(density)))


def _as_u8(seq):
    if isinstance(seq, str):
        seq = seq.encode("latin-1")
    if isinstance(seq, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def pack_reads(reads):
    """list of byte strings -> (bases u8[total], read_off u64[n+1])"""
    arrs = [_as_u8(r) for r in reads]
    off = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        off[1:] = np.cumsum([len(a) for a in arrs], dtype=np.uint64)
    bases = np.concatenate(arrs) if arrs and int(off[-1]) else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(bases, dtype=np.uint8), off


class Engine:
    """One s2k_ctx (one per thread and device, like one KminmersIterator per thread in src/main.rs:65-79)."""

    def __init__(self, device=0):
        self.lib = load_library()
        st = C.c_int(0)
        self.ctx = self.lib.s2k_create(int(device), C.byref(st))
        if not self.ctx:
            raise S2kError(st.value, self.lib.s2k_strerror(st.value).decode())
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.s2k_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise S2kError(st, self.lib.s2k_strerror(st).decode() + " -- " + self.lib.s2k_last_error(self.ctx).decode())

    def set_stream(self, hip_stream_handle):
        self._check(self.lib.s2k_set_stream(self.ctx, C.c_void_p(int(hip_stream_handle))))

    def chain_after(self, prev):
        """Double buffering with two engines on one device: this engine's minimizer kernels wait for those of `prev`'s most recent call (and
        nothing else does), so the tail of one call runs beside the first chunk of the next.  Chain both ways and alternate the calls; None unlinks."""
        self._check(self.lib.s2k_chain_after(self.ctx, prev.ctx if prev is not None else None))

    def set_host_batch(self, bases):
        """Bases per sub-batch of `extract` (host buffers): H2D / kernels / D2H of consecutive sub-batches overlap."""
        self._check(self.lib.s2k_set_host_batch(self.ctx, int(bases)))

    def trim(self):
        """release what the context keeps between calls (idle host result blocks, device workspace)"""
        self._check(self.lib.s2k_trim(self.ctx))

    def enable_timing(self, on=True):
        self._check(self.lib.s2k_enable_timing(self.ctx, 1 if on else 0))

    def last_kernel_ms(self, which):
        ms = C.c_float(0)
        self._check(self.lib.s2k_last_kernel_ms(self.ctx, which, C.byref(ms)))
        return float(ms.value)

    def timing_total(self, which):
        """(sum of HIP-event ms, n_calls) since enable_timing(True); which: 0 pipeline, 1 minimizer kernel, 2 k-min-mer kernel"""
        ms, n = C.c_double(0), C.c_uint32(0)
        self._check(self.lib.s2k_timing_total(self.ctx, which, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    # ---- host-buffer API (s2k_extract) ---------------------------------------------------------------
    def extract(self, bases, read_off, l, k, density, mode=HashMode.Hpc, want_minimizers=False, force_serial=False, pack2=True, legacy=False):
        """Batch counterpart of `for r in reads: KminmersIterator::new(r,l,k,d,mode).collect()`.
        Returns a dict of numpy arrays (copies)."""
        bases = _as_u8(bases)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        if len(read_off) == 0:
            raise ValueError("read_off needs n_reads + 1 entries (at least one)")
        if int(read_off[-1]) > len(bases):
            raise ValueError("read_off[-1] = %d lies past the end of bases (%d bytes)" % (int(read_off[-1]), len(bases)))
        n_reads = len(read_off) - 1  # read_off[0] may be > 0: the library rebases the offsets (bases[read_off[0]:read_off[-1]] is used)
        flags = (FLAG_WANT_MINIMIZERS if want_minimizers else 0) | (FLAG_FORCE_SERIAL if force_serial else 0) | (0 if pack2 else FLAG_NO_PACK2) | (FLAG_LEGACY_PATH if legacy else 0)
        p = Params(int(l), int(k), float(density), int(mode), flags)
        res = Result()
        self._check(self.lib.s2k_extract(self.ctx, bases.ctypes.data_as(C.c_void_p), read_off.ctypes.data_as(C.c_void_p),
                                         n_reads, C.byref(p), C.byref(res)))
        try:
            nk, nm = int(res.n_kminmers), int(res.n_minimizers)

            def take(ptr, n, dt):
                if n == 0:
                    return np.zeros(0, dtype=dt)
                return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

            out = {
                "n": nk, "km_off": take(res.km_off, n_reads + 1, np.uint64), "hash": take(res.hash, nk, np.uint64),
                "start": take(res.start, nk, np.uint32), "end": take(res.end, nk, np.uint32), "rev": take(res.rev, nk, np.uint8),
                "counts": res.counts.as_dict(),
            }
            if want_minimizers:
                out.update({"n_minimizers": nm, "mn_off": take(res.mn_off, n_reads + 1, np.uint64),
                            "mn_j": take(res.mn_j, nm, np.uint32), "mn_jend": take(res.mn_jend, nm, np.uint32),
                            "mn_hash": take(res.mn_hash, nm, np.uint32)})
            return out
        finally:
            self.lib.s2k_result_free(C.byref(res))

    def extract_reads(self, reads, l, k, density, mode=HashMode.Hpc, **kw):
        bases, off = pack_reads(reads)
        return self.extract(bases, off, l, k, density, mode, **kw)

    # ---- device-resident API (s2k_extract_device); pointers are integers (e.g. torch .data_ptr()) ------
    def extract_device(self, d_bases, d_read_off, n_reads, n_bases, l, k, density, mode, out, sync=True, flags=0):
        p = Params(int(l), int(k), float(density), int(mode), int(flags))
        cnt = Counts()
        st = self.lib.s2k_extract_device(self.ctx, C.c_void_p(d_bases), C.c_void_p(d_read_off), n_reads, n_bases, C.byref(p),
                                         C.byref(out), C.byref(cnt) if sync else None)
        self._check(st)
        return cnt.as_dict() if sync else None

    def sync(self):
        cnt = Counts()
        self._check(self.lib.s2k_sync(self.ctx, C.byref(cnt)))
        return cnt.as_dict()

    def synth_bases_device(self, seed, first_base, n, d_ptr):
        self._check(self.lib.s2k_synth_bases_device(self.ctx, seed, first_base, n, C.c_void_p(d_ptr)))

    def synth_hifi_lengths(self, seed, r0, n_reads):
        """lengths of the HiFi-like synthetic reads r0 .. r0+n_reads-1 (BASELINE configs[3]); host array"""
        out = np.empty(n_reads, dtype=np.uint64)
        self.lib.s2k_synth_hifi_lengths(seed, r0, n_reads, C.c_void_p(out.ctypes.data))
        return out

    def synth_hifi_device(self, seed, r0, n_reads, d_read_off_ptr, d_bases_ptr):
        """read r0+i is written at d_bases + d_read_off[i] (device pointers; read_off = prefix of synth_hifi_lengths)"""
        self._check(self.lib.s2k_synth_hifi_device(self.ctx, seed, r0, n_reads, C.c_void_p(d_read_off_ptr), C.c_void_p(d_bases_ptr)))

    def run_file(self, path, l, k, density, mode=HashMode.Regular, batch_bases=256 << 20, flags=0):
        """FASTA/FASTQ file -> totals; the file mode of the reference's CLI (src/main.rs:51-83) on the GPU."""
        p = Params(int(l), int(k), float(density), int(mode), int(flags))
        cnt, sec = Counts(), C.c_double(0)
        self._check(self.lib.s2k_run_file(self.ctx, os.fsencode(path), C.byref(p), batch_bases, C.byref(cnt), C.byref(sec)))
        d = cnt.as_dict()
        d["seconds"] = float(sec.value)
        return d

    def parse_fastx_device(self, d_text, n_bytes, fastq, d_bases=0, bases_capacity=0, d_read_off=0, off_capacity=0):
        """FASTA/FASTQ text in HBM -> (bases, read_off) in HBM (s2k_fastx_parse_device).  Returns (status, n_reads,
        n_bases); status 7 (capacity) still reports the sizes needed."""
        nr, nb = C.c_uint64(0), C.c_uint64(0)
        st = self.lib.s2k_fastx_parse_device(self.ctx, C.c_void_p(d_text), n_bytes, 1 if fastq else 0, C.c_void_p(d_bases or 0),
                                             bases_capacity, C.c_void_p(d_read_off or 0), off_capacity, C.byref(nr), C.byref(nb))
        if st not in (0, 7):
            self._check(st)
        return st, int(nr.value), int(nb.value)

    def hpc_device(self, d_bases, d_read_off, n_reads, n_bases, d_hpc_off, d_hpc, d_pos, capacity, rle=False):
        """rle=True: the run rule of `encode_rle` (only ACTGactgNn collapse, src/hpc.rs:14); else any repeated byte collapses."""
        n = C.c_uint64(0)
        st = self.lib.s2k_hpc_device_ex(self.ctx, C.c_void_p(d_bases), C.c_void_p(d_read_off), n_reads, n_bases,
                                        HPC_RLE_ALPHABET if rle else 0, C.c_void_p(d_hpc_off), C.c_void_p(d_hpc or 0),
                                        C.c_void_p(d_pos or 0), capacity, C.byref(n))
        self._check(st)
        return int(n.value)


    def count_device(self, d_hash, n, d_keys=0, d_counts=0, capacity=0):
        """distinct k-min-mer hashes of d_hash[0..n) and their multiplicities (unordered); returns n_distinct"""
        nd = C.c_uint64(0)
        st = self.lib.s2k_count_device(self.ctx, C.c_void_p(d_hash or 0), n, C.c_void_p(d_keys or 0), C.c_void_p(d_counts or 0), capacity,
                                       C.byref(nd))
        if st != 7 or capacity:
            self._check(st)
        return int(nd.value)

    def partition_device(self, d_hash, n, n_parts, d_out, d_part_off):
        """keys grouped by hash prefix into n_parts ranges (the send buffers of the all-to-all)"""
        self._check(self.lib.s2k_partition_device(self.ctx, C.c_void_p(d_hash or 0), n, n_parts, C.c_void_p(d_out or 0), C.c_void_p(d_part_off)))


class FastxReader:
    """FASTA/FASTQ batches (bases back to back + read_off), the front-end that replaces parallel_fastx
    (src/main.rs:79).  Host-only: works without a GPU."""

    def __init__(self, path):
        self.lib = load_library()
        st = C.c_int(0)
        self.rd = self.lib.s2k_fastx_open(os.fsencode(path), C.byref(st))
        if not self.rd:
            raise S2kError(st.value, "cannot open %s" % path)

    def next_batch(self, max_bases=256 << 20, max_reads=0):
        """-> (bases u8 copy, read_off u64 copy) or None at end of file"""
        b, o, n = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        st = self.lib.s2k_fastx_next(self.rd, max_bases, max_reads, C.byref(b), C.byref(o), C.byref(n))
        if st != 0:
            raise S2kError(st, "malformed FASTA/FASTQ record")
        if n.value == 0:
            return None
        off = np.ctypeslib.as_array(C.cast(o, _u64p), shape=(n.value + 1,)).copy()
        nb = int(off[-1])
        bases = np.ctypeslib.as_array(C.cast(b, _u8p), shape=(max(nb, 1),))[:nb].copy()
        return bases, off

    def close(self):
        if getattr(self, "rd", None):
            self.lib.s2k_fastx_close(self.rd)
            self.rd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


class KminmerHash:
    """src/kminmer.rs:128-135; equality and ordering by `hash` only (src/kminmer.rs:181-204)."""
    __slots__ = ("hash", "start", "end", "offset", "rev")

    def __init__(self, hash, start, end, offset, rev):
        self.hash, self.start, self.end, self.offset, self.rev = hash, start, end, offset, rev

    @classmethod
    def new_from_hash(cls, hash, start, end, offset, rev):  # src/kminmer.rs:169-177
        return cls(hash, start, end, offset, rev)

    def get_hash(self):  # src/kminmer.rs:162-164
        return self.hash

    def __eq__(self, o):
        return self.hash == o.hash

    def __lt__(self, o):
        return self.hash < o.hash

    def __hash__(self):
        return hash(self.hash)

    def __repr__(self):  # Debug formatting of the Rust struct
        return "KminmerHash { hash: %d, start: %d, end: %d, offset: %d, rev: %s }" % (
            self.hash, self.start, self.end, self.offset, "true" if self.rev else "false")


class KminmersIterator:
    """Per-read facade with the reference's constructor signature (src/lib.rs:89) and Iterator shape
    (src/lib.rs:179-181).  One GPU call per read is the wrong granularity for throughput -- use
    Engine.extract for batches; this class exists so reference-style call sites and tests read the same.
    Parameter violations that panic in the reference raise S2kError here."""

    def __init__(self, seq, l, k, density, mode=HashMode.Hpc, engine=None):
        eng = engine or default_engine()
        r = eng.extract_reads([seq], l, k, density, mode)
        self._r = r
        self._i = 0

    @classmethod
    def new(cls, seq, l, k, density, mode, engine=None):
        return cls(seq, l, k, density, mode, engine)

    def __iter__(self):
        return self

    def __next__(self):
        i = self._i
        if i >= self._r["n"]:
            raise StopIteration
        self._i += 1
        r = self._r
        return KminmerHash(int(r["hash"][i]), int(r["start"][i]), int(r["end"][i]), i, bool(r["rev"][i]))


class _MinimizerTriples:
    """The crate's minimizer iterators (re-exported at src/lib.rs:6-13) over S2K_FLAG_WANT_MINIMIZERS: one GPU call per
    sequence -- the wrong granularity for throughput (use Engine.extract(..., want_minimizers=True) for batches); these
    classes exist so reference-style call sites read the same.  Like the reference they take the u32 `hash_bound`, not a
    density.  l > len(seq) raises as KSizeOutOfRange does (src/nthash_hpc.rs:117-121); len(seq) == l is accepted as in the reference."""
    _mode = None

    def __init__(self, seq, l, hash_bound, engine=None):
        a = _as_u8(seq)
        if l > len(a):
            raise S2kError(2, "K size %d is out of range for the given sequence size %d" % (l, len(a)))  # src/nthash_hpc.rs:19-22
        eng = engine or default_engine()
        d = float(eng.lib.s2k_density_for_bound(int(hash_bound)))
        keep = None
        if len(a) == l:
            # seq.len() == l is accepted by the reference's iterators (only l > len is KSizeOutOfRange, src/nthash_hpc.rs:117-121):
            # the scalar Hpc iterator then yields nothing (its only l-mer is the last one, dropped at :265-267); the two SIMD
            # iterators may yield the single l-mer.  KminmersIterator never makes this call (src/lib.rs:97), and the kernels
            # follow it (len <= l: nothing) -- so the l-mer is computed ON THE GPU on the sequence plus one base that starts a new
            # run, and only what lies inside the original sequence is kept.
            if self._mode == HashMode.Hpc:
                self._j = self._jend = self._h = np.empty(0, dtype=np.uint32)
                self._i = 0
                return
            a = np.concatenate((a, np.array([ord("C") if a[-1] == ord("A") else ord("A")], dtype=np.uint8)))
            keep = l
        r = eng.extract_reads([a], l, 1, d, self._mode, want_minimizers=True)
        assert r["counts"]["hash_bound"] == int(hash_bound)
        self._j, self._jend, self._h = r["mn_j"], r["mn_jend"], r["mn_hash"]
        if keep is not None:  # the l-mer that starts at 0 and ends inside the original sequence
            m = (self._j == 0) & (self._jend < keep)
            self._j, self._jend, self._h = self._j[m], self._jend[m], self._h[m]
        self._i = 0

    @classmethod
    def new(cls, seq, l, hash_bound, engine=None):
        return cls(seq, l, hash_bound, engine)

    def __iter__(self):
        return self

    def __len__(self):
        return len(self._h) - self._i

    def _item(self, i):
        return int(self._j[i]), int(self._jend[i]), int(self._h[i])

    def __next__(self):
        if self._i >= len(self._h):
            raise StopIteration
        self._i += 1
        return self._item(self._i - 1)


class NtHashHPCIterator(_MinimizerTriples):
    """src/nthash_hpc.rs:99-283: Item = (start, end, hash) in original space; `<=` against the bound; the last HPC l-mer of the
    sequence is never yielded (:265-267); end = last base of the l-mer's last run (:281)."""
    _mode = HashMode.Hpc


class NtHashHPCSIMDIterator(_MinimizerTriples):
    """src/nthash_hpc_simd.rs:17-68: Item = (start, end, hash); the result semantics of the AVX-512 path: strict `<` against a
    bound re-derived through f32, end = START of the last run (:64), the last l-mer kept, the final block of 16 lost when the
    number of l-mers is a multiple of 16."""
    _mode = HashMode.HpcSimd


class NtHashSIMDIterator(_MinimizerTriples):
    """src/nthash_avx512_32.rs:14-164: Item = (pos, hash) over the sequence as it is (no homopolymer compression)."""
    _mode = HashMode.Simd

    def _item(self, i):
        return int(self._j[i]), int(self._h[i])


class RegularMinimizers(_MinimizerTriples):
    """Not a type of the crate: the (j, jend, hash) triples that the Regular arm of KminmersIterator::next makes from the external
    nthash32::NtHashIterator (src/lib.rs:215-230), for symmetry with the three above."""
    _mode = HashMode.Regular


def _hpc_gpu(seq, engine, rle):
    import torch

    eng = engine or default_engine()
    a = _as_u8(seq)
    n = len(a)
    dev = torch.device("cuda", eng.device)
    d_b = torch.from_numpy(a.copy()).to(dev) if n else torch.zeros(1, dtype=torch.uint8, device=dev)
    d_off = torch.tensor([0, n], dtype=torch.int64, device=dev)
    d_ho = torch.zeros(2, dtype=torch.int64, device=dev)
    d_h = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
    d_p = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    r = eng.hpc_device(d_b.data_ptr(), d_off.data_ptr(), 1, n, d_ho.data_ptr(), d_h.data_ptr(), d_p.data_ptr(), max(n, 1), rle=rle)
    return d_h[:r].cpu().numpy().tobytes(), d_p[:r].cpu().numpy().astype(np.uint32)


def _no_sentinel(a, who):
    # hpc / encode_rle start from prev_char = '#' (src/hpc.rs:9,30): a '#' in the input is dropped or misplaces the next
    # run's position there.  The device op treats '#' as an ordinary byte, so such input is refused instead of answered differently.
    if len(a) and (a == 0x23).any():
        raise ValueError("%s: input contains '#', the reference's start sentinel (src/hpc.rs:9,30); not supported" % who)


def hpc(seq, engine=None):
    """Homopolymer-compressed string of one read (any equal characters collapse): src/hpc.rs:28-41, on the GPU.
    hpc("") == "#" as in the reference (the sentinel is flushed at the end, src/hpc.rs:39-40)."""
    a = _as_u8(seq)
    if len(a) == 0:
        return b"#"
    _no_sentinel(a, "hpc")
    return _hpc_gpu(a, engine, False)[0]


def encode_rle(seq, engine=None):
    """(compressed string, run-start positions) with the rule of src/hpc.rs:7-25: only runs of ACTGactgNn collapse
    (src/hpc.rs:14), every other character stays.  encode_rle("") == ("#", [0]) as in the reference."""
    a = _as_u8(seq)
    if len(a) == 0:
        return b"#", np.zeros(1, dtype=np.uint32)
    _no_sentinel(a, "encode_rle")
    return _hpc_gpu(a, engine, True)


def encode_rle_simd(seq, engine=None):
    """(compressed string, run-start positions) -- src/hpc.rs:44-147 (any equal bytes collapse) -- computed on the GPU."""
    return _hpc_gpu(seq, engine, False)
