"""ctypes loader for the CPU oracle (oracle/s2k_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  Parity status: pinned (see s2k_oracle.h).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REGULAR, HPC, SIMD, HPCSIMD = 0, 1, 2, 3

# whole-run checksums (s2k_oracle_synth_checksums): the fold_* ones are order-sensitive, SUM x[g] (2 g + 1) mod 2^64
CHECKSUM_FIELDS = ("n_minimizers", "n_kminmers", "xor_hash", "sum_start", "sum_end", "n_rev",
                   "fold_hash", "fold_start", "fold_end", "fold_rev", "fold_km_off", "fold_count")

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def build(native=False, out=None):
    """Compile the oracle. `native=True` adds -march=native (used for the timed CPU baseline,
    mirroring the reference's `-Ctarget-cpu=native`, .cargo/config:2)."""
    src = os.path.join(_HERE, "s2k_oracle.c")
    if out is None:
        out = os.path.join(_HERE, "libs2k_oracle.so")
    flags = ["-O3", "-std=c11", "-fPIC", "-shared", "-pthread"]
    if native:
        flags.insert(1, "-march=native")
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "s2k_oracle.h"))
    ):
        tmp = out + ".tmp.%d" % os.getpid()
        subprocess.check_call(["gcc"] + flags + ["-o", tmp, src, "-lm"])
        os.replace(tmp, out)
    return out


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


class Oracle:
    def __init__(self, path=None, native=False):
        if path is None and not native and os.environ.get("S2K_ORACLE_LIB"):
            path = os.environ["S2K_ORACLE_LIB"]  # tests/test_sanitizers.py: the ASan/UBSan build of the same source
        if path is None:
            out = None
            if native:  # -march=native objects must never travel between hosts: key the file by the CPU model
                import hashlib

                model = ""
                try:
                    with open("/proc/cpuinfo") as f:
                        model = next((ln for ln in f if ln.startswith("model name")), "")
                except OSError:
                    pass
                out = os.path.join(_HERE, "libs2k_oracle_native_%s.so" % hashlib.md5(model.encode()).hexdigest()[:8])
            path = build(native=native, out=out)
        L = self.lib = C.CDLL(path)
        L.s2k_oracle_hash_bound.restype = C.c_uint32
        L.s2k_oracle_hash_bound.argtypes = [C.c_double]
        L.s2k_oracle_hash_bound_simd.restype = C.c_uint32
        L.s2k_oracle_hash_bound_simd.argtypes = [C.c_uint32]
        L.s2k_oracle_seed_h.restype = C.c_uint32
        L.s2k_oracle_seed_h.argtypes = [C.c_uint8]
        L.s2k_oracle_seed_rc.restype = C.c_uint32
        L.s2k_oracle_seed_rc.argtypes = [C.c_uint8]
        L.s2k_oracle_mix32.restype = C.c_uint64
        L.s2k_oracle_mix32.argtypes = [C.c_uint32]
        L.s2k_oracle_nthash32_all.restype = C.c_size_t
        L.s2k_oracle_nthash32_all.argtypes = [_u8p, C.c_size_t, C.c_uint, _u32p]
        L.s2k_oracle_hpc.restype = C.c_size_t
        L.s2k_oracle_hpc.argtypes = [_u8p, C.c_size_t, C.c_int, _u8p, _u64p]
        L.s2k_oracle_minimizers.restype = C.c_size_t
        L.s2k_oracle_minimizers.argtypes = [_u8p, C.c_size_t, C.c_uint, C.c_uint32, C.c_int, _u64p, _u64p, _u32p, C.c_size_t]
        L.s2k_oracle_hpc_literal.restype = C.c_size_t
        L.s2k_oracle_hpc_literal.argtypes = [_u8p, C.c_size_t, C.c_uint, C.c_uint32, _u64p, _u64p, _u32p, C.c_size_t]
        L.s2k_oracle_hpc_literal_u64.restype = C.c_size_t
        L.s2k_oracle_hpc_literal_u64.argtypes = [_u8p, C.c_size_t, C.c_uint, C.c_uint64, _u64p, _u64p, C.c_size_t]
        L.s2k_oracle_kminmers.restype = C.c_size_t
        L.s2k_oracle_kminmers.argtypes = [_u8p, C.c_size_t, C.c_uint, C.c_uint, C.c_double, C.c_int, _u64p, _u64p, _u64p, _u8p, C.c_size_t]
        L.s2k_oracle_kminmer_hashes_rolling.restype = C.c_size_t
        L.s2k_oracle_kminmer_hashes_rolling.argtypes = [_u32p, C.c_size_t, C.c_uint, _u64p, _u8p]
        L.s2k_oracle_batch.restype = C.c_uint64
        L.s2k_oracle_batch.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double, C.c_int, C.c_int,
                                       _u64p, _u64p, _u32p, _u32p, _u8p, C.c_uint64]
        L.s2k_oracle_batch_minimizers.restype = C.c_uint64
        L.s2k_oracle_batch_minimizers.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint, C.c_double, C.c_int,
                                                  _u64p, _u32p, _u32p, _u32p, C.c_uint64]
        L.s2k_oracle_synth_bases.restype = None
        L.s2k_oracle_synth_bases.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u8p]
        L.s2k_oracle_synth_checksums.restype = None
        L.s2k_oracle_synth_checksums.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, C.c_double,
                                                 C.c_int, C.c_int, _u64p]
        L.s2k_oracle_hifi_len.restype = C.c_uint64
        L.s2k_oracle_hifi_len.argtypes = [C.c_uint64, C.c_uint64]
        L.s2k_oracle_hifi_lens.restype = None
        L.s2k_oracle_hifi_lens.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u64p]
        L.s2k_oracle_hifi_read.restype = None
        L.s2k_oracle_hifi_read.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u8p]
        L.s2k_oracle_hifi_checksums.restype = None
        L.s2k_oracle_hifi_checksums.argtypes = [C.c_uint64, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double,
                                                C.c_int, C.c_int, _u64p]
        L.s2k_oracle_synth_checksums_off.restype = None
        L.s2k_oracle_synth_checksums_off.argtypes = [C.c_uint64, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double,
                                                     C.c_int, C.c_int, _u64p]

    # -- scalars ---------------------------------------------------------------------------
    def hash_bound(self, d):
        return int(self.lib.s2k_oracle_hash_bound(float(d)))

    def hash_bound_simd(self, b):
        return int(self.lib.s2k_oracle_hash_bound_simd(int(b)))

    def mix32(self, h):
        return int(self.lib.s2k_oracle_mix32(int(h)))

    # -- helpers -----------------------------------------------------------------------------
    @staticmethod
    def _seq(seq):
        if isinstance(seq, (bytes, bytearray)):
            seq = np.frombuffer(bytes(seq), dtype=np.uint8)
        elif isinstance(seq, str):
            seq = np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)
        return np.ascontiguousarray(seq, dtype=np.uint8)

    def nthash32_all(self, seq, l):
        s = self._seq(seq)
        out = np.empty(max(len(s) - l + 1, 0), dtype=np.uint32)
        n = self.lib.s2k_oracle_nthash32_all(_ptr(s, _u8p), len(s), l, _ptr(out, _u32p))
        return out[:n]

    def hpc(self, seq, which=0):
        s = self._seq(seq)
        out = np.empty(len(s) + 1, dtype=np.uint8)
        pos = np.empty(len(s) + 1, dtype=np.uint64)
        r = self.lib.s2k_oracle_hpc(_ptr(s, _u8p), len(s), which, _ptr(out, _u8p), _ptr(pos, _u64p))
        return out[:r].tobytes(), pos[:r].copy()

    def minimizers(self, seq, l, bound, mode):
        s = self._seq(seq)
        n = self.lib.s2k_oracle_minimizers(_ptr(s, _u8p), len(s), l, bound, mode, None, None, None, 0)
        j = np.empty(n, dtype=np.uint64)
        je = np.empty(n, dtype=np.uint64)
        h = np.empty(n, dtype=np.uint32)
        self.lib.s2k_oracle_minimizers(_ptr(s, _u8p), len(s), l, bound, mode, _ptr(j, _u64p), _ptr(je, _u64p), _ptr(h, _u32p), n)
        return j, je, h

    def hpc_literal(self, seq, l, bound):
        s = self._seq(seq)
        n = self.lib.s2k_oracle_hpc_literal(_ptr(s, _u8p), len(s), l, bound, None, None, None, 0)
        j = np.empty(n, dtype=np.uint64)
        je = np.empty(n, dtype=np.uint64)
        h = np.empty(n, dtype=np.uint32)
        self.lib.s2k_oracle_hpc_literal(_ptr(s, _u8p), len(s), l, bound, _ptr(j, _u64p), _ptr(je, _u64p), _ptr(h, _u32p), n)
        return j, je, h

    def hpc_literal_u64(self, seq, l, bound64):
        s = self._seq(seq)
        n = self.lib.s2k_oracle_hpc_literal_u64(_ptr(s, _u8p), len(s), l, bound64, None, None, 0)
        j = np.empty(n, dtype=np.uint64)
        h = np.empty(n, dtype=np.uint64)
        self.lib.s2k_oracle_hpc_literal_u64(_ptr(s, _u8p), len(s), l, bound64, _ptr(j, _u64p), _ptr(h, _u64p), n)
        return j, h

    def kminmers(self, seq, l, k, density, mode):
        """-> dict(hash u64, start u64, end u64, rev u8); offset of item i is i."""
        s = self._seq(seq)
        n = self.lib.s2k_oracle_kminmers(_ptr(s, _u8p), len(s), l, k, density, mode, None, None, None, None, 0)
        hs = np.empty(n, dtype=np.uint64)
        st = np.empty(n, dtype=np.uint64)
        en = np.empty(n, dtype=np.uint64)
        rv = np.empty(n, dtype=np.uint8)
        self.lib.s2k_oracle_kminmers(_ptr(s, _u8p), len(s), l, k, density, mode, _ptr(hs, _u64p), _ptr(st, _u64p),
                                     _ptr(en, _u64p), _ptr(rv, _u8p), n)
        return {"hash": hs, "start": st, "end": en, "rev": rv}

    def kminmer_hashes_rolling(self, mh, k):
        mh = np.ascontiguousarray(mh, dtype=np.uint32)
        n = max(len(mh) - k + 1, 0) if k > 0 else 0
        hs = np.empty(n, dtype=np.uint64)
        rv = np.empty(n, dtype=np.uint8)
        c = self.lib.s2k_oracle_kminmer_hashes_rolling(_ptr(mh, _u32p), len(mh), k, _ptr(hs, _u64p), _ptr(rv, _u8p))
        assert c == n
        return hs, rv

    def batch(self, bases, off, l, k, density, mode, threads=1, count_only=False):
        bases = self._seq(bases)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_reads = len(off) - 1
        km_off = np.zeros(n_reads + 1, dtype=np.uint64)
        if count_only:
            tot = self.lib.s2k_oracle_batch(_ptr(bases, _u8p), _ptr(off, _u64p), n_reads, l, k, density, mode, threads,
                                            _ptr(km_off, _u64p), None, None, None, None, 0)
            return {"n": int(tot), "km_off": km_off}
        tot = self.lib.s2k_oracle_batch(_ptr(bases, _u8p), _ptr(off, _u64p), n_reads, l, k, density, mode, threads,
                                        _ptr(km_off, _u64p), None, None, None, None, 0)
        tot = int(tot)
        hs = np.empty(tot, dtype=np.uint64)
        st = np.empty(tot, dtype=np.uint32)
        en = np.empty(tot, dtype=np.uint32)
        rv = np.empty(tot, dtype=np.uint8)
        self.lib.s2k_oracle_batch(_ptr(bases, _u8p), _ptr(off, _u64p), n_reads, l, k, density, mode, threads,
                                  _ptr(km_off, _u64p), _ptr(hs, _u64p), _ptr(st, _u32p), _ptr(en, _u32p), _ptr(rv, _u8p), tot)
        return {"n": tot, "km_off": km_off, "hash": hs, "start": st, "end": en, "rev": rv}

    def batch_count_timed(self, bases, off, l, k, density, mode, threads=1):
        """count-only pass (what src/main.rs:65-76 does per read); returns total."""
        bases = self._seq(bases)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return int(self.lib.s2k_oracle_batch(_ptr(bases, _u8p), _ptr(off, _u64p), len(off) - 1, l, k, density, mode,
                                             threads, None, None, None, None, None, 0))

    def batch_count_clocked(self, bases, off, l, k, density, mode, threads=1, repeats=1):
        """count-only passes timed inside the library (threads created and warmed before the clock starts);
        returns (k-min-mers of one pass, seconds for `repeats` passes)."""
        bases = self._seq(bases)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        sec = C.c_double(0.0)
        fn = self.lib.s2k_oracle_batch_count_timed
        fn.restype = C.c_uint64
        fn.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        n = fn(_ptr(bases, _u8p), _ptr(off, _u64p), len(off) - 1, l, k, density, mode, int(threads), int(repeats), C.byref(sec))
        return int(n), float(sec.value)

    def synth_checksums(self, seed, n_reads, read_len, l, k, density, mode, threads=1):
        """whole-run checksums of a synthetic batch generated read by read (no big host buffer)"""
        out = np.zeros(12, dtype=np.uint64)
        self.lib.s2k_oracle_synth_checksums(seed, n_reads, read_len, l, k, density, mode, threads, _ptr(out, _u64p))
        return dict(zip(CHECKSUM_FIELDS, map(int, out)))

    def synth_checksums_off(self, seed, off, l, k, density, mode, threads=1):
        """as synth_checksums for ragged reads: read r = synthetic stream [off[r], off[r+1])"""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        self.lib.s2k_oracle_synth_checksums_off(seed, _ptr(off, _u64p), len(off) - 1, l, k, density, mode, threads, _ptr(out, _u64p))
        return dict(zip(CHECKSUM_FIELDS, map(int, out)))

    def batch_minimizers(self, bases, off, l, density, mode):
        bases = self._seq(bases)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_reads = len(off) - 1
        mn_off = np.zeros(n_reads + 1, dtype=np.uint64)
        tot = int(self.lib.s2k_oracle_batch_minimizers(_ptr(bases, _u8p), _ptr(off, _u64p), n_reads, l, density, mode,
                                                       _ptr(mn_off, _u64p), None, None, None, 0))
        j = np.empty(tot, dtype=np.uint32)
        je = np.empty(tot, dtype=np.uint32)
        h = np.empty(tot, dtype=np.uint32)
        self.lib.s2k_oracle_batch_minimizers(_ptr(bases, _u8p), _ptr(off, _u64p), n_reads, l, density, mode,
                                             _ptr(mn_off, _u64p), _ptr(j, _u32p), _ptr(je, _u32p), _ptr(h, _u32p), tot)
        return {"n": tot, "mn_off": mn_off, "j": j, "jend": je, "hash": h}

    def hifi_lengths(self, seed, r0, n):
        """lengths of the HiFi-like reads r0 .. r0+n-1 (s2k_oracle_hifi_len)"""
        out = np.empty(n, dtype=np.uint64)
        self.lib.s2k_oracle_hifi_lens(seed, r0, n, _ptr(out, _u64p))
        return out

    def hifi_read(self, seed, r, n):
        out = np.empty(n, dtype=np.uint8)
        self.lib.s2k_oracle_hifi_read(seed, r, n, _ptr(out, _u8p))
        return out

    def hifi_checksums(self, seed, off, l, k, density, mode, threads=1):
        """as synth_checksums_off for HiFi-like reads: read r = hifi_read(seed, r, off[r+1] - off[r])"""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        out = np.zeros(len(CHECKSUM_FIELDS), dtype=np.uint64)
        self.lib.s2k_oracle_hifi_checksums(seed, _ptr(off, _u64p), len(off) - 1, l, k, density, mode, threads, _ptr(out, _u64p))
        return {f: int(v) for f, v in zip(CHECKSUM_FIELDS, out)}

    def synth_bases(self, seed, first_base, n):
        out = np.empty(n, dtype=np.uint8)
        self.lib.s2k_oracle_synth_bases(seed, first_base, n, _ptr(out, _u8p))
        return out


def build_avx512(out=None):
    """Compile the AVX-512 CPU baseline (oracle/s2k_oracle_avx512.c) next to the scalar oracle.  Returns the path,
    or None when the compiler rejects the flags."""
    if out is None:
        out = os.path.join(_HERE, "libs2k_oracle_avx512.so")
    srcs = [os.path.join(_HERE, "s2k_oracle_avx512.c"), os.path.join(_HERE, "s2k_oracle.c")]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(x) for x in srcs):
        return out
    tmpo = out + ".avx.o.%d" % os.getpid()
    tmps = out + ".sc.o.%d" % os.getpid()
    try:
        subprocess.check_call(["gcc", "-O3", "-std=gnu11", "-fPIC", "-mavx512f", "-mavx512bw", "-mavx512vl", "-mavx512vbmi2",
                               "-c", srcs[0], "-o", tmpo])
        subprocess.check_call(["gcc", "-O3", "-std=c11", "-fPIC", "-c", srcs[1], "-o", tmps])
        subprocess.check_call(["gcc", "-shared", "-pthread", "-o", out + ".tmp", tmpo, tmps, "-lm"])
        os.replace(out + ".tmp", out)
    except subprocess.CalledProcessError:
        return None
    finally:
        for f in (tmpo, tmps):
            if os.path.exists(f):
                os.remove(f)
    return out


class OracleAvx512:
    """AVX-512 restatement of the Simd / HpcSimd result semantics (CPU baseline; checked against Oracle)."""

    def __init__(self):
        path = build_avx512()
        if path is None:
            raise RuntimeError("cannot build the AVX-512 baseline with this compiler")
        L = self.lib = C.CDLL(path)
        L.s2k_avx512_supported.restype = C.c_int
        L.s2k_avx512_minimizers.restype = C.c_size_t
        L.s2k_avx512_minimizers.argtypes = [_u8p, C.c_size_t, C.c_uint, C.c_uint32, C.c_int, _u32p, _u32p, _u32p, C.c_size_t]
        L.s2k_avx512_batch_count.restype = C.c_uint64
        L.s2k_avx512_batch_count.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, C.c_double, C.c_int]
        L.s2k_avx512_batch_count_mt.restype = C.c_uint64
        L.s2k_avx512_batch_count_mt.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double, C.c_int, C.c_int]

    def supported(self):
        return bool(self.lib.s2k_avx512_supported())

    def has_reference_shape(self):
        """variant=1: the reference's own algorithm shape (16-lane rolling scan with the lane-15 carry, compress-stores, 16-base HPC steps)"""
        return hasattr(self.lib, "s2k_avx512_minimizers_v")

    def minimizers(self, seq, l, bound, hpc, variant=0):
        s = Oracle._seq(seq)
        fn = self.lib.s2k_avx512_minimizers_v
        fn.restype = C.c_size_t
        fn.argtypes = [_u8p, C.c_size_t, C.c_uint, C.c_uint32, C.c_int, _u32p, _u32p, _u32p, C.c_size_t, C.c_int]
        n = fn(_ptr(s, _u8p), len(s), l, bound, int(hpc), None, None, None, 0, int(variant))
        j = np.empty(n, dtype=np.uint32)
        je = np.empty(n, dtype=np.uint32)
        h = np.empty(n, dtype=np.uint32)
        fn(_ptr(s, _u8p), len(s), l, bound, int(hpc), _ptr(j, _u32p), _ptr(je, _u32p), _ptr(h, _u32p), n, int(variant))
        return j, je, h

    def batch_count(self, bases, off, l, k, density, hpc, threads=1, variant=0):
        """total k-min-mers, count-only (src/main.rs:65-76 with HashMode::Simd/HpcSimd); pthreads over read shards"""
        bases = Oracle._seq(bases)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        fn = self.lib.s2k_avx512_batch_count_mt_v
        fn.restype = C.c_uint64
        fn.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double, C.c_int, C.c_int, C.c_int]
        return int(fn(_ptr(bases, _u8p), _ptr(off, _u64p), len(off) - 1, l, k, density, int(hpc), int(threads), int(variant)))

    def batch_count_clocked(self, bases, off, l, k, density, hpc, threads=1, repeats=1, variant=0):
        bases = Oracle._seq(bases)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        sec = C.c_double(0.0)
        fn = self.lib.s2k_avx512_batch_count_timed_v
        fn.restype = C.c_uint64
        fn.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
        n = fn(_ptr(bases, _u8p), _ptr(off, _u64p), len(off) - 1, l, k, density, int(hpc), int(threads), int(repeats), C.byref(sec), int(variant))
        return int(n), float(sec.value)


_default = None


def get():
    global _default
    if _default is None:
        _default = Oracle()
    return _default
