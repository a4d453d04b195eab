// Pinned-ring staging between pageable host memory and HBM; see s2k_hostcopy.h.
#include "s2k_hostcopy.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <cctype>
#include <cstdio>
#include <pthread.h>
#include <sched.h>

namespace s2k {

// The CPUs of the NUMA node the current HIP device hangs off (sysfs: the device's PCI address -> numa_node -> the node's cpulist).  A 1-GPU job on a
// two-socket host may run anywhere (its CPU share is a quota, not an affinity mask): copy threads on the far socket read the caller's pages and write the
// pinned ring across the socket link, and s2k_extract moved between 50 and 139 Gbp/s from one process to the next (profiles/r06_e2e_*.txt).  Bound to the
// GPU's node the threads at least agree with the pinned ring and the DMA engine.  false: unknown (no sysfs, one node) -- nothing is bound.
static bool gpu_node_cpus(cpu_set_t *set) {
    int dev = 0;
    char bus[64] = {0};
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) return false;
    for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool ok = fgets(list, (int)sizeof list, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *p = list; *p;) { // "0-63,128-191"
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET((int)c, set), n++;
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    cpu_set_t may; // only CPUs the process may use anyway
    if (sched_getaffinity(0, sizeof may, &may) == 0) {
        int both = 0;
        for (int c = 0; c < CPU_SETSIZE; c++) {
            if (CPU_ISSET(c, set) && !CPU_ISSET(c, &may)) CPU_CLR(c, set);
            if (CPU_ISSET(c, set)) both++;
        }
        n = both;
    }
    return n > 0;
}

CopyPool::CopyPool(int threads) {
    if (threads < 1) threads = 1;
    static const bool bind = !(getenv("S2K_NUMA_BIND") && atoi(getenv("S2K_NUMA_BIND")) == 0);
    cpu_set_t node;
    const bool have_node = bind && gpu_node_cpus(&node);
    for (int i = 1; i < threads; i++) {
        workers_.emplace_back(&CopyPool::worker, this, i);
        if (have_node) (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof node, &node);
    }
}

CopyPool::~CopyPool() {
    {
        std::lock_guard<std::mutex> g(m_);
        stop_ = true;
        gen_++;
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
}

static inline void slice_of(size_t bytes, int parts, int idx, size_t *b, size_t *e) {
    const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
    *b = per * idx < bytes ? per * idx : bytes;
    *e = idx == parts - 1 ? bytes : (per * (idx + 1) < bytes ? per * (idx + 1) : bytes);
}

void CopyPool::worker(int idx) {
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(size_t, size_t)> *fn;
        size_t n;
        int parts;
        {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            fn = fn_;
            n = bytes_;
            parts = parts_;
        }
        size_t b = 0, e = 0;
        if (idx < parts) slice_of(n, parts, idx, &b, &e);
        if (e > b) (*fn)(b, e);
        {
            std::lock_guard<std::mutex> g(m_);
            if (--remaining_ == 0) done_cv_.notify_one();
        }
    }
}

void CopyPool::slices(size_t bytes, const std::function<void(size_t, size_t)> &fn, int max_parts) {
    const int T = threads();
    const int parts = max_parts > 0 && max_parts < T ? max_parts : T;
    if (T == 1 || bytes < (256u << 10)) {
        if (bytes) fn(0, bytes);
        return;
    }
    {
        std::lock_guard<std::mutex> g(m_);
        fn_ = &fn;
        bytes_ = bytes;
        parts_ = parts;
        remaining_ = T - 1;
        gen_++;
    }
    cv_.notify_all();
    size_t b, e;
    slice_of(bytes, parts, 0, &b, &e);
    if (e > b) fn(b, e);
    std::unique_lock<std::mutex> g(m_);
    done_cv_.wait(g, [&] { return remaining_ == 0; });
}

void CopyPool::copy(void *dst, const void *src, size_t bytes) {
    slices(bytes, [&](size_t b, size_t e) { memcpy((char *)dst + b, (const char *)src + b, e - b); }, 8);
}

HostStager::~HostStager() {
    for (int i = 0; i < kSlots; i++) {
        if (ev_[i]) {
            (void)hipEventSynchronize(ev_[i]);
            (void)hipEventDestroy(ev_[i]);
        }
        if (pin_[i]) (void)hipHostFree(pin_[i]);
        if (dpack_[i]) (void)hipFree(dpack_[i]);
    }
    delete pool_;
}

hipError_t HostStager::init() {
    if (ready_) return hipSuccess;
    for (int i = 0; i < kSlots; i++) {
        hipError_t e = pin_[i] ? hipSuccess : hipHostMalloc((void **)&pin_[i], kChunk, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (!ev_[i]) {
            e = hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    }
    if (!pool_) {
        int t = 0;
        if (const char *v = getenv("S2K_COPY_THREADS")) t = atoi(v);
        if (t <= 0) {
            unsigned hw = std::thread::hardware_concurrency();
            cpu_set_t set; // the CPUs this process may use (a 1-GPU job gets a share of the host)
            if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && (unsigned)CPU_COUNT(&set) < hw) hw = (unsigned)CPU_COUNT(&set);
            t = hw >= 16 ? 16 : hw >= 4 ? (int)hw / 2 : 1; // packing bases to 2 bits is compute: it scales past the 8 threads a plain copy needs
        }
        pool_ = new CopyPool(t);
    }
    ready_ = true;
    return hipSuccess;
}

// A recorded-never event reports complete, so waiting on every slot before its first use in a transfer is enough
// to order this transfer after whatever earlier transfer last touched the slot.
hipError_t HostStager::h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (bytes < kSmall) {
        hipError_t e = hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s);
        return e != hipSuccess ? e : hipStreamSynchronize(s); // pageable source: the caller may reuse it on return
    }
    return h2d_fill(dst_dev, bytes, s, [&](char *pin, size_t off, size_t n) {
        memcpy(pin, (const char *)src_host + off, n);
        return true;
    });
}

hipError_t HostStager::h2d_fill(void *dst_dev, size_t bytes, hipStream_t s,
                                const std::function<bool(char *, size_t, size_t)> &fill) {
    if (bytes == 0) return hipSuccess;
    hipError_t e = init();
    if (e != hipSuccess) return e;
    size_t off = 0;
    for (int i = 0; off < bytes; i++) {
        const int slot = i % kSlots;
        const size_t n = bytes - off < kChunk ? bytes - off : kChunk;
        if ((e = hipEventSynchronize(ev_[slot])) != hipSuccess) return e;
        std::atomic<bool> ok{true};
        char *pin = pin_[slot];
        pool_->slices(n, [&](size_t b, size_t en) {
            if (!fill(pin + b, off + b, en - b)) ok = false;
        }, 8);
        if (!ok) return hipErrorUnknown;
        if ((e = hipMemcpyAsync((char *)dst_dev + off, pin, n, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
        if ((e = hipEventRecord(ev_[slot], s)) != hipSuccess) return e;
        off += n;
    }
    return hipSuccess;
}

hipError_t HostStager::d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (bytes < kSmall) {
        hipError_t e = hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s);
        return e != hipSuccess ? e : hipStreamSynchronize(s);
    }
    hipError_t e = init();
    if (e != hipSuccess) return e;
    const size_t chunks = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](size_t c) -> hipError_t {
        const int slot = (int)(c % kSlots);
        const size_t off = c * kChunk, n = bytes - off < kChunk ? bytes - off : kChunk;
        hipError_t e2 = hipMemcpyAsync(pin_[slot], (const char *)src_dev + off, n, hipMemcpyDeviceToHost, s);
        return e2 != hipSuccess ? e2 : hipEventRecord(ev_[slot], s);
    };
    for (int i = 0; i < kSlots; i++) // slots may still be the source of an earlier h2d
        if ((e = hipEventSynchronize(ev_[i])) != hipSuccess) return e;
    for (size_t c = 0; c < chunks && c < (size_t)kSlots; c++)
        if ((e = issue(c)) != hipSuccess) return e;
    for (size_t c = 0; c < chunks; c++) {
        const int slot = (int)(c % kSlots);
        const size_t off = c * kChunk, n = bytes - off < kChunk ? bytes - off : kChunk;
        if ((e = hipEventSynchronize(ev_[slot])) != hipSuccess) return e;
        pool_->copy((char *)dst_host + off, pin_[slot], n);
        if (c + kSlots < chunks && (e = issue(c + kSlots)) != hipSuccess) return e;
    }
    return hipSuccess;
}


// ---- 2-bit packed transfers ----------------------------------------------------------------------------------------
namespace {

struct Exc { // a byte of the stream that is not one of A C G T
    uint32_t pos;  // offset inside the chunk
    uint32_t byte;
};

// code of a base = (ascii >> 1) & 3: A 0, C 1, T 2, G 3
inline int base_code(uint8_t b) { return (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? ((b >> 1) & 3) : -1; }

// packs src[0..n) (n % 4 == 0 except for the very last slice of a stream) into dst[0..ceil(n/4)); exceptions go to
// exc[0..cap) with positions chunk_off + i; returns their number, or SIZE_MAX when there are more than cap
size_t pack2_scalar(const uint8_t *src, size_t n, uint8_t *dst, Exc *exc, size_t cap, uint32_t chunk_off, size_t ne) {
    for (size_t i = 0; i < n; i += 4) {
        uint32_t v = 0;
        for (size_t j = 0; j < 4 && i + j < n; j++) {
            int c = base_code(src[i + j]);
            if (c < 0) {
                if (ne >= cap) return SIZE_MAX;
                exc[ne++] = Exc{chunk_off + (uint32_t)(i + j), src[i + j]};
                c = 0;
            }
            v |= (uint32_t)c << (2 * j);
        }
        dst[i / 4] = (uint8_t)v;
    }
    return ne;
}

__attribute__((target("avx2,bmi2"))) size_t pack2_avx2(const uint8_t *src, size_t n, uint8_t *dst, Exc *exc, size_t cap,
                                                        uint32_t chunk_off, size_t ne0) {
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         0, 0, 0);
    const __m256i m6 = _mm256_set1_epi8(6);
    size_t ne = ne0, i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
        const __m256i c2 = _mm256_and_si256(v, m6);                    // code << 1
        const __m256i idx = _mm256_srli_epi16(c2, 1);                  // bit 0 of every byte of c2 is clear: no cross-byte spill
        const __m256i expect = _mm256_shuffle_epi8(lut, idx);
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, expect)) == -1) { // all 32 are A C G T
            const uint64_t m = 0x0606060606060606ull;
            const uint64_t a = _pext_u64((uint64_t)_mm256_extract_epi64(v, 0), m), b = _pext_u64((uint64_t)_mm256_extract_epi64(v, 1), m);
            const uint64_t c = _pext_u64((uint64_t)_mm256_extract_epi64(v, 2), m), d = _pext_u64((uint64_t)_mm256_extract_epi64(v, 3), m);
            const uint64_t w = a | (b << 16) | (c << 32) | (d << 48);
            memcpy(dst + i / 4, &w, 8);
        } else {
            ne = pack2_scalar(src + i, 32, dst + i / 4, exc, cap, chunk_off + (uint32_t)i, ne);
            if (ne == SIZE_MAX) return SIZE_MAX;
        }
    }
    if (i < n) ne = pack2_scalar(src + i, n - i, dst + i / 4, exc, cap, chunk_off + (uint32_t)i, ne);
    return ne;
}

// AVX-512 (F + BW: every server part since Skylake-X / Zen 4): 64 bases per step.  A byte is one of A C G T iff it equals "ACTG"[(b >> 1) & 3] (one
// in-lane byte shuffle + one compare into a mask register); the four codes of a dword are folded with two multiply-adds (weights 1, 4 per byte pair,
// then 1, 16 per word pair: c0 + 4 c1 + 16 c2 + 64 c3 <= 255) and the sixteen dwords narrowed to sixteen bytes (vpmovdb).  ~10 instructions per 64
// bytes against four pext + extracts per 32: the threads of a 1-GPU job (16 of the host's) pack at the rate they can read pageable memory.
// A 64-byte group with any other byte goes through the scalar routine (its exceptions are listed); groups of clean text never touch it.
__attribute__((target("avx512f,avx512bw"))) size_t pack2_avx512(const uint8_t *src, size_t n, uint8_t *dst, Exc *exc, size_t cap, uint32_t chunk_off,
                                                                size_t ne0) {
    const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0));
    const __m512i m3 = _mm512_set1_epi8(3), w14 = _mm512_set1_epi16(0x0401), w116 = _mm512_set1_epi32(0x00100001);
    size_t ne = ne0, i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m512i v = _mm512_loadu_si512((const void *)(src + i));
        const __m512i code = _mm512_and_si512(_mm512_srli_epi16(v, 1), m3); // (b >> 1) & 3 per byte: the bits a 16-bit shift drags in are masked off
        if (_mm512_cmpeq_epi8_mask(v, _mm512_shuffle_epi8(lut, code)) == ~0ull) {
            const __m512i p16 = _mm512_maddubs_epi16(code, w14);            // c0 + 4 c1 per byte pair
            const __m512i p32 = _mm512_madd_epi16(p16, w116);               // + 16 (c2 + 4 c3) per dword
            _mm_storeu_si128((__m128i *)(dst + i / 4), _mm512_cvtepi32_epi8(p32));
        } else {
            ne = pack2_scalar(src + i, 64, dst + i / 4, exc, cap, chunk_off + (uint32_t)i, ne);
            if (ne == SIZE_MAX) return SIZE_MAX;
        }
    }
    if (i < n) ne = pack2_scalar(src + i, n - i, dst + i / 4, exc, cap, chunk_off + (uint32_t)i, ne);
    return ne;
}

// ne0: exceptions already in exc[] (a slice packed piece by piece)
size_t pack2(const uint8_t *src, size_t n, uint8_t *dst, Exc *exc, size_t cap, uint32_t chunk_off, size_t ne0 = 0) {
    static const int level = getenv("S2K_PACK_ISA") ? atoi(getenv("S2K_PACK_ISA")) // (A/B and tests: 0 scalar, 1 AVX2, 2 AVX-512)
                             : (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) ? 2
                             : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) ? 1 : 0;
    if (level >= 2 && __builtin_cpu_supports("avx512bw")) return pack2_avx512(src, n, dst, exc, cap, chunk_off, ne0);
    if (level >= 1 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) return pack2_avx2(src, n, dst, exc, cap, chunk_off, ne0);
    return pack2_scalar(src, n, dst, exc, cap, chunk_off, ne0);
}

// 16 bases per thread: 4 packed bytes -> 16 ASCII bytes ("ACTG"[code])
__global__ __launch_bounds__(256) void unpack2_kernel(const uint32_t *__restrict__ packed, uint64_t n_bases, uint8_t *__restrict__ out) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; // group of 16 bases
    if (g * 16 >= n_bases) return;
    const uint32_t p = packed[g];
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b = (p >> (8 * k)) & 0xFFu;
        const uint32_t sel = (b & 3u) | ((b & 0xCu) << 6) | ((b & 0x30u) << 12) | ((b & 0xC0u) << 18); // one code per selector byte
        w[k] = __builtin_amdgcn_perm(0u, 0x47544341u /* "ACTG" */, sel);
    }
    if (g * 16 + 16 <= n_bases) *reinterpret_cast<uint4 *>(out + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    else
        for (uint64_t i = g * 16; i < n_bases; i++) out[i] = (uint8_t)(w[(i >> 2) & 3] >> (8 * (i & 3)));
}
__global__ __launch_bounds__(256) void patch_exceptions_kernel(const Exc *__restrict__ exc, uint32_t n, uint8_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[exc[i].pos] = (uint8_t)exc[i].byte;
}

} // namespace

hipError_t HostStager::h2d_packed(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s) {
    return packed_impl(dst_dev, (const uint8_t *)src_host, nullptr, bytes, s, nullptr);
}

hipError_t HostStager::h2d_packed_fill(void *dst_dev, size_t bytes, hipStream_t s, const std::function<bool(char *, size_t, size_t)> &fill,
                                       bool *packed_any) {
    return packed_impl(dst_dev, nullptr, &fill, bytes, s, packed_any);
}

// src != nullptr: the source is host memory; else fill(dst, offset, n) produces source bytes [offset, offset + n) (a file):
// every copy thread reads its slice into a scratch buffer of its own and packs from there
hipError_t HostStager::packed_impl(void *dst_dev, const uint8_t *src, const std::function<bool(char *, size_t, size_t)> *fill, size_t bytes,
                                   hipStream_t s, bool *packed_any) {
    if (packed_any) *packed_any = false;
    if (bytes < (8u << 20) || ((uintptr_t)dst_dev & 15u) != 0) return src ? h2d(dst_dev, src, bytes, s) : h2d_fill(dst_dev, bytes, s, *fill);
    hipError_t e = init();
    if (e != hipSuccess) return e;
    constexpr size_t kHalf = kChunk / 2;  // pinned chunk = [packed bases | exception lists]
    constexpr size_t kSrc = 4 * kHalf;    // source bytes per chunk (64 MiB)
    for (int i = 0; i < kSlots; i++)
        if (!dpack_[i] && (e = hipMalloc((void **)&dpack_[i], kChunk)) != hipSuccess) return e;
    size_t off = 0;
    int overflowed_in_a_row = 0; // chunks that did not pack (soft-masked or N-rich text, FASTQ): each was read and scanned in vain
    for (int i = 0; off < bytes; i++) {
        if (overflowed_in_a_row >= 2) { // stop trying: the rest goes as it is, and the caller is told not to ask again
            if (packed_any) *packed_any = false;
            if (src) return h2d((char *)dst_dev + off, src + off, bytes - off, s);
            const size_t base = off;
            return h2d_fill((char *)dst_dev + off, bytes - off, s, [&](char *d, size_t o, size_t len) { return (*fill)(d, base + o, len); });
        }
        const int slot = i % kSlots;
        const size_t n = bytes - off < kSrc ? bytes - off : kSrc;
        if ((e = hipEventSynchronize(ev_[slot])) != hipSuccess) return e;
        char *pin = pin_[slot];
        std::atomic<bool> overflow{false}, failed{false};
        // every slice packs its part and keeps its exceptions in the matching part of the second half: [count u64][entries]
        pool_->slices(n, [&](size_t b, size_t en) {
            char *region = pin + kHalf + b / 4;
            const size_t cap = (en - b) / 4 >= 16 ? ((en - b) / 4 - 8) / sizeof(Exc) : 0;
            size_t ne = 0;
            if (src) {
                ne = pack2(src + off + b, en - b, (uint8_t *)pin + b / 4, (Exc *)(region + 8), cap, (uint32_t)b);
            } else {
                // a produced source (a file): piece by piece through a scratch buffer that stays in the core's L2 -- read 256 KiB, pack them, read the next --,
                // so that the text crosses DRAM once (page cache -> L2) instead of three times (round 6; rounds 2-5 read the whole 4 MiB slice first)
                constexpr size_t kPiece = 256u << 10;
                thread_local std::vector<uint8_t> scratch;
                if (scratch.size() < kPiece) scratch.resize(kPiece);
                for (size_t o = b; o < en && ne != SIZE_MAX; o += kPiece) {
                    const size_t len = en - o < kPiece ? en - o : kPiece;
                    if (!(*fill)((char *)scratch.data(), off + o, len)) {
                        failed = true;
                        return;
                    }
                    ne = pack2(scratch.data(), len, (uint8_t *)pin + o / 4, (Exc *)(region + 8), cap, (uint32_t)o, ne);
                }
            }
            if (ne == SIZE_MAX) overflow = true;
            else memcpy(region, &ne, 8);
        });
        if (failed) return hipErrorUnknown;
        if (overflow) { // not DNA text: this chunk goes as it is (two pinned chunks' worth at most)
            if ((e = hipEventRecord(ev_[slot], s)) != hipSuccess) return e;
            if (src) {
                e = h2d((char *)dst_dev + off, src + off, n, s);
            } else {
                const size_t base = off;
                e = h2d_fill((char *)dst_dev + off, n, s, [&](char *d, size_t o, size_t len) { return (*fill)(d, base + o, len); });
            }
            if (e != hipSuccess) return e;
            off += n;
            overflowed_in_a_row++;
            continue;
        }
        overflowed_in_a_row = 0;
        if (packed_any) *packed_any = true;
        // gather the slices' exception lists behind the packed bytes (few entries; one thread)
        size_t n_exc = 0;
        {
            const int T = pool_->threads();
            Exc *all = (Exc *)(pin + kHalf);
            std::vector<Exc> tmp;
            for (int t = 0; t < (n < (256u << 10) ? 1 : T); t++) {
                size_t b, en;
                if (n < (256u << 10)) { b = 0; en = n; }
                else slice_of(n, T, t, &b, &en);
                if (en <= b) continue;
                size_t ne;
                memcpy(&ne, pin + kHalf + b / 4, 8);
                const Exc *list = (const Exc *)(pin + kHalf + b / 4 + 8);
                tmp.insert(tmp.end(), list, list + ne);
            }
            n_exc = tmp.size();
            if (n_exc) memcpy(all, tmp.data(), n_exc * sizeof(Exc));
        }
        const size_t packed_bytes = (n + 3) / 4;
        if ((e = hipMemcpyAsync(dpack_[slot], pin, (packed_bytes + 3) & ~(size_t)3, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
        if (n_exc && (e = hipMemcpyAsync(dpack_[slot] + kHalf, pin + kHalf, n_exc * sizeof(Exc), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
        if ((e = hipEventRecord(ev_[slot], s)) != hipSuccess) return e;
        const uint64_t groups = (n + 15) / 16;
        hipLaunchKernelGGL(unpack2_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (const uint32_t *)dpack_[slot], (uint64_t)n,
                           (uint8_t *)dst_dev + off);
        if (n_exc)
            hipLaunchKernelGGL(patch_exceptions_kernel, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, s, (const Exc *)(dpack_[slot] + kHalf),
                               (uint32_t)n_exc, (uint8_t *)dst_dev + off);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        off += n;
    }
    return hipSuccess;
}

} // namespace s2k
