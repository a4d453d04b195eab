// Host <-> HBM transfers for the host-buffer entry points (s2k_extract, the FASTX pipeline).
//
// The reference's boundary hands over pageable `&[u8]` slices (src/lib.rs:89); a plain hipMemcpy from pageable
// memory is bounced through the runtime's own small staging buffers on one thread and reaches only a fraction of
// the PCIe rate.  HostStager keeps a ring of pinned chunks and a few copy threads: while the DMA engine moves
// chunk i, the threads fill (or drain) chunk i+1, so a transfer runs at min(memcpy rate of the pool, PCIe).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace s2k {

class CopyPool { // fixed set of threads that split one memcpy between them
public:
    explicit CopyPool(int threads);
    ~CopyPool();
    void copy(void *dst, const void *src, size_t bytes); // returns when done
    // runs fn(begin, end) on 4 KiB-aligned slices of [0, bytes), one slice per thread; returns when all are done
    // (max_parts > 0 limits the number of threads that take part: a plain copy saturates the link with 8)
    void slices(size_t bytes, const std::function<void(size_t, size_t)> &fn, int max_parts = 0);
    int threads() const { return (int)workers_.size() + 1; }

private:
    void worker(int idx);
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    uint64_t gen_ = 0;
    int remaining_ = 0;
    bool stop_ = false;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    size_t bytes_ = 0;
    int parts_ = 1;
};

class HostStager {
public:
    HostStager() = default;
    ~HostStager();
    HostStager(const HostStager &) = delete;
    HostStager &operator=(const HostStager &) = delete;
    // pageable host -> device.  On return every byte of `src` has been read (the caller may reuse it) and the last
    // DMA is queued on `s`; work queued on `s` afterwards sees the data.
    hipError_t h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s);
    // same, the source being produced piecewise: fill(pinned_dst, offset, n) must write source bytes
    // [offset, offset+n) to pinned_dst (called from several threads on disjoint ranges); false aborts.
    hipError_t h2d_fill(void *dst_dev, size_t bytes, hipStream_t s,
                        const std::function<bool(char *, size_t, size_t)> &fill);
    // pageable host -> device for DNA text: the copy threads pack the bases to 2 bits ("ACTG"[(b >> 1) & 3], four bases
    // per byte) while they fill the pinned chunk, every byte that is not one of A C G T goes to a side list of
    // (position, byte) exceptions, a quarter of the bytes crosses PCIe, and two small kernels on `s` rebuild the exact
    // ASCII stream in dst_dev (which must be 16-byte aligned).  A chunk with more than ~3 % exceptions (lower-case text,
    // quality strings ...) is sent as it is.  SURVEY.md 8f-1: "host-side 2-bit packing to cut PCIe bytes 4x".
    hipError_t h2d_packed(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s);
    // same for a source that is produced piecewise (a file: FASTA text is DNA with ~0.1-2 % other bytes, which travel as
    // exceptions); *packed_any tells whether packing is worth asking for again: false when no chunk packed, or when two chunks in a row did not (FASTQ, soft-masked or N-rich text) and the rest of the call went as it is
    hipError_t h2d_packed_fill(void *dst_dev, size_t bytes, hipStream_t s, const std::function<bool(char *, size_t, size_t)> &fill,
                               bool *packed_any);
    // device -> pageable host, ordered after the work already queued on `s`.  On return `dst` is complete.
    hipError_t d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s);

    static constexpr size_t kChunk = 32u << 20;
    static constexpr int kSlots = 3;
    static constexpr size_t kSmall = 1u << 20; // below this the runtime's own path is as good

private:
    hipError_t init();
    hipError_t packed_impl(void *dst_dev, const uint8_t *src, const std::function<bool(char *, size_t, size_t)> *fill, size_t bytes,
                           hipStream_t s, bool *packed_any);
    bool ready_ = false;
    char *dpack_[kSlots] = {}; // device staging of the packed chunks (h2d_packed)
    char *pin_[kSlots] = {};
    hipEvent_t ev_[kSlots] = {};
    CopyPool *pool_ = nullptr;
};

} // namespace s2k

struct s2k_ctx;
namespace s2k { // context internals shared inside the library (defined in s2k_api.hip)
HostStager &ctx_stager(s2k_ctx *c);
hipStream_t ctx_stream(s2k_ctx *c);
int ctx_device(s2k_ctx *c);
void ctx_set_error(s2k_ctx *c, const char *what);
// grow-only device buffer of the context for s2k_count_device / s2k_partition_device (nullptr: allocation failed)
void *ctx_count_table(s2k_ctx *c, size_t bytes);
} // namespace s2k
