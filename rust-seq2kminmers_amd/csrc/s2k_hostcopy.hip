// Pinned-ring staging between pageable host memory and HBM; see s2k_hostcopy.h.
#include "s2k_hostcopy.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace s2k {

CopyPool::CopyPool(int threads) {
    if (threads < 1) threads = 1;
    for (int i = 1; i < threads; i++) workers_.emplace_back(&CopyPool::worker, this, i);
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
        {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            fn = fn_;
            n = bytes_;
        }
        size_t b, e;
        slice_of(n, threads(), idx, &b, &e);
        if (e > b) (*fn)(b, e);
        {
            std::lock_guard<std::mutex> g(m_);
            if (--remaining_ == 0) done_cv_.notify_one();
        }
    }
}

void CopyPool::slices(size_t bytes, const std::function<void(size_t, size_t)> &fn) {
    const int T = threads();
    if (T == 1 || bytes < (256u << 10)) {
        if (bytes) fn(0, bytes);
        return;
    }
    {
        std::lock_guard<std::mutex> g(m_);
        fn_ = &fn;
        bytes_ = bytes;
        remaining_ = T - 1;
        gen_++;
    }
    cv_.notify_all();
    size_t b, e;
    slice_of(bytes, T, 0, &b, &e);
    if (e > b) fn(b, e);
    std::unique_lock<std::mutex> g(m_);
    done_cv_.wait(g, [&] { return remaining_ == 0; });
}

void CopyPool::copy(void *dst, const void *src, size_t bytes) {
    slices(bytes, [&](size_t b, size_t e) { memcpy((char *)dst + b, (const char *)src + b, e - b); });
}

HostStager::~HostStager() {
    for (int i = 0; i < kSlots; i++) {
        if (ev_[i]) {
            (void)hipEventSynchronize(ev_[i]);
            (void)hipEventDestroy(ev_[i]);
        }
        if (pin_[i]) (void)hipHostFree(pin_[i]);
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
            const unsigned hw = std::thread::hardware_concurrency();
            t = hw >= 16 ? 8 : hw >= 4 ? (int)hw / 2 : 1;
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
        });
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

} // namespace s2k
