// s2k_fastx.hip -- FASTA/FASTQ ingest + batching front-end (host code).
// What it replaces: rust_parallelfastx::parallel_fastx(file, threads, task) + the per-record closure of
// src/main.rs:65-79.  The reference parses on a thread pool and hands each record to a KminmersIterator; here
// records are appended to a batch (bases back to back + read_off) in pinned host memory and the batch is what
// crosses PCIe.  Two batches are in flight: while the GPU works on one, the host parses the next.
#include "../../include/s2k.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct PinnedBuf { // pinned when a GPU is present (fast async H2D), plain otherwise (the parser also runs on CPU-only hosts)
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    bool ensure(size_t need) {
        if (need <= cap) return true;
        size_t want = need + need / 4 + 4096;
        void *q = nullptr;
        bool pin = hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess;
        if (!pin) {
            (void)hipGetLastError();
            q = malloc(want);
            if (!q) return false;
        }
        if (p) {
            memcpy(q, p, cap);
            release();
        }
        p = q;
        cap = want;
        pinned = pin;
        return true;
    }
    void release() {
        if (!p) return;
        if (pinned) (void)hipHostFree(p);
        else free(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace

struct s2k_fastx {
    FILE *f = nullptr;
    std::vector<char> io; // read-ahead window over the file
    size_t io_pos = 0, io_len = 0;
    bool eof = false, fastq = false, started = false;
    PinnedBuf bases[2], offs[2];
    int cur = 0;
    uint64_t n_reads = 0, n_bases = 0;
    std::string line;

    bool fill() {
        if (eof) return false;
        if (io_pos < io_len) memmove(io.data(), io.data() + io_pos, io_len - io_pos);
        io_len -= io_pos;
        io_pos = 0;
        size_t got = fread(io.data() + io_len, 1, io.size() - io_len, f);
        io_len += got;
        if (got == 0) eof = true;
        return got != 0;
    }
    // next line without its terminator ("\n" or "\r\n"); false at end of file
    bool getline(const char *&s, size_t &n) {
        for (;;) {
            const char *b = io.data() + io_pos;
            const char *nl = (const char *)memchr(b, '\n', io_len - io_pos);
            if (nl) {
                s = b;
                n = (size_t)(nl - b);
                io_pos += n + 1;
                if (n && s[n - 1] == '\r') n--;
                return true;
            }
            if (io_len - io_pos == io.size()) io.resize(io.size() * 2); // a line longer than the window
            if (!fill()) {
                if (io_pos < io_len) { // last line without a newline
                    s = io.data() + io_pos;
                    n = io_len - io_pos;
                    io_pos = io_len;
                    if (n && s[n - 1] == '\r') n--;
                    return true;
                }
                return false;
            }
        }
    }
    bool peek(char &c) {
        if (io_pos >= io_len && !fill()) return false;
        if (io_pos >= io_len) return false;
        c = io[io_pos];
        return true;
    }
};

extern "C" {

s2k_fastx *s2k_fastx_open(const char *path, s2k_status *status) {
    s2k_status dummy;
    if (!status) status = &dummy;
    if (!path) {
        *status = S2K_ERR_INVALID_ARG;
        return nullptr;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        *status = S2K_ERR_INVALID_ARG;
        return nullptr;
    }
    s2k_fastx *rd = new (std::nothrow) s2k_fastx();
    if (!rd) {
        fclose(f);
        *status = S2K_ERR_NOMEM;
        return nullptr;
    }
    rd->f = f;
    rd->io.resize(8u << 20);
    rd->fill();
    char c = 0;
    while (rd->peek(c) && (c == '\n' || c == '\r')) rd->io_pos++;
    rd->fastq = rd->peek(c) && c == '@';
    *status = S2K_OK;
    return rd;
}

void s2k_fastx_close(s2k_fastx *rd) {
    if (!rd) return;
    if (rd->f) fclose(rd->f);
    for (int i = 0; i < 2; i++) {
        rd->bases[i].release();
        rd->offs[i].release();
    }
    delete rd;
}

s2k_status s2k_fastx_next(s2k_fastx *rd, uint64_t max_bases, uint64_t max_reads, const uint8_t **bases,
                          const uint64_t **read_off, uint64_t *n_reads) {
    if (!rd || !bases || !read_off || !n_reads) return S2K_ERR_INVALID_ARG;
    rd->cur ^= 1; // the previous batch stays valid while this one is built (double buffering)
    PinnedBuf &B = rd->bases[rd->cur], &O = rd->offs[rd->cur];
    if (!B.ensure((size_t)(max_bases ? max_bases : 1) + 256) || !O.ensure(sizeof(uint64_t) * 1024)) return S2K_ERR_NOMEM;
    uint64_t nb = 0, nr = 0;
    auto push_off = [&](uint64_t v) -> bool {
        if ((nr + 2) * sizeof(uint64_t) > O.cap && !O.ensure((nr + 2) * sizeof(uint64_t) * 2)) return false;
        ((uint64_t *)O.p)[nr] = v;
        return true;
    };
    if (!push_off(0)) return S2K_ERR_NOMEM;
    const char *s;
    size_t n;
    char c;
    while (nr < (max_reads ? max_reads : ~0ull) && (nr == 0 || nb < max_bases)) {
        if (!rd->peek(c)) break;
        if (c == '\n' || c == '\r') { // blank line between records
            rd->getline(s, n);
            continue;
        }
        if (rd->fastq) { // @id / sequence / + / quality  (4-line records)
            if (c != '@') return S2K_ERR_INVALID_ARG;
            rd->getline(s, n);
            if (!rd->getline(s, n)) return S2K_ERR_INVALID_ARG;
            if (!B.ensure(nb + n + 256)) return S2K_ERR_NOMEM;
            memcpy((char *)B.p + nb, s, n);
            nb += n;
            if (!rd->getline(s, n) || n == 0 || s[0] != '+') return S2K_ERR_INVALID_ARG;
            if (!rd->getline(s, n)) return S2K_ERR_INVALID_ARG;
        } else { // >id then sequence lines up to the next '>' (multi-line FASTA)
            if (c != '>') return S2K_ERR_INVALID_ARG;
            rd->getline(s, n);
            while (rd->peek(c) && c != '>') {
                rd->getline(s, n);
                if (!B.ensure(nb + n + 256)) return S2K_ERR_NOMEM;
                memcpy((char *)B.p + nb, s, n);
                nb += n;
            }
        }
        nr++;
        if (!push_off(nb)) return S2K_ERR_NOMEM;
    }
    rd->n_reads += nr;
    rd->n_bases += nb;
    *bases = (const uint8_t *)B.p;
    *read_off = (const uint64_t *)O.p;
    *n_reads = nr;
    return S2K_OK;
}

s2k_status s2k_run_file(s2k_ctx *ctx, const char *path, const s2k_params *params, uint64_t batch_bases,
                        s2k_counts *totals, double *seconds) {
    if (!ctx || !params || !totals) return S2K_ERR_INVALID_ARG;
    auto t_start = std::chrono::steady_clock::now();
    s2k_status st;
    s2k_fastx *rd = s2k_fastx_open(path, &st);
    if (!rd) return st;
    if (batch_bases == 0) batch_bases = 256ull << 20;
    memset(totals, 0, sizeof *totals);
    // two device-side batches so that parsing + H2D of batch i+1 overlap the kernels of batch i
    struct Dev {
        void *bases = nullptr, *off = nullptr, *out = nullptr;
        size_t cb = 0, co = 0, cout = 0;
    } dev[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr};
    bool ok = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 2; i++) ok = hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) == hipSuccess;
    auto grow = [&](void *&p, size_t &cap, size_t need) -> bool {
        if (need <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, need + need / 8 + 4096) != hipSuccess) return false;
        cap = need + need / 8 + 4096;
        return true;
    };
    st = ok ? S2K_OK : S2K_ERR_DEVICE;
    bool pending = false;
    int slot = 0;
    auto collect = [&]() -> s2k_status { // wait for the batch in flight and add its counts
        if (!pending) return S2K_OK;
        s2k_counts c;
        s2k_status s2 = s2k_sync(ctx, &c);
        pending = false;
        if (s2 != S2K_OK) return s2;
        totals->n_reads += c.n_reads;
        totals->n_bases += c.n_bases;
        totals->n_minimizers += c.n_minimizers;
        totals->n_kminmers += c.n_kminmers;
        totals->xor_hash ^= c.xor_hash;
        totals->hash_bound = c.hash_bound;
        totals->path = c.path;
        return S2K_OK;
    };
    while (st == S2K_OK) {
        const uint8_t *hb;
        const uint64_t *ho;
        uint64_t nr = 0;
        st = s2k_fastx_next(rd, batch_bases, 0, &hb, &ho, &nr); // overlaps the GPU work of the previous batch
        if (st != S2K_OK || nr == 0) break;
        const uint64_t nb = ho[nr];
        for (uint64_t r = 0; r < nr; r++)
            if (ho[r + 1] - ho[r] > 0xFFFFFFFEull) st = S2K_ERR_READ_TOO_LONG;
        if (st != S2K_OK) break;
        Dev &d = dev[slot];
        // output capacity: k-min-mers <= minimizers <= bases, so nb + 1 can never overflow (low-complexity input included)
        const uint64_t cap = nb + 1;
        const size_t out_bytes = (((nr + 1) * 8 + 255) & ~(size_t)255) + cap * (8 + 4 + 4 + 1) + 1024;
        if (!grow(d.bases, d.cb, nb + 256) || !grow(d.off, d.co, (nr + 1) * 8) || !grow(d.out, d.cout, out_bytes)) {
            st = S2K_ERR_NOMEM;
            break;
        }
        ok = hipMemcpyAsync(d.bases, hb, nb, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
             hipMemcpyAsync(d.off, ho, (nr + 1) * 8, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
             hipStreamSynchronize(copy_stream) == hipSuccess; // the copy overlaps the previous batch's kernels
        if (!ok) {
            st = S2K_ERR_DEVICE;
            break;
        }
        st = collect(); // previous batch done
        if (st != S2K_OK) break;
        s2k_device_out o;
        memset(&o, 0, sizeof o);
        char *q = (char *)d.out;
        o.km_capacity = cap;
        o.km_off = (uint64_t *)q;
        q += ((nr + 1) * 8 + 255) & ~(size_t)255;
        o.hash = (uint64_t *)q;
        q += cap * 8;
        o.start = (uint32_t *)q;
        q += cap * 4;
        o.end = (uint32_t *)q;
        q += cap * 4;
        o.rev = (uint8_t *)q;
        st = s2k_extract_device(ctx, (const uint8_t *)d.bases, (const uint64_t *)d.off, nr, nb, params, &o, nullptr);
        pending = st == S2K_OK;
        slot ^= 1;
    }
    if (st == S2K_OK) st = collect();
    else (void)s2k_sync(ctx, nullptr);
    for (int i = 0; i < 2; i++) {
        if (dev[i].bases) (void)hipFree(dev[i].bases);
        if (dev[i].off) (void)hipFree(dev[i].off);
        if (dev[i].out) (void)hipFree(dev[i].out);
        if (copied[i]) (void)hipEventDestroy(copied[i]);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    s2k_fastx_close(rd);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return st;
}

} // extern "C"
