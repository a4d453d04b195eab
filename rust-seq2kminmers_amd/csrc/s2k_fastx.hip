// s2k_fastx.hip -- FASTA/FASTQ ingest + batching front-end (host code).
// What it replaces: rust_parallelfastx::parallel_fastx(file, threads, task) + the per-record closure of
// src/main.rs:65-79.  The reference parses on a thread pool and hands each record to a KminmersIterator; here
// records are appended to a batch (bases back to back + read_off) in pinned host memory and the batch is what
// crosses PCIe.  Two batches are in flight: while the GPU works on one, the host parses the next.
#include "../../include/s2k.h"
#include "s2k_dev.h"
#include "s2k_hostcopy.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

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


namespace {

// Read-only window over a file (pread), used to find record starts near a batch boundary without parsing.
struct FileWindow {
    int fd;
    uint64_t fsize, lo = 0, hi = 0;
    std::vector<char> buf;
    FileWindow(int f, uint64_t n) : fd(f), fsize(n) {}
    uint64_t load(uint64_t a, uint64_t b) { // makes [a, min(b, fsize)) available; returns its length
        if (b > fsize) b = fsize;
        if (a >= b) {
            lo = hi = a;
            return 0;
        }
        buf.resize(b - a);
        uint64_t got = 0;
        while (got < b - a) {
            const ssize_t r = pread(fd, buf.data() + got, b - a - got, (off_t)(a + got));
            if (r <= 0) break;
            got += (uint64_t)r;
        }
        lo = a;
        hi = a + got;
        return got;
    }
    char at(uint64_t p) const { return buf[p - lo]; }
    // position after the next '\n' at or after p, or ~0 if the window ends first
    uint64_t next_line(uint64_t p) const {
        if (p >= hi) return ~0ull;
        const char *q = (const char *)memchr(buf.data() + (p - lo), '\n', hi - p);
        return q ? lo + (uint64_t)(q - buf.data()) + 1 : ~0ull;
    }
};

// is there a record starting at p (p is a line start inside the window)?
bool record_at(const FileWindow &w, uint64_t p, bool fastq) {
    if (p >= w.hi) return false;
    if (!fastq) return w.at(p) == '>';
    if (w.at(p) != '@') return false;
    // a quality line may start with '@' too; a header is followed by the sequence line and then a '+' line
    const uint64_t l2 = w.next_line(p);
    if (l2 == ~0ull) return false;
    const uint64_t l3 = w.next_line(l2);
    if (l3 == ~0ull || l3 >= w.hi) return false;
    return w.at(l3) == '+';
}

// Largest record start in (after, want] if there is one, else the first record start after `want`, else the file size.
uint64_t find_record_start(FileWindow &w, uint64_t after, uint64_t want, bool fastq) {
    if (want >= w.fsize) return w.fsize;
    const uint64_t slack = 1u << 20; // look-ahead needed to verify a FASTQ header near `want`
    for (uint64_t span = 4u << 20;; span *= 4) {
        const uint64_t a = want > after + span ? want - span : after;
        w.load(a, want + slack);
        // walk line starts backwards from `want`
        uint64_t p = want + 1 < w.hi ? want + 1 : w.hi; // byte want-1 may be the newline that makes `want` a line start
        while (p > a + 1) {
            const char *base = w.buf.data();
            const void *q = memrchr(base, '\n', p - 1 - w.lo);
            if (!q) break;
            const uint64_t ls = w.lo + (uint64_t)((const char *)q - base) + 1; // a line starts here
            if (ls <= after) break;
            if (ls <= want && record_at(w, ls, fastq)) return ls;
            p = ls;
        }
        if (a == after) break;
    }
    // a single record longer than the batch: extend forward to its end
    uint64_t from = want;
    for (uint64_t span = 16u << 20; from < w.fsize; span *= 2) {
        w.load(from, from + span);
        uint64_t p = from;
        uint64_t last_checked = from;
        for (;;) {
            const uint64_t ls = w.next_line(p);
            if (ls == ~0ull || ls >= w.hi) break;
            if (record_at(w, ls, fastq)) return ls;
            // FASTQ candidates too close to the window end cannot be verified; restart there with a bigger window
            p = ls;
            last_checked = ls;
        }
        if (w.hi >= w.fsize) return w.fsize;
        from = last_checked > from ? last_checked - 1 : w.hi - 1; // keep the '\n' before the next line start in view
    }
    return w.fsize;
}

} // namespace

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

s2k_status s2k_fastx_parse_device(s2k_ctx *ctx, const uint8_t *d_text, uint64_t n_bytes, int format, uint8_t *d_bases,
                                  uint64_t bases_capacity, uint64_t *d_read_off, uint64_t off_capacity,
                                  uint64_t *n_reads, uint64_t *n_bases) {
    if (!ctx || !n_reads || !n_bases || (format != S2K_FASTA && format != S2K_FASTQ)) return S2K_ERR_INVALID_ARG;
    *n_reads = *n_bases = 0;
    if (n_bytes && !d_text) return S2K_ERR_INVALID_ARG;
    if (((uintptr_t)d_text & 15) || n_bytes >= (1ull << 32)) {
        s2k::ctx_set_error(ctx, "text must be 16-byte aligned and shorter than 4 GiB");
        return S2K_ERR_INVALID_ARG;
    }
    if (hipSetDevice(s2k::ctx_device(ctx)) != hipSuccess) return S2K_ERR_DEVICE;
    hipStream_t st = s2k::ctx_stream(ctx);
    void *ws = nullptr;
    uint64_t *d_tot = nullptr, h_tot[3] = {0, 0, 0};
    s2k_status rc = S2K_OK;
    if (hipMalloc(&ws, s2k::fx_ws_bytes(n_bytes)) != hipSuccess || hipMalloc((void **)&d_tot, 3 * sizeof(uint64_t)) != hipSuccess)
        rc = S2K_ERR_NOMEM;
    if (rc == S2K_OK && (s2k::fx_parse_count(d_text, n_bytes, format == S2K_FASTQ, ws, d_tot, st) != hipSuccess ||
                         hipMemcpyAsync(h_tot, d_tot, sizeof h_tot, hipMemcpyDeviceToHost, st) != hipSuccess ||
                         hipStreamSynchronize(st) != hipSuccess))
        rc = S2K_ERR_DEVICE;
    if (rc == S2K_OK) {
        *n_reads = h_tot[0];
        *n_bases = h_tot[1];
        if (h_tot[2]) {
            s2k::ctx_set_error(ctx, "malformed FASTA/FASTQ text (FASTQ must be strict 4-line records)");
            rc = S2K_ERR_INVALID_ARG;
        } else if (h_tot[1] > bases_capacity || h_tot[0] + 1 > off_capacity || !d_bases || !d_read_off) {
            rc = S2K_ERR_CAPACITY; // *n_reads / *n_bases hold what is needed
        } else if (s2k::fx_parse_write(d_text, n_bytes, format == S2K_FASTQ, ws, d_tot, d_bases, bases_capacity, d_read_off,
                                       off_capacity, st) != hipSuccess ||
                   hipStreamSynchronize(st) != hipSuccess)
            rc = S2K_ERR_DEVICE;
    }
    if (ws) (void)hipFree(ws);
    if (d_tot) (void)hipFree(d_tot);
    return rc;
}

s2k_status s2k_run_file(s2k_ctx *ctx, const char *path, const s2k_params *params, uint64_t batch_bases,
                        s2k_counts *totals, double *seconds) {
    if (!ctx || !params || !totals || !path) return S2K_ERR_INVALID_ARG;
    auto t_start = std::chrono::steady_clock::now();
    memset(totals, 0, sizeof *totals);
    if (seconds) *seconds = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        s2k::ctx_set_error(ctx, "cannot open file");
        return S2K_ERR_INVALID_ARG;
    }
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return S2K_ERR_INVALID_ARG;
    }
    const uint64_t fsize = (uint64_t)sb.st_size;
    FileWindow fw(fd, fsize);
    // format from the first non-blank byte (as s2k_fastx_open)
    uint64_t pos = 0;
    int fastq = -1;
    while (pos < fsize && fastq < 0) {
        const uint64_t n = fw.load(pos, pos + (1u << 16));
        if (!n) break;
        for (uint64_t i = 0; i < n && fastq < 0; i++, pos++) {
            const char c = fw.at(pos);
            if (c == '\n' || c == '\r') continue;
            fastq = c == '@' ? 1 : c == '>' ? 0 : 2;
        }
        if (fastq >= 0) pos--; // pos = first byte of the first record
    }
    if (fastq == 2 || hipSetDevice(s2k::ctx_device(ctx)) != hipSuccess) {
        close(fd);
        if (fastq == 2) s2k::ctx_set_error(ctx, "not a FASTA/FASTQ file");
        return fastq == 2 ? S2K_ERR_INVALID_ARG : S2K_ERR_DEVICE;
    }
    if (batch_bases == 0) batch_bases = 256ull << 20;
    // bytes of text per batch: FASTA is ~1 byte per base, FASTQ carries a quality string per read
    uint64_t chunk = fastq == 1 ? 2 * batch_bases + batch_bases / 8 : batch_bases + batch_bases / 64;
    if (chunk < (1u << 20)) chunk = 1u << 20;
    if (chunk > (3ull << 30)) chunk = 3ull << 30;

    struct Slot { // two batches in flight: staging + splitting of batch i+1 overlap the k-min-mer kernels of batch i
        void *raw = nullptr, *ws = nullptr, *bases = nullptr, *off = nullptr, *out = nullptr;
        size_t craw = 0, cws = 0, cb = 0, co = 0, cout = 0;
        uint64_t nr = 0, nb = 0;
    } slot[2];
    auto grow = [&](void *&p, size_t &cap, size_t need) -> bool {
        if (need <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, need + need / 8 + 4096) != hipSuccess) return false;
        cap = need + need / 8 + 4096;
        return true;
    };
    hipStream_t copy_stream = nullptr;
    hipEvent_t parsed = nullptr;
    uint64_t *d_tot = nullptr, *h_tot = nullptr;
    bool ok = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&parsed, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void **)&d_tot, 3 * sizeof(uint64_t)) == hipSuccess &&
              hipHostMalloc((void **)&h_tot, 3 * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
    s2k_status st = ok ? S2K_OK : S2K_ERR_DEVICE;

    auto launch = [&](Slot &d, uint64_t cap, bool wait) -> s2k_status { // k-min-mers of the batch held by slot d
        const size_t out_bytes = (((d.nr + 1) * 8 + 255) & ~(size_t)255) + cap * (8 + 4 + 4 + 1) + 1024;
        if (!grow(d.out, d.cout, out_bytes)) return S2K_ERR_NOMEM;
        s2k_device_out o;
        memset(&o, 0, sizeof o);
        char *q = (char *)d.out;
        o.km_capacity = cap;
        o.km_off = (uint64_t *)q;
        q += ((d.nr + 1) * 8 + 255) & ~(size_t)255;
        o.hash = (uint64_t *)q;
        q += cap * 8;
        o.start = (uint32_t *)q;
        q += cap * 4;
        o.end = (uint32_t *)q;
        q += cap * 4;
        o.rev = (uint8_t *)q;
        s2k_counts c;
        return s2k_extract_device(ctx, (const uint8_t *)d.bases, (const uint64_t *)d.off, d.nr, d.nb, params, &o, wait ? &c : nullptr);
    };
    int pending = -1; // slot whose extraction is in flight
    auto collect = [&]() -> s2k_status {
        if (pending < 0) return S2K_OK;
        Slot &d = slot[pending];
        pending = -1;
        s2k_counts c;
        s2k_status s2 = s2k_sync(ctx, &c);
        if (s2 == S2K_ERR_CAPACITY) { // denser than estimated (low-complexity input): k-min-mers <= bases always fits
            s2 = launch(d, d.nb + 1, false);
            if (s2 == S2K_OK) s2 = s2k_sync(ctx, &c);
        }
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

    int cur = 0;
    const bool may_pack = !(params->flags & S2K_FLAG_NO_PACK2);
    int pack_pause = 0; // batches that go as text after one that did not pack (FASTQ, an N-rich or soft-masked stretch); then packing is tried again
    while (st == S2K_OK && pos < fsize) {
        const uint64_t end = find_record_start(fw, pos, pos + chunk, fastq == 1);
        const uint64_t n = end - pos;
        if (n >= (1ull << 32)) {
            s2k::ctx_set_error(ctx, "a single record of 4 GiB or more");
            st = S2K_ERR_READ_TOO_LONG;
            break;
        }
        Slot &d = slot[cur];
        if (!grow(d.raw, d.craw, n + 64) || !grow(d.ws, d.cws, s2k::fx_ws_bytes(n))) {
            st = S2K_ERR_NOMEM;
            break;
        }
        // file -> pinned ring (several threads pread disjoint ranges) -> HBM, then the record splitter, all on copy_stream
        const uint64_t file_off = pos;
        const std::function<bool(char *, size_t, size_t)> from_file = [&](char *dst, size_t off, size_t len) {
            while (len) {
                const ssize_t got = pread(fd, dst, len, (off_t)(file_off + off));
                if (got <= 0) return false;
                dst += got;
                off += (size_t)got;
                len -= (size_t)got;
            }
            return true;
        };
        // FASTA text is DNA plus a header and a newline now and then: it crosses the link 2-bit packed, the other bytes as
        // exceptions, and is rebuilt byte for byte in HBM before the splitter sees it.  FASTQ (half of it quality strings) does
        // not pack: after the first batch that did not, the text is staged as it is.
        hipError_t e;
        const bool try_pack = may_pack && pack_pause == 0;
        if (pack_pause > 0) pack_pause--;
        if (try_pack) {
            bool packed_any = false;
            e = s2k::ctx_stager(ctx).h2d_packed_fill(d.raw, n, copy_stream, from_file, &packed_any);
            // (a batch that did not pack -- or gave up half-way -- was read and scanned in vain: the next seven go as text, then one batch
            // probes again, so that one soft-masked region of a chromosome does not cost the rest of the file its 4x smaller transfers)
            if (!packed_any && n >= (8u << 20)) pack_pause = 7;
        } else {
            e = s2k::ctx_stager(ctx).h2d_fill(d.raw, n, copy_stream, from_file);
        }
        if (e == hipSuccess) e = s2k::fx_parse_count((const uint8_t *)d.raw, n, fastq == 1, d.ws, d_tot, copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_tot, d_tot, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
        if (e != hipSuccess) {
            st = S2K_ERR_DEVICE;
            break;
        }
        if (h_tot[2]) {
            s2k::ctx_set_error(ctx, "malformed FASTA/FASTQ text (FASTQ must be strict 4-line records)");
            st = S2K_ERR_INVALID_ARG;
            break;
        }
        d.nr = h_tot[0];
        d.nb = h_tot[1];
        if (!grow(d.bases, d.cb, d.nb + 256) || !grow(d.off, d.co, (d.nr + 1) * 8)) {
            st = S2K_ERR_NOMEM;
            break;
        }
        e = s2k::fx_parse_write((const uint8_t *)d.raw, n, fastq == 1, d.ws, d_tot, (uint8_t *)d.bases, d.nb, (uint64_t *)d.off,
                                d.nr + 1, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(parsed, copy_stream);
        if (e != hipSuccess) {
            st = S2K_ERR_DEVICE;
            break;
        }
        st = collect(); // previous batch done (its kernels ran while this one was staged and split)
        if (st != S2K_OK) break;
        if (hipStreamWaitEvent(s2k::ctx_stream(ctx), parsed, 0) != hipSuccess) {
            st = S2K_ERR_DEVICE;
            break;
        }
        // output capacity: three times the expected minimizer density, exact retry in collect() if that is too small
        double dens = params->density < 0 ? 0 : params->density > 1 ? 1 : params->density;
        uint64_t cap = (uint64_t)((double)d.nb * dens * 3.0) + 2 * d.nr + (1u << 16);
        if (cap > d.nb + 1) cap = d.nb + 1;
        st = launch(d, cap, false);
        if (st == S2K_OK) pending = cur;
        pos = end;
        cur ^= 1;
    }
    if (st == S2K_OK) st = collect();
    else (void)s2k_sync(ctx, nullptr);
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    for (auto &d : slot)
        for (void *p : {d.raw, d.ws, d.bases, d.off, d.out})
            if (p) (void)hipFree(p);
    if (parsed) (void)hipEventDestroy(parsed);
    if (d_tot) (void)hipFree(d_tot);
    if (h_tot) (void)hipHostFree(h_tot);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    close(fd);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return st;
}

} // extern "C"
