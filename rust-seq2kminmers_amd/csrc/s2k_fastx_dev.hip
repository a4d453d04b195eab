// s2k_fastx_dev.hip -- FASTA / FASTQ record splitting on the GPU (ingest + batching).
//
// What it replaces: the per-record parsing that rust_parallelfastx does on a CPU thread pool before each record
// reaches KminmersIterator::new (src/main.rs:65-79).  Here the raw file bytes cross PCIe once and the device turns
// them into exactly what s2k_extract_device consumes: the sequence bytes of all records back to back plus the
// n_reads+1 read_off table.  The host only has to cut the file at record starts.
//
// The text is cut into blocks of 4096 bytes (256 threads x 16 bytes).  A byte's role depends on two running
// quantities: the number of newlines before it (FASTQ: line index mod 4 = header / sequence / '+' / quality) and
// the start of its line (FASTA: a line is a header iff it starts with '>').  Both are prefix scans -- a sum and a
// max -- done per thread (16 bytes), per block (LDS) and across blocks (one small single-block kernel), so the
// work is three streaming passes over the text:
//   pass 0  newline count + last line start of every block
//   pass 1  with those carries: classify bytes, count sequence bytes / record starts per block, flag bad syntax
//   pass 2  with all three carries: write sequence bytes compacted and read_off[record] = bases before it
// Each pass is bound by HBM; for a 256 MB chunk the three take well under a millisecond each.
#include "s2k_dev.h"

namespace s2k {

namespace {

constexpr int FX_THREADS = 256;
constexpr int FX_PER = 16;
constexpr uint32_t FX_BLOCK = FX_THREADS * FX_PER;

struct Bytes16 {
    uint8_t c[16];
};

__device__ __forceinline__ Bytes16 load16(const uint8_t *raw, uint64_t pos, uint64_t n) {
    Bytes16 b;
    if (pos + 16 <= n) {
        *(uint4 *)b.c = *(const uint4 *)(raw + pos);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) b.c[j] = pos + j < n ? raw[pos + j] : (uint8_t)'\n';
    }
    return b;
}

// block-wide exclusive scans of one sum and one max (256 threads), plus the block totals
__device__ __forceinline__ void block_scan(uint32_t vs, uint32_t vm, uint32_t *ex_s, uint32_t *ex_m, uint32_t *tot_s,
                                           uint32_t *tot_m, uint32_t *lds_s, uint32_t *lds_m) {
    const int t = threadIdx.x;
    lds_s[t] = vs;
    lds_m[t] = vm;
    __syncthreads();
#pragma unroll
    for (int d = 1; d < FX_THREADS; d <<= 1) {
        uint32_t a = 0, b = 0;
        if (t >= d) {
            a = lds_s[t - d];
            b = lds_m[t - d];
        }
        __syncthreads();
        if (t >= d) {
            lds_s[t] += a;
            lds_m[t] = lds_m[t] > b ? lds_m[t] : b;
        }
        __syncthreads();
    }
    *ex_s = t ? lds_s[t - 1] : 0;
    *ex_m = t ? lds_m[t - 1] : 0;
    *tot_s = lds_s[FX_THREADS - 1];
    *tot_m = lds_m[FX_THREADS - 1];
    __syncthreads();
}

__device__ __forceinline__ void newlines16(const Bytes16 &b, uint64_t pos, uint32_t *cnt, uint32_t *last) {
    uint32_t c = 0, l = 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (b.c[j] == '\n') {
            c++;
            l = (uint32_t)(pos + j + 1); // the line after this newline starts here
        }
    *cnt = c;
    *last = l;
}

struct Roles {
    uint32_t seq, rec, err; // bit j: byte j is a sequence byte / is the first byte of a record
};

// nl_before = newlines before this thread's first byte; line_start = start of the line that byte belongs to
__device__ __forceinline__ Roles classify16(const uint8_t *raw, const Bytes16 &b, uint64_t pos, uint32_t nl_before,
                                            uint32_t line_start, int fastq) {
    Roles r{0, 0, 0};
    bool at_ls = line_start == pos;
    if (fastq) {
        uint32_t line = nl_before & 3;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint8_t c = b.c[j];
            if (c == '\n') {
                line = (line + 1) & 3;
                at_ls = true;
                continue;
            }
            if (c == '\r') continue;
            if (at_ls) {
                if (line == 0) {
                    if (c == '@') r.rec |= 1u << j;
                    else r.err++;
                } else if (line == 2 && c != '+')
                    r.err++;
                at_ls = false;
            }
            if (line == 1) r.seq |= 1u << j;
        }
    } else {
        bool hdr = !at_ls && raw[line_start] == '>';
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint8_t c = b.c[j];
            if (c == '\n') {
                at_ls = true;
                hdr = false;
                continue;
            }
            if (at_ls) {
                hdr = c == '>';
                if (hdr) r.rec |= 1u << j;
                at_ls = false;
            }
            if (!hdr && c != '\r') r.seq |= 1u << j;
        }
    }
    return r;
}

__global__ __launch_bounds__(FX_THREADS) void fx_pass0(const uint8_t *raw, uint64_t n, uint32_t *bnl, uint32_t *blast) {
    __shared__ uint32_t ls[FX_THREADS], lm[FX_THREADS];
    const uint64_t pos = (uint64_t)blockIdx.x * FX_BLOCK + threadIdx.x * FX_PER;
    const Bytes16 b = load16(raw, pos, n);
    uint32_t c, l, es, em, ts, tm;
    newlines16(b, pos, &c, &l);
    block_scan(c, l, &es, &em, &ts, &tm, ls, lm);
    if (threadIdx.x == 0) {
        bnl[blockIdx.x] = ts;
        blast[blockIdx.x] = tm;
    }
}

// exclusive scans over the per-block arrays, in place: sums for a (and c when given), max for m
__global__ __launch_bounds__(1024) void fx_scan_blocks(uint32_t *a, uint32_t *m, uint32_t *c, uint32_t nb, uint64_t *totals) {
    __shared__ uint32_t sa[1024], sm[1024], sc[1024];
    const int t = threadIdx.x;
    uint32_t ca = 0, cm = 0, cc = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + t;
        const uint32_t va = i < nb ? a[i] : 0, vm = (m && i < nb) ? m[i] : 0, vc = (c && i < nb) ? c[i] : 0;
        sa[t] = va;
        sm[t] = vm;
        sc[t] = vc;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint32_t xa = 0, xm = 0, xc = 0;
            if (t >= d) {
                xa = sa[t - d];
                xm = sm[t - d];
                xc = sc[t - d];
            }
            __syncthreads();
            if (t >= d) {
                sa[t] += xa;
                sm[t] = sm[t] > xm ? sm[t] : xm;
                sc[t] += xc;
            }
            __syncthreads();
        }
        if (i < nb) {
            a[i] = ca + sa[t] - va;
            if (m) {
                const uint32_t ex = t ? sm[t - 1] : 0;
                m[i] = ex > cm ? ex : cm;
            }
            if (c) c[i] = cc + sc[t] - vc;
        }
        const uint32_t ta = sa[1023], tm = sm[1023], tc = sc[1023];
        __syncthreads();
        ca += ta;
        cm = tm > cm ? tm : cm;
        cc += tc;
    }
    if (t == 0 && totals) {
        totals[0] = cc; // records   (second sum)
        totals[1] = ca; // sequence bytes (first sum)
    }
}

template <bool WRITE>
__global__ __launch_bounds__(FX_THREADS) void fx_pass12(const uint8_t *raw, uint64_t n, int fastq, const uint32_t *bnl_ex,
                                                       const uint32_t *blast_ex, uint32_t *bseq, uint32_t *brec,
                                                       uint64_t *totals, uint8_t *bases, uint64_t bases_cap,
                                                       uint64_t *read_off, uint64_t off_cap) {
    __shared__ uint32_t ls[FX_THREADS], lm[FX_THREADS];
    const uint64_t pos = (uint64_t)blockIdx.x * FX_BLOCK + threadIdx.x * FX_PER;
    const Bytes16 b = load16(raw, pos, n);
    uint32_t c, l, es, em, ts, tm;
    newlines16(b, pos, &c, &l);
    block_scan(c, l, &es, &em, &ts, &tm, ls, lm);
    const uint32_t nl_before = bnl_ex[blockIdx.x] + es;
    const uint32_t carry = blast_ex[blockIdx.x];
    const uint32_t line_start = em > carry ? em : carry;
    const Roles r = classify16(raw, b, pos, nl_before, line_start, fastq);
    const uint32_t nseq = __popc(r.seq), nrec = __popc(r.rec);
    uint32_t e_seq, e_rec, t_seq, t_rec;
    block_scan(nseq, 0, &e_seq, &em, &t_seq, &tm, ls, lm);
    block_scan(nrec, 0, &e_rec, &em, &t_rec, &tm, ls, lm);
    if constexpr (!WRITE) {
        if (threadIdx.x == 0) {
            bseq[blockIdx.x] = t_seq;
            brec[blockIdx.x] = t_rec;
        }
        if (r.err) atomicAdd((unsigned long long *)&totals[2], (unsigned long long)r.err);
    } else {
        uint64_t so = (uint64_t)bseq[blockIdx.x] + e_seq; // sequence bytes before this thread's first byte
        uint64_t ro = (uint64_t)brec[blockIdx.x] + e_rec;
        if (blockIdx.x == 0 && threadIdx.x == 0 && totals[0] < off_cap) read_off[totals[0]] = totals[1];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if ((r.rec >> j) & 1) {
                if (ro < off_cap) read_off[ro] = so;
                ro++;
            }
            if ((r.seq >> j) & 1) {
                if (so < bases_cap) bases[so] = b.c[j];
                so++;
            }
        }
    }
}

} // namespace

size_t fx_ws_bytes(uint64_t n) {
    const uint64_t nb = (n + FX_BLOCK - 1) / FX_BLOCK + 1;
    return (size_t)(4 * nb * sizeof(uint32_t) + 256);
}

// Phase A (count): after it, d_totals = { n_records, n_sequence_bytes, n_syntax_errors }.
// Phase B (write): fills d_bases[0..n_seq) and d_read_off[0..n_records].  Both are stream-ordered.
hipError_t fx_parse_count(const uint8_t *d_raw, uint64_t n, int fastq, void *ws, uint64_t *d_totals, hipStream_t st) {
    const uint32_t nb = (uint32_t)((n + FX_BLOCK - 1) / FX_BLOCK);
    hipError_t e = hipMemsetAsync(d_totals, 0, 3 * sizeof(uint64_t), st);
    if (e != hipSuccess || nb == 0) return e;
    uint32_t *bnl = (uint32_t *)ws, *blast = bnl + nb, *bseq = blast + nb, *brec = bseq + nb;
    fx_pass0<<<nb, FX_THREADS, 0, st>>>(d_raw, n, bnl, blast);
    fx_scan_blocks<<<1, 1024, 0, st>>>(bnl, blast, nullptr, nb, nullptr);
    fx_pass12<false><<<nb, FX_THREADS, 0, st>>>(d_raw, n, fastq, bnl, blast, bseq, brec, d_totals, nullptr, 0, nullptr, 0);
    fx_scan_blocks<<<1, 1024, 0, st>>>(bseq, nullptr, brec, nb, d_totals);
    return hipGetLastError();
}

hipError_t fx_parse_write(const uint8_t *d_raw, uint64_t n, int fastq, void *ws, uint64_t *d_totals, uint8_t *d_bases,
                          uint64_t bases_cap, uint64_t *d_read_off, uint64_t off_cap, hipStream_t st) {
    const uint32_t nb = (uint32_t)((n + FX_BLOCK - 1) / FX_BLOCK);
    if (nb == 0) return hipMemsetAsync(d_read_off, 0, sizeof(uint64_t), st);
    uint32_t *bnl = (uint32_t *)ws, *blast = bnl + nb, *bseq = blast + nb, *brec = bseq + nb;
    fx_pass12<true><<<nb, FX_THREADS, 0, st>>>(d_raw, n, fastq, bnl, blast, bseq, brec, d_totals, d_bases, bases_cap,
                                               d_read_off, off_cap);
    return hipGetLastError();
}

} // namespace s2k
