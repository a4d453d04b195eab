// s2k_desc.hip -- second half of the descriptor path (the default for k <= 32): from the tiled kernel's 8-byte records and
// per-tile descriptor words to the final k-min-mers (the tail of KminmersIterator::next, src/lib.rs:231-266).
//
//  1. launch_desc_scan: every tile's (G, p) -- k-min-mers that end before the tile, minimizers before it, and
//     p = min(k-1, minimizers the read that continues into the tile has so far) -- by a scan over the tile words; what a run of
//     tiles does to (G, p) composes associatively (agg_then, s2k_dev.h).  Three small kernels, ~8 B read and 16 B written per tile.
//  2. desc_kminmer_kernel: one wave per tile.  The window of k consecutive minimizers of one read that ENDS at the i-th
//     minimizer of the tile (src/lib.rs:235: it exists iff k-1 minimizers of the same read precede it) is written at
//     G + (windows ending earlier in the tile); its hash is the closed form of src/lib.rs:275-288 over the mixed hashes
//     (src/lib.rs:157-169) in a ring in LDS; the up to k-1 minimizers before the tile are fetched from the slabs of the tiles
//     before it.  Read positions: j = (tile start + offset) - (start of the read segment), from the tile's own list of read
//     starts; km_off (and mn_off) of the reads that start in the tile are written here too.  No per-read table is read.
#include "s2k_dev.h"

namespace s2k {
namespace {

constexpr int SCAN_CH = 16;       // tiles per thread in the chunk passes
constexpr int SCAN_TOP = 1024;    // threads of the single block that scans the chunk words
constexpr unsigned long long M48 = (1ull << 48) - 1ull;

struct PairState { // (G, p) and the minimizer count beside G
    uint64_t G, Gmn;
    uint32_t p;
};
__device__ inline void apply(PairState &s, const AggF &a, uint32_t K1) {
    s.G += agg_windows(a, s.p, K1);
    s.Gmn += a.N;
    s.p = agg_p(a, s.p, K1);
}
// chunk words: {m_f, C, N, q | dep << 8 | pass << 9}
__device__ inline void st_agg(unsigned long long *w, const AggF &a) {
    w[0] = a.m_f;
    w[1] = a.C;
    w[2] = a.N;
    w[3] = (unsigned long long)a.q | ((unsigned long long)(a.dep ? 1 : 0) << 8) | ((unsigned long long)(a.pass ? 1 : 0) << 9);
}
__device__ inline AggF ld_agg(const unsigned long long *w) {
    AggF a;
    a.m_f = w[0];
    a.C = w[1];
    a.N = w[2];
    a.q = (uint32_t)w[3] & 63u;
    a.dep = ((w[3] >> 8) & 1u) != 0;
    a.pass = ((w[3] >> 9) & 1u) != 0;
    return a;
}

__global__ __launch_bounds__(256) void desc_scan_chunks(const unsigned long long *__restrict__ agg, uint64_t n_tiles, uint32_t K1,
                                                        unsigned long long *__restrict__ chunk_agg, const Counts *__restrict__ counts) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t t0 = c * SCAN_CH;
    if (t0 >= n_tiles || counts->need_legacy || counts->bad_input || counts->pool_overflow) return;
    AggF a = agg_identity();
    for (int i = 0; i < SCAN_CH && t0 + i < n_tiles; i++) a = agg_then(a, agg_unpack(agg[t0 + i]), K1);
    st_agg(chunk_agg + 4 * c, a);
}

// one block: thread i folds its share of the chunk words, thread 0 walks the SCAN_TOP partial results, then every thread walks
// its share again and leaves the state at the start of every chunk
__global__ __launch_bounds__(SCAN_TOP) void desc_scan_top(const unsigned long long *__restrict__ chunk_agg, uint64_t n_chunks, uint32_t K1,
                                                          unsigned long long *__restrict__ chunk_state, const Counts *__restrict__ counts) {
    __shared__ unsigned long long part[SCAN_TOP][4];
    __shared__ unsigned long long start[SCAN_TOP][3];
    if (counts->need_legacy || counts->bad_input || counts->pool_overflow) return;
    const uint64_t per = (n_chunks + SCAN_TOP - 1) / SCAN_TOP;
    const uint64_t c0 = per * threadIdx.x, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    AggF a = agg_identity();
    for (uint64_t c = c0; c < c1; c++) a = agg_then(a, ld_agg(chunk_agg + 4 * c), K1);
    st_agg(part[threadIdx.x], a);
    __syncthreads();
    if (threadIdx.x == 0) {
        PairState s{0, 0, 0};
        for (int i = 0; i < SCAN_TOP; i++) {
            start[i][0] = s.G;
            start[i][1] = s.Gmn;
            start[i][2] = s.p;
            apply(s, ld_agg(part[i]), K1);
        }
    }
    __syncthreads();
    PairState s{start[threadIdx.x][0], start[threadIdx.x][1], (uint32_t)start[threadIdx.x][2]};
    for (uint64_t c = c0; c < c1; c++) {
        chunk_state[3 * c] = s.G;
        chunk_state[3 * c + 1] = s.Gmn;
        chunk_state[3 * c + 2] = s.p;
        apply(s, ld_agg(chunk_agg + 4 * c), K1);
    }
}

__global__ __launch_bounds__(256) void desc_scan_write(const unsigned long long *__restrict__ agg, uint64_t n_tiles, uint32_t K1,
                                                       const unsigned long long *__restrict__ chunk_state, TileState *__restrict__ state,
                                                       const Counts *__restrict__ counts) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t t0 = c * SCAN_CH;
    if (t0 >= n_tiles || counts->need_legacy || counts->bad_input || counts->pool_overflow) return;
    PairState s{chunk_state[3 * c], chunk_state[3 * c + 1], (uint32_t)chunk_state[3 * c + 2]};
    for (int i = 0; i < SCAN_CH && t0 + i < n_tiles; i++) {
        state[t0 + i].g = ((unsigned long long)s.p << 48) | s.G;
        state[t0 + i].gmn = s.Gmn;
        apply(s, agg_unpack(agg[t0 + i]), K1);
    }
    if (t0 + SCAN_CH >= n_tiles) { // the totals
        state[n_tiles].g = s.G;
        state[n_tiles].gmn = s.Gmn;
    }
}

// ---- k-min-mers of one tile ------------------------------------------------------------------------------------------------
constexpr int DK_WAVES = 4;
constexpr int DK_KMAX = 32;

__device__ inline uint32_t dk_incl_scan(uint32_t v, int lane) { // inclusive scan over the wave (shuffles; this kernel is not the hot one)
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

__global__ __launch_bounds__(64 * DK_WAVES, 4) void desc_kminmer_kernel(uint64_t n_tiles, uint64_t n_reads, Desc dz, Records rec, Counts *counts) {
    __shared__ unsigned long long s_ring[DK_WAVES][64 + DK_KMAX];
    __shared__ unsigned long long s_rs[DK_WAVES][META_SEGS];
    __shared__ int32_t s_adj[DK_WAVES][META_SEGS];
    __shared__ uint32_t s_segb[DK_WAVES][META_SEGS];
    __shared__ uint32_t s_jcar[DK_WAVES][DK_KMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t t = (uint64_t)blockIdx.x * DK_WAVES + w;
    if (t >= n_tiles) return; // whole waves leave together; no block-level barrier below
    if (counts->need_legacy || counts->bad_input || counts->pool_overflow) return;
    const uint32_t k = dz.k, K1 = k - 1;
    const AggF ag = agg_unpack(dz.agg[t]);
    const uint32_t N = (uint32_t)ag.N;
    const TileState st = dz.state[t];
    const uint64_t G = st.g & M48, Gmn = st.gmn;
    const uint32_t p_in = ag.dep ? (uint32_t)(st.g >> 48) & 63u : 0u;
    const TileMeta *m = &dz.meta[t];
    const uint64_t base = m->rec_base, rs0 = m->rs0, t0 = t * (uint64_t)TILE_BASES;
    const uint32_t r0 = m->r0, nb = m->nb, nrd = m->nrd;
    // read segments: hits before segment s, where its read starts, and what turns a hit index into an output offset
    uint32_t segstart = 0;
    if ((uint32_t)lane <= nb) segstart = m->segstart[lane];
    const uint32_t nextstart = __shfl_down(segstart, 1);
    const uint32_t mine = (uint32_t)lane <= nb ? ((uint32_t)lane < nb ? nextstart : N) - segstart : 0u;
    const uint32_t skip = lane == 0 ? (ag.dep ? K1 - p_in : K1) : K1; // minimizers of a segment that end no k-min-mer
    const uint32_t wseg = (uint32_t)lane <= nb && mine > skip ? mine - skip : 0u;
    const uint32_t winc = dk_incl_scan(wseg, lane);
    const uint32_t Wb = winc - wseg, Wt = __shfl(winc, 63);
    if ((uint32_t)lane <= nb) {
        s_segb[w][lane] = segstart;
        s_adj[w][lane] = (int32_t)Wb - (int32_t)segstart - (int32_t)skip;
        s_rs[w][lane] = lane ? t0 + m->rs16[lane] : rs0;
    }
    // km_off / mn_off of the reads that start in (t0, end of the tile]: the first nb inside, the rest exactly at the end
    if (lane >= 1 && (uint32_t)lane <= nrd) {
        dz.o_km_off[(uint64_t)r0 + lane] = (uint32_t)lane <= nb ? G + Wb : G + Wt;
        if (dz.mn_capacity) dz.o_mn_off[(uint64_t)r0 + lane] = (uint32_t)lane <= nb ? Gmn + segstart : Gmn + N;
    }
    if (t == 0) // reads 0 .. r0 start at position 0
        for (uint64_t r = lane; r <= (uint64_t)r0; r += 64) {
            dz.o_km_off[r] = 0;
            if (dz.mn_capacity) dz.o_mn_off[r] = 0;
        }
    if (t + 1 == n_tiles && lane == 0) {
        dz.o_km_off[n_reads] = G + Wt;
        if (dz.mn_capacity) dz.o_mn_off[n_reads] = Gmn + N;
    }
    if (N == 0) return;
    // the up to k-1 minimizers of the continuing read that lie before the tile (lane q: the (q+1)-th counted backwards): the last
    // records of the tiles before this one -- a tile that is one stretch of the read hands on to the tile before it
    if ((uint32_t)lane < p_in) {
        uint32_t rem = (uint32_t)lane;
        uint64_t u = t;
        uint32_t idx = 0;
        bool found = false;
        while (u > 0) {
            u--;
            const uint32_t Nu = (uint32_t)((dz.agg[u] >> 28) & 0x3FFFu);
            if (rem < Nu) {
                idx = Nu - 1 - rem;
                found = true;
                break;
            }
            rem -= Nu;
        }
        if (found) { // (always: p counted these minimizers)
            const uint64_t bu = dz.meta[u].rec_base;
            const uint32_t pos = rec.j[bu + idx];
            s_ring[w][K1 - 1 - lane] = mix32(rec.hash[bu + idx]);
            s_jcar[w][K1 - 1 - lane] = (uint32_t)(u * (uint64_t)TILE_BASES + (pos & 0x3FFFu) - rs0); // the same read: it starts at rs0
        }
    }
    wave_sync();
    uint32_t jprev = (uint32_t)lane >= 64u - K1 ? s_jcar[w][lane - (64u - K1)] : 0u; // "the round before the first": the k-1 minimizers before the tile
    uint64_t xacc = 0;
    for (uint32_t i0 = 0; i0 < N; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool act = i < N;
        uint32_t h32 = 0, pos = 0;
        if (act) {
            h32 = rec.hash[base + i];
            pos = rec.j[base + i];
        }
        uint32_t c = 0;
        for (uint32_t s = 1; s <= nb; s++) c += (s_segb[w][s] <= i); // wave-uniform trip count, LDS broadcast
        const uint64_t rstart = s_rs[w][c];
        const uint32_t j = (uint32_t)(t0 + (pos & 0x3FFFu) - rstart);
        const uint32_t jend = j + (pos >> 14);
        const uint64_t xm = mix32(h32); // src/lib.rs:157-169
        wave_sync();
        s_ring[w][K1 + lane] = xm;
        wave_sync();
        // rank of the hit inside its read, capped: the window that ENDS here exists iff k-1 minimizers of the read precede it
        const uint32_t before = (c == 0 ? p_in : 0u) + (i - s_segb[w][c]);
        const bool win = act && before >= K1;
        uint64_t f = 0, r = 0;
        for (uint32_t mm = 0; mm < k; mm++) { // ring[lane + mm] = hit i - (k-1) + mm
            const uint64_t xw = s_ring[w][lane + mm];
            f = ((f << 1) | (f >> 63)) ^ xw; // F  = XOR rotl(x_m, k-1-m)   (src/lib.rs:238-249, closed form :275-288)
            r = ((r >> 1) | (r << 63)) ^ xw; // Rv = rotl(XOR rotr(x_m, k-1-m), k-1) = XOR rotl(x_m, m)
        }
        const uint64_t rvv = rotl64(r, K1);
        // start = j of the window's first minimizer: k-1 hits back, in this round or the one before it
        const uint32_t jsame = (uint32_t)__shfl((int)j, (lane - (int)K1) & 63), jbefore = (uint32_t)__shfl((int)jprev, (lane - (int)K1) & 63);
        const uint32_t jstart = (uint32_t)lane >= K1 ? jsame : jbefore;
        const uint64_t hmin = f < rvv ? f : rvv;
        if (win) {
            const uint64_t o = G + (uint64_t)(int64_t)((int32_t)i + s_adj[w][c]);
            xacc ^= hmin;
            if (o < dz.km_capacity) {
                if (dz.o_hash) dz.o_hash[o] = hmin;
                if (dz.o_start) dz.o_start[o] = jstart;
                if (dz.o_end) dz.o_end[o] = jend;
                if (dz.o_rev) dz.o_rev[o] = (uint8_t)(rvv < f); // src/lib.rs:250-251
            }
        }
        if (dz.mn_capacity && act) { // optional minimizer triples (NtHashHPCIterator::Item, src/nthash_hpc.rs:193)
            const uint64_t g = Gmn + i;
            if (g < dz.mn_capacity) {
                dz.o_mn_j[g] = j;
                dz.o_mn_jend[g] = jend;
                dz.o_mn_hash[g] = h32;
            }
        }
        wave_sync();
        if ((uint32_t)lane >= 64u - K1) s_ring[w][lane - (64u - K1)] = xm; // the last k-1 hits of a full round lead the next one
        jprev = j;
    }
    for (int o = 32; o > 0; o >>= 1) xacc ^= __shfl_xor(xacc, o);
    if (lane == 0 && xacc) atomicXor((unsigned long long *)&dz.xor_shards[t & (XOR_SHARDS - 1)], (unsigned long long)xacc);
}

} // namespace

size_t desc_scan_tmp_words(uint64_t n_tiles) { return 7 * ((n_tiles + SCAN_CH - 1) / SCAN_CH) + 8; }

hipError_t launch_desc_scan(uint64_t n_tiles, Desc dz, unsigned long long *scan_tmp, const Counts *counts, hipStream_t st) {
    if (n_tiles == 0) return hipSuccess;
    const uint64_t n_chunks = (n_tiles + SCAN_CH - 1) / SCAN_CH;
    unsigned long long *chunk_agg = scan_tmp, *chunk_state = scan_tmp + 4 * n_chunks;
    const uint32_t K1 = dz.k - 1;
    const unsigned blocks = (unsigned)((n_chunks + 255) / 256);
    hipLaunchKernelGGL(desc_scan_chunks, dim3(blocks), dim3(256), 0, st, dz.agg, n_tiles, K1, chunk_agg, counts);
    hipLaunchKernelGGL(desc_scan_top, dim3(1), dim3(SCAN_TOP), 0, st, chunk_agg, n_chunks, K1, chunk_state, counts);
    hipLaunchKernelGGL(desc_scan_write, dim3(blocks), dim3(256), 0, st, dz.agg, n_tiles, K1, chunk_state, dz.state, counts);
    return hipGetLastError();
}

hipError_t launch_desc_kminmers(uint64_t n_tiles, uint64_t n_reads, Desc dz, Records rec, Counts *counts, hipStream_t st) {
    if (n_tiles == 0) return hipSuccess;
    if (dz.k == 0 || dz.k > (uint32_t)DK_KMAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(desc_kminmer_kernel, dim3((unsigned)((n_tiles + DK_WAVES - 1) / DK_WAVES)), dim3(64 * DK_WAVES), 0, st, n_tiles, n_reads, dz,
                       rec, counts);
    return hipGetLastError();
}

} // namespace s2k
