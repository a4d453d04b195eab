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
#include <cstdlib>
#include <type_traits>

namespace s2k {
namespace {

constexpr unsigned long long M48 = (1ull << 48) - 1ull;

struct PairState { // (G, p) and the minimizer count beside G
    uint64_t G, Gmn;
    uint32_t p;
};
__device__ inline PairState applied(PairState s, const AggF &a, uint32_t K1) {
    s.G += agg_windows(a, s.p, K1);
    s.Gmn += a.N;
    s.p = agg_p(a, s.p, K1);
    return s;
}
// words of the upper scan levels: {m_f, C, N, q | dep << 8 | pass << 9}; states: {G, Gmn, p}
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
template <bool PACKED>
__device__ inline AggF ld_level(const unsigned long long *in, uint64_t i) { // level 0: the tiles' packed words; above: 32-byte words
    if constexpr (PACKED) return agg_unpack(in[i]);
    else return ld_agg(in + 4 * i);
}
__device__ inline uint64_t shfl_up64(uint64_t v, int o) {
    return ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), o) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, o);
}
// inclusive scan of `a` over the 64 lanes under agg_then (lane i: lanes 0 .. i in order)
__device__ inline AggF wave_scan_agg(AggF a, int lane, uint32_t K1) {
    for (int o = 1; o < 64; o <<= 1) {
        AggF b;
        b.m_f = shfl_up64(a.m_f, o);
        b.C = shfl_up64(a.C, o);
        b.N = shfl_up64(a.N, o);
        const uint32_t f = (uint32_t)__shfl_up((int)(a.q | (a.dep ? 256u : 0u) | (a.pass ? 512u : 0u)), o);
        b.q = f & 63u;
        b.dep = (f & 256u) != 0;
        b.pass = (f & 512u) != 0;
        if (lane >= o) a = agg_then(b, a, K1);
    }
    return a;
}
// The scan and k-min-mer kernels of chunk c run on the second stream while the minimizer kernel of chunk c+1 may be SETTING these flags on
// the caller's stream (plain loads here, plain stores there).  That is benign because of three invariants, which any change must keep:
// (1) the flags only ever go 0 -> 1 inside a call (memset per enqueue); (2) every one of them makes the host run the call again
// (s2k_api.hip: finish()), so (3) no output of a call in which a flag was raised is ever consumed -- kernels of one chunk that disagree about
// scan_off leave garbage states and partly written outputs, which the re-run overwrites.
__device__ inline bool scan_off(const Counts *counts) { return counts->need_legacy || counts->need_runs || counts->bad_input || counts->pool_overflow; }

// one wave per 64 entries of a level: what the 64 together do (coalesced loads, log-step scan)
template <bool PACKED>
__global__ __launch_bounds__(256) void desc_scan_reduce(const unsigned long long *__restrict__ in, uint64_t n, uint32_t K1,
                                                        unsigned long long *__restrict__ out, const Counts *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), i = c * 64 + lane;
    if (c * 64 >= n || scan_off(counts)) return;
    const AggF a = wave_scan_agg(i < n ? ld_level<PACKED>(in, i) : agg_identity(), lane, K1);
    if (lane == 63) st_agg(out + 4 * c, a);
}
// top level (at most a few hundred words): one wave walks it in batches of 64 and leaves the state BEFORE every entry
__global__ __launch_bounds__(64) void desc_scan_top(const unsigned long long *__restrict__ in, uint64_t n, uint32_t K1,
                                                    unsigned long long *__restrict__ states, const TileState *__restrict__ init,
                                                    const Counts *__restrict__ counts) {
    const int lane = threadIdx.x;
    if (scan_off(counts)) return;
    PairState s{0, 0, 0};
    if (init) { // the totals the scan of the tiles before this range left: {p << 48 | G, minimizers}
        s.G = init->g & M48;
        s.p = (uint32_t)(init->g >> 48) & 63u;
        s.Gmn = init->gmn;
    }
    for (uint64_t b0 = 0; b0 < n; b0 += 64) {
        const uint64_t i = b0 + lane;
        const AggF inc = wave_scan_agg(i < n ? ld_agg(in + 4 * i) : agg_identity(), lane, K1);
        AggF exc;
        exc.m_f = shfl_up64(inc.m_f, 1);
        exc.C = shfl_up64(inc.C, 1);
        exc.N = shfl_up64(inc.N, 1);
        const uint32_t f = (uint32_t)__shfl_up((int)(inc.q | (inc.dep ? 256u : 0u) | (inc.pass ? 512u : 0u)), 1);
        exc.q = f & 63u;
        exc.dep = (f & 256u) != 0;
        exc.pass = (f & 512u) != 0;
        if (lane == 0) exc = agg_identity();
        const PairState mine = applied(s, exc, K1);
        if (i < n) {
            states[3 * i] = mine.G;
            states[3 * i + 1] = mine.Gmn;
            states[3 * i + 2] = mine.p;
        }
        const PairState after = applied(s, inc, K1); // lane 63: the whole batch
        s.G = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(after.G >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)after.G, 63);
        s.Gmn = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(after.Gmn >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)after.Gmn, 63);
        s.p = (uint32_t)__shfl((int)after.p, 63);
    }
}
// one wave per 64 entries: from the state before the 64 (chunk_states) to the state before every one of them.  Level 0 writes
// the TileState records (and the totals behind the last tile), the upper level its 24-byte states.
template <bool PACKED>
__global__ __launch_bounds__(256) void desc_scan_expand(const unsigned long long *__restrict__ in, uint64_t n, uint32_t K1,
                                                        const unsigned long long *__restrict__ chunk_states, unsigned long long *__restrict__ out_states,
                                                        TileState *__restrict__ out_tiles, const Counts *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), i = c * 64 + lane;
    if (c * 64 >= n || scan_off(counts)) return;
    const PairState s0{chunk_states[3 * c], chunk_states[3 * c + 1], (uint32_t)chunk_states[3 * c + 2]};
    const AggF inc = wave_scan_agg(i < n ? ld_level<PACKED>(in, i) : agg_identity(), lane, K1);
    AggF exc;
    exc.m_f = shfl_up64(inc.m_f, 1);
    exc.C = shfl_up64(inc.C, 1);
    exc.N = shfl_up64(inc.N, 1);
    const uint32_t f = (uint32_t)__shfl_up((int)(inc.q | (inc.dep ? 256u : 0u) | (inc.pass ? 512u : 0u)), 1);
    exc.q = f & 63u;
    exc.dep = (f & 256u) != 0;
    exc.pass = (f & 512u) != 0;
    if (lane == 0) exc = agg_identity();
    const PairState mine = applied(s0, exc, K1);
    if (i < n) {
        if constexpr (PACKED) {
            out_tiles[i].g = ((unsigned long long)mine.p << 48) | mine.G;
            out_tiles[i].gmn = mine.Gmn;
            if (i + 1 == n) { // the totals, in the form of a tile state: the scan of the next range of tiles starts from them
                const PairState tot = applied(s0, inc, K1);
                out_tiles[n].g = ((unsigned long long)tot.p << 48) | tot.G;
                out_tiles[n].gmn = tot.Gmn;
            }
        } else {
            out_states[3 * i] = mine.G;
            out_states[3 * i + 1] = mine.Gmn;
            out_states[3 * i + 2] = mine.p;
        }
    }
}

// ---- k-min-mers of one tile ------------------------------------------------------------------------------------------------
constexpr int DK_WAVES = 4;
constexpr int DK_KMAX = 32;

// inclusive scan over the 64 lanes with DPP row shifts / broadcasts (as in the tiled kernel; no LDS round trips)
__device__ inline uint32_t dk_incl_scan(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    uint32_t a = v + t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    a += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false); // row_shr:3
    a += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xf, 0xe, false); // row_shr:4, banks 1-3
    a += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xc, false); // row_shr:8, banks 2-3
    a += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
    a += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
    a += t;
    return a;
}
// the same steps with XOR: lane i = XOR of lanes 0 .. i
__device__ inline uint32_t dk_incl_xor(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    uint32_t a = v ^ t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    a ^= t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false); // row_shr:3
    a ^= t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xf, 0xe, false); // row_shr:4, banks 1-3
    a ^= t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xc, false); // row_shr:8, banks 2-3
    a ^= t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
    a ^= t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
    a ^= t;
    return a;
}
__device__ inline uint32_t dk_lane(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

// KT: compile-time k (the window loop is unrolled and reads the ring at constant offsets), or 0: run-time k <= 32
// FULL: the caller wants all four k-min-mer arrays and no minimizer triples (the common call): no per-array tests in the round loop, and a tile whose
// windows all fit the arrays' capacity (every tile of a call that fits) stores at 32-bit offsets from per-tile scalar bases without a per-lane test
// COAL (compile-time k, FULL): the kernel has the CU to itself (the Regular family's k-min-mer stage runs BEHIND its minimizer kernel) and LDS to spare: a
// lane's consecutive windows go through a per-wave staging area so that every store instruction writes 64 CONSECUTIVE windows (whole 128-byte lines per
// array).  Without it a lane writes its RB windows side by side and a store instruction touches RB times as many lines, each only partly: beside the
// persistent Hpc kernel (where there is no LDS for the area) that is hidden, alone it made the kernel 20 % slower than the one-window-per-lane rounds
// (profiles/r06_ab_km1.txt).
constexpr int DK_STAGE_RECS = 192, DK_STAGE_BYTES = DK_STAGE_RECS * 20; // per wave: {hash, start, end} 16 B + {window index | rev << 31} 4 B per minimizer of a batch
template <int KT, bool FULL, bool COAL = false>
__global__ __launch_bounds__(64 * DK_WAVES, 8) void desc_kminmer_kernel(uint64_t tile_begin, uint64_t tile_end, uint64_t n_tiles, uint64_t n_reads, Desc dz,
                                                                        Records rec, Counts *counts) {
    // compile-time k (lane-serial windows, below): the ring only carries the k-1 minimizers before a batch (+ the 64 counts of the slow look-back, which
    // share its upper entries); run-time k: a round's 64 mixed hashes behind them
    __shared__ unsigned long long s_ring[DK_WAVES][((KT > 0 && FULL) ? 32 : 64) + DK_KMAX];
    __shared__ uint4 s_seg[DK_WAVES][META_SEGS]; // per read segment of the tile: {hits before it, hit index -> window index, (tile start - read start) mod 2^32, -}
    __shared__ uint32_t s_jcar[DK_WAVES][DK_KMAX];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // w in an SGPR: per-tile values are scalar
    // cumulative minimizer counts of the 63 tiles before this one (round trip 2 only): they share the ring's upper 64 entries, which the rounds
    // overwrite afterwards (the entries below DK_KMAX take the minimizers found with them) -- 6.5 -> 5.5 KiB per block, so that two blocks fit
    // beside the persistent minimizer kernel's block whatever that one's rows cost (s2k_tile_impl.h: HpcLds)
    uint32_t *const s_cum_w = reinterpret_cast<uint32_t *>(&s_ring[w][DK_KMAX]);
    const uint64_t t = tile_begin + (uint64_t)blockIdx.x * DK_WAVES + w;
    if (t >= tile_end) return; // whole waves leave together; no block-level barrier below
    // ---- round trip 1: everything whose address depends on nothing loaded -- the words of this tile and of the 63 before it,
    //      its state, its segment list, and its first 64 records under the assumption that they sit in the tile's own slab
    const uint32_t off_flags = counts->need_legacy | counts->need_runs | counts->bad_input | counts->pool_overflow;
    const unsigned long long agw = t >= (uint64_t)lane ? dz.agg[t - lane] : 0ull; // lane 0: this tile; lane i: tile t - i
    const TileState st = dz.state[t];
    const TileMeta *m = &dz.meta[t];
    const uint64_t base = m->rec_base, rs0 = m->rs0, t0 = t * (uint64_t)TILE_BASES;
    const uint32_t r0 = m->r0, nb = m->nb, nrd = m->nrd;
    const uint32_t segstart_raw = lane < META_SEGS ? m->segstart[lane] : 0u, rs16 = lane < META_SEGS ? m->rs16[lane] : 0u;
    const uint64_t slab = t * rec.slab_cap;
    uint32_t nh32 = rec.hash[slab + lane], npos = rec.j[slab + lane]; // (slab_cap >= 64: always inside the pool)  [run-time k only]
    constexpr bool KFIX = KT > 0;
    // SERIAL: lane-serial windows (below) for the common call -- compile-time k, all four k-min-mer arrays, no minimizer triples.  The other calls keep the
    // one-minimizer-per-lane rounds of rounds 3-5: their extra arrays (triples) would be written with a stride too, and alone that costs more than the
    // rolling windows save (minimizers beside the k-min-mers, Regular: 1760 -> 1539 Gbp/s with the lane-serial kernel, bench line of round 6's first runs)
    constexpr bool SERIAL = KFIX && FULL;
    // compile-time k: the lane's three consecutive records, fetched under the assumptions that hold for all but a few tiles of a call -- 129 .. 192 minimizers
    // (three per lane, below), in the tile's own slab -- so that the records travel with the first round trip instead of behind the tile's word
    // (the addresses stay inside the pool whatever the slab holds: the overflow region and the arena's slack lie behind the last slab)
    uint32_t sp_h0 = 0, sp_h1 = 0, sp_h2 = 0, sp_p0 = 0, sp_p1 = 0, sp_p2 = 0; // (scalars: as arrays the lambda below would keep them in scratch)
    if constexpr (SERIAL) {
        if (3u * (uint32_t)lane < dz.spec_n) { // (the lanes that the expected number of records reaches: Desc::spec_n)
            const uint32_t *ph = rec.hash + slab + 3u * (uint32_t)lane, *pp = rec.j + slab + 3u * (uint32_t)lane;
            sp_h0 = ph[0], sp_h1 = ph[1], sp_h2 = ph[2];
            sp_p0 = pp[0], sp_p1 = pp[1], sp_p2 = pp[2];
        }
    }
    if (off_flags) return;
    const uint32_t k = KFIX ? (uint32_t)KT : dz.k, K1 = k - 1;
    const unsigned long long ag0 = ((unsigned long long)dk_lane((uint32_t)(agw >> 32), 0) << 32) | dk_lane((uint32_t)agw, 0);
    const AggF ag = agg_unpack(ag0);
    const uint32_t N = (uint32_t)ag.N;
    const uint64_t G = st.g & M48, Gmn = st.gmn;
    const uint32_t p_in = ag.dep ? (uint32_t)(st.g >> 48) & 63u : 0u;
    if (base != slab && (uint32_t)lane < N) { // (rare: the records are in the overflow region, not where they were fetched from)
        nh32 = rec.hash[base + lane];
        npos = rec.j[base + lane];
    }
    // read segments: hits before segment s, where its read starts, and what turns a hit index into an output offset
    const uint32_t segstart = (uint32_t)lane <= nb ? segstart_raw : 0u;
    const uint32_t nextstart = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, (int)segstart);
    const uint32_t mine = (uint32_t)lane <= nb ? ((uint32_t)lane < nb ? nextstart : N) - segstart : 0u;
    const uint32_t skip = lane == 0 ? (ag.dep ? K1 - p_in : K1) : K1; // minimizers of a segment that end no k-min-mer
    const uint32_t wseg = (uint32_t)lane <= nb && mine > skip ? mine - skip : 0u;
    // windows before each segment / in the whole tile: one or two read starts per tile as a rule -- three readlanes instead of a scan over the wave
    uint32_t Wb, Wt;
    if (nb <= 2u) { // (wave-uniform; wseg is 0 beyond lane nb)
        const uint32_t w0 = dk_lane(wseg, 0), w1 = dk_lane(wseg, 1), w2 = dk_lane(wseg, 2);
        Wt = w0 + w1 + w2;
        Wb = lane == 0 ? 0u : (lane == 1 ? w0 : w0 + w1); // (only lanes 0 .. nb read theirs)
    } else {
        const uint32_t winc = dk_incl_scan(wseg);
        Wb = winc - wseg;
        Wt = dk_lane(winc, 63);
    }
    if ((uint32_t)lane <= nb) // (read-relative positions are 32-bit: only the low word of tile start - read start is ever needed)
        s_seg[w][lane] = make_uint4(segstart, (uint32_t)((int32_t)Wb - (int32_t)segstart - (int32_t)skip), lane ? 0u - rs16 : (uint32_t)(t0 - rs0), 0u);
    // km_off / mn_off of the reads that start in (t0, end of the tile]: the first nb inside, the rest exactly at the end
    if (lane >= 1 && (uint32_t)lane <= nrd) {
        dz.o_km_off[(uint64_t)r0 + lane] = (uint32_t)lane <= nb ? G + Wb : G + Wt;
        if (!FULL && dz.mn_capacity) dz.o_mn_off[(uint64_t)r0 + lane] = (uint32_t)lane <= nb ? Gmn + segstart : Gmn + N;
    }
    if (t == 0) // reads 0 .. r0 start at position 0
        for (uint64_t r = lane; r <= (uint64_t)r0; r += 64) {
            dz.o_km_off[r] = 0;
            if (!FULL && dz.mn_capacity) dz.o_mn_off[r] = 0;
        }
    if (t + 1 == n_tiles && lane == 0) {
        dz.o_km_off[n_reads] = G + Wt;
        if (!FULL && dz.mn_capacity) dz.o_mn_off[n_reads] = Gmn + N;
    }
    if (N == 0) return;
    // ---- round trip 2: the up to k-1 minimizers of the continuing read that lie before the tile (lane q: the (q+1)-th counted
    //      backwards) = the last records of the tiles before this one; a tile that is one stretch of the read hands on to the tile
    //      before it.  The counts of the 63 tiles before this one are in the lanes already.
    // (the tile before holds all of them as a rule -- it has ~140 minimizers, k - 1 are wanted: then no counts are scanned and nothing is searched)
    const uint32_t N1 = (dk_lane((uint32_t)agw, 1) >> 28 | dk_lane((uint32_t)(agw >> 32), 1) << 4) & 0x3FFFu; // minimizers of tile t - 1 (0 before tile 0)
    if (p_in && N1 >= p_in) { // wave-uniform
        if ((uint32_t)lane < p_in) {
            const uint32_t q = (uint32_t)lane, idx = N1 - 1u - q;
            const uint64_t u = t - 1; // (p_in > 0: a read continues into the tile, so there is a tile before it)
            const uint64_t bu = N1 <= rec.slab_cap ? slab - rec.slab_cap : dz.meta[u].rec_base;
            const uint32_t pos = rec.j[bu + idx];
            s_ring[w][K1 - 1 - q] = mix32(rec.hash[bu + idx]);
            s_jcar[w][K1 - 1 - q] = (uint32_t)(u * (uint64_t)TILE_BASES + (pos & 0x3FFFu) - rs0); // the same read: it starts at rs0
        }
        wave_sync();
    } else if (p_in) { // wave-uniform
        const uint32_t Nl = lane >= 1 ? (uint32_t)((agw >> 28) & 0x3FFFu) : 0u; // lane i >= 1: minimizers of tile t - i
        s_cum_w[lane] = dk_incl_scan(Nl);                                        // ... of tiles t-1 .. t-i together
        wave_sync();
        if ((uint32_t)lane < p_in) {
            const uint32_t q = (uint32_t)lane;
            int ti = 1;
#pragma nounroll // (the first tile back as a rule: unrolled 63 times this search was most of the kernel's code)
            while (ti < 64 && s_cum_w[ti] <= q) ti++; // first tile back whose cumulative count exceeds q
            uint64_t u = 0;
            uint32_t idx = 0, Nu = 0;
            bool found = false;
            if (ti < 64 && (uint64_t)ti <= t) {
                u = t - ti;
                Nu = s_cum_w[ti] - s_cum_w[ti - 1];
                idx = Nu - 1 - (q - s_cum_w[ti - 1]);
                found = true;
            } else { // more than 63 tiles back (very sparse minimizers): walk
                uint32_t rem = q - s_cum_w[63];
                u = t >= 63 ? t - 63 : 0;
                while (u > 0) {
                    u--;
                    Nu = (uint32_t)((dz.agg[u] >> 28) & 0x3FFFu);
                    if (rem < Nu) {
                        idx = Nu - 1 - rem;
                        found = true;
                        break;
                    }
                    rem -= Nu;
                }
            }
            if (found) { // (always: p counted these minimizers)
                const uint64_t bu = Nu <= rec.slab_cap ? u * rec.slab_cap : dz.meta[u].rec_base;
                const uint32_t pos = rec.j[bu + idx];
                s_ring[w][K1 - 1 - q] = mix32(rec.hash[bu + idx]);
                s_jcar[w][K1 - 1 - q] = (uint32_t)(u * (uint64_t)TILE_BASES + (pos & 0x3FFFu) - rs0); // the same read: it starts at rs0
            }
        }
        wave_sync();
    }
    uint64_t xacc = 0;
    // one or two read starts per tile as a rule: their hit indices as scalars (a segment that does not exist starts "never"), no LDS walk per round
    const uint32_t sb1 = nb >= 1u ? dk_lane(segstart, 1) : 0xFFFFFFFFu, sb2 = nb >= 2u ? dk_lane(segstart, 2) : 0xFFFFFFFFu;
    // FULL: do all windows of the tile fit the caller's arrays?  (G + Wt <= capacity: then nothing is tested per lane)
    const bool tile_fits = FULL && G + (uint64_t)Wt <= dz.km_capacity;
    unsigned long long *const t_hash = dz.o_hash + G;
    uint32_t *const t_start = dz.o_start + G, *const t_end = dz.o_end + G;
    uint8_t *const t_rev = dz.o_rev + G;
    if constexpr (SERIAL) {
    // ---- compile-time k: LANE-SERIAL windows (round 6).  A lane takes RB CONSECUTIVE minimizers (RB = 1 .. 4 by the tile's count: 64 RB of them per
    //      batch, one batch as a rule), so that only its first window pays the closed form (src/lib.rs:275-288: 2 k rotations of 64 bits) and the next
    //      ones ROLL (src/lib.rs:243-249: F' = rotl(F, 1) ^ rotl(out, k) ^ in, Rv' = rotr(Rv ^ out, 1) ^ rotl(in, k-1): 14 instructions instead of 58).
    //      The k-1 minimizers before a lane's first are the last ones of the lanes before it: they come over DPP (wave_shr:1, chained; lane 0 is
    //      handed the minimizers before the batch through the instruction's "old" operand), not through LDS -- no ring of a round's mixed hashes, no
    //      barriers in the loop, 5.5 -> 3.5 KiB of LDS per block.  (Rounds 3-5: one minimizer per lane and round, every window by the closed form over
    //      ten ring reads: 130 vector + 15 LDS instructions per round of 64, three rounds per Hpc tile.)
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4v lds_u4;
    typedef __attribute__((address_space(3))) uint32_t lds_u1;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[]; // COAL: DK_WAVES staging areas
    const uint32_t stage = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_dyn + (uint32_t)DK_STAGE_BYTES * (uint32_t)w;
    (void)stage;
    static_assert(!COAL || FULL, "the staged stores write all four arrays");
    auto batch = [&](auto rb_c, uint32_t b0) {
        constexpr int RB = decltype(rb_c)::value;
        static_assert(64 * RB <= DK_STAGE_RECS, "a batch fits the staging area");
        int lane_b = lane; // (opaque per batch: what is derived from the lane index -- LDS addresses, offsets -- is otherwise hoisted out of the batch loop and spilled)
        asm volatile("" : "+v"(lane_b));
        constexpr int S = ((int)KT - 1 + RB - 1) / RB; // lanes back that hold the k-1 minimizers before a lane's first
        constexpr int K1c = KT - 1;
        const uint32_t i0 = b0 + (uint32_t)RB * (uint32_t)lane_b; // this lane's first minimizer
        uint32_t h32[RB], pos[RB];
        // the lane's records: RB consecutive entries of each array (reads past the tile's last record stay inside the pool: the arena keeps slack)
        bool pre = false;
        if constexpr (RB == 3) {
            if (b0 == 0u && base == slab && N <= dz.spec_n) { // (wave-uniform) what the first round trip fetched is what this batch wants
                pre = true;
                h32[0] = sp_h0, h32[1] = sp_h1, h32[2] = sp_h2;
                pos[0] = sp_p0, pos[1] = sp_p1, pos[2] = sp_p2;
            }
        }
        if (pre) {
        } else if (i0 < N) {
#pragma unroll
            for (int r = 0; r < RB; r++) {
                h32[r] = rec.hash[base + i0 + r];
                pos[r] = rec.j[base + i0 + r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < RB; r++) h32[r] = pos[r] = 0u;
        }
        // Z[u], u = -(k-1) .. RB-1: mixed hash (src/lib.rs:157-169) of the minimizer u places from the lane's first.  u >= 0: the lane's own; u < 0: they
        // come down the lanes level by level (level s = the values of the lane s places back, slot r -> u = r - RB s), and every level is folded into the
        // first window's closed form  F = XOR rotl(Z[u], -u),  Rv = XOR rotl(Z[u], u + k - 1)  (src/lib.rs:275-288) as it arrives: one level is live at a time
        uint32_t xl[RB], xh[RB], jj[RB];     // the lane's own
        uint32_t cl[RB], ch[RB], cj[RB];     // the level in flight
        uint32_t outl[RB], outh[RB], js[RB]; // what the rolls and the records need later: Z[r - 1 - (k-1)] (r >= 1), j of Z[r - (k-1)]
        uint32_t jend[RB];
        uint32_t before[RB], adj[RB]; // rank of the hit inside its read; hit index -> index among the tile's windows
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const uint64_t xm = mix32(h32[r]);
            xl[r] = cl[r] = (uint32_t)xm;
            xh[r] = ch[r] = (uint32_t)(xm >> 32);
            const uint32_t i = i0 + (uint32_t)r;
            uint32_t c = (uint32_t)(sb1 <= i) + (uint32_t)(sb2 <= i);
#pragma nounroll
            for (uint32_t sq = 3; sq <= nb; sq++) c += (s_seg[w][sq].x <= i); // (more than two read starts in a tile: rare; wave-uniform trip count)
            const uint4 sg = s_seg[w][c];
            before[r] = (c == 0 ? p_in : 0u) + (i - sg.x);
            adj[r] = sg.y;
            jj[r] = cj[r] = sg.z + (pos[r] & 0x3FFFu);
            jend[r] = jj[r] + (pos[r] >> 14);
            outl[r] = outh[r] = js[r] = 0u;
        }
        auto rot_lo = [](uint32_t lo, uint32_t hi, int c) { return c ? __builtin_amdgcn_alignbit(lo, hi, 32 - c) : lo; }; // low word of rotl64(x, c), 0 <= c < 32
        auto rot_hi = [](uint32_t lo, uint32_t hi, int c) { return c ? __builtin_amdgcn_alignbit(hi, lo, 32 - c) : hi; };
        // F / Rv of the lane's first window, terms folded two at a time with the three-input xor (v_bitop3_b32)
        uint32_t fl = xl[0], fh = xh[0], rl = rot_lo(xl[0], xh[0], K1c), rh = rot_hi(xl[0], xh[0], K1c); // u = 0
        uint32_t pfl = 0, pfh = 0, prl = 0, prh = 0; // a term waiting for its partner
        bool have = false;                            // (compile-time once unrolled)
#pragma unroll
        for (int sl = 1; sl <= S; sl++) {
#pragma unroll
            for (int r = RB - 1; r >= 0; r--) {
                const int m = RB * sl - r; // the level's slot r: Z[-m]; lane 0's value: the m-th minimizer before the batch
                unsigned long long cx = 0;
                uint32_t cjv = 0;
                if (m <= K1c) { // (every lane reads the same address: a broadcast)
                    cx = s_ring[w][K1c - m];
                    cjv = s_jcar[w][K1c - m];
                }
                // (a slot is carried down only as far as somebody needs it: Z[-m] for m <= k-1, its j for the records' start positions)
                bool deeper = false;
                for (int s2 = sl; s2 <= S; s2++) deeper = deeper || RB * s2 - r <= K1c;
                if (!deeper) continue;
                cl[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)cx, (int)cl[r], 0x138, 0xf, 0xf, false); // wave_shr:1 -- lane 0 keeps "old"
                ch[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(cx >> 32), (int)ch[r], 0x138, 0xf, 0xf, false);
                bool j_deeper = false;
                for (int s2 = sl; s2 <= S; s2++) j_deeper = j_deeper || (RB * s2 - r <= K1c && RB * s2 - r > K1c - RB);
                if (j_deeper) cj[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)cjv, (int)cj[r], 0x138, 0xf, 0xf, false);
                if (m > K1c) continue;
                const uint32_t tfl = rot_lo(cl[r], ch[r], m), tfh = rot_hi(cl[r], ch[r], m), trl = rot_lo(cl[r], ch[r], K1c - m), trh = rot_hi(cl[r], ch[r], K1c - m);
                if (have) {
                    fl = xor3(fl, pfl, tfl);
                    fh = xor3(fh, pfh, tfh);
                    rl = xor3(rl, prl, trl);
                    rh = xor3(rh, prh, trh);
                    have = false;
                } else {
                    pfl = tfl, pfh = tfh, prl = trl, prh = trh;
                    have = true;
                }
                const int rr = K1c - m; // record rr of the lane starts its window at Z[-m]; record rr + 1 drops it when it rolls
                if (rr >= 0 && rr < RB) js[rr] = cj[r];
                if (rr + 1 >= 1 && rr + 1 < RB) outl[rr + 1] = cl[r], outh[rr + 1] = ch[r];
            }
        }
        if (have) {
            fl ^= pfl;
            fh ^= pfh;
            rl ^= prl;
            rh ^= prh;
        }
        // (records whose window starts at one of the lane's own: k - 1 < RB never happens for k >= 5, RB <= 4)
        static_assert(KT - 1 >= 4, "a window starts before the lane's own minimizers");
#pragma unroll
        for (int r = 0; r < RB; r++) {
            if (r > 0) { // roll: the window loses Z[r-1-(k-1)] and takes Z[r]
                const uint32_t ol = outl[r], oh = outh[r], il = xl[r], ih = xh[r];
                static_assert(KT < 32, "rotations below are by less than a word");
                const uint32_t nfl = xor3(__builtin_amdgcn_alignbit(fl, fh, 31), rot_lo(ol, oh, KT), il); // rotl(F, 1) ^ rotl(out, k) ^ in
                const uint32_t nfh = xor3(__builtin_amdgcn_alignbit(fh, fl, 31), rot_hi(ol, oh, KT), ih);
                const uint32_t tl = rl ^ ol, th = rh ^ oh;                                                  // rotr(Rv ^ out, 1) ^ rotl(in, k-1)
                const uint32_t nrl = __builtin_amdgcn_alignbit(th, tl, 1) ^ rot_lo(il, ih, K1c);
                const uint32_t nrh = __builtin_amdgcn_alignbit(tl, th, 1) ^ rot_hi(il, ih, K1c);
                fl = nfl, fh = nfh, rl = nrl, rh = nrh;
            }
            const uint32_t i = i0 + (uint32_t)r;
            const bool act = i < N;
            // rank of the hit inside its read, capped: the window that ENDS here exists iff k-1 minimizers of the read precede it (src/lib.rs:235)
            const bool win = act && before[r] >= K1;
            const uint64_t f = ((uint64_t)fh << 32) | fl, rvv = ((uint64_t)rh << 32) | rl;
            const uint64_t hmin = f < rvv ? f : rvv;
            const uint32_t jstart = js[r]; // start = j of the window's first minimizer, k-1 hits back
            if constexpr (COAL) {
                const uint32_t e = (uint32_t)RB * (uint32_t)lane_b + (uint32_t)r; // the minimizer's place in the batch
                if (win) xacc ^= hmin;
                *reinterpret_cast<lds_u4 *>(stage + 16u * e) = u4v{(uint32_t)hmin, (uint32_t)(hmin >> 32), jstart, jend[r]};
                *reinterpret_cast<lds_u1 *>(stage + 16u * DK_STAGE_RECS + 4u * e) = win ? ((i + adj[r]) | (rvv < f ? 0x80000000u : 0u)) : 0xFFFFFFFFu;
            } else
            if (win) {
                const uint32_t ot = i + adj[r]; // index among the tile's windows (>= 0 for a window)
                xacc ^= hmin;
                if (tile_fits) { // (wave-uniform) FULL and everything fits: scalar base + 32-bit offset, no tests
                    t_hash[ot] = hmin;
                    t_start[ot] = jstart;
                    t_end[ot] = jend[r];
                    t_rev[ot] = (uint8_t)(rvv < f); // src/lib.rs:250-251
                } else {
                    const uint64_t o = G + (uint64_t)ot;
                    if (o < dz.km_capacity) {
                        if (dz.o_hash) dz.o_hash[o] = hmin;
                        if (dz.o_start) dz.o_start[o] = jstart;
                        if (dz.o_end) dz.o_end[o] = jend[r];
                        if (dz.o_rev) dz.o_rev[o] = (uint8_t)(rvv < f);
                    }
                }
            }
            if (!FULL && dz.mn_capacity && act) { // optional minimizer triples (NtHashHPCIterator::Item, src/nthash_hpc.rs:193)
                const uint64_t g = Gmn + i;
                if (g < dz.mn_capacity) {
                    dz.o_mn_j[g] = jj[r];
                    dz.o_mn_jend[g] = jend[r];
                    dz.o_mn_hash[g] = h32[r];
                }
            }
        }
        if constexpr (COAL) { // every store instruction: 64 consecutive minimizers of the batch -> (all but) consecutive windows
            wave_sync();
#pragma unroll
            for (int jr = 0; jr < RB; jr++) {
                const uint32_t e = 64u * (uint32_t)jr + (uint32_t)lane_b;
                const u4v v = *reinterpret_cast<lds_u4 *>(stage + 16u * e);
                const uint32_t oo = *reinterpret_cast<lds_u1 *>(stage + 16u * DK_STAGE_RECS + 4u * e);
                if (oo != 0xFFFFFFFFu) {
                    const uint32_t ot = oo & 0x7FFFFFFFu;
                    const unsigned long long hm = ((unsigned long long)v.y << 32) | v.x;
                    if (tile_fits) {
                        t_hash[ot] = hm;
                        t_start[ot] = v.z;
                        t_end[ot] = v.w;
                        t_rev[ot] = (uint8_t)(oo >> 31); // src/lib.rs:250-251
                    } else if (G + (uint64_t)ot < dz.km_capacity) {
                        dz.o_hash[G + ot] = hm;
                        dz.o_start[G + ot] = v.z;
                        dz.o_end[G + ot] = v.w;
                        dz.o_rev[G + ot] = (uint8_t)(oo >> 31);
                    }
                }
            }
            wave_sync();
        }
        if (b0 + 64u * (uint32_t)RB < N) { // (rare) another batch follows: its "minimizers before" are the last k-1 of this one
            wave_sync();
            const int back = 63 - lane_b; // lanes 63, 62, ...: the m-th before the next batch, m = RB back + RB - r
            if (back < S) {
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    const int m = RB * back + RB - r;
                    if (m <= K1c) {
                        s_ring[w][K1c - m] = ((unsigned long long)xh[r] << 32) | xl[r];
                        s_jcar[w][K1c - m] = jj[r];
                    }
                }
            }
            wave_sync();
        }
    };
    for (uint32_t b0 = 0; b0 < N;) { // one batch as a rule: 137 +- 11 minimizers per Hpc tile of uniform ACGT, 184 +- 13 per Regular tile (then 192 + the rest)
        const uint32_t rem = N - b0;
        if (rem <= 64u) batch(std::integral_constant<int, 1>{}, b0), b0 += 64u;
        else if (rem <= 128u) batch(std::integral_constant<int, 2>{}, b0), b0 += 128u;
        else batch(std::integral_constant<int, 3>{}, b0), b0 += 192u;
    }
    } else {
    uint32_t jprev = p_in && (uint32_t)lane >= 64u - K1 ? s_jcar[w][lane - (64u - K1)] : 0u; // "the round before the first": the k-1 minimizers before the tile
    for (uint32_t i0 = 0; i0 < N; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool act = i < N;
        const uint32_t h32 = nh32, pos = npos;
        if (i0 + 64 < N && i + 64 < N) { // the next round's records, in flight while this one is worked on
            nh32 = rec.hash[base + i + 64];
            npos = rec.j[base + i + 64];
        }
        uint32_t c = (uint32_t)(sb1 <= i) + (uint32_t)(sb2 <= i);
#pragma nounroll // (more than two read starts in a tile: the walk; unrolled 31 times, the loop was most of the kernel's code)
        for (uint32_t s = 3; s <= nb; s++) c += (s_seg[w][s].x <= i); // wave-uniform trip count, LDS broadcast
        const uint4 sg = s_seg[w][c];
        const uint32_t j = sg.z + (pos & 0x3FFFu);
        const uint32_t jend = j + (pos >> 14);
        const uint64_t xm = mix32(h32); // src/lib.rs:157-169
        wave_sync();
        s_ring[w][K1 + lane] = xm;
        wave_sync();
        // rank of the hit inside its read, capped: the window that ENDS here exists iff k-1 minimizers of the read precede it
        const uint32_t before = (c == 0 ? p_in : 0u) + (i - sg.x);
        const bool win = act && before >= K1;
        // F = XOR rotl(x_m, k-1-m), Rv = XOR rotl(x_m, m) over the window (src/lib.rs:238-249, closed form :275-288); ring[lane + m] = hit i - (k-1) + m
        uint32_t fl = 0, fh = 0, rl = 0, rh = 0;
        uint64_t rvv;
        if constexpr (KFIX) {
            // compile-time k: every x_m is rotated by its own constant (two v_alignbit each for F and for Rv, none for a rotation by 0)
            // and the terms are folded two at a time with the three-input xor of gfx950 (v_bitop3_b32)
            uint32_t xl[KT], xh[KT];
#pragma unroll
            for (int mm = 0; mm < KT; mm++) {
                const uint64_t xw = s_ring[w][lane + mm];
                xl[mm] = (uint32_t)xw;
                xh[mm] = (uint32_t)(xw >> 32);
            }
            auto rot_lo = [](uint32_t lo, uint32_t hi, int c) { return c ? __builtin_amdgcn_alignbit(lo, hi, 32 - c) : lo; }; // low word of rotl64(x, c), c < 32
            auto rot_hi = [](uint32_t lo, uint32_t hi, int c) { return c ? __builtin_amdgcn_alignbit(hi, lo, 32 - c) : hi; };
            uint32_t al[KT], ah[KT], bl[KT], bh[KT];
#pragma unroll
            for (int mm = 0; mm < KT; mm++) {
                al[mm] = rot_lo(xl[mm], xh[mm], KT - 1 - mm); // F: rotl(x_m, k-1-m)
                ah[mm] = rot_hi(xl[mm], xh[mm], KT - 1 - mm);
                bl[mm] = rot_lo(xl[mm], xh[mm], mm);          // Rv: rotl(x_m, m)
                bh[mm] = rot_hi(xl[mm], xh[mm], mm);
            }
            fl = al[0], fh = ah[0], rl = bl[0], rh = bh[0];
#pragma unroll
            for (int mm = 1; mm + 1 < KT; mm += 2) {
                fl = xor3(fl, al[mm], al[mm + 1]);
                fh = xor3(fh, ah[mm], ah[mm + 1]);
                rl = xor3(rl, bl[mm], bl[mm + 1]);
                rh = xor3(rh, bh[mm], bh[mm + 1]);
            }
            if constexpr (KT % 2 == 0) {
                fl ^= al[KT - 1];
                fh ^= ah[KT - 1];
                rl ^= bl[KT - 1];
                rh ^= bh[KT - 1];
            }
            rvv = ((uint64_t)rh << 32) | rl;
        } else {
            // run-time k: Horner with rotations by one: f <- rotl(f, 1) ^ x_m;  r <- rotr(r, 1) ^ x_m, Rv = rotl(r, k-1)
            for (uint32_t mm = 0; mm < k; mm++) {
                const uint64_t xw = s_ring[w][lane + mm];
                const uint32_t xl = (uint32_t)xw, xh = (uint32_t)(xw >> 32);
                const uint32_t nfh = __builtin_amdgcn_alignbit(fh, fl, 31), nfl = __builtin_amdgcn_alignbit(fl, fh, 31);
                const uint32_t nrl = __builtin_amdgcn_alignbit(rh, rl, 1), nrh = __builtin_amdgcn_alignbit(rl, rh, 1);
                fl = nfl ^ xl;
                fh = nfh ^ xh;
                rl = nrl ^ xl;
                rh = nrh ^ xh;
            }
            rvv = rotl64(((uint64_t)rh << 32) | rl, K1);
        }
        const uint64_t f = ((uint64_t)fh << 32) | fl;
        // start = j of the window's first minimizer: k-1 hits back, in this round or the one before it
        const int src = ((lane - (int)K1) & 63) << 2;
        const uint32_t jsame = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)j), jbefore = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)jprev);
        const uint32_t jstart = (uint32_t)lane >= K1 ? jsame : jbefore;
        const uint64_t hmin = f < rvv ? f : rvv;
        if (win) {
            const uint32_t ot = i + sg.y; // index among the tile's windows (>= 0 for a window)
            xacc ^= hmin;
            if (tile_fits) { // (wave-uniform) FULL and everything fits: scalar base + 32-bit offset, no tests
                t_hash[ot] = hmin;
                t_start[ot] = jstart;
                t_end[ot] = jend;
                t_rev[ot] = (uint8_t)(rvv < f); // src/lib.rs:250-251
            } else {
                const uint64_t o = G + (uint64_t)ot;
                if (o < dz.km_capacity) {
                    if (dz.o_hash) dz.o_hash[o] = hmin;
                    if (dz.o_start) dz.o_start[o] = jstart;
                    if (dz.o_end) dz.o_end[o] = jend;
                    if (dz.o_rev) dz.o_rev[o] = (uint8_t)(rvv < f);
                }
            }
        }
        if (!FULL && dz.mn_capacity && act) { // optional minimizer triples (NtHashHPCIterator::Item, src/nthash_hpc.rs:193)
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
    }
    // XOR of the tile's k-min-mer hashes (s2k_counts.xor_hash): folded over the lanes with the DPP steps of dk_incl_scan -- lane 63 ends up with the XOR of all
    // (round 6; the butterfly of 64-bit shuffles it replaces was twelve ds_bpermute and their waits: a tenth of the kernel's issue slots)
    const uint32_t xl_all = dk_incl_xor((uint32_t)xacc), xh_all = dk_incl_xor((uint32_t)(xacc >> 32));
    if (lane == 63 && (xl_all | xh_all))
        atomicXor((unsigned long long *)&dz.xor_shards[t & (XOR_SHARDS - 1)], ((unsigned long long)xh_all << 32) | xl_all);
}

} // namespace

// scan_tmp: level-1 words (4 per 64 tiles) + states (3 each), level-2 words (4 per 4096 tiles) + states (3 each)
size_t desc_scan_tmp_words(uint64_t n_tiles) {
    const uint64_t n1 = (n_tiles + 63) / 64, n2 = (n1 + 63) / 64;
    return 7 * n1 + 7 * n2 + 16;
}

hipError_t launch_desc_scan(uint64_t tile_begin, uint64_t tile_end, Desc dz, unsigned long long *scan_tmp, const Counts *counts, hipStream_t st) {
    if (tile_end <= tile_begin) return hipSuccess;
    const uint64_t n = tile_end - tile_begin;
    const uint64_t n1 = (n + 63) / 64, n2 = (n1 + 63) / 64;
    unsigned long long *w1 = scan_tmp, *s1 = w1 + 4 * n1, *w2 = s1 + 3 * n1, *s2 = w2 + 4 * n2;
    const uint32_t K1 = dz.k - 1;
    const unsigned long long *agg = dz.agg + tile_begin;
    TileState *state = dz.state + tile_begin;
    hipLaunchKernelGGL(desc_scan_reduce<true>, dim3((unsigned)((n1 + 3) / 4)), dim3(256), 0, st, agg, n, K1, w1, counts);
    hipLaunchKernelGGL(desc_scan_reduce<false>, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, st, w1, n1, K1, w2, counts);
    hipLaunchKernelGGL(desc_scan_top, dim3(1), dim3(64), 0, st, w2, n2, K1, s2, tile_begin ? (const TileState *)state : (const TileState *)nullptr, counts);
    hipLaunchKernelGGL(desc_scan_expand<false>, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, st, w1, n1, K1, s2, s1, (TileState *)nullptr, counts);
    hipLaunchKernelGGL(desc_scan_expand<true>, dim3((unsigned)((n1 + 3) / 4)), dim3(256), 0, st, agg, n, K1, s1, (unsigned long long *)nullptr, state,
                       counts);
    return hipGetLastError();
}

hipError_t launch_desc_kminmers(uint64_t tile_begin, uint64_t tile_end, uint64_t n_tiles, uint64_t n_reads, Desc dz, Records rec, Counts *counts,
                                hipStream_t st, bool alone) {
    if (tile_end <= tile_begin) return hipSuccess;
    if (dz.k == 0 || dz.k > (uint32_t)DK_KMAX || rec.slab_cap < 64) return hipErrorInvalidValue;
    const dim3 g((unsigned)((tile_end - tile_begin + DK_WAVES - 1) / DK_WAVES)), b(64 * DK_WAVES);
    unsigned pad = 0; // unused dynamic LDS: bounds how many of these blocks fit a CU beside the persistent minimizer kernel
#ifdef S2K_DEBUG_KNOBS
    if (const char *e = getenv("S2K_DEBUG_KM_LDS")) pad = (unsigned)atoi(e);
#endif
    const bool full = dz.o_hash && dz.o_start && dz.o_end && dz.o_rev && dz.mn_capacity == 0;
    // alone: nothing runs beside this kernel (the caller launched it behind the minimizer kernel): staged, line-filling stores (COAL)
    static const bool no_coal = getenv("S2K_KM_NO_COAL") != nullptr; // (A/B runs)
    const bool coal = alone && full && !no_coal && (dz.k == 10 || dz.k == 5);
#define S2K_KM_GO(KT, F, C) \
    hipLaunchKernelGGL((desc_kminmer_kernel<KT, F, C>), g, b, (C) ? (unsigned)(DK_WAVES * DK_STAGE_BYTES) : pad, st, tile_begin, tile_end, n_tiles, n_reads, dz, rec, counts)
    switch (dz.k) { // the benchmark's k and the reference demo's (src/main.rs:13-48) get an unrolled window loop
    case 10: if (coal) S2K_KM_GO(10, true, true); else if (full) S2K_KM_GO(10, true, false); else S2K_KM_GO(10, false, false); break;
    case 5: if (coal) S2K_KM_GO(5, true, true); else if (full) S2K_KM_GO(5, true, false); else S2K_KM_GO(5, false, false); break;
    default: if (full) S2K_KM_GO(0, true, false); else S2K_KM_GO(0, false, false); break;
    }
#undef S2K_KM_GO
    return hipGetLastError();
}

} // namespace s2k
