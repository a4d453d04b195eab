// s2k_tile.hip -- the tiled minimizer kernel: fused homopolymer compression + 32-bit canonical ntHash1
// + density threshold + original-space back-map, for gfx950.
//
// What it replaces: the per-base loop of the reference's scalar iterators
// (NtHashHPCIterator::next, src/nthash_hpc.rs:241-278; the Regular arm, src/lib.rs:215-230), which
// is ~100 % of the reference's time.  It is NOT a translation of either the
// scalar loop (loop-carried over the whole read) or the AVX-512 code (16-lane Hillis-Steele scan):
//
//  * The batch of reads is one byte stream; the stream is cut into fixed tiles of 64 x 144 = 9216
//    bases, one wave per tile, independent of read lengths.  l-mers that straddle a read boundary
//    are simply discarded when a hit is validated, so ragged reads cost nothing in the hot loop.
//  * A tile is staged in LDS with coalesced 16 B/lane loads.  Each lane then owns 144 consecutive
//    l-mer start positions and *rolls* the hash privately (fh' = rotl(fh,1) ^ OUT[s[p]] ^ IN[s[p+l]]),
//    reading its bytes from LDS in 16 B pieces and its seeds from two 256-entry LDS tables that have
//    the rotations pre-applied (one ds_read_b64 each).  The lane stride of 16*odd bytes makes the
//    piece reads bank-conflict free.  A lane pays an l-base warm-up instead of a cross-lane scan.
//  * Minimizers are rare (~2 % of positions), but a per-position branch is taken by almost every wave
//    (64 lanes x 2 %).  So the hot loop is branch-free: it records a hit bit per position and keeps
//    the hash of the last hit of each 8-position piece.  After the loop the wave turns the bitmasks
//    into an ordered, dense list (prefix sums + select-nth-set-bit), validates every hit against the
//    read table, back-maps positions and writes records; the rare second hit of a piece has its hash
//    re-derived cooperatively by the whole wave.
//  * Hpc mode first compacts the tile's run heads in place in LDS (SWAR byte compares, per-lane
//    popcounts, one wave scan, overwrite-style byte stores), appends the l run heads that follow the
//    tile (a loop, so arbitrarily long homopolymers are fine), and then runs the same hash loop over
//    the compacted bytes.  Raw positions are recovered for hits only, from the per-lane flag masks.
//    Read starts are forced run heads: they are marked with bit 7 of the staged byte, which is why
//    Hpc tiles require 7-bit input (a byte >= 0x80 raises `non_ascii` and the host re-runs the call on
//    the exact serial kernels).
#include "s2k_dev.h"

namespace s2k {
namespace {

constexpr int TW = 2;                                  // waves per block
constexpr int HS_OFF = 16;                             // data starts here; byte HS_OFF-1 absorbs "slot -1" stores
constexpr int BUF_BYTES = HS_OFF + TILE_BASES + 128;   // tile + halo / window slack
constexpr int NPC = TILE_T / 8;                        // 18 capture pieces per lane
constexpr int MAX_L_TILED = 64;

struct alignas(16) WaveLds {
    uint8_t buf[BUF_BYTES];
    uint32_t caps[NPC][64];  // hash of the last hit of each 8-position piece
    uint32_t hm[64][5];      // raw hit masks, bit x = lane-local hash position x (stored bytewise)
    uint32_t hoff[64];       // exclusive prefix of per-lane hit counts
    uint32_t fm[64][5];      // Hpc: run-head flags of the lane's 144 raw bytes, 32-byte groups, bit 8b+d <-> byte 4d+b
    uint32_t hbase[64];      // Hpc: exclusive prefix of per-lane run-head counts
    uint32_t halo_pos[64];   // Hpc: tile-relative raw offsets of the run heads that follow the tile
};
struct BlockLds {
    uint2 t_in[256];   // {h[c], rotl(rc[c], l-1)}
    uint2 t_out[256];  // {rotl(h[c], l), rotr(rc[c], 1)}
    WaveLds w[TW];
};
static_assert(sizeof(BlockLds) <= 64 * 1024, "static LDS limit");

__device__ inline uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ inline uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
__device__ inline uint32_t bcast(uint32_t v, int src) { return __shfl(v, src); }

// index of the n-th (0-based) set bit of w; n < popc(w)
__device__ inline uint32_t select_nth_32(uint32_t w, uint32_t n) {
    uint32_t r = 0, c;
    c = __popc(w & 0xFFFFu); if (n >= c) { n -= c; r += 16; w >>= 16; }
    c = __popc(w & 0xFFu);   if (n >= c) { n -= c; r += 8;  w >>= 8; }
    c = __popc(w & 0xFu);    if (n >= c) { n -= c; r += 4;  w >>= 4; }
    c = __popc(w & 0x3u);    if (n >= c) { n -= c; r += 2;  w >>= 2; }
    c = w & 1u;              if (n >= c) { r += 1; }
    return r;
}
// n-th set bit over 5 words (160 bits)
__device__ inline uint32_t select_nth_160(const uint32_t *w5, uint32_t n) {
    uint32_t base = 0, word = w5[0];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t c = __popc(word);
        if (n >= c && base == 32u * d) {
            n -= c;
            base += 32;
            word = w5[d + 1];
        }
    }
    return base + select_nth_32(word, n);
}
// bit i of a byte -> bit 4i
__device__ inline uint32_t spread4(uint32_t x) {
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x;
}
// flag-mask group (bit 8b+d <-> byte 4d+b) -> natural order (bit 4d+b)
__device__ inline uint32_t untranspose(uint32_t u) {
    return spread4(u & 0xFFu) | (spread4((u >> 8) & 0xFFu) << 1) | (spread4((u >> 16) & 0xFFu) << 2) |
           (spread4(u >> 24) << 3);
}
// bits of a group mask that belong to bytes at or before (d, b) in byte order
__host__ __device__ constexpr uint32_t at_or_before(int d, int b) {
    uint32_t m = 0;
    for (int bb = 0; bb < 4; bb++)
        for (int dd = 0; dd < 8; dd++)
            if (dd < d || (dd == d && bb <= b)) m |= 1u << (8 * bb + dd);
    return m;
}

__device__ inline uint32_t byte_of(const uint32_t *W, int idx) { return (W[idx >> 2] >> (8 * (idx & 3))) & 0xFFu; }

// ------------------------------------------------------------------------------------------------
// Hash loop, compile-time l, NP 16-byte pieces per lane.  Lane q owns hash positions
// [16*NP*q, 16*NP*(q+1)) of the byte array D (LDS).  Branch-free: per position
//   hv = min(fh, rh); hit = hv <= bound; cap = hit ? hv : cap; bits = bits<<1 | hit; roll.
// ------------------------------------------------------------------------------------------------
template <int L, int NP>
__device__ inline void hash_loop_static(const uint8_t *D, const uint2 *__restrict__ t_in,
                                        const uint2 *__restrict__ t_out, uint32_t bound, int lane, WaveLds &S) {
    constexpr int NWP = (L + 16 + 15) / 16; // pieces that must be resident to serve IN bytes of the current piece
    uint32_t W[NWP * 4];
    const uint4 *src = reinterpret_cast<const uint4 *>(D + 16 * NP * lane);
#pragma unroll
    for (int p = 0; p < NWP; p++) {
        uint4 v = src[p];
        W[4 * p] = v.x; W[4 * p + 1] = v.y; W[4 * p + 2] = v.z; W[4 * p + 3] = v.w;
    }
    uint32_t fh = 0, rh = 0;
#pragma unroll
    for (int i = 0; i < L; i++) { // warm-up: first l-mer of the lane (src/nthash_hpc.rs:138-150,158-174)
        uint2 ti = t_in[byte_of(W, i)];
        fh = __builtin_rotateleft32(fh, 1) ^ ti.x;
        rh = __builtin_rotateright32(rh, 1) ^ ti.y;
    }
    uint32_t cap = 0, bits = 0;
    uint8_t *hmb = reinterpret_cast<uint8_t *>(S.hm[lane]);
#pragma unroll
    for (int m = 0; m < NP; m++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t hv = fh < rh ? fh : rh;      // canonical (src/nthash_hpc.rs:276)
            bool hit = hv <= bound;               // src/nthash_hpc.rs:277 / src/lib.rs:228
            cap = hit ? hv : cap;
            bits = (bits << 1) | (hit ? 1u : 0u);
            uint2 to = t_out[byte_of(W, j)];
            uint2 ti = t_in[byte_of(W, L + j)];
            fh = __builtin_rotateleft32(fh, 1) ^ to.x ^ ti.x;  // src/nthash_hpc.rs:245
            rh = __builtin_rotateright32(rh, 1) ^ to.y ^ ti.y; // src/nthash_hpc.rs:247-249
            if ((j & 7) == 7) {
                const int pc = 2 * m + (j >> 3);
                hmb[pc] = (uint8_t)(__builtin_bitreverse32(bits) >> 24);
                S.caps[pc][lane] = cap;
                bits = 0;
            }
        }
        if (m + 1 < NP) { // slide the window by one piece
#pragma unroll
            for (int i = 0; i < (NWP - 1) * 4; i++) W[i] = W[i + 4];
            uint4 v = src[m + NWP];
            W[4 * NWP - 4] = v.x; W[4 * NWP - 3] = v.y; W[4 * NWP - 2] = v.z; W[4 * NWP - 1] = v.w;
        }
    }
}

// Same loop for a run-time l (1..64): bytes are fetched one by one from LDS.  Slower; only l values
// without a static instantiation come here.
__device__ inline void hash_loop_dynamic(const uint8_t *D, const uint2 *__restrict__ t_in,
                                         const uint2 *__restrict__ t_out, uint32_t bound, int lane, WaveLds &S,
                                         uint32_t l, int np) {
    const uint8_t *q = D + 16 * np * lane;
    uint32_t fh = 0, rh = 0;
    for (uint32_t i = 0; i < l; i++) {
        uint2 ti = t_in[q[i]];
        fh = __builtin_rotateleft32(fh, 1) ^ ti.x;
        rh = __builtin_rotateright32(rh, 1) ^ ti.y;
    }
    uint32_t cap = 0, bits = 0;
    uint8_t *hmb = reinterpret_cast<uint8_t *>(S.hm[lane]);
    for (int pos = 0; pos < 16 * np; pos++) {
        uint32_t hv = fh < rh ? fh : rh;
        bool hit = hv <= bound;
        cap = hit ? hv : cap;
        bits = (bits << 1) | (hit ? 1u : 0u);
        uint2 to = t_out[q[pos]];
        uint2 ti = t_in[q[pos + l]];
        fh = __builtin_rotateleft32(fh, 1) ^ to.x ^ ti.x;
        rh = __builtin_rotateright32(rh, 1) ^ to.y ^ ti.y;
        if ((pos & 7) == 7) {
            hmb[pos >> 3] = (uint8_t)(__builtin_bitreverse32(bits) >> 24);
            S.caps[pos >> 3][lane] = cap;
            bits = 0;
        }
    }
}

template <int L>
__device__ inline void hash_stage(const uint8_t *D, const uint2 *t_in, const uint2 *t_out, uint32_t bound, int lane,
                                  WaveLds &S, uint32_t l, int np) {
    if constexpr (L > 0) {
        switch (np) { // wave-uniform
        case 1: hash_loop_static<L, 1>(D, t_in, t_out, bound, lane, S); break;
        case 3: hash_loop_static<L, 3>(D, t_in, t_out, bound, lane, S); break;
        case 5: hash_loop_static<L, 5>(D, t_in, t_out, bound, lane, S); break;
        case 7: hash_loop_static<L, 7>(D, t_in, t_out, bound, lane, S); break;
        default: hash_loop_static<L, 9>(D, t_in, t_out, bound, lane, S); break;
        }
    } else {
        hash_loop_dynamic(D, t_in, t_out, bound, lane, S, l, np);
    }
}

// ------------------------------------------------------------------------------------------------
// Hpc pre-stage: in-place run-head compaction of the staged tile.  Returns R_t (run heads owned by
// the tile) and leaves D[0..R_t) = head bytes, D[R_t..R_t+halo_n) = following heads, S.fm / S.hbase /
// S.halo_pos for the back-map.  `na` accumulates bytes with bit 7 set.
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t hpc_compact(uint8_t *D, WaveLds &S, const uint8_t *__restrict__ bases,
                                       const uint64_t *__restrict__ read_off, uint64_t n_bases, uint64_t t0,
                                       uint32_t tile_len, uint32_t r0, uint32_t r1, uint32_t l, int lane,
                                       uint32_t &na, uint32_t &halo_n_out) {
    // 1. mark read starts strictly inside the tile (forced run heads: every read starts a new run,
    //    src/nthash_hpc.rs:138-150 runs per read)
    for (uint64_t r = (uint64_t)r0 + 1 + lane; r <= r1; r += 64) {
        uint64_t s = read_off[r];
        if (s > t0 && s < t0 + tile_len) {
            uint32_t o = (uint32_t)(s - t0);
            atomicOr(reinterpret_cast<unsigned int *>(D + (o & ~3u)), 0x80u << (8 * (o & 3u)));
        }
    }
    wave_sync();
    // 2. lane chunk -> registers
    uint32_t c[36];
    const uint4 *src = reinterpret_cast<const uint4 *>(D + TILE_T * lane);
#pragma unroll
    for (int p = 0; p < 9; p++) {
        uint4 v = src[p];
        c[4 * p] = v.x; c[4 * p + 1] = v.y; c[4 * p + 2] = v.z; c[4 * p + 3] = v.w;
    }
    uint32_t prevw;
    if (lane == 0) {
        bool forced = (t0 == 0) || (read_off[r0] == t0);
        if (forced) c[0] |= 0x80u;
        prevw = (t0 > 0) ? ((uint32_t)bases[t0 - 1] << 24) : 0u;
    } else {
        prevw = (uint32_t)D[TILE_T * lane - 1] << 24;
    }
    const uint32_t last_raw = bcast(c[35] >> 24, 63); // last raw byte of a full tile
    const int vb = (int)tile_len - TILE_T * lane;      // valid bytes in this lane's chunk (may be <=0 or >=144)
    const bool partial = tile_len < (uint32_t)TILE_BASES;
    // 3. pass 1: SWAR head flags -> transposed group masks + count
    uint32_t fmk[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 36; d++) {
        uint32_t cur = c[d];
        uint32_t prv = (cur << 8) | (prevw >> 24);
        uint32_t x = (cur ^ prv) & 0x7F7F7F7Fu;
        uint32_t t = ((x + 0x7F7F7F7Fu) | cur) & 0x80808080u; // bit7: differs from predecessor, or marked read start
        if (partial) {
            int v = vb - 4 * d;
            uint32_t keep = v >= 4 ? 0xFFFFFFFFu : (v <= 0 ? 0u : ((1u << (8 * v)) - 1u));
            t &= keep;
        }
        fmk[d >> 3] |= t >> (7 - (d & 7));
        prevw = cur;
    }
    uint32_t cnt = __popc(fmk[0]) + __popc(fmk[1]) + __popc(fmk[2]) + __popc(fmk[3]) + __popc(fmk[4]);
    uint32_t incl = wave_incl_scan(cnt, lane);
    uint32_t base = incl - cnt;
    const uint32_t R = bcast(incl, 63);
#pragma unroll
    for (int g = 0; g < 5; g++) S.fm[lane][g] = fmk[g];
    S.hbase[lane] = base;
    // all lanes hold their raw chunk in registers now -> the buffer may be overwritten in place
    wave_sync();
    // 4. pass 2: every byte is stored at slot (#heads at or before it) - 1; bytes of one run carry the
    //    same value, so only the slot matters.  Slot -1 of lane 0 lands on the scratch byte D[-1].
    uint32_t gbase = base - 1;
#pragma unroll
    for (int d = 0; d < 36; d++) {
        const int g = d >> 3, dd = d & 7;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t slot = gbase + __popc(fmk[g] & at_or_before(dd, b));
            if (!partial || 4 * d + b < vb) D[(int)slot] = (uint8_t)(c[d] >> (8 * b));
        }
        if (dd == 7) gbase += __popc(fmk[g]);
    }
    // 5. run heads that follow the tile: up to l of them (hash needs l-1, the end position one more)
    uint32_t halo_n = 0;
    if (!partial) {
        uint64_t q = t0 + TILE_BASES;
        uint32_t pb = last_raw;
        while (halo_n < l && q < n_bases) { // wave-uniform
            uint64_t a = q + 4 * (uint64_t)lane;
            int nval = a >= n_bases ? 0 : (n_bases - a >= 4 ? 4 : (int)(n_bases - a));
            uint32_t wv = 0;
            if (nval == 4) wv = *reinterpret_cast<const uint32_t *>(bases + a);
            else
                for (int b = 0; b < nval; b++) wv |= (uint32_t)bases[a + b] << (8 * b);
            na |= wv & 0x80808080u;
            uint32_t pw = __shfl_up(wv, 1);
            if (lane == 0) pw = pb << 24;
            uint32_t prv = (wv << 8) | (pw >> 24);
            uint32_t x = (wv ^ prv) & 0x7F7F7F7Fu;
            uint32_t t = (x + 0x7F7F7F7Fu) & 0x80808080u;
            t &= nval >= 4 ? 0xFFFFFFFFu : (nval <= 0 ? 0u : ((1u << (8 * nval)) - 1u));
            uint32_t cn = __popc(t);
            uint32_t in2 = wave_incl_scan(cn, lane);
            uint32_t idx = halo_n + in2 - cn;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (t & (0x80u << (8 * b))) {
                    if (idx < l) {
                        D[R + idx] = (uint8_t)(wv >> (8 * b));
                        S.halo_pos[idx] = (uint32_t)(a - t0) + b;
                    }
                    idx++;
                }
            }
            halo_n += bcast(in2, 63);
            pb = bcast(wv >> 24, 63);
            q += 256;
        }
        if (halo_n > l) halo_n = l;
    }
    halo_n_out = halo_n;
    wave_sync();
    return R;
}

// tile-relative raw offset of run head x (x < R: inside the tile, else halo head x-R); false if it does not exist
__device__ inline bool hpc_rawpos(const WaveLds &S, uint32_t x, uint32_t R, uint32_t halo_n, uint32_t &raw) {
    if (x >= R) {
        uint32_t hx = x - R;
        if (hx >= halo_n) return false;
        raw = S.halo_pos[hx];
        return true;
    }
    uint32_t o = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1)
        if (S.hbase[o + step] <= x) o += step;
    uint32_t n = x - S.hbase[o];
    uint32_t g = 0, word = S.fm[o][0];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t c = __popc(word);
        if (n >= c && g == (uint32_t)d) {
            n -= c;
            g++;
            word = S.fm[o][d + 1];
        }
    }
    raw = TILE_T * o + 32 * g + select_nth_32(untranspose(word), n);
    return true;
}

template <int L, bool HPC>
__global__ __launch_bounds__(64 * TW) void tile_minimizer_kernel(
    const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
    uint64_t n_tiles, const uint32_t *__restrict__ tile_read0, Sem sem, Records rec, uint64_t *pool_cursor,
    uint64_t *__restrict__ tile_rec_off, uint32_t *__restrict__ tile_cnt, uint32_t *mn_cnt, Counts *counts) {
    __shared__ BlockLds B;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t l = L > 0 ? (uint32_t)L : sem.l;
    for (int c = threadIdx.x; c < 256; c += 64 * TW) {
        // Hpc tiles carry read-start marks in bit 7 (input is 7-bit there), so the table ignores it
        uint32_t cc = HPC ? (c & 0x7F) : c;
        uint32_t h = seed_h_scalar(cc), r = seed_rc_scalar(cc);
        B.t_in[c] = make_uint2(h, rotl32(r, l - 1));
        B.t_out[c] = make_uint2(rotl32(h, l), rotr32(r, 1));
    }
    __syncthreads(); // the only workgroup barrier; waves are independent from here on
    const uint64_t t = (uint64_t)blockIdx.x * TW + w;
    if (t >= n_tiles) return;
    WaveLds &S = B.w[w];
    uint8_t *D = S.buf + HS_OFF;
    const uint64_t t0 = t * (uint64_t)TILE_BASES;
    const uint64_t rem = n_bases - t0;
    const uint32_t avail = rem > (uint64_t)(TILE_BASES + 128) ? (uint32_t)(TILE_BASES + 128) : (uint32_t)rem;
    const uint32_t tile_len = rem > (uint64_t)TILE_BASES ? (uint32_t)TILE_BASES : (uint32_t)rem;
    const uint32_t r0 = tile_read0[t], r1 = tile_read0[t + 1];

    // ---- stage the tile (+128 B look-ahead) in LDS: 1 KiB per wave-instruction, zero past the end ----
    uint32_t na = 0;
    {
        const uint8_t *g = bases + t0;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint32_t off = 16 * lane + 1024 * r;
            if (r == 9 && lane >= 8) break;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off + 16 <= avail) {
                v = *reinterpret_cast<const uint4 *>(g + off);
            } else if (off < avail) {
                uint32_t tmp[4] = {0, 0, 0, 0};
                for (uint32_t b = 0; off + b < avail && b < 16; b++) tmp[b >> 2] |= (uint32_t)g[off + b] << (8 * (b & 3));
                v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
            if (r < 9) na |= (v.x | v.y | v.z | v.w);
            *reinterpret_cast<uint4 *>(D + off) = v;
        }
        if (lane == 0) S.buf[HS_OFF - 1] = 0;
#pragma unroll
        for (int g2 = 0; g2 < 5; g2++) S.hm[lane][g2] = 0;
    }
    wave_sync();

    uint32_t nh = tile_len; // number of hash positions owned by this tile
    uint32_t halo_n = 0;
    int np = 9;
    if (HPC) {
        nh = hpc_compact(D, S, bases, read_off, n_bases, t0, tile_len, r0, r1, l, lane, na, halo_n);
        if (__any((na & 0x80808080u) != 0)) { // bytes >= 0x80: the exact path is the serial kernel
            if (lane == 0) counts->non_ascii = 1;
        }
        int need = (int)((nh + 1023) >> 10);
        np = need <= 1 ? 1 : need <= 3 ? 3 : need <= 5 ? 5 : need <= 7 ? 7 : 9;
    }
    const uint32_t Tq = 16 * np;
    if (nh == 0 || !sem.enabled) {
        if (lane == 0) {
            tile_cnt[t] = 0;
            tile_rec_off[t] = 0;
        }
        return;
    }

    // ---- the hot loop ------------------------------------------------------------------------------
    hash_stage<L>(D, B.t_in, B.t_out, sem.bound_le, lane, S, l, np);
    wave_sync();

    // ---- dense phase: bitmasks -> ordered hit list -> validated records ------------------------------
    uint32_t cnt = 0;
    {
        int vc = (int)nh - (int)(Tq * lane); // hash positions of this lane that exist
#pragma unroll
        for (int d = 0; d < 5; d++) {
            int v = vc - 32 * d;
            uint32_t keep = v >= 32 ? 0xFFFFFFFFu : (v <= 0 ? 0u : ((1u << v) - 1u));
            uint32_t word = S.hm[lane][d] & keep;
            S.hm[lane][d] = word;
            cnt += __popc(word);
        }
    }
    const uint32_t incl = wave_incl_scan(cnt, lane);
    S.hoff[lane] = incl - cnt;
    const uint32_t N = bcast(incl, 63); // raw hits in this tile
    if (N == 0) {
        if (lane == 0) {
            tile_cnt[t] = 0;
            tile_rec_off[t] = 0;
        }
        return;
    }
    uint64_t base = 0;
    if (lane == 0) base = atomicAdd((unsigned long long *)pool_cursor, (unsigned long long)N);
    base = ((uint64_t)bcast((uint32_t)(base >> 32), 0) << 32) | bcast((uint32_t)base, 0);
    if (base + N > rec.capacity) { // record pool exhausted: the host re-runs with pool_needed
        if (lane == 0) {
            counts->pool_overflow = 1;
            atomicMax((unsigned long long *)&counts->pool_needed, (unsigned long long)(base + N));
            tile_cnt[t] = 0;
            tile_rec_off[t] = 0;
        }
        return;
    }
    wave_sync();

    uint32_t n_valid = 0;
    for (uint32_t k0 = 0; k0 < N; k0 += 64) {
        const uint32_t kk = k0 + lane;
        const bool act = kk < N;
        // owner lane and lane-local bit of hit kk
        uint32_t o = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (S.hoff[o + step] <= kk) o += step;
        uint32_t bit = 0, x = 0, hv = 0;
        bool need_re = false;
        if (act) {
            bit = select_nth_160(S.hm[o], kk - S.hoff[o]);
            x = Tq * o + bit;
            const uint32_t piece = bit >> 3;
            const uint32_t pbyte = reinterpret_cast<const uint8_t *>(S.hm[o])[piece];
            const uint32_t later = pbyte >> ((bit & 7) + 1);
            // the kept hash belongs to the piece's last raw hit; pieces cut by nh may hold a stale one
            need_re = later != 0 || (Tq * o + 8 * piece + 8 > nh);
            hv = S.caps[piece][o];
        }
        // re-derive the hash of hits that were not the last of their piece, one at a time, whole wave
        uint64_t jobs = __ballot(need_re);
        while (jobs) {
            const int z = __builtin_ctzll(jobs);
            jobs &= jobs - 1;
            const uint32_t xz = bcast(x, z);
            uint32_t f = 0, r = 0;
            if ((uint32_t)lane < l) { // closed form: src/nthash_hpc.rs:144,168
                const uint32_t c = D[xz + lane];
                f = rotl32(B.t_in[c].x, l - 1 - lane);
                r = rotl32(rotl32(B.t_out[c].y, 1), lane);
            }
            f = wave_xor(f);
            r = wave_xor(r);
            if (lane == z) hv = f < r ? f : r;
        }
        // positions in the stream, read lookup, validation
        uint64_t p = 0, e = 0; // l-mer start; one past the last position that must belong to the same read
        bool ok = act;
        if (act) {
            if (HPC) {
                uint32_t rp = 0, re = 0;
                hpc_rawpos(S, x, nh, halo_n, rp);
                ok = hpc_rawpos(S, x + l, nh, halo_n, re); // head p+l must exist (src/nthash_hpc.rs:265-267)
                p = t0 + rp;
                e = t0 + re + 1;
            } else {
                p = t0 + x;
                e = p + l;
            }
        }
        uint32_t rid = r0;
        if (act) { // last r in [r0, r1] with read_off[r] <= p
            uint32_t lo = r0, hi = r1;
            while (lo < hi) {
                uint32_t mid = lo + (hi - lo + 1) / 2;
                if (read_off[mid] <= p) lo = mid;
                else hi = mid - 1;
            }
            rid = lo;
        }
        uint32_t j = 0, je = 0;
        if (act) {
            const uint64_t rs = read_off[rid], rend = read_off[rid + 1];
            ok = ok && e <= rend;
            j = (uint32_t)(p - rs);
            je = HPC ? (uint32_t)(e - 2 - rs) : j + l - 1; // src/nthash_hpc.rs:281 / src/lib.rs:226
        }
        const uint64_t vmask = __ballot(ok);
        if (ok) {
            const uint32_t rank = __popcll(vmask & ((1ull << lane) - 1ull));
            const uint64_t slot = base + n_valid + rank;
            rec.j[slot] = j;
            rec.jend[slot] = je;
            rec.hash[slot] = hv;
            rec.rid[slot] = rid;
        }
        n_valid += __popcll(vmask);
        // per-read minimizer counts: one atomic per (wave round, read)
        uint64_t remm = vmask;
        while (remm) {
            const int z = __builtin_ctzll(remm);
            const uint32_t rz = bcast(rid, z);
            const uint64_t same = __ballot(ok && rid == rz);
            if (lane == z) atomicAdd(&mn_cnt[rz], (uint32_t)__popcll(same));
            remm &= ~same;
        }
    }
    if (lane == 0) {
        tile_cnt[t] = n_valid;
        tile_rec_off[t] = base;
    }
}

// tile_read0[t] = last read index r (0 <= r < n_reads) with read_off[r] <= min(t*TILE, n_bases)
__global__ __launch_bounds__(256) void tile_index_kernel(const uint64_t *__restrict__ read_off, uint64_t n_reads,
                                                         uint64_t n_bases, uint64_t n_tiles,
                                                         uint32_t *__restrict__ tile_read0) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint64_t pos = t * (uint64_t)TILE_BASES;
    if (pos > n_bases) pos = n_bases;
    uint64_t lo = 0, hi = n_reads - 1;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo + 1) / 2;
        if (read_off[mid] <= pos) lo = mid;
        else hi = mid - 1;
    }
    tile_read0[t] = (uint32_t)lo;
}

template <int L>
hipError_t launch_tiles_l(bool hpc, dim3 g, dim3 b, hipStream_t st, const uint8_t *bases, const uint64_t *read_off,
                          uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles, const uint32_t *tile_read0, Sem sem,
                          Records rec, uint64_t *pool_cursor, uint64_t *tile_rec_off, uint32_t *tile_cnt,
                          uint32_t *mn_cnt, Counts *counts) {
    if (hpc)
        hipLaunchKernelGGL((tile_minimizer_kernel<L, true>), g, b, 0, st, bases, read_off, n_reads, n_bases, n_tiles,
                           tile_read0, sem, rec, pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts);
    else
        hipLaunchKernelGGL((tile_minimizer_kernel<L, false>), g, b, 0, st, bases, read_off, n_reads, n_bases, n_tiles,
                           tile_read0, sem, rec, pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts);
    return hipGetLastError();
}

} // namespace

hipError_t launch_tile_index(const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles,
                             uint32_t *tile_read0, hipStream_t st) {
    if (n_reads == 0) return hipSuccess;
    uint64_t n = n_tiles + 1;
    hipLaunchKernelGGL(tile_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, read_off, n_reads, n_bases,
                       n_tiles, tile_read0);
    return hipGetLastError();
}

hipError_t launch_tile_minimizers(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                                  uint64_t n_tiles, const uint32_t *tile_read0, Sem sem, Records rec,
                                  uint64_t *pool_cursor, uint64_t *tile_rec_off, uint32_t *tile_cnt, uint32_t *mn_cnt,
                                  Counts *counts, hipStream_t st) {
    if (n_tiles == 0 || n_reads == 0) return hipSuccess;
    if (sem.l > (uint32_t)MAX_L_TILED || sem.simd_seeds) return hipErrorInvalidValue;
    dim3 g((unsigned)((n_tiles + TW - 1) / TW)), b(64 * TW);
    switch (sem.l) {
    case 31:
        return launch_tiles_l<31>(sem.hpc, g, b, st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec,
                                  pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts);
    default:
        return launch_tiles_l<0>(sem.hpc, g, b, st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec,
                                 pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts);
    }
}

} // namespace s2k
