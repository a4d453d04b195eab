// s2k_kminmer.hip -- k-min-mer emission (the tail of KminmersIterator::next, src/lib.rs:231-266).
//
// The reference keeps a rolling pair (kminmer_fhash, kminmer_rhash) per read (lib.rs:238-249).  The
// k-min-mer hash is a pure function of k consecutive mixed minimizer hashes (closed form:
// lib.rs:275-288), so here every window is computed independently:
//     F  = XOR_i rotl64(mix(m[c+i]), k-1-i)     Rv = XOR_i rotl64(mix(m[c+i]), i)
//     item = KminmerHash{ min(F,Rv), start = j[c], end = jend[c+k-1], offset = c, rev = Rv < F }   (lib.rs:250-258)
//
// Input: minimizer records grouped in "tiles" (a stream tile of the tiled kernel, or a whole read for
// the serial kernels).  Tiles are in stream order and records inside a tile are in position order, so
// the concatenation of tiles is the global minimizer sequence sorted by (read, position); tile_goff is
// its exclusive prefix.  One wave per tile.  Two kernels: `kminmer_kernel_fast` (k <= 65, the normal case: 128
// records per pass, three global round trips per tile) and `kminmer_kernel` (any k up to 4096; k > 65 walks the
// following tiles per window instead of staging them in LDS).
#include "s2k_dev.h"

namespace s2k {
namespace {

constexpr int KM_WAVES = 4;       // waves per block
constexpr int KM_LDS_K = 65;      // windows up to this k are served from LDS
constexpr int KM_WIN = 64 + KM_LDS_K - 1;

struct Cursor {
    uint64_t tile;
    uint64_t idx;
};

// advance (tile, idx) so that idx < tile_cnt[tile]; returns false past the last tile
__device__ inline bool normalize(Cursor &c, const uint32_t *__restrict__ tile_cnt, uint64_t n_tiles) {
    while (c.tile < n_tiles) {
        uint32_t n = tile_cnt[c.tile];
        if (c.idx < n) return true;
        c.idx -= n;
        c.tile++;
    }
    return false;
}

__global__ __launch_bounds__(64 * KM_WAVES) void kminmer_kernel(
    uint64_t n_tiles, const uint64_t *__restrict__ tile_rec_off, const uint32_t *__restrict__ tile_cnt,
    const uint64_t *__restrict__ tile_goff, Records rec, const uint64_t *__restrict__ mn_off,
    const uint64_t *__restrict__ km_off, uint32_t k, uint64_t km_capacity, uint64_t *__restrict__ o_hash,
    uint32_t *__restrict__ o_start, uint32_t *__restrict__ o_end, uint8_t *__restrict__ o_rev, uint64_t mn_capacity,
    uint32_t *__restrict__ o_mn_j, uint32_t *__restrict__ o_mn_jend, uint32_t *__restrict__ o_mn_hash,
    uint64_t *__restrict__ xor_shards, const Counts *__restrict__ counts) {
    __shared__ uint64_t s_x[KM_WAVES][KM_WIN];
    __shared__ uint32_t s_je[KM_WAVES][KM_WIN];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t t = (uint64_t)blockIdx.x * KM_WAVES + w;
    if (t >= n_tiles) return; // whole wave exits together; no block-level barrier below
    if (counts->pool_overflow || counts->bad_input) return; // record pool too small (the host re-runs) / malformed read table
    const uint32_t cnt = tile_cnt[t];
    if (cnt == 0) return;
    const uint64_t roff = tile_rec_off[t];
    const uint64_t g0 = tile_goff[t];
    const bool lds_path = k <= (uint32_t)KM_LDS_K;
    uint64_t xacc = 0;

    for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t i = base + lane;
        const bool have = i < cnt;
        uint32_t j = 0, je = 0, hv = 0, rid = 0;
        if (have) {
            j = rec.j[roff + i];
            je = rec.jend[roff + i];
            hv = rec.hash[roff + i];
            rid = rec.rid[roff + i];
        }
        const uint64_t g = g0 + i;
        if (lds_path) {
            wave_sync(); // LDS rows are wave-private: order this round's writes after the previous round's reads
            // window entry e <-> record (base + e) counted from this tile's first record; entries past the
            // tile's own records continue in the following tiles (the global sequence is contiguous)
            for (uint32_t e = lane; e < 64 + k - 1; e += 64) {
                uint64_t x = 0;
                uint32_t jj = 0;
                if (e < 64 && have) {
                    x = mix32(hv);
                    jj = je;
                } else {
                    Cursor c{t, (uint64_t)base + e};
                    if (normalize(c, tile_cnt, n_tiles)) {
                        uint64_t a = tile_rec_off[c.tile] + c.idx;
                        x = mix32(rec.hash[a]);
                        jj = rec.jend[a];
                    }
                }
                s_x[w][e] = x;
                s_je[w][e] = jj;
            }
            wave_sync();
        }
        if (have) {
            const uint64_t m0 = mn_off[rid];
            const uint64_t c = g - m0;          // rank of this minimizer inside its read
            const uint64_t M = mn_off[rid + 1] - m0;
            if (o_mn_j && g < mn_capacity) {
                o_mn_j[g] = j;
                o_mn_jend[g] = je;
                o_mn_hash[g] = hv;
            }
            if (c + k <= M) { // window c..c+k-1 lies inside the read: item `c` exists (lib.rs:235)
                uint64_t F = 0, Rv = 0;
                uint32_t end;
                if (lds_path) {
                    for (uint32_t m = 0; m < k; m++) {
                        uint64_t x = s_x[w][lane + m];
                        F ^= rotl64(x, k - 1 - m);
                        Rv ^= rotl64(x, m);
                    }
                    end = s_je[w][lane + k - 1];
                } else {
                    Cursor cur{t, i};
                    end = je;
                    for (uint32_t m = 0; m < k; m++) {
                        normalize(cur, tile_cnt, n_tiles);
                        uint64_t a = tile_rec_off[cur.tile] + cur.idx;
                        uint64_t x = mix32(rec.hash[a]);
                        F ^= rotl64(x, k - 1 - m);
                        Rv ^= rotl64(x, m);
                        if (m == k - 1) end = rec.jend[a];
                        cur.idx++;
                    }
                }
                const uint64_t o = km_off[rid] + c;
                const uint64_t hmin = F < Rv ? F : Rv;
                xacc ^= hmin;
                if (o < km_capacity) {
                    if (o_hash) o_hash[o] = hmin;
                    if (o_start) o_start[o] = j;
                    if (o_end) o_end[o] = end;
                    if (o_rev) o_rev[o] = (uint8_t)(Rv < F);
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) xacc ^= __shfl_down(xacc, o);
    if (lane == 0 && xacc) atomicXor((unsigned long long *)&xor_shards[t & (XOR_SHARDS - 1)], (unsigned long long)xacc);
}


// ---- fast path (k <= 65): 256 records per pass, three global round trips per tile -------------------------------
// The generic kernel above pays a dependent chain of global loads per round of 64 records (records -> read table
// -> following tiles).  Here a wave first fetches the meta data of its tile AND of the four tiles that follow
// (where the k-1 records after the tile's last one almost always live), then all records of the pass plus those
// k-1 "tail" records, then the per-read offsets; everything else runs out of registers and LDS.
constexpr int KF_WAVES = 4;
constexpr int KF_U = 2;              // records per lane per pass
constexpr int KF_PASS = 64 * KF_U;   // records per pass
constexpr int KF_TAIL = 64;  // k - 1 <= 64

__global__ __launch_bounds__(64 * KF_WAVES, 8) void kminmer_kernel_fast(
    uint64_t n_tiles, const uint64_t *__restrict__ tile_rec_off, const uint32_t *__restrict__ tile_cnt,
    const uint64_t *__restrict__ tile_goff, Records rec, const uint64_t *__restrict__ mn_off,
    const uint64_t *__restrict__ km_off, uint32_t k, uint64_t km_capacity, uint64_t *__restrict__ o_hash,
    uint32_t *__restrict__ o_start, uint32_t *__restrict__ o_end, uint8_t *__restrict__ o_rev, uint64_t mn_capacity,
    uint32_t *__restrict__ o_mn_j, uint32_t *__restrict__ o_mn_jend, uint32_t *__restrict__ o_mn_hash,
    uint64_t *__restrict__ xor_shards, const Counts *__restrict__ counts) {
    __shared__ uint64_t s_x[KF_WAVES][KF_PASS + KF_TAIL];
    __shared__ uint32_t s_je[KF_WAVES][KF_PASS + KF_TAIL];
    __shared__ uint64_t s_tx[KF_WAVES][KF_TAIL];
    __shared__ uint32_t s_tje[KF_WAVES][KF_TAIL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t t = (uint64_t)blockIdx.x * KF_WAVES + w;
    if (t >= n_tiles) return;
    // round trip 1: meta data of tiles t .. t+4
    uint32_t mc = 0;
    uint64_t mo = 0;
    if (lane < 5 && t + lane < n_tiles) {
        mc = tile_cnt[t + lane];
        mo = tile_rec_off[t + lane];
    }
    const uint64_t g0 = tile_goff[t];
    const uint32_t pool_bad = counts->pool_overflow | counts->bad_input;
    const uint32_t cnt = __shfl(mc, 0);
    if (cnt == 0 || pool_bad) return;
    const uint64_t roff = ((uint64_t)__shfl((uint32_t)(mo >> 32), 0) << 32) | __shfl((uint32_t)mo, 0);
    const uint32_t K1 = k - 1;
    // where do the K1 records after this tile's last one live?  prefix over tiles t+1 .. t+4
    uint32_t c1 = __shfl(mc, 1), c2 = __shfl(mc, 2), c3 = __shfl(mc, 3), c4 = __shfl(mc, 4);
    const uint32_t p1 = c1, p2 = p1 + c2, p3 = p2 + c3, p4 = p3 + c4;
    uint64_t on[5]; // record offsets of tiles t+1 .. t+4 (shuffles stay outside divergent code)
#pragma unroll
    for (int jt = 1; jt <= 4; jt++) on[jt] = ((uint64_t)__shfl((uint32_t)(mo >> 32), jt) << 32) | __shfl((uint32_t)mo, jt);
    uint64_t tx = 0;
    uint32_t tje = 0;
    if ((uint32_t)lane < K1) {
        const uint32_t e = lane;
        uint64_t a = ~0ull;
        if (e < p4) {
            const int jt = e < p1 ? 1 : e < p2 ? 2 : e < p3 ? 3 : 4;
            const uint32_t before = jt == 1 ? 0 : jt == 2 ? p1 : jt == 3 ? p2 : p3;
            const uint64_t o2 = jt == 1 ? on[1] : jt == 2 ? on[2] : jt == 3 ? on[3] : on[4];
            a = o2 + (e - before);
        } else { // sparse stretch: walk further (rare)
            Cursor c{t + 5, (uint64_t)(e - p4)};
            if (t + 5 < n_tiles && normalize(c, tile_cnt, n_tiles)) a = tile_rec_off[c.tile] + c.idx;
        }
        if (a != ~0ull) { // round trip 2 (together with the records below)
            tx = mix32(rec.hash[a]);
            tje = rec.jend[a];
        }
    }
    uint64_t xacc = 0;
    for (uint32_t base = 0; base < cnt; base += KF_PASS) {
        uint32_t j[KF_U], je[KF_U], hv[KF_U], rid[KF_U];
        bool have[KF_U];
#pragma unroll
        for (int u = 0; u < KF_U; u++) { // round trip 2: up to 256 records
            const uint32_t i = base + 64 * u + lane;
            have[u] = i < cnt;
            j[u] = je[u] = hv[u] = rid[u] = 0;
            if (have[u]) {
                j[u] = rec.j[roff + i];
                je[u] = rec.jend[roff + i];
                hv[u] = rec.hash[roff + i];
                rid[u] = rec.rid[roff + i];
            }
        }
        uint64_t m0[KF_U], m1[KF_U], ko[KF_U];
#pragma unroll
        for (int u = 0; u < KF_U; u++) { // round trip 3: per-read offsets (same address for most lanes)
            m0[u] = m1[u] = ko[u] = 0;
            if (have[u]) {
                m0[u] = mn_off[rid[u]];
                m1[u] = mn_off[rid[u] + 1];
                ko[u] = km_off[rid[u]];
            }
        }
        wave_sync();
        if (base == 0) {
            s_tx[w][lane] = tx;
            s_tje[w][lane] = tje;
        }
#pragma unroll
        for (int u = 0; u < KF_U; u++) {
            s_x[w][64 * u + lane] = mix32(hv[u]);
            s_je[w][64 * u + lane] = je[u];
        }
        wave_sync();
        // entries past this pass's 256: the next records of the tile (next pass) or the tail
        {
            const uint32_t e = KF_PASS + lane, i = base + e;
            uint64_t x = 0;
            uint32_t jj = 0;
            if ((uint32_t)lane < K1) {
                if (i < cnt) {
                    x = mix32(rec.hash[roff + i]);
                    jj = rec.jend[roff + i];
                } else if (i - cnt < K1) {
                    x = s_tx[w][i - cnt];
                    jj = s_tje[w][i - cnt];
                }
            }
            s_x[w][e] = x;
            s_je[w][e] = jj;
        }
        // entries inside the 256 but past the tile's last record: tail
#pragma unroll
        for (int u = 0; u < KF_U; u++) {
            const uint32_t i = base + 64 * u + lane;
            if (!have[u] && i - cnt < K1) {
                s_x[w][64 * u + lane] = s_tx[w][i - cnt];
                s_je[w][64 * u + lane] = s_tje[w][i - cnt];
            }
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < KF_U; u++) {
            if (!have[u]) continue;
            const uint32_t i = base + 64 * u + lane;
            const uint64_t g = g0 + i;
            const uint64_t c = g - m0[u]; // rank of this minimizer inside its read
            const uint64_t M = m1[u] - m0[u];
            if (o_mn_j && g < mn_capacity) {
                o_mn_j[g] = j[u];
                o_mn_jend[g] = je[u];
                o_mn_hash[g] = hv[u];
            }
            if (c + k <= M) { // window c..c+k-1 lies inside the read: item `c` exists (lib.rs:235)
                uint64_t F = 0, Rv = 0;
                const int e0 = 64 * u + lane;
                for (uint32_t m = 0; m < k; m++) {
                    const uint64_t x = s_x[w][e0 + m];
                    F ^= rotl64(x, k - 1 - m);
                    Rv ^= rotl64(x, m);
                }
                const uint32_t end = s_je[w][e0 + k - 1];
                const uint64_t o = ko[u] + c;
                const uint64_t hmin = F < Rv ? F : Rv;
                xacc ^= hmin;
                if (o < km_capacity) {
                    if (o_hash) o_hash[o] = hmin;
                    if (o_start) o_start[o] = j[u];
                    if (o_end) o_end[o] = end;
                    if (o_rev) o_rev[o] = (uint8_t)(Rv < F);
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) xacc ^= __shfl_down(xacc, o);
    if (lane == 0 && xacc) atomicXor((unsigned long long *)&xor_shards[t & (XOR_SHARDS - 1)], (unsigned long long)xacc);
}

} // namespace

hipError_t launch_kminmers(uint64_t n_tiles, const uint64_t *tile_rec_off, const uint32_t *tile_cnt,
                           const uint64_t *tile_goff, Records rec, const uint64_t *mn_off, const uint64_t *km_off,
                           uint32_t k, uint64_t km_capacity, uint64_t *o_hash, uint32_t *o_start, uint32_t *o_end,
                           uint8_t *o_rev, uint64_t mn_capacity, uint32_t *o_mn_j, uint32_t *o_mn_jend,
                           uint32_t *o_mn_hash, uint64_t *xor_shards, const Counts *counts, hipStream_t st) {
    if (n_tiles == 0) return hipSuccess;
    if (k <= (uint32_t)KF_TAIL + 1) {
        dim3 g((unsigned)((n_tiles + KF_WAVES - 1) / KF_WAVES)), b(64 * KF_WAVES);
        hipLaunchKernelGGL(kminmer_kernel_fast, g, b, 0, st, n_tiles, tile_rec_off, tile_cnt, tile_goff, rec, mn_off, km_off, k,
                           km_capacity, o_hash, o_start, o_end, o_rev, mn_capacity, o_mn_j, o_mn_jend, o_mn_hash, xor_shards, counts);
        return hipGetLastError();
    }
    dim3 g((unsigned)((n_tiles + KM_WAVES - 1) / KM_WAVES)), b(64 * KM_WAVES);
    hipLaunchKernelGGL(kminmer_kernel, g, b, 0, st, n_tiles, tile_rec_off, tile_cnt, tile_goff, rec, mn_off, km_off, k,
                       km_capacity, o_hash, o_start, o_end, o_rev, mn_capacity, o_mn_j, o_mn_jend, o_mn_hash, xor_shards, counts);
    return hipGetLastError();
}

} // namespace s2k
