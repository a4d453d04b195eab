// s2k_hpc_seg.hip -- standalone homopolymer compression at stream speed.
// What it replaces: hpc() / encode_rle_simd() of the reference (src/hpc.rs:28-41, :44-147): per read the compressed
// string and the start of every run.  The reference walks a read sequentially (16 bytes per AVX-512 step); here the
// batch is cut into 4096-byte segments regardless of read boundaries.  A segment's first output slot follows from two
// prefixes made by launch_read_run_counts -- runs before the segment start inside its read -- plus hpc_off of that
// read; inside the segment a block scan places every run head, and a running max carries the start of the current
// read so that positions come out read-relative.  Outputs of the whole batch are contiguous in read order, so the
// stores are coalesced.
#include "s2k_dev.h"

namespace s2k {
namespace {

constexpr int HS_THREADS = 256;
constexpr uint32_t HS_SEG = HS_THREADS * 16;

// Per-segment index, one thread per segment (and one entry past the last): seg_read0[s] = last read r with read_off[r] <= s * HS_SEG
// (non-empty by construction), seg_a[s] = where that read starts, seg_g[s] = output slot of the segment's first run head (hpc_off of
// that read + its runs before the segment, from the two prefixes of launch_read_run_counts).  (Round 3 let thread 0 of every block
// do the search and the dependent loads behind it -- five round trips to memory in front of everything the block does, 85 % of the
// pipeline's time in that kernel; now a block starts with three scalar loads.)
__global__ __launch_bounds__(256) void hpc_segment_index_kernel(const uint8_t *__restrict__ s, const uint64_t *__restrict__ read_off, uint64_t n_reads,
                                                                uint64_t n_segs, const uint64_t *__restrict__ hpc_off, const uint64_t *__restrict__ blk_off,
                                                                const uint64_t *__restrict__ read_c0, uint64_t *__restrict__ seg_g,
                                                                uint64_t *__restrict__ seg_a, uint32_t *__restrict__ seg_read0, bool rle) {
    const uint64_t sidx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx > n_segs) return;
    const uint64_t seg = sidx * HS_SEG;
    uint64_t lo = 0, hi = n_reads - 1;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (read_off[mid] <= seg) lo = mid;
        else hi = mid - 1;
    }
    seg_read0[sidx] = (uint32_t)lo;
    if (sidx == n_segs) return; // (the entry past the last segment only bounds the reads that start inside the last one)
    const uint64_t a = read_off[lo];
    uint64_t g = hpc_off[lo];
    if (seg > a) { // runs of that read before the segment
        const bool neq_a = a == 0 || run_head(s[a], s[a - 1], rle);
        g += blk_off[seg / 256] - read_c0[lo] + (neq_a ? 0u : 1u);
    }
    seg_g[sidx] = g;
    seg_a[sidx] = a;
}

// inclusive sum / max scans over the 64 lanes of a wave (log steps of shuffles)
__device__ inline void wave_scan_sum_max(uint32_t &s, uint32_t &m, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t a = (uint32_t)__shfl_up((int)s, d), b = (uint32_t)__shfl_up((int)m, d);
        if (lane >= d) {
            s += a;
            m = m > b ? m : b;
        }
    }
}

__global__ __launch_bounds__(HS_THREADS) void hpc_segment_kernel(
    const uint8_t *__restrict__ s, const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
    const uint64_t *__restrict__ seg_g, const uint64_t *__restrict__ seg_a, const uint32_t *__restrict__ seg_read0, uint8_t *__restrict__ o_hpc,
    uint32_t *__restrict__ o_pos, uint64_t capacity, bool rle) {
    __shared__ uint32_t starts[HS_SEG / 32]; // bit i: a non-empty read starts at seg + i
    __shared__ uint32_t ws[HS_THREADS / 64], wm[HS_THREADS / 64];
    __shared__ uint32_t out_p[HS_SEG]; // (staged as 16-bit offsets and resolved at the write -- 14 KiB, eight blocks per CU instead of seven -- the kernel was 8 % SLOWER:
                                       // the write loop of a segment that holds read starts pays a dozen instructions per run head; profiles/r06_hpc_two_pass_trims.txt)
    __shared__ __attribute__((aligned(16))) uint8_t out_b[HS_SEG + 16];
    const int t = threadIdx.x;
    const uint64_t seg = (uint64_t)blockIdx.x * HS_SEG, seg_end = seg + HS_SEG;
    // block-uniform (scalar loads): the read that contains the first byte of the segment, the last read that starts at or before
    // the end of the segment, the segment's first output slot
    const uint64_t r_s = seg_read0[blockIdx.x], r_e = seg_read0[blockIdx.x + 1];
    const uint64_t g = seg_g[blockIdx.x], a_s = seg_a[blockIdx.x];
    // this thread's 16 bytes and the byte before them: in flight while the read starts are marked
    const uint64_t q0 = seg + 16 * (uint64_t)t;
    uint4 v = make_uint4(0, 0, 0, 0);
    uint32_t prev = 0x100u;
    int nval = 0;
    if (q0 < n_bases) {
        nval = n_bases - q0 >= 16 ? 16 : (int)(n_bases - q0);
        if (q0) prev = s[q0 - 1]; // (taken from the lane before instead -- one byte load per wave -- the kernel was 2 % SLOWER: profiles/r06_hpc_two_pass_trims.txt)
        if (nval == 16) v = *reinterpret_cast<const uint4 *>(s + q0);
    }
    // starts of the non-empty reads inside the segment (read r_s itself when it starts exactly here); none in most segments of long reads
    const uint64_t r_first = r_s + (a_s == seg ? 0 : 1);
    const bool any_starts = r_first <= r_e; // block-uniform
    if (any_starts) {
        if (t < (int)(HS_SEG / 32)) starts[t] = 0;
        __syncthreads();
        for (uint64_t r = r_first + (uint64_t)t; r <= r_e; r += HS_THREADS) {
            const uint64_t a = read_off[r];
            if (a >= seg && a < seg_end && read_off[r + 1] > a) atomicOr(&starts[(a - seg) >> 5], 1u << ((a - seg) & 31));
        }
        __syncthreads();
    }
    uint32_t c[16];
    if (nval == 16) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = j < nval ? s[q0 + j] : 0u;
    }
    const uint32_t sb = any_starts ? (starts[(16 * t) >> 5] >> ((16 * t) & 31)) & 0xFFFFu : 0u;
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j < nval && (prev == 0x100u || run_head(c[j], prev, rle) || ((sb >> j) & 1))) heads |= 1u << j;
        prev = j < nval ? c[j] : prev;
    }
    const uint32_t last_start = sb ? (uint32_t)(16 * t + (31 - __clz(sb)) + 1) : 0u; // segment-relative + 1
    // block-wide exclusive sum of head counts and exclusive max of start positions: a scan inside every wave, then the four wave totals
    uint32_t isum = __popc(heads), imax = last_start;
    const int lane = t & 63, wv = t >> 6;
    wave_scan_sum_max(isum, imax, lane);
    if (lane == 63) {
        ws[wv] = isum;
        wm[wv] = imax;
    }
    __syncthreads();
    uint32_t wsum = 0, wmax = 0, total = 0;
#pragma unroll
    for (int i = 0; i < HS_THREADS / 64; i++) {
        if (i < wv) {
            wsum += ws[i];
            wmax = wmax > wm[i] ? wmax : wm[i];
        }
        total += ws[i];
    }
    // exclusive values of this thread: the inclusive ones of the lane before it (or of the waves before)
    uint32_t esum = (uint32_t)__shfl_up((int)isum, 1), emax = (uint32_t)__shfl_up((int)imax, 1);
    if (lane == 0) esum = 0, emax = 0;
    esum += wsum;
    emax = emax > wmax ? emax : wmax;
    // compact into LDS first, then write whole lines: the outputs of a segment are one contiguous range.  The bytes are staged
    // at the alignment they will have in o_hpc (g mod 4), so that the copy below moves aligned dwords on both sides.
    const uint32_t mis = (uint32_t)(((uintptr_t)o_hpc + g) & 3u);
    uint32_t slot = esum; // segment-relative
    const uint32_t carry = emax;
    uint64_t cur = carry ? seg + carry - 1 : a_s; // start of the read the current byte belongs to
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if ((sb >> j) & 1) cur = q0 + j;
        if ((heads >> j) & 1) {
            out_b[mis + slot] = (uint8_t)c[j];
            out_p[slot] = (uint32_t)(q0 + j - cur);
            slot++;
        }
    }
    __syncthreads();
    const uint64_t room = g < capacity ? capacity - g : 0;
    const uint32_t n_out = total < room ? total : (uint32_t)room; // (a too small output: what fits is written, the caller is told)
    if (o_pos)
        for (uint32_t i = t; i < n_out; i += HS_THREADS) o_pos[g + i] = out_p[i];
    if (o_hpc && n_out) {
        // bytes [g, g + n_out): a head of < 4 bytes up to the first aligned dword, whole dwords, a tail of < 4 bytes
        uint8_t *dst = o_hpc + g - mis; // 4-byte aligned; staged byte k belongs at dst[k], k in [mis, mis + n_out)
        const uint32_t lo_b = mis, hi_b = mis + n_out;
        const uint32_t first_dw = (lo_b + 3) >> 2, end_dw = hi_b >> 2;
        if (first_dw < end_dw) {
            for (uint32_t d = first_dw + t; d < end_dw; d += HS_THREADS)
                reinterpret_cast<uint32_t *>(dst)[d] = reinterpret_cast<const uint32_t *>(out_b)[d];
            if ((uint32_t)t < 4 * first_dw - lo_b) dst[lo_b + t] = out_b[lo_b + t];
            if ((uint32_t)t < hi_b - 4 * end_dw) dst[4 * end_dw + t] = out_b[4 * end_dw + t];
        } else { // fewer than one aligned dword
            if ((uint32_t)t < n_out) dst[lo_b + t] = out_b[lo_b + t];
        }
    }
}

// ---- single pass (round 6): the segments' first output slots by a decoupled look-back instead of a counting pre-pass ------------------------
// The output of the whole batch is contiguous in read order and every non-empty read starts a run, so a segment's first slot is simply the number of
// run heads in all the bytes before it -- a prefix over the segments that the kernel can form itself: every block publishes its head count in one
// 64-bit word ({flag, value}: AGGREGATE as soon as it knows its own count, PREFIX once it knows the counts of everything before it too) and looks
// back over the words of the blocks before it (one wave, 64 words per load) to the nearest PREFIX.  The bases are read ONCE: rounds 3-5 counted the
// runs of every read in a pass of their own (10 % more traffic, 0.44 of 2.6 ms per 2 Gbp) and scanned the counts.  hpc_off (runs before every read)
// is written on the way, by the block in which a read starts.  Words are self-contained (no other memory is published through them): relaxed
// atomics at agent scope.  Polls are bounded: a block that gives up raises *fail and the host runs the two-pass path.
constexpr unsigned long long LB_AGG = 1ull << 62, LB_PRE = 2ull << 62, LB_VAL = (1ull << 62) - 1ull;

// per segment (and one entry past the last): seg_f[s] = first entry r of the read table (0 .. n_reads) with read_off[r] >= s * HS_SEG -- block s owns the
// entries [seg_f[s], seg_f[s + 1]) --, seg_a[s] = start of the read that holds the segment's first byte
__global__ __launch_bounds__(256) void hpc_segment_index2_kernel(const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_segs,
                                                                 uint32_t *__restrict__ seg_f, uint64_t *__restrict__ seg_a) {
    const uint64_t sidx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx > n_segs) return;
    if (sidx == n_segs) {
        seg_f[sidx] = (uint32_t)(n_reads + 1);
        return;
    }
    const uint64_t seg = sidx * HS_SEG;
    uint64_t lo = 0, hi = n_reads + 1; // first r in [0, n_reads] with read_off[r] >= seg (entry n_reads = n_bases > seg: exists)
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (read_off[mid] < seg) lo = mid + 1;
        else hi = mid;
    }
    seg_f[sidx] = (uint32_t)lo;
    uint64_t l2 = lo, h2 = n_reads; // first r in [lo, n_reads] with read_off[r] > seg; the read before it holds byte seg
    while (l2 < h2) {
        const uint64_t mid = l2 + (h2 - l2) / 2;
        if (read_off[mid] <= seg) l2 = mid + 1;
        else h2 = mid;
    }
    seg_a[sidx] = read_off[l2 - 1]; // (l2 >= 1: read_off[0] = 0 <= seg)
}

template <bool WRITE>
__global__ __launch_bounds__(HS_THREADS) void hpc_segment_lb_kernel(
    const uint8_t *__restrict__ s, const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases, const uint32_t *__restrict__ seg_f,
    const uint64_t *__restrict__ seg_a, unsigned long long *status, uint32_t *fail, uint64_t *__restrict__ o_hpc_off, uint8_t *__restrict__ o_hpc,
    uint32_t *__restrict__ o_pos, uint64_t capacity, bool rle) {
    __shared__ uint32_t starts[HS_SEG / 32]; // bit i: a non-empty read starts at seg + i
    __shared__ uint32_t ws[HS_THREADS / 64], wm[HS_THREADS / 64];
    __shared__ uint32_t pre[HS_THREADS];  // run heads of the segment before the thread's 16 bytes
    __shared__ uint16_t hdm[HS_THREADS];  // ... and which of its bytes are run heads
    __shared__ unsigned long long s_g;
    __shared__ uint32_t out_p[WRITE ? HS_SEG : 4];
    __shared__ __attribute__((aligned(16))) uint8_t out_b[WRITE ? HS_SEG + 16 : 16];
    const int t = threadIdx.x;
    const uint64_t b = blockIdx.x, seg = b * HS_SEG, seg_end = seg + HS_SEG;
    const uint64_t f0 = seg_f[b], f1 = seg_f[b + 1], a_s = seg_a[b];
    const uint64_t q0 = seg + 16 * (uint64_t)t;
    uint4 v = make_uint4(0, 0, 0, 0);
    uint32_t prev = 0x100u;
    int nval = 0;
    if (q0 < n_bases) {
        nval = n_bases - q0 >= 16 ? 16 : (int)(n_bases - q0);
        if (q0) prev = s[q0 - 1];
        if (nval == 16) v = *reinterpret_cast<const uint4 *>(s + q0);
    }
    const bool any_entries = f0 < f1; // block-uniform: entries of the read table (read starts, the end of the stream) inside the segment
    if (any_entries) {
        if (t < (int)(HS_SEG / 32)) starts[t] = 0;
        __syncthreads();
        for (uint64_t r = f0 + (uint64_t)t; r < f1 && r < n_reads; r += HS_THREADS) {
            const uint64_t a = read_off[r];
            if (a < seg_end && read_off[r + 1] > a) atomicOr(&starts[(a - seg) >> 5], 1u << ((a - seg) & 31));
        }
        __syncthreads();
    }
    uint32_t c[16];
    if (nval == 16) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = j < nval ? s[q0 + j] : 0u;
    }
    const uint32_t sb = any_entries ? (starts[(16 * t) >> 5] >> ((16 * t) & 31)) & 0xFFFFu : 0u;
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j < nval && (prev == 0x100u || run_head(c[j], prev, rle) || ((sb >> j) & 1))) heads |= 1u << j;
        prev = j < nval ? c[j] : prev;
    }
    const uint32_t last_start = sb ? (uint32_t)(16 * t + (31 - __clz(sb)) + 1) : 0u; // segment-relative + 1
    uint32_t isum = __popc(heads), imax = last_start;
    const int lane = t & 63, wv = t >> 6;
    wave_scan_sum_max(isum, imax, lane);
    if (lane == 63) {
        ws[wv] = isum;
        wm[wv] = imax;
    }
    __syncthreads();
    uint32_t wsum = 0, wmax = 0, total = 0;
#pragma unroll
    for (int i = 0; i < HS_THREADS / 64; i++) {
        if (i < wv) {
            wsum += ws[i];
            wmax = wmax > wm[i] ? wmax : wm[i];
        }
        total += ws[i];
    }
    uint32_t esum = (uint32_t)__shfl_up((int)isum, 1), emax = (uint32_t)__shfl_up((int)imax, 1);
    if (lane == 0) esum = 0, emax = 0;
    esum += wsum;
    emax = emax > wmax ? emax : wmax;
    pre[t] = esum;
    hdm[t] = (uint16_t)heads;
    // ---- the segment's first output slot: publish, look back (wave 0) ---------------------------------------------------------------------
    if (wv == 0) {
        if (lane == 0) __hip_atomic_store(status + b, (b == 0 ? LB_PRE : LB_AGG) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long excl = 0;
        if (b > 0) {
            // 256 words per round trip (four per lane): a block's PREFIX can only be formed once a PREFIX lies inside what it looks at, so the prefixes
            // spread over the grid at (window) blocks per round trip -- with 64 that chain, not the kernel's work, set the time (4.6 ms per 2 Gbp)
            long long base_i = (long long)b - 1;
            uint32_t polls = 0;
            for (bool done = false; !done;) {
                unsigned long long w[4];
#pragma unroll
                for (int jw = 0; jw < 4; jw++) {
                    const long long idx = base_i - 64 * jw - lane;
                    w[jw] = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PRE;
                }
                unsigned long long part = 0;
                bool retry = false;
#pragma unroll
                for (int jw = 0; jw < 4; jw++) {
                    if (done || retry) continue; // (wave-uniform)
                    const uint32_t fl = (uint32_t)(w[jw] >> 62);
                    const uint64_t pm = __ballot(fl == 2u), valid = __ballot(fl != 0u);
                    const int fs = pm ? __builtin_ctzll(pm) : 64; // the nearest block of this window whose PREFIX is known; every block nearer than it must have published its count
                    const uint64_t need = fs >= 63 ? ~0ull : ((2ull << fs) - 1ull);
                    if ((valid & need) != need) {
                        retry = true;
                    } else {
                        part += lane <= fs ? (w[jw] & LB_VAL) : 0ull;
                        if (fs < 64) done = true;
                    }
                }
                if (retry) {
                    if (++polls > (1u << 16) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { // (~100 ms: the block before is not coming)
                        if (lane == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue; // (what the complete windows of this attempt gave is dropped: everything is read again)
                }
                for (int o = 32; o > 0; o >>= 1)
                    part += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(part >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)part, o);
                excl += part;
                base_i -= 256;
            }
            if (lane == 0) __hip_atomic_store(status + b, LB_PRE | ((excl + total) & LB_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_g = excl;
    }
    __syncthreads();
    const uint64_t g = s_g;
    // hpc_off of the table entries inside the segment: the run heads before them (an entry at or behind the end of the data: all of the segment's)
    for (uint64_t r = f0 + (uint64_t)t; r < f1; r += HS_THREADS) {
        const uint64_t rel = read_off[r] - seg;
        const uint32_t ix = (uint32_t)(rel >> 4);
        o_hpc_off[r] = g + (ix >= (uint32_t)HS_THREADS ? total : pre[ix] + __popc((uint32_t)hdm[ix] & ((1u << (rel & 15u)) - 1u)));
    }
    if constexpr (WRITE) {
        const uint32_t mis = (uint32_t)(((uintptr_t)o_hpc + g) & 3u);
        uint32_t slot = esum; // segment-relative
        const uint32_t carry = emax;
        uint64_t cur = carry ? seg + carry - 1 : a_s; // start of the read the current byte belongs to
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if ((sb >> j) & 1) cur = q0 + j;
            if ((heads >> j) & 1) {
                out_b[mis + slot] = (uint8_t)c[j];
                out_p[slot] = (uint32_t)(q0 + j - cur);
                slot++;
            }
        }
        __syncthreads();
        const uint64_t room = g < capacity ? capacity - g : 0;
        const uint32_t n_out = total < room ? total : (uint32_t)room; // (a too small output: what fits is written, the caller is told)
        if (o_pos)
            for (uint32_t i = t; i < n_out; i += HS_THREADS) o_pos[g + i] = out_p[i];
        if (o_hpc && n_out) {
            uint8_t *dst = o_hpc + g - mis; // 4-byte aligned; staged byte k belongs at dst[k], k in [mis, mis + n_out)
            const uint32_t lo_b = mis, hi_b = mis + n_out;
            const uint32_t first_dw = (lo_b + 3) >> 2, end_dw = hi_b >> 2;
            if (first_dw < end_dw) {
                for (uint32_t d = first_dw + t; d < end_dw; d += HS_THREADS)
                    reinterpret_cast<uint32_t *>(dst)[d] = reinterpret_cast<const uint32_t *>(out_b)[d];
                if ((uint32_t)t < 4 * first_dw - lo_b) dst[lo_b + t] = out_b[lo_b + t];
                if ((uint32_t)t < hi_b - 4 * end_dw) dst[4 * end_dw + t] = out_b[4 * end_dw + t];
            } else { // fewer than one aligned dword
                if ((uint32_t)t < n_out) dst[lo_b + t] = out_b[lo_b + t];
            }
        }
    }
}

} // namespace

// single pass: workspace words (32-bit) of the index (seg_f: one per segment + 1; seg_a: 64-bit each) and of the look-back's status words (64-bit each) + the fail flag
size_t hpc_single_pass_words(uint64_t n_bases) {
    const size_t segs = (size_t)((n_bases + HS_SEG - 1) / HS_SEG);
    return (segs + 2) + 2 * (segs + 1) + 2 * (segs + 1) + 8;
}
// d_hpc_off[0 .. n_reads], and (o_hpc / o_pos != nullptr) the compressed bytes and read-relative run starts, in one pass over the bases.
// *fail (device word inside ws) != 0 afterwards: a look-back gave up, the outputs are incomplete -- run the two-pass path.
hipError_t launch_hpc_single_pass(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint32_t *ws /* 8-byte aligned */,
                                  uint64_t *o_hpc_off, uint8_t *o_hpc, uint32_t *o_pos, uint64_t capacity, uint32_t **fail_word, hipStream_t st, bool rle) {
    const uint64_t segs = (n_bases + HS_SEG - 1) / HS_SEG;
    unsigned long long *status = reinterpret_cast<unsigned long long *>(ws);
    uint64_t *seg_a = reinterpret_cast<uint64_t *>(status + segs + 1);
    uint32_t *fail = reinterpret_cast<uint32_t *>(seg_a + segs + 1);
    uint32_t *seg_f = fail + 2;
    *fail_word = fail;
    S2K_HIP_CHECK(hipMemsetAsync(status, 0, (segs + 1) * 8 + (segs + 1) * 8 + 8, st)); // status words (and the fail flag behind seg_a)
    hipLaunchKernelGGL(hpc_segment_index2_kernel, dim3((unsigned)((segs + 1 + 255) / 256)), dim3(256), 0, st, read_off, n_reads, segs, seg_f, seg_a);
    if (o_hpc || o_pos)
        hipLaunchKernelGGL(hpc_segment_lb_kernel<true>, dim3((unsigned)segs), dim3(HS_THREADS), 0, st, bases, read_off, n_reads, n_bases, seg_f, seg_a, status, fail,
                           o_hpc_off, o_hpc, o_pos, capacity, rle);
    else
        hipLaunchKernelGGL(hpc_segment_lb_kernel<false>, dim3((unsigned)segs), dim3(HS_THREADS), 0, st, bases, read_off, n_reads, n_bases, seg_f, seg_a, status, fail,
                           o_hpc_off, o_hpc, o_pos, capacity, rle);
    return hipGetLastError();
}

// workspace of the per-segment index, in 32-bit words: seg_g and seg_a (64-bit each), then seg_read0 with one entry past the end
size_t hpc_segment_index_words(uint64_t n_bases) { return 5 * (size_t)((n_bases + HS_SEG - 1) / HS_SEG) + 8; }

hipError_t launch_hpc_segments(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                               const uint64_t *hpc_off, const uint64_t *blk_off, const uint64_t *read_c0, uint32_t *seg_index /* hpc_segment_index_words, 8-byte aligned */,
                               uint8_t *o_hpc, uint32_t *o_pos, uint64_t capacity, hipStream_t st, bool rle) {
    if (n_reads == 0 || n_bases == 0) return hipSuccess;
    const uint64_t segs = (n_bases + HS_SEG - 1) / HS_SEG;
    uint64_t *seg_g = reinterpret_cast<uint64_t *>(seg_index), *seg_a = seg_g + segs;
    uint32_t *seg_read0 = reinterpret_cast<uint32_t *>(seg_a + segs);
    hipLaunchKernelGGL(hpc_segment_index_kernel, dim3((unsigned)((segs + 1 + 255) / 256)), dim3(256), 0, st, bases, read_off, n_reads, segs, hpc_off, blk_off,
                       read_c0, seg_g, seg_a, seg_read0, rle);
    hipLaunchKernelGGL(hpc_segment_kernel, dim3((unsigned)segs), dim3(HS_THREADS), 0, st, bases, read_off, n_reads, n_bases, seg_g, seg_a, seg_read0, o_hpc,
                       o_pos, capacity, rle);
    return hipGetLastError();
}

} // namespace s2k
