// s2k_api.hip -- the C ABI of include/s2k.h: context, workspace arena, and the two kernel pipelines
// (tiled = fast path, serial = exact fallback).  There is NO CPU path in this library: without a GPU
// s2k_create() fails with S2K_ERR_NO_DEVICE.
#include "../../include/s2k.h"
#include "s2k_dev.h"
#include "s2k_hostcopy.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <memory>
#include <mutex>
#include <sys/mman.h>
#include <vector>

using namespace s2k;

static_assert(offsetof(Counts, path) == offsetof(s2k_counts, path), "Counts must start with s2k_counts");

namespace {

// Host memory of s2k_result arrays.  A fresh malloc of a gigabyte is a gigabyte of page faults in the copy threads that
// drain the D2H ring (measured: 1.3 GB of results took 104 ms to land instead of 35), so blocks given back by
// s2k_result_free are kept by the context that made them and reused by its next call; new blocks are 2 MiB-aligned and
// advised to transparent huge pages.  The pool outlives the context while results still refer to it.
// Round 6: large blocks are PINNED (hipHostRegister) when they are made, so that the drain of a sub-batch is a plain DMA straight into the caller-visible
// arrays -- no staging ring, no copy threads: the 16 CPUs of a 1-GPU job are left to the packing of the next sub-batch's bases.  Pinning costs ~0.1 s per
// GB once; the pool keeps the blocks across calls.  A block that cannot be pinned (the budget below, a refusal by the driver) is drained through the
// staging ring as before.  S2K_PIN_RESULTS=0 turns it off.
struct HostPool {
    struct Blk {
        void *p;
        size_t cap;
        bool pinned;
    };
    std::mutex mu;
    std::vector<Blk> idle;
    size_t pinned_bytes = 0;                          // of all blocks this pool has made and not freed
    static constexpr size_t kKeep = 18;               // two calls' worth of arrays
    static constexpr size_t kPinBudget = 24ull << 30; // pinned host memory this pool may hold
    static bool pin_enabled() {
        static const bool on = !(getenv("S2K_PIN_RESULTS") && atoi(getenv("S2K_PIN_RESULTS")) == 0);
        return on;
    }
    void release(const Blk &b) {
        if (!b.p) return;
        if (b.pinned) {
            (void)hipHostUnregister(b.p);
            std::lock_guard<std::mutex> lk(mu);
            pinned_bytes -= b.cap;
        }
        free(b.p);
    }
    void *get(size_t bytes, size_t *cap, bool *pinned = nullptr) {
        if (pinned) *pinned = false;
        if (bytes == 0) bytes = 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++) // smallest block that fits and is not grossly oversized
                if (idle[i].cap >= bytes && idle[i].cap / 4 <= bytes + (1u << 20) && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
            if (best != idle.size()) {
                Blk b = idle[best];
                idle.erase(idle.begin() + (long)best);
                *cap = b.cap;
                if (pinned) *pinned = b.pinned;
                return b.p;
            }
        }
        void *q = nullptr;
        size_t want = bytes;
        bool pin = false;
        if (bytes >= (4u << 20)) {
            want = (bytes + bytes / 16 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            if (posix_memalign(&q, 2u << 20, want) != 0) q = nullptr;
            if (q) (void)madvise(q, want, MADV_HUGEPAGE);
            if (q && pin_enabled()) {
                bool room;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    room = pinned_bytes + want <= kPinBudget;
                    if (room) pinned_bytes += want;
                }
                if (room) {
                    pin = hipHostRegister(q, want, hipHostRegisterPortable) == hipSuccess;
                    if (!pin) {
                        (void)hipGetLastError();
                        std::lock_guard<std::mutex> lk(mu);
                        pinned_bytes -= want;
                    }
                }
            }
        } else {
            q = malloc(want);
        }
        *cap = q ? want : 0;
        if (pinned) *pinned = q && pin;
        if (!pinned && pin) { // (a caller that does not ask cannot hand the flag back: keep the accounting right)
            (void)hipHostUnregister(q);
            std::lock_guard<std::mutex> lk(mu);
            pinned_bytes -= want;
        }
        return q;
    }
    void put(void *q, size_t cap, bool pinned = false) {
        if (!q) return;
        Blk drop{nullptr, 0, false};
        {
            std::lock_guard<std::mutex> lk(mu);
            idle.push_back(Blk{q, cap, pinned});
            if (idle.size() > kKeep) { // drop the smallest
                size_t m = 0;
                for (size_t i = 1; i < idle.size(); i++)
                    if (idle[i].cap < idle[m].cap) m = i;
                drop = idle[m];
                idle.erase(idle.begin() + (long)m);
            }
        }
        release(drop);
    }
    size_t trim() { // s2k_trim: give every idle block back to the allocator
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(idle);
        }
        size_t bytes = 0;
        for (Blk &b : drop) {
            bytes += b.cap;
            release(b);
        }
        return bytes;
    }
    ~HostPool() {
        for (Blk &b : idle) {
            if (b.pinned) (void)hipHostUnregister(b.p);
            free(b.p);
        }
    }
};

struct HostOwner { // backing store of one s2k_result (no zero fill -- the arrays are overwritten whole)
    std::shared_ptr<HostPool> pool;
    void *p[9] = {};
    size_t cap[9] = {};
    bool pinned[9] = {};
    explicit HostOwner(std::shared_ptr<HostPool> hp) : pool(std::move(hp)) {}
    template <typename T> T *take(int slot, uint64_t n) {
        p[slot] = pool->get((size_t)n * sizeof(T), &cap[slot], &pinned[slot]);
        return (T *)p[slot];
    }
    ~HostOwner() {
        for (int i = 0; i < 9; i++) pool->put(p[i], cap[i], pinned[i]);
    }
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = need + need / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return e;
        bytes = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

constexpr uint32_t MAX_DESC_CHUNKS = 64;
constexpr size_t CTRL_XOR_OFF = (sizeof(Counts) + 255) & ~(size_t)255;
constexpr size_t CTRL_CURSORS_OFF = CTRL_XOR_OFF + XOR_SHARDS * sizeof(uint64_t);

struct Arena {
    char *base;
    size_t off, cap;
    template <class T> T *take(size_t n) {
        size_t o = (off + 255) & ~(size_t)255;
        off = o + n * sizeof(T);
        return base ? reinterpret_cast<T *>(base + o) : nullptr;
    }
};

struct Call { // everything needed to (re-)enqueue one extraction
    const uint8_t *d_bases = nullptr;
    const uint64_t *d_read_off = nullptr;
    uint64_t n_reads = 0, n_bases = 0;
    s2k_params params{};
    s2k_device_out out{};
    Sem sem{};
    uint32_t bound = 0;
    bool serial = false;
    bool desc_run = false; // what the last enqueue() took
    bool full_runs = false; // HpcSimd: the runs of every read are counted in a pre-pass (a look-back from tile to tile gave up: need_runs; S2K_FULL_RUNS)
    bool legacy = false;   // S2K_FLAG_LEGACY_PATH, or the descriptor path met something it does not handle: the (re-)run takes the legacy records
    uint64_t pool_cap = 0; // serial: dense record capacity; tiled: capacity of the overflow region
    uint64_t slab_cap = 0; // tiled: records per tile slab
    bool valid = false;
};

} // namespace

struct s2k_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf ws, in_bases, in_off, outbuf, realign; // realign: aligned copy of a caller's misaligned device stream
    DevBuf in_bases2, in_off2, outbuf2;           // second set: s2k_extract double-buffers its sub-batches
    DevBuf count_tab;                             // s2k_count_device: the hash table (grow-only, like ws)
    hipStream_t s_in = nullptr, s_out = nullptr;  // s2k_extract: H2D of the next / D2H of the previous sub-batch
    hipStream_t s_km = nullptr;                   // descriptor path: scan + k-min-mer kernel of chunk c run here, beside the minimizer kernel of chunk c+1
    std::vector<hipEvent_t> chunk_ev;             // fork / per-chunk / join events of that pipeline (no timing)
    uint32_t rec_per_tile_hint = 0;               // minimizers per tile of the last descriptor-path call (+ margin): Desc::spec_n
    s2k_ctx *chain_prev = nullptr;                // s2k_chain_after: this context's minimizer kernels wait for tiles_done of that one
    hipEvent_t tiles_done = nullptr;              // recorded behind the last minimizer kernel of a call (created on first use)
    bool tiles_done_valid = false;
    uint32_t desc_chunks = 0;                     // chunks of tiles per call: 0 = default (6 for Hpc modes, 8 otherwise; S2K_DESC_CHUNKS overrides; 1 = no overlap)
    uint32_t lookback_giveups = 0;                // HpcSimd calls that were run again because a look-back gave up; two make the pre-pass the default of the context
    bool force_full_runs = false;                 // S2K_FULL_RUNS=1 (A/B, tests): HpcSimd counts the runs of every read in a pre-pass instead of looking back from tile to tile
    bool trace = false;                           // S2K_TRACE: one line on stderr whenever a call is run again (record pool too small, a fall-back to another path)
    uint64_t host_batch = 1ull << 29;             // bases per sub-batch of s2k_extract (s2k_set_host_batch)
    // counters, XOR shards and the tile cursors of up to MAX_DESC_CHUNKS launches are ONE allocation (d_counts is its start): one memset
    // per call instead of three (every launch in front of the first minimizer kernel is ~7 us of an otherwise idle device)
    Counts *d_counts = nullptr;
    Counts *h_counts = nullptr; // pinned
    uint64_t *d_xor = nullptr;
    uint64_t *d_cursors = nullptr;
    std::vector<hipEvent_t> evs; // 6 events per timed call, in call order since timing was enabled
    size_t ev_used = 0;           // sets handed out
    hipEvent_t *ev = nullptr;     // set of the current / last call
    bool timing = false, timed = false;
    Call call;
    bool pending = false;
    s2k_status pending_status = S2K_OK;
    std::string err;
    s2k::HostStager stager; // pinned ring + copy threads of the host-buffer entry points
    s2k::HostStager stager_out; // ... of the result drain, which runs beside the next sub-batch's H2D
    std::shared_ptr<HostPool> host_pool = std::make_shared<HostPool>(); // result arrays of s2k_extract
};

// Every live context, so that s2k_destroy can take the context it destroys out of the chains of the others (s2k_chain_after): a link
// to a destroyed context must never be followed.  One mutex; contexts are created and destroyed rarely.
namespace {
std::mutex g_ctx_mu;
std::vector<s2k_ctx *> g_live_ctx;
} // namespace

// context internals for the other translation units of the library (declared in s2k_hostcopy.h)
namespace s2k {
HostStager &ctx_stager(s2k_ctx *c) { return c->stager; }
hipStream_t ctx_stream(s2k_ctx *c) { return c->stream; }
int ctx_device(s2k_ctx *c) { return c->device; }
void ctx_set_error(s2k_ctx *c, const char *what) { c->err = what; }
void *ctx_count_table(s2k_ctx *c, size_t bytes) { return c->count_tab.ensure(bytes) == hipSuccess ? c->count_tab.p : nullptr; }
} // namespace s2k

namespace {

s2k_status fail(s2k_ctx *c, s2k_status st, const char *what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return st;
}

#define S2K_TRY(expr, what)                                               \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return fail(ctx, S2K_ERR_DEVICE, what, _e); \
    } while (0)

// src/lib.rs:91 -- ((density as FH) * (H::MAX as FH)) as H ; float->int `as` saturates, NaN -> 0
uint32_t hash_bound(double density) {
    double v = density * 4294967295.0;
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
// src/nthash_avx512_32.rs:46-48 -- density re-derived from the bound, then through f32
uint32_t hash_bound_simd(uint32_t b) {
    double density = (double)b / 4294967295.0;
    float f = (float)density * 4294967296.0f;
    if (!(f == f) || f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return 4294967295u;
    return (uint32_t)f;
}

s2k_status resolve_sem(s2k_ctx *ctx, const s2k_params *p, Sem *s, uint32_t *bound_out) {
    if (!p) return fail(ctx, S2K_ERR_INVALID_ARG, "params is NULL");
    if (p->l == 0 || p->l >= 256) return fail(ctx, S2K_ERR_L_RANGE, "l must be in [1,255] (src/nthash_hpc.rs:123-133)");
    if (p->k == 0 || p->k > 4096) return fail(ctx, S2K_ERR_K_RANGE, "k must be in [1,4096]");
    uint32_t b = hash_bound(p->density);
    *bound_out = b;
    memset(s, 0, sizeof *s);
    s->l = p->l;
    s->k = p->k;
    switch (p->mode) {
    case S2K_MODE_REGULAR: // src/lib.rs:215-230
        s->bound_le = b; s->enabled = 1; s->hpc = 0; s->simd_seeds = 0; s->keep_last = 1; s->end_kind = 0; s->tail_quirk = 0;
        break;
    case S2K_MODE_HPC: // src/nthash_hpc.rs:115-283
        s->bound_le = b; s->enabled = 1; s->hpc = 1; s->simd_seeds = 0; s->keep_last = 0; s->end_kind = 1; s->tail_quirk = 0;
        break;
    case S2K_MODE_SIMD:    // src/nthash_avx512_32.rs:32-164
    case S2K_MODE_HPCSIMD: // src/nthash_hpc_simd.rs:35-68
    {
        if (p->l > 31) return fail(ctx, S2K_ERR_L_RANGE, "Simd modes need l <= 31 (src/nthash_avx512_32.rs:33)");
        uint32_t b2 = hash_bound_simd(b);
        s->enabled = b2 != 0; // strict '<' (nthash_avx512_32.rs:55,130)
        s->bound_le = b2 ? b2 - 1 : 0;
        s->hpc = p->mode == S2K_MODE_HPCSIMD;
        s->simd_seeds = 1;
        s->keep_last = 1;
        s->end_kind = s->hpc ? 2 : 0;
        s->tail_quirk = 1;
        break;
    }
    default: return fail(ctx, S2K_ERR_INVALID_ARG, "unknown mode");
    }
    return S2K_OK;
}

// density as the size estimators use it: NaN -> 0 (as hash_bound does, src/lib.rs:91), clamped to [0, 1]
double sane_density(double density) {
    if (!(density == density)) return 0.0;
    return density < 0 ? 0.0 : (density > 1 ? 1.0 : density);
}

// expected number of minimizers, padded: canonical min of two strands passes with prob ~ 1-(1-d)^2
uint64_t pool_estimate(uint64_t n_bases, uint64_t n_units, double density, bool hpc) {
    double d = sane_density(density);
    double p = 1.0 - (1.0 - d) * (1.0 - d);
    double est = (double)n_bases * p * (hpc ? 0.85 : 1.0) * 1.15 + 64.0 * sqrt((double)n_bases * p + 1.0);
    uint64_t cap = (uint64_t)est + 4 * n_units + (uint64_t)TILE_BASES + 4096;
    if (cap > n_bases + 4096) cap = n_bases + 4096;
    return cap;
}

// records per tile slab: mean + 6 sigma of a binomial(TILE_BASES, p) + margin
uint64_t slab_estimate(double density) {
    double d = sane_density(density);
    double mu = (double)TILE_BASES * (1.0 - (1.0 - d) * (1.0 - d));
    uint64_t cap = (uint64_t)(mu + 6.0 * sqrt(mu + 1.0)) + 32;
    cap = (cap + 15) & ~(uint64_t)15;
    if (cap < 64) cap = 64; // (the descriptor path's k-min-mer kernel fetches a tile's first 64 records before it knows how many there are)
    if (cap > (uint64_t)TILE_BASES) cap = TILE_BASES;
    return cap;
}
// overflow region of the tiled path: 2 % of the expected records, at least a few tiles' worth
uint64_t overflow_estimate(uint64_t n_bases, double density) {
    double d = sane_density(density);
    double p = 1.0 - (1.0 - d) * (1.0 - d);
    return (uint64_t)((double)n_bases * p * 0.02) + 4 * (uint64_t)TILE_BASES;
}

bool tiled_supported(const Sem &s) {
    // all four HashModes; HpcSimd first counts the runs of every read (its tail rule depends on that number)
    return s.l <= 64;
}

s2k_status enqueue(s2k_ctx *ctx) {
    Call &c = ctx->call;
    const uint64_t n_reads = c.n_reads, n_bases = c.n_bases;
    const uint64_t n_tiles = c.serial ? n_reads : (n_bases + TILE_BASES - 1) / TILE_BASES;
    const s2k_device_out &o = c.out;
    hipStream_t st = ctx->stream;

    // ---- carve the workspace (first pass sizes, second pass pointers) --------------------------
    Arena a{nullptr, 0, 0};
    uint32_t *mn_cnt = nullptr, *tile_read0 = nullptr, *tile_cnt = nullptr;
    uint64_t *mn_off = nullptr, *tile_rec_off = nullptr, *tile_goff = nullptr, *scan_tmp = nullptr, *pool_cursor = ctx->d_cursors;
    const bool want_runs = !c.serial && c.sem.hpc && c.sem.tail_quirk; // HpcSimd on the tiled kernel
    // descriptor path (default): 8-byte tile-relative records + one word and a segment list per tile, a scan over the tile words,
    // and a k-min-mer kernel that needs no per-read table (s2k_desc.hip).  Legacy path: 16-byte records with the read index,
    // per-read counters and three scans (k > 32, tiles with more than 30 read starts, S2K_FLAG_LEGACY_PATH).
    const bool use_desc = !c.serial && !c.legacy && n_tiles >= 1 && n_reads >= 1 && c.sem.k <= 32;
    c.desc_run = use_desc;
    // The call is cut into chunks of tiles: while the minimizer kernel works on chunk c+1, the scan and the k-min-mer kernel of
    // chunk c run on a second stream (they need ~1/6 of the time and, being bound by latency, fit beside a kernel that leaves
    // a quarter of the issue slots and some LDS free).  A chunk is at least a few tiles per resident wave.
    uint32_t n_chunks = 1;
    bool chained = false; // this context's calls are chained behind another's (s2k_chain_after)
    if (use_desc) {
        // Hpc, how many chunks: more chunks = a smaller exposed tail (the last chunk's k-min-mer kernel, alone on the device) and chunks whose records are
        // still near when their k-min-mer kernel reads them; fewer = fewer chunk boundaries (~40 us each: waves run dry, a launch, a first tile's unhidden load).
        // What pays is a matter of the chunk's SIZE, and differently for a chained context (s2k_chain_after), whose tail is covered by the other context's next
        // call: measured on calls of 1 - 25 Gbp of three read shapes (profiles/r05_chunks_sizes.txt, r05_chunks_workloads.txt, r05_chunks_workloads_b.txt), a
        // chained call does best with chunks of ~4 Gbp (10 Gbp: 2-3 chunks 6.10 ms, six 6.17-6.25; 25 Gbp: six 15.0, three 15.5), an unchained one with
        // ~1.7 Gbp (10 Gbp: six 6.35-6.41, three 6.48-6.51), either way between two and eight.  (Rounds 3-4 used fixed counts: 6, and 2 for chained calls.)
        // Regular family: its minimizer kernel runs 16 waves per CU and leaves the k-min-mer kernel no room beside it (s2k_tile_impl.h: tw()): one
        // launch of each, the k-min-mer stage behind the minimizer kernel on the caller's stream
        { // (the link is another context's to cut -- s2k_destroy(prev) on any thread --: looked at under the registry's lock, here and at the wait below)
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            chained = ctx->chain_prev != nullptr;
        }
        uint32_t by_size = (uint32_t)((double)n_bases / (chained ? 4.0e9 : 1.7e9) + 0.5);
        by_size = by_size < 2u ? 2u : (by_size > 8u ? 8u : by_size);
        n_chunks = ctx->desc_chunks ? ctx->desc_chunks : (c.sem.hpc ? by_size : 1u);
        const uint64_t min_chunk = ctx->desc_chunks ? 64 : 12 * 3072; // tiles: a dozen per resident wave (a forced count -- tests -- only needs 64)
        if ((uint64_t)n_chunks * min_chunk > n_tiles) n_chunks = (uint32_t)(n_tiles / min_chunk);
        if (n_chunks < 1) n_chunks = 1;
        if (n_chunks > MAX_DESC_CHUNKS) n_chunks = MAX_DESC_CHUNKS; // (the cursors of the launches live in the context's control block)
    }
    unsigned long long *d_agg = nullptr, *d_scan = nullptr;
    TileMeta *d_meta = nullptr;
    TileState *d_state = nullptr;
    const uint64_t n_runblk = n_bases / 256 + 1;
    uint32_t *run_blk = nullptr, *read_runs = nullptr, *tile_heads = nullptr;
    uint64_t *run_off = nullptr, *run_tmp = nullptr;
    Records rec{};
    for (int pass = 0; pass < 2; pass++) {
        a.off = 0;
        mn_cnt = a.take<uint32_t>(n_reads + 1);
        mn_off = o.mn_off ? o.mn_off : a.take<uint64_t>(n_reads + 1);
        scan_tmp = a.take<uint64_t>(scan_tmp_bytes(n_reads > n_tiles ? n_reads : n_tiles) / sizeof(uint64_t) + 1);
        if (!c.serial) {
            tile_read0 = a.take<uint32_t>(n_tiles + 1);
            tile_cnt = a.take<uint32_t>(n_tiles + 1);
            tile_rec_off = a.take<uint64_t>(n_tiles + 1);
            tile_goff = a.take<uint64_t>(n_tiles + 1);
        }
        if (want_runs && c.full_runs) {
            run_blk = a.take<uint32_t>(n_runblk + 1);
            run_off = a.take<uint64_t>(n_runblk + 2);
            run_tmp = a.take<uint64_t>(scan_tmp_bytes(n_runblk) / sizeof(uint64_t) + 1);
            read_runs = a.take<uint32_t>(n_reads + 1);
        } else if (want_runs) {
            tile_heads = a.take<uint32_t>(n_tiles + 1);
        }
        if (use_desc) {
            d_agg = a.take<unsigned long long>(n_tiles);
            d_meta = a.take<TileMeta>(n_tiles);
            d_state = a.take<TileState>(n_tiles + 1);
            d_scan = a.take<unsigned long long>(desc_scan_tmp_words(n_tiles));
        }
        const uint64_t rec_total = c.serial ? c.pool_cap : n_tiles * c.slab_cap + c.pool_cap;
        rec.j = a.take<uint32_t>(rec_total + 4); // (+4: the k-min-mer kernel's lanes fetch up to four consecutive records at a time, the last lane past the tile's count)
        rec.hash = a.take<uint32_t>(rec_total + 4);
        rec.jend = use_desc ? nullptr : a.take<uint32_t>(rec_total); // (descriptor path: rec.j holds {offset in the tile, span})
        rec.rid = use_desc ? nullptr : a.take<uint32_t>(rec_total);
        rec.capacity = rec_total;
        rec.slab_cap = c.serial ? 0 : c.slab_cap;
        rec.ovf_base = c.serial ? 0 : n_tiles * c.slab_cap;
        rec.ovf_cursor = (unsigned long long *)pool_cursor; // word 0 of the first block of cursors (all zeroed below)
        if (pass == 0) {
            S2K_TRY(ctx->ws.ensure(a.off + 256), "workspace allocation");
            a.base = (char *)ctx->ws.p;
        }
    }
    if (c.serial) {
        tile_cnt = mn_cnt;
        tile_goff = mn_off;
        tile_rec_off = mn_off;
    }

    const bool tm = ctx->timing;
    ctx->timed = tm;
    if (tm) {
        if (ctx->evs.size() < (ctx->ev_used + 1) * 6) {
            for (int i = 0; i < 6; i++) {
                hipEvent_t e;
                S2K_TRY(hipEventCreate(&e), "event create");
                ctx->evs.push_back(e);
            }
        }
        ctx->ev = &ctx->evs[ctx->ev_used * 6];
        ctx->ev_used++;
        S2K_TRY(hipEventRecord(ctx->ev[0], st), "event");
    }
    S2K_TRY(hipMemsetAsync(ctx->d_counts, 0, CTRL_CURSORS_OFF + (size_t)CURSOR_WORDS * (use_desc ? n_chunks : 1) * sizeof(uint64_t), st),
            "memset counters, xor shards, cursors");
    // the read table is caller memory in HBM: checked on the device, first thing in the stream; the kernels below look at
    // the verdict (or clamp what they read from the table) and the host reports it in finish()
    if (c.serial) {
        S2K_TRY(launch_validate_read_off(c.d_read_off, n_reads, n_bases, &ctx->d_counts->bad_input, st), "read table validation");
        if (tm) S2K_TRY(hipEventRecord(ctx->ev[1], st), "event");
        S2K_TRY(launch_serial_count(c.d_bases, c.d_read_off, n_reads, n_bases, c.sem, mn_cnt, st), "serial count kernel");
        S2K_TRY(launch_scan_u32(mn_cnt, n_reads, mn_off, scan_tmp, 0, st), "scan");
        S2K_TRY(launch_serial_write(c.d_bases, c.d_read_off, n_reads, n_bases, c.sem, mn_off, rec, ctx->d_counts, st), "serial write kernel");
        if (tm) S2K_TRY(hipEventRecord(ctx->ev[2], st), "event");
    } else {
        if (!use_desc) S2K_TRY(hipMemsetAsync(mn_cnt, 0, (n_reads + 1) * sizeof(uint32_t), st), "memset mn_cnt"); // (per-read counts: legacy path only)
        // Descriptor path: every tile writes its word (dense_phase<DESC>); the array still starts as the IDENTITY of the scan -- dep and pass
        // set: "no minimizers, p handed on" (agg_identity, s2k_dev.h) -- so that a tile which left none would drop nothing.  (Zero is NOT the
        // identity: it unpacks to dep = pass = false, q = 0, i.e. "a read ends here", and would cut every window spanning the tile.)  The
        // read-table kernel writes it on its way: validation, tile index and this fill are one launch.
        S2K_TRY(launch_read_table(c.d_read_off, n_reads, n_bases, n_tiles, &ctx->d_counts->bad_input, tile_read0,
                                  use_desc ? (unsigned long long *)d_agg : nullptr, agg_pack(0, 0, 0, 0, true, true), st),
                "read table kernel");
        // s2k_chain_after: this call's minimizer kernels start when those of the other context's call have ended (the launches above
        // -- a memset, the read table -- need not wait)
        // (under the registry's lock: s2k_destroy(prev) cuts the link under the same lock BEFORE it destroys the event, so the wait is either enqueued
        // on a live event or not at all)
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            s2k_ctx *const prev = ctx->chain_prev;
            if (prev && prev->tiles_done_valid) S2K_TRY(hipStreamWaitEvent(st, prev->tiles_done, 0), "stream wait (chain)");
        }
        if (!ctx->tiles_done) S2K_TRY(hipEventCreateWithFlags(&ctx->tiles_done, hipEventDisableTiming), "event create");
        Sem sem = c.sem;
        sem.read_runs = nullptr;
        // HpcSimd: the tail rule needs the run count of the whole read.  Default: the tiles tell each other (Sem::tile_heads, a
        // word per tile published right after its compaction; a tile in which a read ends that began earlier looks back) -- no
        // second pass over the bases.  Fall-back (a look-back gave up, S2K_FULL_RUNS): the runs of every read counted first.
        sem.tile_heads = nullptr;
        if (want_runs) {
            if (c.full_runs) {
                S2K_TRY(launch_read_run_counts(c.d_bases, c.d_read_off, n_reads, n_bases, run_blk, run_off, run_tmp, read_runs, nullptr, st),
                        "run count kernels");
                sem.read_runs = read_runs;
            } else {
                S2K_TRY(hipMemsetAsync(tile_heads, 0, n_tiles * sizeof(uint32_t), st), "memset tile head words");
                sem.tile_heads = tile_heads;
            }
        }
        if (use_desc) {
            Desc dz{};
            dz.agg = d_agg;
            dz.meta = d_meta;
            dz.state = d_state;
            dz.k = c.sem.k;
            // the k-min-mer kernel fetches a tile's records before it knows how many there are: as many as the tiles of this context's last call of the same
            // shape held on average, + 2.5 sigma of a Poisson count (first call: a full 192) -- a tile with more fetches again, one with fewer wasted little
            dz.spec_n = ctx->rec_per_tile_hint ? ctx->rec_per_tile_hint : 192u;
            dz.km_capacity = o.km_capacity;
            dz.mn_capacity = o.mn_capacity;
            dz.o_km_off = (unsigned long long *)o.km_off;
            dz.o_hash = (unsigned long long *)o.hash;
            dz.o_start = o.start;
            dz.o_end = o.end;
            dz.o_rev = o.rev;
            dz.o_mn_off = (unsigned long long *)mn_off; // the caller's, or workspace
            dz.o_mn_j = o.mn_j;
            dz.o_mn_jend = o.mn_jend;
            dz.o_mn_hash = o.mn_hash;
            dz.xor_shards = (unsigned long long *)ctx->d_xor;
            if (tm) S2K_TRY(hipEventRecord(ctx->ev[1], st), "event");
            if (n_chunks == 1) {
                S2K_TRY(launch_tile_minimizers(c.d_bases, c.d_read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec, pool_cursor, nullptr,
                                               nullptr, nullptr, ctx->d_counts, &dz, 0, st),
                        "tiled minimizer kernel");
                S2K_TRY(hipEventRecord(ctx->tiles_done, st), "event");
                ctx->tiles_done_valid = true;
                if (tm) S2K_TRY(hipEventRecord(ctx->ev[2], st), "event");
                if (tm) S2K_TRY(hipEventRecord(ctx->ev[3], st), "event");
                S2K_TRY(launch_desc_scan(0, n_tiles, dz, d_scan, ctx->d_counts, st), "tile word scan");
                // (alone: the Regular family's minimizer kernel -- this context's or a chained one's -- uses the whole LDS of a CU: none of its blocks is ever
                // resident beside a block of this kernel, whatever that one's size)
                S2K_TRY(launch_desc_kminmers(0, n_tiles, n_tiles, n_reads, dz, rec, ctx->d_counts, st, true), "k-min-mer kernel");
                if (tm) S2K_TRY(hipEventRecord(ctx->ev[4], st), "event");
            } else {
                // fork: the second stream starts behind everything that is in the caller's stream so far
                if (!ctx->s_km) S2K_TRY(hipStreamCreateWithFlags(&ctx->s_km, hipStreamNonBlocking), "stream create");
                while (ctx->chunk_ev.size() < (size_t)n_chunks + 2) {
                    hipEvent_t e;
                    S2K_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event create");
                    ctx->chunk_ev.push_back(e);
                }
                hipStream_t s2 = ctx->s_km;
                S2K_TRY(hipEventRecord(ctx->chunk_ev[n_chunks], st), "event");
                S2K_TRY(hipStreamWaitEvent(s2, ctx->chunk_ev[n_chunks], 0), "stream wait");
                if (tm) S2K_TRY(hipEventRecord(ctx->ev[3], s2), "event");
                // (Every minimizer kernel on the caller's stream.  A chunk boundary costs ~40 us -- the waves of a persistent launch run dry
                // within a tile time or two of each other, then the launch and the first tile's unhidden load: 0.22 ms of 6.6 for six chunks,
                // profiles/r04_chunk_boundaries.txt.  Alternating the chunks between two streams, so that the blocks of chunk c+1 take every
                // CU its block of chunk c has left, was measured SLOWER, 7.30 vs 7.09 ms per step: profiles/r04_ab_tile_streams.txt.)
                // (equal chunks: tapering the last ones -- their k-min-mer kernel is the one nothing runs beside -- measured no better)
                for (uint32_t ch = 0; ch < n_chunks; ch++) {
                    // chunk boundaries: equal chunks, or (S2K_DESC_TAIL = fraction of the tiles in the LAST chunk, whose k-min-mer stage nothing
                    // runs beside) the chunks before it share the rest equally
                    auto cut = [&](uint32_t i) -> uint64_t {
                        static const double tail = getenv("S2K_DESC_TAIL") ? atof(getenv("S2K_DESC_TAIL")) : 0.0;
                        if (i >= n_chunks) return n_tiles;
                        if (tail <= 0.0 || tail >= 1.0 || n_chunks < 2) return n_tiles * i / n_chunks;
                        return (uint64_t)((double)n_tiles * (1.0 - tail) * (double)i / (double)(n_chunks - 1));
                    };
                    const uint64_t T0 = cut(ch), T1 = cut(ch + 1);
                    hipStream_t ts = st;
                    S2K_TRY(launch_tile_minimizers(c.d_bases, c.d_read_off, n_reads, n_bases, T1, tile_read0, sem, rec,
                                                   pool_cursor + (size_t)CURSOR_WORDS * ch, nullptr, nullptr, nullptr, ctx->d_counts, &dz, T0, ts),
                            "tiled minimizer kernel");
                    S2K_TRY(hipEventRecord(ctx->chunk_ev[ch], ts), "event");
                    S2K_TRY(hipStreamWaitEvent(s2, ctx->chunk_ev[ch], 0), "stream wait");
#ifdef S2K_DEBUG_KNOBS // S2K_DEBUG_NOKM=1: the chunked minimizer kernels alone (what do the launch boundaries cost?); results are wrong
                    static const bool nokm = getenv("S2K_DEBUG_NOKM") != nullptr;
                    if (nokm) continue;
#endif
                    S2K_TRY(launch_desc_scan(T0, T1, dz, d_scan, ctx->d_counts, s2), "tile word scan");
                    // (the last chunk's k-min-mer kernel has the device to itself unless a chained context's next call follows: staged stores,
                    // +0.3 % for an unchained call, profiles/r06_chunks_prio_sweep.txt)
                    S2K_TRY(launch_desc_kminmers(T0, T1, n_tiles, n_reads, dz, rec, ctx->d_counts, s2, ch + 1 == n_chunks && !chained), "k-min-mer kernel");
                }
                S2K_TRY(hipEventRecord(ctx->tiles_done, st), "event");
                ctx->tiles_done_valid = true;
                if (tm) S2K_TRY(hipEventRecord(ctx->ev[2], st), "event");
                if (tm) S2K_TRY(hipEventRecord(ctx->ev[4], s2), "event");
                // join: what follows in the caller's stream (finalize, the next call) comes after the last k-min-mer kernel
                S2K_TRY(hipEventRecord(ctx->chunk_ev[n_chunks + 1], s2), "event");
                S2K_TRY(hipStreamWaitEvent(st, ctx->chunk_ev[n_chunks + 1], 0), "stream wait");
            }
            S2K_TRY(launch_finalize(ctx->d_counts, ctx->d_xor, (const uint64_t *)&d_state[n_tiles].gmn, (const uint64_t *)&d_state[n_tiles].g,
                                    o.km_capacity, o.mn_capacity, st),
                    "finalize kernel");
            S2K_TRY(hipMemcpyAsync(ctx->h_counts, ctx->d_counts, sizeof(Counts), hipMemcpyDeviceToHost, st), "counts copy");
            if (tm) S2K_TRY(hipEventRecord(ctx->ev[5], st), "event");
            ctx->pending = true;
            return S2K_OK;
        }
        if (tm) S2K_TRY(hipEventRecord(ctx->ev[1], st), "event");
        S2K_TRY(launch_tile_minimizers(c.d_bases, c.d_read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec, pool_cursor,
                                       tile_rec_off, tile_cnt, mn_cnt, ctx->d_counts, nullptr, 0, st),
                "tiled minimizer kernel");
        S2K_TRY(hipEventRecord(ctx->tiles_done, st), "event");
        ctx->tiles_done_valid = true;
        if (tm) S2K_TRY(hipEventRecord(ctx->ev[2], st), "event");
        S2K_TRY(launch_scan_u32(tile_cnt, n_tiles, tile_goff, scan_tmp, 0, st), "scan");
        S2K_TRY(launch_scan_u32(mn_cnt, n_reads, mn_off, scan_tmp, 0, st), "scan");
    }
    S2K_TRY(launch_scan_u32(mn_cnt, n_reads, o.km_off, scan_tmp, c.sem.k, st), "scan");
    if (tm) S2K_TRY(hipEventRecord(ctx->ev[3], st), "event");
    S2K_TRY(launch_kminmers(n_tiles, tile_rec_off, tile_cnt, tile_goff, rec, mn_off, o.km_off, c.sem.k, o.km_capacity, o.hash,
                            o.start, o.end, o.rev, o.mn_capacity, o.mn_capacity ? o.mn_j : nullptr, o.mn_jend, o.mn_hash,
                            ctx->d_xor, ctx->d_counts, st),
            "k-min-mer kernel");
    if (tm) S2K_TRY(hipEventRecord(ctx->ev[4], st), "event");
    S2K_TRY(launch_finalize(ctx->d_counts, ctx->d_xor, mn_off + n_reads, o.km_off + n_reads, o.km_capacity, o.mn_capacity, st),
            "finalize kernel");
    S2K_TRY(hipMemcpyAsync(ctx->h_counts, ctx->d_counts, sizeof(Counts), hipMemcpyDeviceToHost, st), "counts copy");
    if (tm) S2K_TRY(hipEventRecord(ctx->ev[5], st), "event");
    ctx->pending = true;
    return S2K_OK;
}

s2k_status finish(s2k_ctx *ctx, s2k_counts *counts) {
    if (!ctx->pending) {
        if (counts && ctx->call.valid) memcpy(counts, ctx->h_counts, sizeof(s2k_counts));
        return ctx->pending_status;
    }
    for (int attempt = 0; attempt < 4; attempt++) {
        S2K_TRY(hipStreamSynchronize(ctx->stream), "stream synchronize");
        ctx->pending = false;
        Counts *h = ctx->h_counts;
        Call &c = ctx->call;
        if (h->bad_input) { // the device-resident read table failed validate_read_off_kernel: nothing was computed
            ctx->pending_status = (h->bad_input & ~(uint32_t)BAD_LONG) ? S2K_ERR_INVALID_ARG : S2K_ERR_READ_TOO_LONG;
            ctx->err = (h->bad_input & BAD_FIRST)   ? "d_read_off[0] != 0"
                       : (h->bad_input & BAD_ORDER) ? "d_read_off is not non-decreasing"
                       : (h->bad_input & BAD_END)   ? "d_read_off[n_reads] != n_bases"
                                                    : "a read is longer than 2^32-2 bases";
            c.valid = false;
            return ctx->pending_status;
        }
        if (h->need_legacy && c.desc_run) { // the descriptor path met a tile with more read starts than it lists (reads shorter than
                                            // ~300 bases) or a span that does not fit its records: the whole call takes the legacy path
            c.legacy = true;
            if (ctx->trace) fprintf(stderr, "[s2k] re-run: the descriptor path met a tile it does not handle -> legacy path\n");
            s2k_status st = enqueue(ctx);
            if (st != S2K_OK) return st;
            continue;
        }
        if (h->need_runs && !c.full_runs) { // HpcSimd: a tile waited too long for the word of an earlier one (see enqueue())
            c.full_runs = true;
            // (sticky after the second give-up of a context: waves that are not all resident -- another process on the GPU, CU masking --
            // stay that way, and every later call would pay the bounded polls and the second run again)
            if (++ctx->lookback_giveups >= 2) ctx->force_full_runs = true;
            if (ctx->trace) fprintf(stderr, "[s2k] re-run: a look-back for run heads gave up -> runs of every read counted first\n");
            s2k_status st = enqueue(ctx);
            if (st != S2K_OK) return st;
            continue;
        }
        if (h->pool_overflow) { // record pool (serial) / overflow region (tiled) too small: re-run with the exact size
            c.pool_cap = h->pool_needed + h->pool_needed / 64 + (uint64_t)TILE_BASES + 4096;
            if (ctx->trace) fprintf(stderr, "[s2k] re-run: record pool too small -> %llu records\n", (unsigned long long)c.pool_cap);
            s2k_status st = enqueue(ctx);
            if (st != S2K_OK) return st;
            continue;
        }
        if (c.desc_run && c.n_bases >= 64u * (uint64_t)TILE_BASES) { // (see enqueue: Desc::spec_n of the next call)
            const double avg = (double)h->n_minimizers / ((double)c.n_bases / (double)TILE_BASES);
            const double want = avg + 2.5 * sqrt(avg + 1.0) + 3.0; // (a tile in ~100 holds more and fetches again; every lane not fetched saves 24 B per tile)
            ctx->rec_per_tile_hint = want >= 192.0 ? 192u : (uint32_t)want;
        }
        h->n_reads = c.n_reads;
        h->n_bases = c.n_bases;
        h->hash_bound = c.bound;
        h->path = c.serial ? 1u : (c.desc_run ? 0u : 2u);
        if (c.sem.dbg_skip & 32) // KNOBS builds: spread of the waves' finishing times in the tiled kernel
        {
            fprintf(stderr, "[s2k dbg] last wave finished %.1f us after the first; shader clock under the tiled kernel %.0f MHz (waves alive %.3f ms on average)\n",
                    (double)(h->dbg_cycles[0][0] - ~h->dbg_cycles[0][1]) * 0.01,
                    h->dbg_cycles[1][1] ? 100.0 * (double)h->dbg_cycles[1][0] / (double)h->dbg_cycles[1][1] : 0.0,
                    (double)h->dbg_cycles[1][1] * 1e-5 / (256.0 * 12.0));
#ifdef S2K_DEBUG_KNOBS
            if (const char *path = getenv("S2K_DEBUG_WAVE_DUMP")) {
                if (FILE *f = fopen(path, "w")) {
                    for (int i = 0; i < 4096; i++)
                        if (h->dbg_wave[i][0]) fprintf(f, "%d %llu %u %u\n", i, (unsigned long long)(h->dbg_wave[i][0] - ~h->dbg_cycles[0][1]), (unsigned)(h->dbg_wave[i][1] >> 32), (unsigned)h->dbg_wave[i][1]);
                    fclose(f);
                }
            }
#endif
        }
        if (c.sem.dbg_skip & 8) {
            fprintf(stderr, "[s2k dbg] phase cycles (sum over waves):");
            for (int i = 0; i < 16; i++) {
                unsigned long long sum = 0;
                for (int sh = 0; sh < 64; sh++) sum += h->dbg_cycles[sh][i];
                fprintf(stderr, " %llu", sum);
            }
            fprintf(stderr, "\n");
        }
        if (counts) memcpy(counts, h, sizeof(s2k_counts));
        ctx->pending_status = (h->km_overflow || h->mn_overflow) ? S2K_ERR_CAPACITY : S2K_OK;
        if (ctx->pending_status != S2K_OK) ctx->err = "device output capacity too small; see counts";
        return ctx->pending_status;
    }
    return fail(ctx, S2K_ERR_DEVICE, "workspace retry did not converge");
}

} // namespace

// =================================================================================================
extern "C" {

int s2k_abi_version(void) { return S2K_ABI_VERSION; }

double s2k_density_for_bound(uint32_t bound) {
    if (bound == 0xFFFFFFFFu) return 1.0;
    const double d = ((double)bound + 0.5) / 4294967295.0; // d * u32::MAX lands half way between bound and bound + 1: truncates to bound
    return hash_bound(d) == bound ? d : ((double)bound + 0.25) / 4294967295.0;
}

int s2k_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *s2k_strerror(s2k_status st) {
    switch (st) {
    case S2K_OK: return "ok";
    case S2K_ERR_INVALID_ARG: return "invalid argument";
    case S2K_ERR_L_RANGE: return "minimizer length l out of range";
    case S2K_ERR_K_RANGE: return "k-min-mer order k out of range";
    case S2K_ERR_READ_TOO_LONG: return "a read is longer than 2^32-2 bases";
    case S2K_ERR_DEVICE: return "HIP runtime error";
    case S2K_ERR_NOMEM: return "out of memory";
    case S2K_ERR_CAPACITY: return "device output capacity too small";
    case S2K_ERR_NO_DEVICE: return "no GPU visible (this library has no CPU path)";
    case S2K_ERR_NON_ASCII: return "reserved (not returned since ABI 2: every byte value is accepted)";
    }
    return "unknown status";
}

const char *s2k_last_error(const s2k_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

uint32_t s2k_hash_bound(double density) { return hash_bound(density); }

s2k_ctx *s2k_create(int device, s2k_status *status) {
    s2k_status dummy;
    if (!status) status = &dummy;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        *status = S2K_ERR_NO_DEVICE;
        return nullptr;
    }
    if (device < 0 || device >= n) {
        *status = S2K_ERR_INVALID_ARG;
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        *status = S2K_ERR_DEVICE;
        return nullptr;
    }
    s2k_ctx *ctx = new (std::nothrow) s2k_ctx();
    if (!ctx) {
        *status = S2K_ERR_NOMEM;
        return nullptr;
    }
    ctx->device = device;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    ctx->own_stream = ok;
    ok = ok && hipMalloc((void **)&ctx->d_counts, CTRL_CURSORS_OFF + (size_t)CURSOR_WORDS * MAX_DESC_CHUNKS * sizeof(uint64_t)) == hipSuccess;
    if (ok) {
        ctx->d_xor = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(ctx->d_counts) + CTRL_XOR_OFF);
        ctx->d_cursors = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(ctx->d_counts) + CTRL_CURSORS_OFF);
    }
    ok = ok && hipHostMalloc((void **)&ctx->h_counts, sizeof(Counts), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        *status = S2K_ERR_DEVICE;
        s2k_destroy(ctx);
        return nullptr;
    }
    memset(ctx->h_counts, 0, sizeof(Counts));
    if (const char *e = getenv("S2K_DESC_CHUNKS")) // tuning / A-B runs: chunks of tiles per call on the descriptor path (1 = no second stream)
        if (atoi(e) >= 1 && atoi(e) <= 64) ctx->desc_chunks = (uint32_t)atoi(e);
    if (const char *e = getenv("S2K_TRACE")) ctx->trace = atoi(e) != 0;
    if (const char *e = getenv("S2K_FULL_RUNS")) ctx->force_full_runs = atoi(e) != 0;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        g_live_ctx.push_back(ctx);
    }
    *status = S2K_OK;
    return ctx;
}

void s2k_destroy(s2k_ctx *ctx) {
    if (!ctx) return;
    { // no other context may keep a link to this one (s2k_chain_after): its next call would wait on an event that no longer exists
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (size_t i = 0; i < g_live_ctx.size();) {
            if (g_live_ctx[i] == ctx) {
                g_live_ctx[i] = g_live_ctx.back();
                g_live_ctx.pop_back();
                continue;
            }
            if (g_live_ctx[i]->chain_prev == ctx) g_live_ctx[i]->chain_prev = nullptr;
            i++;
        }
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->ws.release();
    ctx->in_bases.release();
    ctx->realign.release();
    ctx->in_off.release();
    ctx->outbuf.release();
    ctx->in_bases2.release();
    ctx->in_off2.release();
    ctx->outbuf2.release();
    ctx->count_tab.release();
    if (ctx->s_km) (void)hipStreamDestroy(ctx->s_km);
    if (ctx->tiles_done) (void)hipEventDestroy(ctx->tiles_done);
    for (hipEvent_t e : ctx->chunk_ev) (void)hipEventDestroy(e);
    if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
    if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
    if (ctx->d_counts) (void)hipFree(ctx->d_counts); // (d_xor and d_cursors lie inside it)
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    for (hipEvent_t e : ctx->evs) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

s2k_status s2k_set_stream(s2k_ctx *ctx, void *hip_stream) {
    // (the descriptor path's second stream forks from and joins this one with events, call by call)
    if (!ctx) return S2K_ERR_INVALID_ARG;
    if (ctx->pending) {
        s2k_status st = finish(ctx, nullptr);
        (void)st;
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return S2K_OK;
}

s2k_status s2k_chain_after(s2k_ctx *ctx, s2k_ctx *prev) {
    if (!ctx || prev == ctx) return S2K_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (prev) { // a context that is not (or no longer) alive cannot be chained to
        bool live = false;
        for (s2k_ctx *c : g_live_ctx) live = live || c == prev;
        if (!live || prev->device != ctx->device) return S2K_ERR_INVALID_ARG;
    }
    ctx->chain_prev = prev;
    return S2K_OK;
}

s2k_status s2k_enable_timing(s2k_ctx *ctx, int on) {
    if (!ctx) return S2K_ERR_INVALID_ARG;
    ctx->timing = on != 0;
    ctx->ev_used = 0; // restart the per-call event log
    ctx->timed = false;
    return S2K_OK;
}

s2k_status s2k_timing_total(s2k_ctx *ctx, int which, double *ms_sum, uint32_t *n_calls) {
    if (!ctx || !ms_sum || !n_calls || which < 0 || which > 2) return S2K_ERR_INVALID_ARG;
    if (ctx->pending) {
        s2k_status st = finish(ctx, nullptr);
        if (st != S2K_OK && st != S2K_ERR_CAPACITY) return st;
    }
    int a = which == 0 ? 0 : which == 1 ? 1 : 3, b = which == 0 ? 5 : which == 1 ? 2 : 4;
    double sum = 0;
    for (size_t i = 0; i < ctx->ev_used; i++) {
        float ms = 0;
        S2K_TRY(hipEventElapsedTime(&ms, ctx->evs[i * 6 + a], ctx->evs[i * 6 + b]), "event elapsed");
        sum += ms;
    }
    *ms_sum = sum;
    *n_calls = (uint32_t)ctx->ev_used;
    return S2K_OK;
}

s2k_status s2k_last_kernel_ms(s2k_ctx *ctx, int which, float *ms) {
    if (!ctx || !ms || which < 0 || which > 2) return S2K_ERR_INVALID_ARG;
    if (!ctx->timed || !ctx->ev) return fail(ctx, S2K_ERR_INVALID_ARG, "timing was not enabled for the last call");
    if (ctx->pending) {
        s2k_status st = finish(ctx, nullptr);
        if (st != S2K_OK && st != S2K_ERR_CAPACITY) return st;
    }
    int a = which == 0 ? 0 : which == 1 ? 1 : 3, b = which == 0 ? 5 : which == 1 ? 2 : 4;
    S2K_TRY(hipEventElapsedTime(ms, ctx->ev[a], ctx->ev[b]), "event elapsed");
    return S2K_OK;
}

s2k_status s2k_extract_device(s2k_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                              uint64_t n_bases, const s2k_params *params, const s2k_device_out *out,
                              s2k_counts *counts) {
    if (!ctx) return S2K_ERR_INVALID_ARG;
    if (ctx->pending) {
        s2k_status st = finish(ctx, nullptr);
        if (st != S2K_OK && st != S2K_ERR_CAPACITY) return st;
    }
    if (!out || !out->km_off || !d_read_off || (!d_bases && n_bases)) return fail(ctx, S2K_ERR_INVALID_ARG, "NULL device pointer");
    if (n_reads >= 0xFFFFFFFFull) return fail(ctx, S2K_ERR_INVALID_ARG, "too many reads in one call");
    if (out->mn_capacity && !(out->mn_j && out->mn_jend && out->mn_hash)) return fail(ctx, S2K_ERR_INVALID_ARG, "mn_capacity set but minimizer arrays missing");
    S2K_TRY(hipSetDevice(ctx->device), "set device");
    Call &c = ctx->call;
    c = Call();
    s2k_status st = resolve_sem(ctx, params, &c.sem, &c.bound);
    if (st != S2K_OK) return st;
    c.d_bases = d_bases;
    c.d_read_off = d_read_off;
    c.n_reads = n_reads;
    c.n_bases = n_bases;
    c.params = *params;
    c.out = *out;
#ifdef S2K_DEBUG_KNOBS // `make KNOBS=1` / `make PROFILE=1` builds only: ablation timing (results are wrong when set)
    if (const char *dbg = getenv("S2K_DEBUG_SKIP")) c.sem.dbg_skip = (uint32_t)atoi(dbg);
#endif
    c.serial = (params->flags & S2K_FLAG_FORCE_SERIAL) || !tiled_supported(c.sem);
    c.legacy = (params->flags & S2K_FLAG_LEGACY_PATH) != 0;
    c.full_runs = ctx->force_full_runs;
    // The tiled kernel stages tiles with 16 B vector loads, i.e. it needs a 16 B aligned base pointer.  A misaligned
    // stream is first copied to an aligned buffer (one device-to-device pass, ~3 ms per 10 GB) instead of being handed to
    // the read-serial kernels as in round 1 (20-40x slower for the whole call).
    if (!c.serial && n_bases && (((uintptr_t)d_bases) & 15u) != 0) {
        S2K_TRY(ctx->realign.ensure(n_bases + 256), "aligned copy of a misaligned input stream");
        S2K_TRY(hipMemcpyAsync(ctx->realign.p, d_bases, n_bases, hipMemcpyDeviceToDevice, ctx->stream), "realign copy");
        c.d_bases = (const uint8_t *)ctx->realign.p;
    }
    c.pool_cap = c.serial ? pool_estimate(n_bases, n_reads, params->density, c.sem.hpc) : overflow_estimate(n_bases, params->density);
    c.slab_cap = slab_estimate(params->density);
    c.valid = true;
    ctx->pending_status = S2K_OK;
    st = enqueue(ctx);
    if (st != S2K_OK) return st;
    if (counts) return finish(ctx, counts);
    return S2K_OK;
}

s2k_status s2k_trim(s2k_ctx *ctx) {
    if (!ctx) return S2K_ERR_INVALID_ARG;
    if (ctx->pending) {
        s2k_status st = finish(ctx, nullptr);
        if (st != S2K_OK && st != S2K_ERR_CAPACITY) return st;
    }
    (void)ctx->host_pool->trim();
    S2K_TRY(hipSetDevice(ctx->device), "set device");
    S2K_TRY(hipStreamSynchronize(ctx->stream), "stream synchronize");
    for (DevBuf *b : {&ctx->ws, &ctx->in_bases, &ctx->in_off, &ctx->outbuf, &ctx->realign, &ctx->in_bases2, &ctx->in_off2, &ctx->outbuf2, &ctx->count_tab})
        b->release();
    return S2K_OK;
}

s2k_status s2k_sync(s2k_ctx *ctx, s2k_counts *counts) {
    if (!ctx) return S2K_ERR_INVALID_ARG;
    return finish(ctx, counts);
}

// ---- s2k_extract: host buffers in, host SoA out ----------------------------------------------------------------------
// A call is cut into sub-batches of whole reads (~0.5 Gbp each) and pipelined: while the kernels of sub-batch b run, the
// copy threads already pack and send b+1 (own stream, own pinned ring) and a second host thread drains the results of
// b-1 into the result arrays (third stream, second ring).  PCIe is full duplex: the call then costs max(H2D, D2H) instead
// of their sum.  Everything a sub-batch yields is read-relative, so the pieces concatenate; only km_off / mn_off need
// the running totals added.
namespace {

s2k_status carve_out(s2k_ctx *ctx, DevBuf &buf, uint64_t n_reads, uint64_t cap, bool want_mn, s2k_device_out *o) {
    Arena a{nullptr, 0, 0};
    for (int pass = 0; pass < 2; pass++) { // km <= minimizers <= cap, so one capacity serves every output array
        a.off = 0;
        memset(o, 0, sizeof *o);
        o->km_capacity = cap;
        o->km_off = a.take<uint64_t>(n_reads + 1);
        o->hash = a.take<uint64_t>(cap);
        o->start = a.take<uint32_t>(cap);
        o->end = a.take<uint32_t>(cap);
        o->rev = a.take<uint8_t>(cap);
        if (want_mn) {
            o->mn_capacity = cap;
            o->mn_off = a.take<uint64_t>(n_reads + 1);
            o->mn_j = a.take<uint32_t>(cap);
            o->mn_jend = a.take<uint32_t>(cap);
            o->mn_hash = a.take<uint32_t>(cap);
        }
        if (pass == 0) {
            S2K_TRY(buf.ensure(a.off + 256), "output allocation");
            a.base = (char *)buf.p;
        }
    }
    return S2K_OK;
}

struct SubBatch { // one finished sub-batch waiting for its results to be drained
    int slot = 0;
    uint64_t r_lo = 0, n_reads = 0, n_bases = 0;
    s2k_counts cnt{};
    s2k_device_out o{};
};

struct Drain { // the s2k_result under construction and the thread that fills it
    s2k_ctx *ctx;
    HostOwner *ow;
    s2k_result *res;
    bool want_mn;
    uint64_t cap_k = 0, cap_m = 0, used_k = 0, used_m = 0; // elements
    std::atomic<hipError_t> err{hipSuccess};
    std::atomic<bool> nomem{false};
    // hand-over
    std::mutex mu;
    std::condition_variable cv;
    std::vector<SubBatch> queue;
    bool closed = false;
    bool slot_busy[2] = {false, false};

    bool grow(int slot, size_t es, uint64_t used, uint64_t want) {
        size_t ncap = 0;
        bool npin = false;
        void *q = ow->pool->get((size_t)want * es, &ncap, &npin);
        if (!q) return false;
        if (used) memcpy(q, ow->p[slot], (size_t)used * es);
        ow->pool->put(ow->p[slot], ow->cap[slot], ow->pinned[slot]);
        ow->p[slot] = q;
        ow->cap[slot] = ncap;
        ow->pinned[slot] = npin;
        return true;
    }
    bool ensure(uint64_t nk, uint64_t nm) { // the estimate of the whole call was too small (rare): grow, keep what is there
        if (nk > cap_k) {
            const uint64_t w = nk + nk / 8 + 4096;
            if (!grow(1, 8, used_k, w) || !grow(2, 4, used_k, w) || !grow(3, 4, used_k, w) || !grow(4, 1, used_k, w)) return false;
            cap_k = w;
            res->hash = (uint64_t *)ow->p[1];
            res->start = (uint32_t *)ow->p[2];
            res->end = (uint32_t *)ow->p[3];
            res->rev = (uint8_t *)ow->p[4];
        }
        if (want_mn && nm > cap_m) {
            const uint64_t w = nm + nm / 8 + 4096;
            if (!grow(6, 4, used_m, w) || !grow(7, 4, used_m, w) || !grow(8, 4, used_m, w)) return false;
            cap_m = w;
            res->mn_j = (uint32_t *)ow->p[6];
            res->mn_jend = (uint32_t *)ow->p[7];
            res->mn_hash = (uint32_t *)ow->p[8];
        }
        return true;
    }
    void drain(const SubBatch &b) { // D2H of one sub-batch, appended to the result
        if (err != hipSuccess || nomem) return;
        const uint64_t nk = b.cnt.n_kminmers, nm = b.cnt.n_minimizers;
        if (!ensure(used_k + nk, used_m + nm)) {
            nomem = true;
            return;
        }
        hipStream_t s = ctx->s_out;
        s2k::HostStager &hs = ctx->stager_out;
        // an array whose block is pinned takes its part by DMA, asynchronously (one wait for all of them below); the others go through the staging ring
        auto down = [&](int slot, void *dst, const void *src, size_t bytes) -> hipError_t {
            if (bytes == 0) return hipSuccess;
            if (ow->pinned[slot]) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
            return hs.d2h(dst, src, bytes, s);
        };
        hipError_t e = down(0, res->km_off + b.r_lo, b.o.km_off, (b.n_reads + 1) * 8);
        if (e == hipSuccess) e = down(1, res->hash + used_k, b.o.hash, nk * 8);
        if (e == hipSuccess) e = down(2, res->start + used_k, b.o.start, nk * 4);
        if (e == hipSuccess) e = down(3, res->end + used_k, b.o.end, nk * 4);
        if (e == hipSuccess) e = down(4, res->rev + used_k, b.o.rev, nk);
        if (want_mn) {
            if (e == hipSuccess) e = down(5, res->mn_off + b.r_lo, b.o.mn_off, (b.n_reads + 1) * 8);
            if (e == hipSuccess) e = down(6, res->mn_j + used_m, b.o.mn_j, nm * 4);
            if (e == hipSuccess) e = down(7, res->mn_jend + used_m, b.o.mn_jend, nm * 4);
            if (e == hipSuccess) e = down(8, res->mn_hash + used_m, b.o.mn_hash, nm * 4);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            err.store(e);
            return;
        }
        if (used_k)
            for (uint64_t r = 0; r <= b.n_reads; r++) res->km_off[b.r_lo + r] += used_k;
        if (want_mn && used_m)
            for (uint64_t r = 0; r <= b.n_reads; r++) res->mn_off[b.r_lo + r] += used_m;
        used_k += nk;
        used_m += nm;
    }
    void run() { // consumer thread
        (void)hipSetDevice(ctx->device);
        for (;;) {
            SubBatch b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closed || !queue.empty(); });
                if (queue.empty()) return;
                b = queue.front();
                queue.erase(queue.begin());
            }
            drain(b);
            {
                std::lock_guard<std::mutex> lk(mu);
                slot_busy[b.slot] = false;
            }
            cv.notify_all();
        }
    }
    void push(const SubBatch &b) {
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(b);
        }
        cv.notify_all();
    }
    void wait_slot_free(int slot) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !slot_busy[slot]; });
    }
    void mark_busy(int slot) {
        std::lock_guard<std::mutex> lk(mu);
        slot_busy[slot] = true;
    }
    void close() {
        {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
        }
        cv.notify_all();
    }
};

} // namespace

s2k_status s2k_set_host_batch(s2k_ctx *ctx, uint64_t bases) {
    if (!ctx) return S2K_ERR_INVALID_ARG;
    ctx->host_batch = bases ? bases : (1ull << 29);
    return S2K_OK;
}

s2k_status s2k_extract(s2k_ctx *ctx, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                       const s2k_params *params, s2k_result *res) {
    if (!ctx || !res) return S2K_ERR_INVALID_ARG;
    memset(res, 0, sizeof *res);
    if (!read_off) return fail(ctx, S2K_ERR_INVALID_ARG, "read_off is NULL");
    Sem sem;
    uint32_t bound;
    s2k_status st = resolve_sem(ctx, params, &sem, &bound);
    if (st != S2K_OK) return st;
    for (uint64_t r = 0; r < n_reads; r++) {
        if (read_off[r + 1] < read_off[r]) return fail(ctx, S2K_ERR_INVALID_ARG, "read_off is not monotone");
        if (read_off[r + 1] - read_off[r] > 0xFFFFFFFEull) return fail(ctx, S2K_ERR_READ_TOO_LONG, "read too long");
    }
    const uint64_t first = read_off[0], n_bases = read_off[n_reads] - first;
    if (n_bases && !bases) return fail(ctx, S2K_ERR_INVALID_ARG, "bases is NULL");
    S2K_TRY(hipSetDevice(ctx->device), "set device");
    if (ctx->pending) (void)finish(ctx, nullptr);
#ifdef S2K_DEBUG_KNOBS // wall time of this call on stderr (tools/pcie_trace.sh with a KNOBS build)
    const bool trace = getenv("S2K_TRACE_EXTRACT") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
#endif

    // ---- sub-batches of whole reads, balanced by bases ------------------------------------------------------------
    std::vector<uint64_t> cut{0};
    {
        const uint64_t bb = ctx->host_batch;
        const uint64_t nb = n_bases < bb + bb / 2 ? 1 : (n_bases + bb / 2) / bb;
        for (uint64_t b = 1; b < nb; b++) {
            const uint64_t target = first + (uint64_t)((unsigned __int128)n_bases * b / nb);
            const uint64_t r = (uint64_t)(std::lower_bound(read_off, read_off + n_reads + 1, target) - read_off);
            if (r > cut.back() && r < n_reads) cut.push_back(r);
        }
        cut.push_back(n_reads);
    }
    const size_t B = cut.size() - 1;
    if (B > 1) {
        if (!ctx->s_in) S2K_TRY(hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking), "stream create");
    }
    if (!ctx->s_out) S2K_TRY(hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking), "stream create");
    hipStream_t s_in = B > 1 ? ctx->s_in : ctx->stream;

    // ---- the result arrays, sized by the estimate for the whole call (they grow if it was too small) -----------------
    const bool want_mn = params->flags & S2K_FLAG_WANT_MINIMIZERS;
    HostOwner *ow = new (std::nothrow) HostOwner(ctx->host_pool);
    if (!ow) return fail(ctx, S2K_ERR_NOMEM, "host result allocation");
    Drain dr{ctx, ow, res, want_mn};
    {
        // one sub-batch: allocated at the exact size when the counts are known; several: the bound the device arrays use
        // (pages that are never touched cost nothing)
        const uint64_t est = B == 1 ? 0 : pool_estimate(n_bases, n_reads, params->density, sem.hpc);
        res->km_off = ow->take<uint64_t>(0, n_reads + 1);
        if (want_mn) res->mn_off = ow->take<uint64_t>(5, n_reads + 1);
        bool ok = res->km_off && (!want_mn || res->mn_off);
        if (ok && est) {
            res->hash = ow->take<uint64_t>(1, est);
            res->start = ow->take<uint32_t>(2, est);
            res->end = ow->take<uint32_t>(3, est);
            res->rev = ow->take<uint8_t>(4, est);
            dr.cap_k = est;
            ok = res->hash && res->start && res->end && res->rev;
            if (ok && want_mn) {
                res->mn_j = ow->take<uint32_t>(6, est);
                res->mn_jend = ow->take<uint32_t>(7, est);
                res->mn_hash = ow->take<uint32_t>(8, est);
                dr.cap_m = est;
                ok = res->mn_j && res->mn_jend && res->mn_hash;
            }
        }
        if (!ok) {
            delete ow;
            memset(res, 0, sizeof *res);
            return fail(ctx, S2K_ERR_NOMEM, "host result allocation");
        }
    }
    std::thread consumer;
    if (B > 1) consumer = std::thread([&dr] { dr.run(); });
    auto bail = [&](s2k_status code) { // leave nothing behind: consumer joined, result freed
        if (consumer.joinable()) {
            dr.close();
            consumer.join();
        }
        (void)hipStreamSynchronize(ctx->stream);
        ctx->pending = false;
        delete ow;
        memset(res, 0, sizeof *res);
        return code;
    };

    DevBuf *in_b[2] = {&ctx->in_bases, &ctx->in_bases2}, *in_o[2] = {&ctx->in_off, &ctx->in_off2};
    DevBuf *out_b[2] = {&ctx->outbuf, &ctx->outbuf2};
    s2k_counts total{};
    total.hash_bound = bound;
    std::vector<uint64_t> off;
    SubBatch prev;
    bool have_prev = false;
    // finishes sub-batch `prev` (its kernels ran while the next one was being sent) and hands it to the drain
    auto complete_prev = [&]() -> s2k_status {
        s2k_counts cnt;
        s2k_status r = finish(ctx, &cnt);
        for (int attempt = 0; r == S2K_ERR_CAPACITY && attempt < 3; attempt++) { // device arrays too small: exact size, again
            const uint64_t cap = (cnt.n_minimizers > cnt.n_kminmers ? cnt.n_minimizers : cnt.n_kminmers) + 4096;
            r = carve_out(ctx, *out_b[prev.slot], prev.n_reads, cap, want_mn, &prev.o);
            if (r != S2K_OK) return r;
            r = s2k_extract_device(ctx, (const uint8_t *)in_b[prev.slot]->p, (const uint64_t *)in_o[prev.slot]->p, prev.n_reads,
                                   prev.n_bases, params, &prev.o, &cnt);
        }
        if (r != S2K_OK) return r;
        prev.cnt = cnt;
        total.n_minimizers += cnt.n_minimizers;
        total.n_kminmers += cnt.n_kminmers;
        total.xor_hash ^= cnt.xor_hash;
        total.path |= cnt.path;
        if (B > 1) dr.push(prev);
        else dr.drain(prev);
        return S2K_OK;
    };
    for (size_t b = 0; b < B; b++) {
        const int slot = (int)(b & 1);
        const uint64_t r_lo = cut[b], r_hi = cut[b + 1], nr = r_hi - r_lo;
        const uint64_t b0 = read_off[r_lo], nbase = read_off[r_hi] - b0;
        // inputs -> HBM (+ slack so vector loads of the last tile stay inside the allocation); slot b&1 was last read by the
        // kernels of sub-batch b-2, which complete_prev() of iteration b-1 has waited for
        hipError_t e = in_b[slot]->ensure(nbase + 256);
        if (e == hipSuccess) e = in_o[slot]->ensure((nr + 1) * sizeof(uint64_t));
        if (e != hipSuccess) return bail(fail(ctx, S2K_ERR_DEVICE, "input allocation", e));
        off.resize(nr + 1);
        for (uint64_t r = 0; r <= nr; r++) off[r] = read_off[r_lo + r] - b0;
        // bases cross PCIe packed to 2 bits (exact: every non-ACGT byte travels in a side list), unless the caller opts out
        if (params->flags & S2K_FLAG_NO_PACK2) e = ctx->stager.h2d(in_b[slot]->p, bases + b0, nbase, s_in);
        else e = ctx->stager.h2d_packed(in_b[slot]->p, bases + b0, nbase, s_in);
        if (e == hipSuccess) e = hipMemcpyAsync(in_o[slot]->p, off.data(), (nr + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s_in);
        if (e == hipSuccess) e = hipStreamSynchronize(s_in);
        if (e != hipSuccess) return bail(fail(ctx, S2K_ERR_DEVICE, "H2D", e));
        if (have_prev) {
            st = complete_prev();
            if (st != S2K_OK) return bail(st);
        }
        dr.wait_slot_free(slot); // the drain of sub-batch b-2 reads out_b[slot]
        if (dr.err != hipSuccess || dr.nomem) break;
        prev = SubBatch();
        prev.slot = slot;
        prev.r_lo = r_lo;
        prev.n_reads = nr;
        prev.n_bases = nbase;
        st = carve_out(ctx, *out_b[slot], nr, pool_estimate(nbase, nr, params->density, sem.hpc), want_mn, &prev.o);
        if (st != S2K_OK) return bail(st);
        dr.mark_busy(slot);
        st = s2k_extract_device(ctx, (const uint8_t *)in_b[slot]->p, (const uint64_t *)in_o[slot]->p, nr, nbase, params, &prev.o, nullptr);
        if (st != S2K_OK) return bail(st);
        have_prev = true;
    }
    if (have_prev && dr.err == hipSuccess && !dr.nomem) {
        st = complete_prev();
        if (st != S2K_OK) return bail(st);
    }
    if (consumer.joinable()) {
        dr.close();
        consumer.join();
    }
    if (dr.nomem) return bail(fail(ctx, S2K_ERR_NOMEM, "host result allocation"));
    if (dr.err != hipSuccess) return bail(fail(ctx, S2K_ERR_DEVICE, "D2H results", dr.err));
    total.n_reads = n_reads;
    total.n_bases = n_bases;
    res->n_reads = n_reads;
    res->n_kminmers = total.n_kminmers;
    res->n_minimizers = want_mn ? total.n_minimizers : 0;
    res->counts = total;
    res->_owner = ow;
#ifdef S2K_DEBUG_KNOBS
    if (trace)
        fprintf(stderr, "s2k_extract: %zu sub-batch(es), %.1f ms (%.2f Gbp)\n", B,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), n_bases * 1e-9);
#endif
    return S2K_OK;
}

void s2k_result_free(s2k_result *res) {
    if (!res || !res->_owner) return;
    delete reinterpret_cast<HostOwner *>(res->_owner);
    memset(res, 0, sizeof *res);
}

s2k_status s2k_synth_bases_device(s2k_ctx *ctx, uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *d_bases) {
    if (!ctx || (!d_bases && n)) return S2K_ERR_INVALID_ARG;
    S2K_TRY(hipSetDevice(ctx->device), "set device");
    S2K_TRY(launch_synth(seed, first_base, n, d_bases, ctx->stream), "synth kernel");
    return S2K_OK;
}

void s2k_synth_hifi_lengths(uint64_t seed, uint64_t r0, uint64_t n_reads, uint64_t *lengths) {
    if (!lengths) return;
    for (uint64_t i = 0; i < n_reads; i++) lengths[i] = synth_hifi_len(seed, r0 + i);
}

s2k_status s2k_synth_hifi_device(s2k_ctx *ctx, uint64_t seed, uint64_t r0, uint64_t n_reads, const uint64_t *d_read_off, uint8_t *d_bases) {
    if (!ctx || ((!d_read_off || !d_bases) && n_reads)) return S2K_ERR_INVALID_ARG;
    S2K_TRY(hipSetDevice(ctx->device), "set device");
    S2K_TRY(launch_synth_hifi(seed, r0, n_reads, d_read_off, d_bases, ctx->stream), "synth hifi kernel");
    return S2K_OK;
}

s2k_status s2k_hpc_device(s2k_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                          uint64_t n_bases, uint64_t *d_hpc_off, uint8_t *d_hpc, uint32_t *d_pos, uint64_t capacity,
                          uint64_t *n_runs) {
    return s2k_hpc_device_ex(ctx, d_bases, d_read_off, n_reads, n_bases, 0, d_hpc_off, d_hpc, d_pos, capacity, n_runs);
}

s2k_status s2k_hpc_device_ex(s2k_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                             uint64_t n_bases, uint32_t flags, uint64_t *d_hpc_off, uint8_t *d_hpc, uint32_t *d_pos,
                             uint64_t capacity, uint64_t *n_runs) {
    if (!ctx || !d_read_off || !d_hpc_off || (!d_bases && n_bases)) return S2K_ERR_INVALID_ARG;
    if (flags & ~(uint32_t)S2K_HPC_RLE_ALPHABET) return fail(ctx, S2K_ERR_INVALID_ARG, "unknown s2k_hpc_device_ex flag");
    const bool rle = (flags & S2K_HPC_RLE_ALPHABET) != 0; // encode_rle: only ACTGactgNn collapse (src/hpc.rs:14)
    S2K_TRY(hipSetDevice(ctx->device), "set device");
    if (ctx->pending) (void)finish(ctx, nullptr);
    // segment-parallel path: 16-byte vector loads, so a misaligned stream is first copied to an aligned buffer
    if (n_reads && n_bases && ((uintptr_t)d_bases & 15) != 0) {
        S2K_TRY(ctx->realign.ensure(n_bases + 256), "aligned copy of a misaligned input stream");
        S2K_TRY(hipMemcpyAsync(ctx->realign.p, d_bases, n_bases, hipMemcpyDeviceToDevice, ctx->stream), "realign copy");
        d_bases = (const uint8_t *)ctx->realign.p;
    }
    { // the read table is caller memory in HBM: checked on the device before any kernel walks it (this call blocks anyway)
        S2K_TRY(hipMemsetAsync(&ctx->d_counts->bad_input, 0, sizeof(uint32_t), ctx->stream), "memset");
        S2K_TRY(launch_validate_read_off(d_read_off, n_reads, n_bases, &ctx->d_counts->bad_input, ctx->stream), "read table validation");
        uint32_t bad = 0;
        S2K_TRY(hipMemcpyAsync(&bad, &ctx->d_counts->bad_input, sizeof bad, hipMemcpyDeviceToHost, ctx->stream), "D2H");
        S2K_TRY(hipStreamSynchronize(ctx->stream), "stream synchronize");
        if (bad & ~(uint32_t)BAD_LONG) return fail(ctx, S2K_ERR_INVALID_ARG, "d_read_off must start at 0, be non-decreasing and end at n_bases");
        if (bad) return fail(ctx, S2K_ERR_READ_TOO_LONG, "a read is longer than 2^32-2 bases");
    }
    const bool seg_path = n_reads && n_bases && ((uintptr_t)d_bases & 15) == 0;
    const uint64_t nblk = n_bases / 256 + 1;
    Arena a{nullptr, 0, 0};
    uint32_t *cnt = nullptr, *blk_cnt = nullptr;
    uint64_t *tmp = nullptr, *blk_off = nullptr, *blk_tmp = nullptr, *read_c0 = nullptr;
    uint32_t *seg_index = nullptr, *lb_ws = nullptr;
    for (int pass = 0; pass < 2; pass++) {
        a.off = 0;
        cnt = a.take<uint32_t>(n_reads + 1);
        tmp = a.take<uint64_t>(scan_tmp_bytes(n_reads) / sizeof(uint64_t) + 1);
        if (seg_path) {
            blk_cnt = a.take<uint32_t>(nblk + 1);
            blk_off = a.take<uint64_t>(nblk + 2);
            blk_tmp = a.take<uint64_t>(scan_tmp_bytes(nblk) / sizeof(uint64_t) + 1);
            read_c0 = a.take<uint64_t>(n_reads + 1);
            seg_index = a.take<uint32_t>(hpc_segment_index_words(n_bases));
            lb_ws = a.take<uint32_t>(hpc_single_pass_words(n_bases));
        }
        if (pass == 0) {
            S2K_TRY(ctx->ws.ensure(a.off + 256), "workspace allocation");
            a.base = (char *)ctx->ws.p;
        }
    }
    // Two passes (default): the runs of every read counted first, then scanned (rounds 3-5).  One pass (S2K_HPC_SINGLE_PASS=1, round 6): every segment
    // finds its first output slot by a decoupled look-back over the run-head counts of the segments before it -- bit-exact (tools/fuzz_hpc.py) and
    // 1.7-2.1 x SLOWER on MI355X: a block's prefix can only be formed behind its predecessors', and words published on one XCD reach the others through
    // memory, so the prefixes spread over the 488 k blocks of 2 Gbp at that latency whatever the window (64 words per look: 4.58 ms; 256: 5.64 ms;
    // two passes: 2.66 ms -- profiles/r06_hpc_single_pass_*.txt).  Kept selectable, not used.
    static const bool single_pass_env = getenv("S2K_HPC_SINGLE_PASS") != nullptr;
    if (seg_path && single_pass_env) {
        uint32_t *fail_word = nullptr;
        S2K_TRY(launch_hpc_single_pass(d_bases, d_read_off, n_reads, n_bases, lb_ws, d_hpc_off, d_hpc, d_pos, capacity, &fail_word, ctx->stream, rle),
                "hpc single-pass kernels");
        uint32_t failed = 0;
        uint64_t total1 = 0;
        S2K_TRY(hipMemcpyAsync(&failed, fail_word, sizeof failed, hipMemcpyDeviceToHost, ctx->stream), "D2H");
        S2K_TRY(hipMemcpyAsync(&total1, d_hpc_off + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream), "D2H");
        S2K_TRY(hipStreamSynchronize(ctx->stream), "sync");
        if (!failed) {
            if (n_runs) *n_runs = total1;
            return total1 > capacity && (d_hpc || d_pos) ? S2K_ERR_CAPACITY : S2K_OK;
        }
    }
    if (seg_path) {
        S2K_TRY(launch_read_run_counts(d_bases, d_read_off, n_reads, n_bases, blk_cnt, blk_off, blk_tmp, cnt, read_c0, ctx->stream, rle),
                "run count kernels");
        S2K_TRY(launch_scan_u32(cnt, n_reads, d_hpc_off, tmp, 0, ctx->stream), "scan");
        if (d_hpc || d_pos)
            S2K_TRY(launch_hpc_segments(d_bases, d_read_off, n_reads, n_bases, d_hpc_off, blk_off, read_c0, seg_index, d_hpc, d_pos, capacity,
                                        ctx->stream, rle),
                    "hpc segment kernel");
    } else {
        S2K_TRY(launch_hpc_count(d_bases, d_read_off, n_reads, cnt, ctx->stream, rle), "hpc count kernel");
        S2K_TRY(launch_scan_u32(cnt, n_reads, d_hpc_off, tmp, 0, ctx->stream), "scan");
        if (d_hpc || d_pos) S2K_TRY(launch_hpc_write(d_bases, d_read_off, n_reads, d_hpc_off, d_hpc, d_pos, capacity, ctx->stream, rle), "hpc write kernel");
    }
    uint64_t total = 0;
    S2K_TRY(hipMemcpyAsync(&total, d_hpc_off + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream), "D2H");
    S2K_TRY(hipStreamSynchronize(ctx->stream), "sync");
    if (n_runs) *n_runs = total;
    return total > capacity && (d_hpc || d_pos) ? S2K_ERR_CAPACITY : S2K_OK;
}

} // extern "C"
