// s2k_tile_impl.h -- the tiled minimizer kernel (included by s2k_tile.hip and, once per compile-time l, by s2k_tile_inst.hip): fused homopolymer compression + 32-bit canonical ntHash1
// + density threshold + original-space back-map, for gfx950.
//
// What it replaces: the per-base loop of the reference's scalar iterators
// (NtHashHPCIterator::next, src/nthash_hpc.rs:241-278; the Regular arm, src/lib.rs:215-230), which
// is ~100 % of the reference's time.  It is NOT a translation of either the
// scalar loop (loop-carried over the whole read) or the AVX-512 code (16-lane Hillis-Steele scan):
//
//  * The batch of reads is one byte stream; the stream is cut into fixed tiles of 64 x 144 = 9216
//    bases, one wave per tile, independent of read lengths.  l-mers that straddle a read boundary
//    are cleared from the hit masks afterwards, so ragged reads cost nothing in the hot loop.
//  * Persistent waves, tiles dealt DYNAMICALLY: a wave takes its first three tiles statically (the software pipeline
//    is three deep) and every later one from one of 64 cursors in global memory (one atomic per tile, issued at the
//    top of an iteration and looked at after the hash loop).  Everything per tile lives in SGPRs.  One block per CU: 12
//    waves for the Hpc modes, 16 for the Regular family (tw<HPC>()).
//  * A tile's 9344 bytes (9216 + 128 B look-ahead) go from global memory STRAIGHT into the wave's LDS buffer
//    (global_load_lds_dwordx4, issued from inline asm, 1 KiB per wave-instruction, no registers, no ds_write) while
//    the previous tile's one-lane-per-hit rounds run; the top of an iteration waits for them with a COUNTED
//    s_waitcnt vmcnt(n) that leaves the record stores issued after them in flight (see stores_after_dma below and
//    tests/test_isa_invariants.py, which checks the count against the disassembly).
//  * Each lane owns 144 consecutive l-mer start positions and *rolls* the hash privately
//    (fh' = rotl(fh,1) ^ OUT[s[p]] ^ IN[s[p+l]]), reading its bytes from LDS in 16 B pieces (lane stride 16*odd bytes:
//    bank-conflict free) and its seeds from two 256-entry LDS tables that have the rotations pre-applied (one
//    ds_read_b64 each, one SDWA instruction per base for the address).  A lane pays an l-base warm-up instead of a
//    cross-lane scan.
//  * Minimizers are rare (~2 % of positions), but a per-position branch would be taken by almost every wave
//    (64 lanes x 2 %).  So the hot loop is branch-free: per position it shifts a hit bit into a register and keeps the
//    hash of the last hit of each 16-position piece (compare -> select -> add-with-carry through VCC).  Hit bits
//    (5 registers) and kept hashes (9 registers) stay in VGPRs.
//  * Dense phase: read starts become hash-space boundaries and invalid l-mers are cleared from the masks by range;
//    popcounts + one DPP scan give every lane its output offset; lanes list their hits (16-bit entries in LDS); the
//    ~14 % of hits that were not the last of their piece are queued and re-derived from their l bytes, four lanes per
//    hit, BEFORE the rounds (from then on the tile's bytes are dead and the next tile is loaded into the same buffer);
//    then one lane per hit back-maps its position and writes ONE word per minimizer: {offset of the l-mer's first base in the
//    tile : 14, span to its last base : 18}.  The tile also leaves a 64-bit word (what it does to the running pair "k-min-mers
//    so far, minimizers of the read that continues") and its list of read segments; k-min-mers are made from these by
//    s2k_desc.hip (descriptor path).  The legacy path (k > 32, > 30 read starts per tile) instead finds each hit's read among
//    the read starts kept in LDS and writes 16-byte records for s2k_kminmer.hip.  No global LOAD sits in the round loop: one
//    would make the compiler drain every outstanding store of the previous round.
//  * Hpc mode first compacts the tile's run heads in place in LDS (SWAR byte compares -> v_dot4 nibbles -> per-lane
//    flag masks in natural bit order, popcounts, one wave scan; then per EIGHT raw bytes two v_perm_b32 pack their run
//    heads, and aligned LDS operations put them in place at the lane's fill level -- round 5; rounds 1-4: one byte
//    store per raw byte), appends the l run heads that follow the tile (first from the staged look-ahead, then by a
//    loop over the stream, so arbitrarily long homopolymers are fine), and then runs the same hash loop over the
//    compacted bytes.  Raw positions are recovered for hits only: a directory of every 64th run head names the raw
//    lane, that lane's row (flag words, prefix, cumulative word counts) names the flag word, a 2 KiB table the bit
//    (HpcLds, hpc_rawpos2).  Read starts are forced run heads: they are OR-ed into the flag masks; the staged bytes
//    are never modified, so ANY byte value is fine.
//  * HpcSimd result semantics need the run count of the whole read where it ends: every tile publishes how many run heads of
//    the read continuing past its end it holds, the tile in which the read ends looks back (lookback_heads) -- no second pass.
//  * Records go to a fixed per-tile slab (mean + 6 sigma); only a tile with more hits takes space from a
//    shared overflow region with one atomic.  (One shared cursor for every tile serialised the kernel.)
//
// Diagnostics: S2K_DEBUG_SKIP (bit 1 skip hash loop, 2 skip dense phase, 4 skip compaction, 8 per-phase cycle
// stamps printed by the host -- only in builds made with `make PROFILE=1` --, 16/64 skip stores / re-derivation, 128 skip
// listing + re-derivation + rounds) and
// S2K_DEBUG_BLOCKS_PER_CU are timing ablations only -- results are wrong when a skip bit is set.  The library reads
// these environment variables only when built with `make KNOBS=1` (or PROFILE=1); the shipped build ignores them.
#pragma once
#include "s2k_dev.h"
#include "s2k_static_l.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace s2k {
namespace {

#ifndef S2K_TW
#define S2K_TW 12
#endif
#ifndef S2K_TW_REG
#ifdef S2K_PROFILE
#define S2K_TW_REG 12 // PROFILE builds: the phase accumulators need 2 KiB of LDS, and 16 Regular waves use all 160 KiB -- the Regular family is profiled at three waves per SIMD
#else
#define S2K_TW_REG 16
#endif
#endif
// waves per block = all the waves of a CU: ONE block per CU shares the seed tables.  Hpc: 12 = three per SIMD (12 KB of LDS per wave: the rows of the
// back-map beside the tile buffer), and the k-min-mer kernel's blocks run BESIDE it in what is left of the CU.  Regular family (Regular, Simd): 16 = four
// per SIMD (9984 B per wave), nothing fits beside them, and the k-min-mer stage runs BEHIND the minimizer kernel, one launch of each per call:
// the fourth wave buys the minimizer kernel 8 % (4.32 -> 3.96 ms per 10 Gbp), which since round 5 is more than running the k-min-mer kernel beside a
// three-wave block saves (profiles/r05_regular_occupancy.txt; round 4 measured the two equal).
template <bool HPC> constexpr int tw() { return HPC ? S2K_TW : S2K_TW_REG; }
template <bool HPC> constexpr int waves_per_simd() { return tw<HPC>() / 4; }
static_assert(S2K_TW % 4 == 0 && S2K_TW_REG % 4 == 0, "whole waves per SIMD");
constexpr int HS_OFF = 16;                             // data starts here; byte HS_OFF-1 absorbs "slot -1" stores
constexpr int BUF_BYTES = HS_OFF + TILE_BASES + 128;   // tile + halo / window slack
#ifndef S2K_CAPP
#define S2K_CAPP 16
#endif
constexpr int CAPP = S2K_CAPP;                         // positions per capture piece: 16 or 8 (8: half the re-derivations, nine more registers)
constexpr int NPC = TILE_T / CAPP;                     // 9 capture pieces per lane
constexpr int MAX_L_TILED = 64;
#ifndef S2K_LISTCAP
#define S2K_LISTCAP 256
#endif
#ifndef S2K_JOBCAP
#define S2K_JOBCAP 32
#endif
#ifndef S2K_REDERIVE_GROUP
#define S2K_REDERIVE_GROUP 8 // bases a lane of a re-derivation quad fetches at a time (8 = all of them at once; 4: +0.2 % kernel time, profiles/r05_ab_knobs_steady.txt)
#endif
constexpr int LISTCAP_REG = S2K_TW_REG > 12 ? 216 : S2K_LISTCAP; // hits handled per dense batch (Regular: 184 +- 13 per tile; 216 = what 16 waves' LDS leaves)
constexpr int LISTCAP_HPC = 192;                                      // ... Hpc: 137 +- 11 per tile of uniform ACGT; the 128 bytes pay for the wider rows of the back-map
template <bool HPC> constexpr int listcap() { return HPC ? LISTCAP_HPC : LISTCAP_REG; }
template <bool HPC> constexpr int jobcap() { return !HPC && S2K_TW_REG > 12 ? 16 : S2K_JOBCAP; } // queued hash re-derivations per flush
#ifndef S2K_REG_LA
#define S2K_REG_LA 1
#endif
#ifndef S2K_HPC_LA
#define S2K_HPC_LA 2
#endif
constexpr int REG_LA = S2K_REG_LA;                     // seed look-ahead (positions) of the hash loop: the LDS round trip of a look-up is covered by
constexpr int HPC_LA = S2K_HPC_LA;                     // LA positions of the lane's own arithmetic (4 registers per position of look-ahead)
constexpr int NBL = 32;                                // read starts of a tile kept in LDS (hb / rs16); tiles with more search the read table
// vector-memory operations one round of 64 hits issues: the counted vmcnt wait relies on it (tools/isa/check_vmcnt.py checks it).
// Descriptor path: one store ({offset in the tile, span}); legacy path: three (j, jend, read index).
template <bool DESC> constexpr int stores_per_round() { return DESC ? 1 : 3; }
static_assert(NBL == META_SEGS, "TileMeta keeps NBL segment starts");
constexpr int NPRE = 10;                               // 16 B/lane loads that stage one tile + 128 B look-ahead

// Back-map data of a tile, written by hpc_compact.  Everything a hit needs to find the raw offset of a run head x is TWO small reads
// away from x itself: sd[x >> 6] names the raw lane that owns head 64 (x >> 6) -- the owner of x is that lane or the next one unless a lane
// holds fewer than 64 heads --, and that lane's row says how many heads lie before it (prefix) and before each of its flag words (cum),
// so that ONE flag word is fetched and searched (sel8).  (Rounds 2-4: a hint per hash lane, a search over the prefixes, two whole rows of
// five words and a popcount walk over them per head -- nine dependent LDS round trips and ~250 instructions per round of 64 hits.)
#ifndef S2K_SD_SHIFT
#define S2K_SD_SHIFT 6
#endif
constexpr int ROW_W = 6, ROW_PFX = 4, ROW_CUM = 5, N_ROWS = 66, SD_SHIFT = S2K_SD_SHIFT;
struct HpcLds {
    uint32_t row[N_ROWS][ROW_W]; // raw lane o: words 0-3 = run-head flags of its bytes 0..127 (bit i of word g <-> byte 32 g + i), word 4 = flags of bytes
                                 // 128..143 | exclusive prefix of the lanes' head counts << 16, word 5 = heads in words 0..g, one byte per g = 0..3.
                                 // Rows 64, 65: sentinels (prefix = R, the tile's head count)
    uint32_t halo_pos[64];       // tile-relative raw offsets of the run heads that follow the tile
    uint8_t sd[SD_SHIFT == 6 ? 160 : 288]; // sd[m] = raw lane that owns run head 64 m (m <= (R - 1) / 64 <= 143)
};
static_assert(sizeof(HpcLds) % 16 == 0, "the tile buffer behind it is read 16 bytes at a time");
struct NoHpcLds {};

template <bool HPC>
struct alignas(16) WaveLdsT : std::conditional<HPC, HpcLds, NoHpcLds>::type {
    uint8_t buf[BUF_BYTES];
    uint16_t list[listcap<HPC>()]; // validated hits of the current batch: tile-local hash position (bits 0-13), ascending; bit 15 = hash must be re-derived
    uint16_t jobx[jobcap<HPC>()];    // hits whose hash must be re-derived: tile-local position ...
    uint16_t jobslot[jobcap<HPC>()]; // ... and index of the hit among the tile's hits
    int16_t hb[NBL];         // read starts inside the tile, as hash-space positions (ascending; 0 .. TILE_BASES)
    uint16_t rs16[NBL];      // rs16[i] = read_off[r0 + i] - t0 for the read starts inside the tile (i >= 1; read r0 starts at rs0, kept in a register)
};
#ifdef S2K_PROFILE // phase accumulators of the PROFILE build (see S2K_STAMP): 16 x 8 B per wave behind the waves' buffers
typedef __attribute__((address_space(3))) unsigned long long *ph_ptr_t;
constexpr int PROFILE_LDS_BYTES = 16 * 16 * 8;
#else
typedef uint64_t *ph_ptr_t;
constexpr int PROFILE_LDS_BYTES = 0;
#endif
constexpr int SEED_TABLE_BYTES = 2 * 256 * 8; // IN table at 0: {h[c], rotl(rc[c], l-1)};  OUT table at 2048: {rotl(h[c], l), rotr(rc[c], 1)}
#ifndef S2K_SEL8
#define S2K_SEL8 1
#endif
// Hpc: behind the seed tables, sel8[m][n] = index of the n-th set bit of the byte m, 0x0C for n >= popcount(m) (2 KiB): the last three levels of the
// back-map's select-nth-set-bit are one look-up -- and entry m, read as two dwords, is the pair of v_perm_b32 selectors with which pass 2 of the
// compaction packs the run heads of eight raw bytes whose flag byte is m (hpc_compact)
constexpr int TABLE_BYTES = SEED_TABLE_BYTES; // (tools/experiments)
constexpr int SEL8_OFF = SEED_TABLE_BYTES, SEL8_BYTES = S2K_SEL8 ? 256 * 8 : 0;
#ifndef S2K_PASS2_ACC
#define S2K_PASS2_ACC 1 // Hpc compaction, pass 2: 1 = eight raw bytes per step, packed by v_perm_b32 and placed with aligned LDS operations (round 5), 0 = one byte store per raw byte (rounds 1-4)
#endif
static_assert(!S2K_PASS2_ACC || S2K_SEL8, "pass 2 reads its v_perm_b32 selectors from sel8");
#ifndef S2K_LPH
#define S2K_LPH 0 // 1 = Hpc hits listed one lane per hit (round 6 experiment, dense_phase: 2.4 % SLOWER, profiles/r06_steady_tile1.txt); 0 = every lane lists its own in a loop
#endif
#ifndef S2K_WARM2
#define S2K_WARM2 1 // Hpc: a third seed table with the rotations of a warm-up PAIR's first base pre-applied (round 6; the Regular block has no LDS left for it)
#endif
// Hpc: behind sel8, WARM[c] = {rotl(h[c], 1), rotr(IN[c].y, 1)}: the first l-mer of a lane is then built two bases per step as
// fh = rotl(fh, 2) ^ WARM[s[i]] ^ IN[s[i+1]] -- one rotation and one three-input xor per strand and pair instead of two rotations and the xor
constexpr int WARM_OFF = SEL8_OFF + SEL8_BYTES, WARM_BYTES = S2K_WARM2 ? 256 * 8 : 0;
template <bool HPC> constexpr int table_bytes() { return SEED_TABLE_BYTES + (HPC ? SEL8_BYTES + WARM_BYTES : 0); }
template <bool HPC>
constexpr int block_lds_bytes() { return table_bytes<HPC>() + tw<HPC>() * (int)sizeof(WaveLdsT<HPC>) + PROFILE_LDS_BYTES; }
// one block of tw<HPC>() waves per CU: it may use the whole 160 KiB (MI355X_MICROARCH.md: "a single workgroup may declare all 160 KiB")
#ifndef S2K_EXPERIMENT
static_assert(block_lds_bytes<true>() <= 160 * 1024 && block_lds_bytes<false>() <= 160 * 1024, "the block of a CU must fit its 160 KiB of LDS");
#endif

// inclusive scan over the 64 lanes with DPP row shifts / broadcasts (no LDS round trips)
__device__ inline uint32_t wave_incl_scan(uint32_t v, int lane) {
    (void)lane;
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
// inclusive MAX scan over the 64 lanes (values >= 0), same DPP steps
__device__ inline uint32_t wave_incl_max(uint32_t v) {
    uint32_t t, a;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    a = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    a = a > t ? a : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false); // row_shr:3
    a = a > t ? a : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xf, 0xe, false); // row_shr:4, banks 1-3
    a = a > t ? a : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xc, false); // row_shr:8, banks 2-3
    a = a > t ? a : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
    a = a > t ? a : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
    a = a > t ? a : t;
    return a;
}
// value of lane `src` (wave-uniform index) in every lane: v_readlane, no LDS round trip
__device__ inline uint32_t bcast(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
}

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
// the same with the table sel8 (LDS, see SEL8_OFF) for the last eight bits
__device__ __forceinline__ uint32_t select_nth_32_lut(uint32_t w, uint32_t n) {
    typedef __attribute__((address_space(3))) const uint8_t *lds_cu8;
    uint32_t r = 0, c;
    c = __popc(w & 0xFFFFu); if (n >= c) { n -= c; r += 16; w >>= 16; }
    c = __popc(w & 0xFFu);   if (n >= c) { n -= c; r += 8;  w >>= 8; }
    return r + reinterpret_cast<lds_cu8>((uint32_t)SEL8_OFF)[8u * (w & 0xFFu) + n];
}
// a * b + c for a, b < 2^24 as v_mad_u32_u24.  The compiler turns plain index arithmetic of this shape -- and __umul24(a, b) + c as well -- into
// v_mad_u64_u32 with a 64-bit result nobody reads: a register pair more, and 3.37 instead of 3.10 issue cycles at three waves per SIMD
// (tools/experiments/valu_rate.hip; the 64-bit and multiply instructions are NOT quarter rate on gfx950: v_mul_lo_u32, v_mul_hi_u32, v_lshl_add_u64 3.10).
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// x / Tq for a hash position x < 64 Tq, Tq = 16 np, np in {1, 3, 5, 7, 9}: (x >> 4) / np by a 16-bit reciprocal (rcp = 65536 / np + 1:
// exact while (x >> 4) (rcp np - 65536) < 65536, i.e. for all x < 2^16).  One 24-bit multiply and two shifts instead of the 64-bit
// multiply-add pair the compiler made of __umulhi with a scalar operand.
__device__ __forceinline__ uint32_t div_tq(uint32_t x, uint32_t rcp) { return __umul24(x >> 4, rcp) >> 16; }
// x / TILE_T for x < 20 000 (a tile-relative offset): one 24-bit multiply and a shift (3641 / 2^19 = 1 / 144.0002; checked for every x below 20 000)
static_assert(TILE_T == 144, "div_tile_t's reciprocal is that of 144");
__device__ __forceinline__ uint32_t div_tile_t(uint32_t x) { return __umul24(x, 3641u) >> 19; }
// bits [0, v) of a word: none for v <= 0, all for v >= 32: v_med3_i32, a 64-bit shift (1 << 32 leaves a zero low word) and a decrement -- three instructions
// (rounds 2-5 selected between a 32-bit shift's result and all ones: five; the dense phase forms fifteen of these masks per tile)
__device__ __forceinline__ uint32_t bits_below(int v) {
    const int t = v < 0 ? 0 : (v > 32 ? 32 : v);
    return (uint32_t)(1ull << (uint32_t)t) - 1u;
}
// heads of a raw lane before its flag word g (0 .. 4), from the lane's cum word (byte k = heads in words 0 .. k): byte g - 1, or 0 for g = 0 --
// one v_perm_b32 whose selector picks a zero byte for g = 0
__device__ __forceinline__ uint32_t heads_before_word(uint32_t cum, uint32_t g) {
    return __builtin_amdgcn_perm(cum, 0u, g + 0x0C0C0C03u); // selector bytes 1-3: 0x0C = constant zero; byte 0: 3 -> zero (src1), 4 + k -> byte k of cum
}
// flag word of a raw lane that holds its n-th run head (0-based, n < heads of the lane): number of cum bytes <= n (the bytes ascend)
__device__ __forceinline__ uint32_t word_of_head(uint32_t cum, uint32_t n) {
    uint32_t g = 0;
    asm("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
        "v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
        : "+v"(g)
        : "v"(n), "v"(cum)
        : "vcc");
    return g;
}
// Seed look-ups of the hot loop.  The two 2 KiB tables sit at LDS byte offsets 0 (IN pairs) and 2048 (OUT
// pairs); the kernel has no static LDS, so the dynamic region starts at 0 (checked at kernel entry).  The byte
// offset of a base's entries is formed ONCE, when the base enters the window, by one v_lshlrev_b32_sdwa (byte
// select + x8); its IN pair is read right away and its OUT pair l positions later with the same offset
// register and the table base in the ds_read immediate.  (Reading both pairs at once would be one LDS
// instruction fewer, but the 4-register result tuple then stays pinned for ~40 positions: 256 VGPRs + spills.)
template <int BYTE>
__device__ __forceinline__ uint32_t byte_x8(uint32_t w) {
    uint32_t off;
    if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(3u), "v"(w));
    if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(3u), "v"(w));
    if constexpr (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(3u), "v"(w));
    if constexpr (BYTE == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(3u), "v"(w));
    return off;
}
template <int TABLE_OFF>
__device__ __forceinline__ uint2 seed_pair(uint32_t off) { // ds_read_b64 v, off offset:TABLE_OFF
#ifdef HX_NOLDS // (tools/experiments/hash_loop_bench.hip only: the loop without its table look-ups)
    return make_uint2(off, off);
#endif
    typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
    const unsigned long long v = *reinterpret_cast<lds_cu64 *>(off + TABLE_OFF);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ uint2 tab_in(const uint2 *tab, uint32_t c) { return tab[c]; }
__device__ __forceinline__ uint2 tab_out(const uint2 *tab, uint32_t c) { return tab[256 + c]; }

// ------------------------------------------------------------------------------------------------
// Hash loop, compile-time l.  Lane q owns hash positions [Tq*q, Tq*(q+1)), Tq = 16*np (np odd: the lane
// stride is then an odd number of 16 B pieces and the piece reads are bank-conflict free), of the byte array D
// (LDS).  Branch-free per position:
//   hv = min(fh, rh); hit = hv <= bound; cap = hit ? hv : cap; bits = bits<<1 | hit; roll.
// Everything the dense phase needs stays in REGISTERS: `raw[w]` = hit bits of positions 32w .. 32w+31 of the lane,
// `caps[g]` = hash of the last hit of the 16-position piece g.  (Round 1 kept both in LDS plus one LDS address per base
// of the window in registers: 256 VGPRs and 19 KB of LDS per wave, i.e. two waves per SIMD.  With the next tile's
// staging loads issued after the hash loop instead of before it, the address history fits three waves per SIMD.)
// ------------------------------------------------------------------------------------------------
// One position of the hot loop.  The hit test, the capture of the hit's hash and the hit bit are three
// VALU instructions chained through VCC (compare -> select -> add-with-carry shifts the bit in).
__device__ __forceinline__ void hit_track(uint32_t hv, uint32_t bound, uint32_t &cap, uint32_t &bits) {
#ifdef HX_NOTRACK // (tools/experiments/hash_loop_bench.hip only: the loop without compare / select / add-with-carry)
    bits ^= hv;
    return;
#endif
    asm("v_cmp_ge_u32_e32 vcc, %2, %3\n\t"
        "v_cndmask_b32_e32 %0, %0, %3, vcc\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "+v"(cap), "+v"(bits)
        : "s"(bound), "v"(hv)
        : "vcc");
}
// hit bits of a finished piece -> raw[]: `bits` holds position 32w at its highest used bit
template <int P>
__device__ __forceinline__ void close_piece(uint32_t &bits, uint32_t (&raw)[5]) {
    if constexpr (P % 32 == 15) raw[P / 32] = __builtin_bitreverse32(bits) >> 16; // first half of the word (final if the loop ends here)
    if constexpr (P % 32 == 31) {
        raw[P / 32] = __builtin_bitreverse32(bits);
        bits = 0;
    }
}

// Step s of the lane's stream (s = 0 .. T+L-1, all compile-time): base s enters the window; for s >= L the
// l-mer at position p = s - L is complete, so it is tested and then rolled forward with OUT[p], IN[s].
// W[] holds the lane's stream as dwords (piece j = W[4j .. 4j+3], read one piece ahead of its first use); only the
// ~12 dwords between the outgoing and the incoming base are live at any time.
#ifdef HX_PIECE4 // (tools/experiments/hash_stream_bench.hip: the loop below with P4 forced on)
constexpr bool HX_P4_DEFAULT = true;
#else
constexpr bool HX_P4_DEFAULT = false;
#endif
// P4: the lane's stream comes from GLOBAL memory (stream_minimizer_kernel): four pieces = 64 bytes per lane at a time, so that a lane's loads of one 128-byte
// line follow each other (16 bytes at a time the strided access bounds the loop, profiles/r04_hash_loop_without_tile_buffer.txt)
template <int L, int T, int LA, bool W2, bool P4, int S>
__device__ __forceinline__ void hash_steps(const uint4 *src, uint32_t (&W)[4 * ((T + L + 15) / 16)], uint32_t (&A)[T + L],
                                           uint2 (&EI)[T + L], uint2 (&EO)[T], uint32_t &fh, uint32_t &rh, uint32_t (&caps)[NPC], uint32_t (&raw)[5],
                                           uint32_t &bits, uint32_t bound, int np) {
    if constexpr (S < T + L - 1) {
        if constexpr (P4) {
            if constexpr (S % 64 == 0) {
#pragma unroll
                for (int pj = 0; pj < 4; pj++) {
                    constexpr int PB = S / 16 + 4;
                    if (PB + pj < (T + L + 15) / 16) {
                        const uint4 v = src[PB + pj];
                        W[4 * (PB + pj)] = v.x; W[4 * (PB + pj) + 1] = v.y; W[4 * (PB + pj) + 2] = v.z; W[4 * (PB + pj) + 3] = v.w;
                    }
                }
            }
        } else if constexpr (S % 16 == 0 && S / 16 + 2 < (T + L + 15) / 16) { // piece S/16+2: its first base enters >= 16 steps from now
            const uint4 v = src[S / 16 + 2];
            W[4 * (S / 16 + 2)] = v.x; W[4 * (S / 16 + 2) + 1] = v.y; W[4 * (S / 16 + 2) + 2] = v.z; W[4 * (S / 16 + 2) + 3] = v.w;
        }
        // seeds are fetched LA steps ahead of their use, a group of LA at a time
        if constexpr (S % LA == 0) {
            __builtin_amdgcn_sched_barrier(0); // keep the scheduler from hoisting later groups' look-ups (register pressure)
            constexpr int G = S + LA; // first base of the group entering LA steps from now
#define S2K_IN(J)                                                        \
    if constexpr (J < LA && G + J < T + L - 1) {                          \
        A[G + J] = byte_x8<(G + J) & 3>(W[(G + J) >> 2]);                  \
        if constexpr (W2 && (G + J) % 2 == 0 && G + J + 1 < L)             \
            EI[G + J] = seed_pair<WARM_OFF>(A[G + J]); /* first base of a warm-up pair */ \
        else                                                               \
            EI[G + J] = seed_pair<0>(A[G + J]);                            \
    }
            S2K_IN(0) S2K_IN(1) S2K_IN(2) S2K_IN(3) S2K_IN(4) S2K_IN(5) S2K_IN(6) S2K_IN(7)
#undef S2K_IN
            constexpr int H = S + LA - L; // first base of the group leaving LA steps from now
#define S2K_OUT(J) \
    if constexpr (J < LA && H + J >= 0 && H + J < T - 1) EO[H + J] = seed_pair<2048>(A[H + J]); // offset formed l positions ago
            S2K_OUT(0) S2K_OUT(1) S2K_OUT(2) S2K_OUT(3) S2K_OUT(4) S2K_OUT(5) S2K_OUT(6) S2K_OUT(7)
#undef S2K_OUT
        }
        if constexpr (S < L) { // warm-up: first l-mer of the lane (src/nthash_hpc.rs:138-150,158-174), in closed form: base i enters
            // as rotl(h, l-1-i) / rotl(rc-entry, i) -- EI.y holds rotl(rc, l-1), so rotr by l-1-i --, two bases per three-input XOR
            // (three instructions per strand and pair instead of four for two rolling steps)
            if constexpr (W2) { // Horner, two bases per step: the pair's first base comes from the pre-rotated table (see S2K_IN)
                if constexpr (S % 2 == 1) {
                    fh = xor3(S == 1 ? 0u : __builtin_rotateleft32(fh, 2), EI[S - 1].x, EI[S].x);
                    rh = xor3(S == 1 ? 0u : __builtin_rotateright32(rh, 2), EI[S - 1].y, EI[S].y);
                } else if constexpr (S == L - 1) { // odd l: the last base alone
                    fh = __builtin_rotateleft32(fh, 1) ^ EI[S].x;
                    rh = __builtin_rotateright32(rh, 1) ^ EI[S].y;
                }
            } else if constexpr (S % 2 == 1) {
                fh = xor3(fh, __builtin_rotateleft32(EI[S - 1].x, (L - S) & 31), __builtin_rotateleft32(EI[S].x, (L - 1 - S) & 31));
                rh = xor3(rh, __builtin_rotateright32(EI[S - 1].y, (L - S) & 31), __builtin_rotateright32(EI[S].y, (L - 1 - S) & 31));
            } else if constexpr (S == L - 1) { // odd l: the last base alone, rotation 0
                fh ^= EI[S].x;
                rh ^= EI[S].y;
            }
        } else {
            constexpr int P = S - L;
            const uint32_t hv = fh < rh ? fh : rh;                              // canonical (src/nthash_hpc.rs:276)
            hit_track(hv, bound, caps[P / CAPP], bits);                         // hv <= bound (src/nthash_hpc.rs:277 / src/lib.rs:228)
            fh = xor3(__builtin_rotateleft32(fh, 1), EO[P].x, EI[S].x);         // src/nthash_hpc.rs:245
            rh = xor3(__builtin_rotateright32(rh, 1), EO[P].y, EI[S].y);        // src/nthash_hpc.rs:247-249
            if constexpr (P % 16 == 15) {
                close_piece<P>(bits, raw);
                if (P / 16 + 1 >= np) return; // wave-uniform: the (compacted) tile is shorter than 144 bases per lane
            }
        }
        hash_steps<L, T, LA, W2, P4, S + 1>(src, W, A, EI, EO, fh, rh, caps, raw, bits, bound, np);
    } else { // last position: test only, nothing left to roll into
        const uint32_t hv = fh < rh ? fh : rh;
        hit_track(hv, bound, caps[(T - 1) / CAPP], bits);
        close_piece<T - 1>(bits, raw);
    }
}

template <int L, int LA, bool W2, bool P4 = HX_P4_DEFAULT>
__device__ __forceinline__ void hash_loop_static(const uint8_t *D, uint32_t bound, int lane, int np, uint32_t (&caps)[NPC],
                                                 uint32_t (&raw)[5]) {
    constexpr int T = TILE_T;
    static_assert(L > LA && L >= 4 && L <= 32, "the static schedule looks LA bases ahead and keeps one address register per base of the l-mer");
    static_assert((CAPP == 16 || CAPP == 8) && LA <= 8, "capture pieces of 16 or 8 positions");
    uint32_t W[4 * ((T + L + 15) / 16)];
    const uint4 *src = reinterpret_cast<const uint4 *>(D + 16 * np * lane);
    {
        const uint4 v0 = src[0], v1 = src[1];
        W[0] = v0.x; W[1] = v0.y; W[2] = v0.z; W[3] = v0.w;
        W[4] = v1.x; W[5] = v1.y; W[6] = v1.z; W[7] = v1.w;
        if constexpr (P4) {
            const uint4 v2 = src[2], v3 = src[3];
            W[8] = v2.x; W[9] = v2.y; W[10] = v2.z; W[11] = v2.w;
            W[12] = v3.x; W[13] = v3.y; W[14] = v3.z; W[15] = v3.w;
        }
    }
    uint32_t A[T + L]; // LDS byte offset of each base's table entries: formed when the base enters, reused when it leaves
    uint2 EI[T + L], EO[T];
    A[0] = byte_x8<0>(W[0]); A[1] = byte_x8<1>(W[0]); A[2] = byte_x8<2>(W[0]); A[3] = byte_x8<3>(W[0]);
    if constexpr (LA > 4) { A[4] = byte_x8<0>(W[1]); A[5] = byte_x8<1>(W[1]); A[6] = byte_x8<2>(W[1]); A[7] = byte_x8<3>(W[1]); }
#pragma unroll
    for (int i = 0; i < LA; i++) EI[i] = (W2 && i % 2 == 0 && i + 1 < L) ? seed_pair<WARM_OFF>(A[i]) : seed_pair<0>(A[i]);
    uint32_t fh = 0, rh = 0, bits = 0;
    hash_steps<L, T, LA, W2, P4, 0>(src, W, A, EI, EO, fh, rh, caps, raw, bits, bound, np);
}

// Same loop for a run-time l (1 .. 64; only l values without a static instantiation come here), unrolled like the static one.  The
// outgoing base of position p is byte p of the lane's stream: static offsets, 16-byte pieces, one SDWA instruction per base as above.
// The incoming base is byte p + l: the same stream read as dwords from the dword-aligned offset l & ~3 and funnel-shifted by l & 3
// bytes (one v_alignbyte_b32 per four bases), after which its byte offsets are static too.  What the run-time l costs against a
// compile-time one: the second SDWA instruction per position (the static loop keeps a base's table offset in a register from the step
// it enters to the step it leaves -- l registers, l a constant) and the warm-up as a rolled loop: 10.3 instead of 9 vector
// instructions per position.  (Round 3 fetched both bases byte by byte from LDS: 1.4x the static loop's time.)
__device__ __forceinline__ uint32_t byte_x8_sel(uint32_t w, int b) { // b is a constant once the loops below are unrolled
    switch (b & 3) {
    case 0: return byte_x8<0>(w);
    case 1: return byte_x8<1>(w);
    case 2: return byte_x8<2>(w);
    default: return byte_x8<3>(w);
    }
}
__device__ __forceinline__ void hash_loop_dynamic(const uint8_t *D, const uint2 *__restrict__ tab, uint32_t bound, int lane,
                                                  uint32_t l, int np, uint32_t (&caps)[NPC], uint32_t (&raw)[5]) {
    constexpr int T = TILE_T, NP = TILE_T / 16;
    const uint8_t *q = D + 16 * np * lane;
    uint32_t fh = 0, rh = 0;
    for (uint32_t i = 0; i < l; i++) { // first l-mer of the lane (src/nthash_hpc.rs:138-150,158-174)
        uint2 ti = tab_in(tab, q[i]);
        fh = __builtin_rotateleft32(fh, 1) ^ ti.x;
        rh = __builtin_rotateright32(rh, 1) ^ ti.y;
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(q);
    const uint32_t *rin = reinterpret_cast<const uint32_t *>(q + (l & ~3u));
    const uint32_t lb = l & 3u;
    uint32_t Wo[4 * NP], Wi[4 * NP + 4], R[4 * NP + 5];
    auto load_out = [&](int g) { // piece g of the outgoing stream
        const uint4 v = src[g];
        Wo[4 * g] = v.x; Wo[4 * g + 1] = v.y; Wo[4 * g + 2] = v.z; Wo[4 * g + 3] = v.w;
    };
    auto load_in = [&](int g) { // piece g of the incoming stream: dwords 4g+1 .. 4g+4 of the aligned read, shifted into place
#pragma unroll
        for (int j = 1; j <= 4; j++) R[4 * g + j] = rin[4 * g + j];
#pragma unroll
        for (int j = 0; j < 4; j++) Wi[4 * g + j] = __builtin_amdgcn_alignbyte(R[4 * g + j + 1], R[4 * g + j], lb);
    };
    R[0] = rin[0];
    load_out(0);
    load_in(0);
    load_out(1);
    load_in(1);
    uint2 eo = seed_pair<2048>(byte_x8<0>(Wo[0])), ei = seed_pair<0>(byte_x8<0>(Wi[0])); // seeds of position 0
    uint32_t bits = 0;
#pragma unroll
    for (int g = 0; g < NP; g++) {
        if (g < np) { // wave-uniform: the (compacted) tile may be shorter than 144 bases per lane
            if (g + 2 < NP) load_out(g + 2);
            if (g + 2 < NP) load_in(g + 2);
#pragma unroll
            for (int h = 0; h < 16 / CAPP; h++) {
                uint32_t cap = 0;
#pragma unroll
                for (int i = 0; i < CAPP; i++) {
                    const int P = 16 * g + CAPP * h + i;
                    __builtin_amdgcn_sched_barrier(0); // (as in the static loop: keep later positions' look-ups where they are)
                    uint2 eo_n = eo, ei_n = ei;
                    if (P + 1 < T) { // the next position's seeds, one step ahead of their use
                        eo_n = seed_pair<2048>(byte_x8_sel(Wo[(P + 1) >> 2], P + 1));
                        ei_n = seed_pair<0>(byte_x8_sel(Wi[(P + 1) >> 2], P + 1));
                    }
                    const uint32_t hv = fh < rh ? fh : rh;                    // canonical (src/nthash_hpc.rs:276)
                    hit_track(hv, bound, cap, bits);                          // hv <= bound (src/nthash_hpc.rs:277 / src/lib.rs:228)
                    fh = xor3(__builtin_rotateleft32(fh, 1), eo.x, ei.x);     // src/nthash_hpc.rs:245
                    rh = xor3(__builtin_rotateright32(rh, 1), eo.y, ei.y);    // src/nthash_hpc.rs:247-249
                    eo = eo_n;
                    ei = ei_n;
                }
                caps[(16 / CAPP) * g + h] = cap;
            }
            if (g % 2 == 0) raw[g / 2] = __builtin_bitreverse32(bits) >> 16;
            else {
                raw[g / 2] = __builtin_bitreverse32(bits);
                bits = 0;
            }
        }
    }
}

// Everything below is force-inlined into the kernel so that the LDS operands keep their address space
// (a generic pointer costs a 64-bit add, a null compare and a select per table lookup).
template <int L, bool HPC>
__device__ __forceinline__ void hash_stage(const uint8_t *D, const uint2 *tab, uint32_t bound, int lane, uint32_t l, int np,
                                           uint32_t (&caps)[NPC], uint32_t (&raw)[5]) {
#pragma unroll
    for (int g = 0; g < NPC; g++) caps[g] = 0;
#pragma unroll
    for (int w = 0; w < 5; w++) raw[w] = 0;
    if constexpr (L > 0) hash_loop_static<L, HPC ? HPC_LA : REG_LA, HPC && S2K_WARM2 != 0>(D, bound, lane, np, caps, raw);
    else hash_loop_dynamic(D, tab, bound, lane, l, np, caps, raw);
}

// Tiles are dealt DYNAMICALLY.  The SIMD's issue arbiter serves the oldest wave first: with a static deal (every wave the
// same number of tiles) the first-launched wave of each SIMD ran out of tiles 3.7 ms before the last-launched one, and the
// SIMDs spent the last third of the kernel with two, then one wave (tools/tail_spread.sh, profiles/r02_zz_tail_spread.txt).
// A wave takes its first three tiles statically (the software pipeline is three deep) and every later one from one of
// TILE_CURSORS cursors in global memory, two iterations before it processes it; when its cursor runs dry the wave is done
// (moving on to a cursor that is not was measured again in round 3: -1 %).  Which cursor a wave draws from is decorrelated from its
// slot in the block (see cur_g).  (ONE cursor for all waves was measured first: 1.085 M atomics on one address are serialised at ~12 ns
// each and the kernel took 13.2 ms; 64 addresses, 128 B apart, are not a bottleneck.)
// (s_setprio per phase -- dense phase high and hash loop low, and the reverse -- moved the kernel by <= 1 % either way: not used.)

// Phase stamps (cycles per phase, flushed once per wave to one of 64 shards) exist only in builds with -DS2K_PROFILE
// (tools/phases.sh).  The 16 accumulators of a wave live in LDS (128 B per wave behind the waves' buffers; lane 0 adds to them):
// in registers (rounds 2-3) they cost 32 VGPRs, i.e. the third wave per SIMD -- and what a phase costs at two waves per SIMD is
// not what it costs at three.  A stamp is ~8 instructions and drains lgkmcnt; 16 of them per tile perturb the kernel by ~3 %.
#ifdef S2K_PROFILE
#define S2K_STAMP(i)                                                                     \
    do {                                                                                 \
        if (sem.dbg_skip & 8) {                                                          \
            uint64_t _n = __builtin_amdgcn_s_memtime();                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
            if (lane == 0) ph[i] += _n - stamp;                                          \
            stamp = _n;                                                                  \
        }                                                                                \
    } while (0)
#else
#ifdef S2K_ISA_MARKS // analysis builds only (tools/isa/phase_hist.py): the phase boundaries as comments in the ISA
#define S2K_STAMP(i) do { (void)ph; (void)stamp; asm volatile("; S2K_PHASE %0" ::"i"(i)); } while (0)
#else
#define S2K_STAMP(i) do { (void)ph; (void)stamp; } while (0)
#endif
#endif

// ------------------------------------------------------------------------------------------------
// Hpc pre-stage: in-place run-head compaction of the staged tile.  Returns R_t (run heads owned by
// the tile) and leaves D[0..R_t) = head bytes, D[R_t..R_t+halo_n) = following heads, S.row / S.sd /
// S.halo_pos for the back-map.
//
// Run-head flags are kept in NATURAL order: bit i of the lane's 144-bit mask (five words) <-> raw byte i of the
// lane's chunk.  Read starts are forced run heads (every read starts a new run, src/nthash_hpc.rs:138-150 runs per
// read); they are OR-ed into the masks, the staged bytes are never modified -- any byte value is fine.
// ------------------------------------------------------------------------------------------------
template <class WL>
__device__ __forceinline__ uint32_t hpc_compact(uint8_t *D, WL &S, const uint8_t *__restrict__ bases,
                                                const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
                                                uint64_t t0, uint32_t tile_len, uint32_t r0, uint32_t r1, uint32_t l,
                                                int lane, uint32_t &halo_n_out, uint64_t bpos0,
                                                uint32_t prev_byte0, bool forced0, const Sem &sem, ph_ptr_t ph,
                                                uint64_t &stamp) {
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    // 1. read starts strictly inside the tile -> forced run heads, OR-ed into S.row (words 0-4, cleared by the caller before the
    //    tile was staged).  bpos0 = read_off[r0 + 1 + lane] was fetched ahead.
    {
        uint64_t sp = bpos0;
        for (uint64_t c0 = 0;; c0 += 64) {
            if (sp > t0 && sp < t0 + tile_len) {
                const uint32_t rel = (uint32_t)(sp - t0), o = div_tile_t(rel), i = rel - __umul24(o, (uint32_t)TILE_T);
                atomicOr(&S.row[o][i >> 5], 1u << (i & 31u));
            }
            if ((uint64_t)r0 + 1 + c0 + 64 > (uint64_t)r1) break; // wave-uniform: all starts up to r1 covered
            const uint64_t ri = (uint64_t)r0 + 1 + c0 + 64 + lane;
            sp = ri <= n_reads ? read_off[ri] : ~0ull;
            // (> 63 read starts in a tile: rare.)  Waited for HERE: pending across the back edge, the load would put a vmcnt(0)
            // at the loop header, i.e. in front of every tile's compaction, and drain the previous tile's stores (see the kernel)
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    }
    wave_sync();
    S2K_STAMP(13); // compaction: read-start marks
    // 2. lane chunk -> registers
    uint32_t c[36];
    const uint32_t lane_off = __umul24((uint32_t)TILE_T, (uint32_t)lane);
    const uint4 *src = reinterpret_cast<const uint4 *>(D + lane_off);
#pragma unroll
    for (int p = 0; p < 9; p++) {
        uint4 v = src[p];
        c[4 * p] = v.x; c[4 * p + 1] = v.y; c[4 * p + 2] = v.z; c[4 * p + 3] = v.w;
    }
    uint32_t fmk[5]; // forced heads
    {
        const uint2 *rr = reinterpret_cast<const uint2 *>(&S.row[lane][0]);
        const uint2 r01 = rr[0], r23 = rr[1];
        fmk[0] = r01.x; fmk[1] = r01.y; fmk[2] = r23.x; fmk[3] = r23.y;
        fmk[4] = S.row[lane][4];
    }
    uint32_t prevw;
    if (lane == 0) {
        if (forced0) fmk[0] |= 1u; // the tile starts a read
        prevw = prev_byte0 << 24;
    } else {
        prevw = (uint32_t)D[lane_off - 1] << 24;
    }
    const uint32_t last_raw = bcast(c[35] >> 24, 63); // last raw byte of a full tile
    // the 128 staged look-ahead bytes serve the first round of the run-head search after the tile; read them
    // before the buffer is compacted in place (lanes 0..31, 4 bytes each)
    const uint32_t la_a = D[TILE_BASES + lane], la_b = D[TILE_BASES + 64 + lane]; // (one byte per lane and round)
    const int vb = (int)tile_len - (int)lane_off;      // valid bytes in this lane's chunk (may be <=0 or >=144)
    const bool partial = tile_len < (uint32_t)TILE_BASES;
    // 3. pass 1: SWAR "differs from its predecessor" per byte -> four flags per dword -> natural-order masks
    uint32_t pair7 = 0;
#pragma unroll
    for (int d = 0; d < 36; d++) {
        const uint32_t cur = c[d];
        const uint32_t prv = __builtin_amdgcn_alignbyte(cur, d == 0 ? prevw : c[d - 1], 3); // predecessor of every byte
        const uint32_t x = cur ^ prv;
        const uint32_t y = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; // 0x80 per byte that differs
        // the four flags as a nibble: one dot product with the weights 1, 2, 4, 8 (x 128: the flags sit at bit 7); the odd dword of a
        // pair ADDS its nibble above the even one's (weights x 16 and the accumulator operand), so that two dwords cost one shift-or
        if ((d & 1) == 0) {
            pair7 = __builtin_amdgcn_udot4(y, 0x08040201u, 0u, false);
        } else {
            pair7 = __builtin_amdgcn_udot4(y, 0x80402010u, pair7, false);
            const int sh = 8 * ((d & 7) >> 1) - 7; // (constants once the loop is unrolled)
            // (v_lshl_or_b32: the opaque copy keeps the compiler from re-associating the ors into a tree of separate shifts and ors.
            // The instruction itself must stay the compiler's: on gfx950 a VALU instruction may read a v_dot4 result only three wait
            // states after it, which the hazard recognizer cannot arrange for an instruction hidden in inline asm -- measured: wrong flags)
            if (sh >= 0) {
                fmk[d >> 3] = (pair7 << (sh >= 0 ? sh : 0)) | fmk[d >> 3];
                asm("" : "+v"(fmk[d >> 3]));
            }
            else fmk[d >> 3] |= pair7 >> 7;
        }
    }
    if (partial) { // only the stream's last tile: bytes past the end are no run heads
#pragma unroll
        for (int g = 0; g < 5; g++) {
            const int v = vb - 32 * g;
            fmk[g] &= v >= 32 ? 0xFFFFFFFFu : (v <= 0 ? 0u : ((1u << v) - 1u));
        }
    }
    const uint32_t p0 = __popc(fmk[0]), p1 = p0 + __popc(fmk[1]), p2 = p1 + __popc(fmk[2]), p3 = p2 + __popc(fmk[3]);
    uint32_t cnt = p3 + __popc(fmk[4]);
    uint32_t incl = wave_incl_scan(cnt, lane);
    uint32_t base = incl - cnt;
    const uint32_t R = bcast(incl, 63);
    { // the lane's row (HpcLds), three 8-byte stores; lanes 0 and 1 also leave the sentinels behind the last lane
        uint2 *rw = reinterpret_cast<uint2 *>(&S.row[lane][0]);
        rw[0] = make_uint2(fmk[0], fmk[1]);
        rw[1] = make_uint2(fmk[2], fmk[3]);
        rw[2] = make_uint2(fmk[4] | (base << 16), p0 | (p1 << 8) | (p2 << 16) | (p3 << 24));
        if (lane < N_ROWS - 64) *reinterpret_cast<uint2 *>(&S.row[64 + lane][ROW_PFX]) = make_uint2(R << 16, 0u);
        // directory: the raw lane that owns run head 64 m, for every m the lane's range [base, base + cnt) holds (at most three)
        if (cnt)
            for (uint32_t m = (base + (1u << SD_SHIFT) - 1u) >> SD_SHIFT; (m << SD_SHIFT) < base + cnt; m++) S.sd[m] = (uint8_t)lane;
    }
    // all lanes hold their raw chunk in registers now -> the buffer may be overwritten in place
    wave_sync();
    S2K_STAMP(14); // compaction: chunk load + flags + scan
    // 4. pass 2: every byte is stored at slot (#heads at or before it) - 1; bytes of one run carry the same value, so
    //    only the slot matters and nothing is predicated.  Slot -1 of lane 0 lands on the scratch byte D[-1].
    //    (Byte stores on purpose: a ds_write_b32 at an address that is not a multiple of 4 is executed one lane per cycle --
    //    64 cycles per wave instruction against ~4 for ds_write_b8, tools/experiments/lds_rate.hip -- so packing the heads
    //    of a dword with v_perm_b32 and storing them with one unaligned dword store was 2x slower than this.)
#if S2K_PASS2_ACC
    // 4. pass 2 (round 5): eight raw bytes (two dwords) per step.  Their flag BYTE indexes sel8 -- whose entry, the positions of the byte's set bits in
    //    order with 0x0C behind them, is at the same time the pair of v_perm_b32 selectors that packs the run heads of the eight bytes to the low end of
    //    a 64-bit value.  Shifted to the lane's fill level that value spans up to three dwords of the (zeroed) buffer, all at ALIGNED addresses:
    //    18 steps of 14 vector + 3 LDS instructions per lane instead of 144 byte stores whose data-dependent slots collided in the banks (a byte
    //    store cost 5.5 units of a plain vector instruction's 1.0 where it ran, 3.1 without the conflicts: profiles/r05_instruction_costs.txt; one
    //    dword per step through a 16-entry table, this round's first version: 22 + 4 per eight bytes, profiles/r05_ab_pass2_pairs.txt).  The dwords two
    //    lanes share (a lane's run heads begin and end at any byte) are merged in one of two ways:
    //    DENSE tiles (>= 0.6 run heads per base; uniform-random ACGT has 0.75): the first two dwords are OR-ed in place every step (ds_or_b32) -- partial
    //      states of a dword are subsets of its final state, so storing early and often is harmless, and shared dwords merge by themselves -- and only
    //      the bits beyond them are carried to the next step;
    //    SPARSE tiles (HiFi-like reads, 0.29): with few run heads per step the same dword would be OR-ed over and over (7.7 -> 8.9 ms per 15 Gbp:
    //      atomics on one address serialise), so a dword is written plainly by the lane that COMPLETES it, the moment it does -- with zeros where other
    //      lanes' bytes go --, every other store of a step goes to a per-lane dump slot (the hit list's space, dead during the compaction), the incomplete
    //      dword is carried whole, and what a lane holds of a dword it does not complete is OR-ed in after the loop.  (Dense tiles that way: +1 % kernel
    //      time, profiles/r05_ab_pass2_sparse.txt.)
    {
        typedef __attribute__((address_space(3))) const unsigned long long *lds_cu64;
        uint4 *zp = reinterpret_cast<uint4 *>(D + lane_off);
#pragma unroll
        for (int p = 0; p < 9; p++) zp[p] = make_uint4(0, 0, 0, 0); // (every lane holds its raw chunk in registers)
        asm volatile("" ::: "memory");
        const uint32_t(&fmk_)[5] = fmk;
        auto pass2 = [&](auto sparse_c) {
            constexpr bool SPARSE = decltype(sparse_c)::value;
            // (opaque copies of the masks per instantiation: what the two have in common -- table look-ups, packed dwords -- is otherwise hoisted above
            // the branch and held across it: 118 -> 143 registers)
            uint32_t fm[5] = {fmk_[0], fmk_[1], fmk_[2], fmk_[3], fmk_[4]};
#pragma unroll
            for (int g = 0; g < 5; g++) asm volatile("" : "+v"(fm[g]));
            uint32_t waddr = (uint32_t)(uintptr_t)(lds_u8 *)D + (base & ~3u); // aligned LDS address of the dword being filled
            uint32_t fill8 = 8u * (base & 3u), lo = 0;                        // bits of it that lie before this lane's next run head; this lane's share of them
            // (sparse tiles: a dump slot of two dwords, so that "offset:4" works on it as on the buffer; lanes i and i + 32 share one -- nobody reads it)
            const uint32_t dump = (uint32_t)(uintptr_t)(lds_u8 *)reinterpret_cast<uint8_t *>(&S.list[0]) + 8u * ((uint32_t)lane & 31u);
            constexpr int GRPQ = 9; // table entries fetched at a time: one LDS round trip per group instead of one per step
#pragma unroll
            for (int q0 = 0; q0 < 18; q0 += GRPQ) {
                unsigned long long e[GRPQ];
                uint32_t ix[GRPQ];
#pragma unroll
                for (int qq = 0; qq < GRPQ; qq++) {
                    const int q = q0 + qq;
                    ix[qq] = byte_x8_sel(fm[q >> 2], q & 3); // 8 x the flag byte of raw bytes 8 q .. 8 q + 7 (one SDWA instruction)
                    e[qq] = *reinterpret_cast<lds_cu64>((uint32_t)SEL8_OFF + ix[qq]);
                }
#pragma unroll
                for (int qq = 0; qq < GRPQ; qq++) {
                    const int q = q0 + qq;
                    const uint32_t plo = __builtin_amdgcn_perm(c[2 * q + 1], c[2 * q], (uint32_t)e[qq]);         // run heads 1-4 of the eight bytes
                    const uint32_t phi = __builtin_amdgcn_perm(c[2 * q + 1], c[2 * q], (uint32_t)(e[qq] >> 32)); // ... 5-8 (selector 0x0C: a zero byte)
                    const unsigned long long win = (((unsigned long long)phi << 32) | plo) << fill8;             // fill8 <= 24
                    const uint32_t w0 = lo | (uint32_t)win, w1 = (uint32_t)(win >> 32);
                    const uint32_t w2 = (phi >> 8) >> (24u - fill8);           // bits 64 .. of the window: phi >> (32 - fill8), and 0 for fill8 = 0
                    const uint32_t nb = fill8 + 8u * (uint32_t)__popc(ix[qq]); // <= 88
                    if constexpr (!SPARSE) {
                        asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4" ::"v"(waddr), "v"(w0), "v"(w1) : "memory");
                        lo = w2; // (the first two dwords are in place; only what lies beyond them is carried)
                    } else {
                        const uint32_t a0 = nb >= 32u ? waddr : dump, a1 = nb >= 64u ? waddr : dump;
                        asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3 offset:4" ::"v"(a0), "v"(a1), "v"(w0), "v"(w1) : "memory");
                        lo = nb >= 64u ? w2 : (nb >= 32u ? w1 : w0);
                    }
                    waddr += (nb >> 3) & 0xCu; // 4 x (dwords completed: 0, 1 or 2)
                    fill8 = nb & 24u;
                }
            }
            asm volatile("ds_or_b32 %0, %1" ::"v"(waddr), "v"(lo) : "memory"); // what the lane holds of a dword it did not complete / beyond the last one it OR-ed
        };
        static_assert(sizeof(S.list) >= 32 * 8, "the dump slots of the sparse variant live in the hit list");
        if (10u * R < 6u * tile_len) pass2(std::true_type{}); // (wave-uniform)
        else pass2(std::false_type{});
    }
#else
    {
        uint32_t gaddr = (uint32_t)(uintptr_t)(lds_u8 *)D + base - 1; // LDS byte address of slot -1
        auto pass2 = [&](auto partial_c) {
            constexpr bool PARTIAL = decltype(partial_c)::value;
#pragma unroll
            for (int d = 0; d < 36; d++) {
                const int g = d >> 3;
                const uint32_t hi = c[d] >> 8; // bytes 1 and 3 (the d16_hi store takes bits 16..23)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int i = (4 * d + b) & 31;
                    const uint32_t a = gaddr + __popc(fmk[g] & (i == 31 ? 0xFFFFFFFFu : ((2u << i) - 1u))); // v_and + v_bcnt(+gaddr)
                    if (!PARTIAL || 4 * d + b < vb) {
                        if (b == 0) asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(c[d]) : "memory");
                        if (b == 1) asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(hi) : "memory");
                        if (b == 2) asm volatile("ds_write_b8_d16_hi %0, %1" ::"v"(a), "v"(c[d]) : "memory");
                        if (b == 3) asm volatile("ds_write_b8_d16_hi %0, %1" ::"v"(a), "v"(hi) : "memory");
                    }
                }
                if ((d & 7) == 7) gaddr += __popc(fmk[g]);
            }
        };
        // full tiles (all but the last of the stream) take the branch-free instantiation
        if (partial) pass2(std::true_type{});
        else pass2(std::false_type{});
    }
#endif
    S2K_STAMP(15); // compaction: stores
    // 5. run heads that follow the tile: up to l of them (hash needs l-1, the end position one more); the HpcSimd
    //    tail rule looks 16 heads further for the end of the read
    const uint32_t hl = sem.tail_quirk ? l + 16 : l;
    uint32_t halo_n = 0;
    if (!partial) {
        uint32_t pb = last_raw;
        // the staged look-ahead first, ONE BYTE PER LANE and round of 64 bytes: a run head is a byte that differs from the lane before's (DPP), its index
        // among the heads one ballot and a bit count away -- a dozen instructions per round, one round as a rule (uniform ACGT: 48 run heads in 64 bytes;
        // rounds 1-4 looked at four bytes per lane with a wave scan and four guarded stores: ~65 + 30 scalar)
#pragma unroll
        for (int rnd = 0; rnd < 2; rnd++) {
            const uint64_t qr = t0 + TILE_BASES + 64 * rnd;
            if (halo_n >= hl || qr >= n_bases) break; // wave-uniform
            const uint32_t by = rnd ? la_b : la_a;
            const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)pb, (int)by, 0x138, 0xf, 0xf, false); // wave_shr:1 -- lane 0 keeps pb, the byte before the round
            const bool hd = qr + (uint64_t)lane < n_bases && by != pv; // (bytes past the end of the stream were staged as zeros: masked here)
            const uint64_t mk = __ballot(hd);
            const uint32_t idx = halo_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (hd && idx < hl) {
                D[R + idx] = (uint8_t)by;
                S.halo_pos[idx] = (uint32_t)(TILE_BASES + 64 * rnd) + (uint32_t)lane;
            }
            halo_n += (uint32_t)__popcll(mk);
            pb = bcast(by, 63);
        }
        // ... then the stream itself, 256 bytes per round (long homopolymers, sparse run heads)
        uint64_t q = t0 + TILE_BASES + 128;
        while (halo_n < hl && q < n_bases) { // wave-uniform
            const uint32_t span = 256u; // bytes examined this round
            uint64_t a = q + 4 * (uint64_t)lane;
            int nval = a >= n_bases ? 0 : (n_bases - a >= 4 ? 4 : (int)(n_bases - a));
            uint32_t wv = 0;
            if (nval == 4) wv = *reinterpret_cast<const uint32_t *>(bases + a);
            else
                for (int b = 0; b < nval; b++) wv |= (uint32_t)bases[a + b] << (8 * b);
            uint32_t pw = __shfl_up(wv, 1);
            if (lane == 0) pw = pb << 24;
            uint32_t prv = (wv << 8) | (pw >> 24);
            uint32_t x = wv ^ prv;
            uint32_t t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
            t &= nval >= 4 ? 0xFFFFFFFFu : (nval <= 0 ? 0u : ((1u << (8 * nval)) - 1u));
            uint32_t cn = __popc(t);
            uint32_t in2 = wave_incl_scan(cn, lane);
            uint32_t idx = halo_n + in2 - cn;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (t & (0x80u << (8 * b))) {
                    if (idx < hl) {
                        D[R + idx] = (uint8_t)(wv >> (8 * b));
                        S.halo_pos[idx] = (uint32_t)(a - t0) + b;
                    }
                    idx++;
                }
            }
            halo_n += bcast(in2, 63);
            pb = bcast(wv >> 24, 63);
            q += span;
        }
        if (halo_n > hl) halo_n = hl;
    }
    halo_n_out = halo_n;
    wave_sync();
    return R;
}


// Back-map of one Hpc hit (v2, see HpcLds): tile-relative raw offsets of run heads x and y = x + l (x < R; y may be one of the run heads
// that follow the tile).  Four LDS round trips, both heads side by side: directory -> {prefix, cum} of the three candidate lanes ->
// the one flag word that holds the head -> sel8.
struct BmFirst { // what the first round trip of a hit's back-map brings: the directory entries of its two run heads and the position behind the tile
    uint32_t x, yy, ox0, oy0, he;
    bool y_in;
};
template <class WL>
__device__ __forceinline__ BmFirst hpc_rawpos_first(const WL &S, uint32_t x, uint32_t l, uint32_t R, uint32_t halo_n) {
    typedef __attribute__((address_space(3))) const uint8_t *lds_cu8;
    BmFirst f;
    const uint32_t y = x + l;
    f.x = x;
    f.y_in = y < R;
    f.yy = f.y_in ? y : x;     // (a head of the tile either way; what it gives is not used when y lies behind the tile)
    const uint32_t hx = y - R; // index among the run heads that follow the tile; validated hits guarantee hx < halo_n then
    f.he = S.halo_pos[(!f.y_in && hx < halo_n) ? hx : 0];
    const uint32_t sd0 = (uint32_t)(uintptr_t)(lds_cu8)&S.sd[0];
    f.ox0 = reinterpret_cast<lds_cu8>(sd0)[x >> SD_SHIFT];
    f.oy0 = reinterpret_cast<lds_cu8>(sd0)[f.yy >> SD_SHIFT];
    return f;
}
template <class WL>
__device__ __forceinline__ void hpc_rawpos_rest(const WL &S, const BmFirst &f, uint32_t &raw_x, uint32_t &raw_e) {
    typedef __attribute__((address_space(3))) const uint8_t *lds_cu8;
    const uint32_t x = f.x, yy = f.yy, ox0 = f.ox0, oy0 = f.oy0, he = f.he;
    const bool y_in = f.y_in;
    const uint32_t row0 = (uint32_t)(uintptr_t)(lds_cu8)&S.row[0][0];
    // 2: {flags 128.. | prefix << 16, cum} of the lane the directory names and of the two after it (rows 64, 65 are sentinels)
    typedef __attribute__((address_space(3))) const unsigned long long *lds_cu64;
    typedef __attribute__((address_space(3))) const uint32_t *lds_cu32;
    auto ld2 = [](uint32_t a) { // ds_read_b64
        const unsigned long long v = *reinterpret_cast<lds_cu64>(a);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    const uint32_t rx = mad24(ox0, 4u * ROW_W, row0), ry = mad24(oy0, 4u * ROW_W, row0);
    const uint2 ax = ld2(rx + 4 * ROW_PFX), bx = ld2(rx + 4 * (ROW_PFX + ROW_W)), cx = ld2(rx + 4 * (ROW_PFX + 2 * ROW_W));
    const uint2 ay = ld2(ry + 4 * ROW_PFX), by = ld2(ry + 4 * (ROW_PFX + ROW_W)), cy = ld2(ry + 4 * (ROW_PFX + 2 * ROW_W));
    const bool sx = (bx.x >> 16) <= x, sy = (by.x >> 16) <= yy;
    uint32_t rox = sx ? rx + 4 * ROW_W : rx, roy = sy ? ry + 4 * ROW_W : ry; // LDS address of the owner's row
    uint2 px = sx ? bx : ax, py = sy ? by : ay;
    if (__ballot((cx.x >> 16) <= x || (cy.x >> 16) <= yy)) { // (wave-uniform, rare: a raw lane with fewer than 64 run heads -- long homopolymers: walk on)
        while ((*reinterpret_cast<lds_cu32>(rox + 4 * (ROW_PFX + ROW_W)) >> 16) <= x) rox += 4 * ROW_W;
        while ((*reinterpret_cast<lds_cu32>(roy + 4 * (ROW_PFX + ROW_W)) >> 16) <= yy) roy += 4 * ROW_W;
        px = ld2(rox + 4 * ROW_PFX);
        py = ld2(roy + 4 * ROW_PFX);
    }
    // 3: the flag word that holds the head (word 4 carries the prefix above its 16 flags: a head's rank inside the word never reaches them)
    const uint32_t nx = x - (px.x >> 16), ny = yy - (py.x >> 16);
    const uint32_t gx = word_of_head(px.y, nx), gy = word_of_head(py.y, ny);
    const uint32_t wx = *reinterpret_cast<lds_cu32>(rox + 4u * gx), wy = *reinterpret_cast<lds_cu32>(roy + 4u * gy);
    // 4: the head's bit inside it
    const uint32_t bx_ = select_nth_32_lut(wx, nx - heads_before_word(px.y, gx)), by_ = select_nth_32_lut(wy, ny - heads_before_word(py.y, gy));
    // row address -> lane: (ro - row0) / 24, then x 144: one multiply by 6
    raw_x = (rox - row0) * (uint32_t)(TILE_T / (4 * ROW_W)) + 32u * gx + bx_;
    const uint32_t re = (roy - row0) * (uint32_t)(TILE_T / (4 * ROW_W)) + 32u * gy + by_;
    raw_e = y_in ? re : he;
}
template <class WL>
__device__ __forceinline__ void hpc_rawpos2(const WL &S, uint32_t x, uint32_t l, uint32_t R, uint32_t halo_n,
                                            uint32_t Tq, uint32_t rcpTq, uint32_t &raw_x, uint32_t &raw_e) {
    (void)Tq;
    (void)rcpTq;
    const BmFirst f = hpc_rawpos_first(S, x, l, R, halo_n);
    hpc_rawpos_rest(S, f, raw_x, raw_e);
}
static_assert(TILE_T % (4 * ROW_W) == 0, "raw offset of a lane = its row offset times a whole number");
// run heads of the tile before the raw byte at tile-relative offset rel (< TILE_BASES): prefix of its lane + heads of the lane's earlier
// flag words + those of its own word below its bit -- two reads, one round trip
template <class WL>
__device__ __forceinline__ uint32_t heads_before_raw(const WL &S, uint32_t rel) {
    const uint32_t o = div_tile_t(rel), wi = rel - __umul24(o, (uint32_t)TILE_T), g = wi >> 5;
    const uint2 pc = *reinterpret_cast<const uint2 *>(&S.row[o][ROW_PFX]);
    const uint32_t wd = S.row[o][g];
    return (pc.x >> 16) + heads_before_word(pc.y, g) + __popc(wd & ((1u << (wi & 31u)) - 1u)); // (g = 4: wi & 31 < 16, the prefix bits are masked off)
}

// HpcSimd without a second pass over the bases: run heads that the tiles before tile t hold of the read that continues into it.
// Every tile publishes one word right after its compaction (publish_tile_heads): HW_VALID | count of its run heads that belong
// to the read continuing past its end, HW_PASS when that is the whole tile (no read starts in it and the read it lies in began
// earlier).  The sum runs back over PASS tiles to the first tile that is not one, 64 tiles per load.  A word is published early in
// a tile's life and is looked at late in the life of a later tile, and nothing a tile publishes depends on anything it waits for:
// no chains, mostly no waiting.  Polls are bounded (a wave that is not resident, CU masking): then need_runs is raised and the
// host re-runs the call with the runs of every read counted by a pre-pass (s2k_api.hip).
__device__ __forceinline__ uint32_t lookback_heads(const uint32_t *W, uint64_t t, int lane, Counts *counts, uint32_t early, bool &dead) {
    uint32_t sum = 0;
    if (dead) return 0; // this wave (or, at its start, anybody: need_runs) has given up once: the call is re-run anyway, do not wait again
    for (uint64_t tb = t;;) { // tiles tb-1 .. tb-64, lane i looks at tile tb-1-i
        const bool exists = tb >= 1 + (uint64_t)lane;
        const uint32_t *p = W + (exists ? tb - 1 - (uint64_t)lane : 0);
        uint32_t w = HW_VALID; // before tile 0: the chain ends, nothing to add
        int fs = 64;
        for (int polls = 0;; polls++) {
            // (the first look is at what was loaded before the hash loop -- lookback_early --: by now a global round trip old)
            if (exists) w = (polls == 0 && tb == t) ? early : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t valid = __ballot((w & HW_VALID) != 0u);
            const uint64_t stop = __ballot((w & HW_VALID) != 0u && (w & HW_PASS) == 0u);
            fs = stop ? __builtin_ctzll(stop) : 64; // the nearest tile known to end the chain; every tile nearer than it must be known too
            const uint64_t need = fs >= 63 ? ~0ull : ((2ull << fs) - 1ull);
            if ((valid & need) == need) break;
            // (every 64th poll also looks at the flag itself: once one wave has given up the call is run again anyway, and the others
            // need not wait out their own 4096 polls)
            const bool anybody = (polls & 63) == 63 && __hip_atomic_load(&counts->need_runs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (polls >= 4096 || __builtin_amdgcn_readfirstlane((int)anybody)) { // ~10 ms: hundreds of tile times
                if (lane == 0) counts->need_runs = 1;
                dead = true;
                return 0;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        const uint32_t mine = (lane <= fs) ? (w & HW_COUNT) : 0u;
        sum += bcast(wave_incl_scan(mine, lane), 63);
        if (fs < 64) break;
        tb -= 64; // (64 PASS tiles in a row: a read of more than half a million bases)
    }
    return sum;
}

// ... and the word itself (HpcSimd only; after hpc_compact: S.row holds the tile's run heads).  last_start: where the
// last read that starts in (t0, tile end] starts, if there is one.
template <class WL>
__device__ __forceinline__ void publish_tile_heads(uint32_t *W, uint64_t t, const WL &S, uint32_t nh, uint64_t t0, uint32_t tile_len,
                                                   bool any_start, uint64_t last_start, bool first_begins_here, int lane) {
    uint32_t word;
    if (!any_start) {
        word = HW_VALID | (first_begins_here ? 0u : HW_PASS) | nh; // one read all through (it began at t0, or earlier: PASS)
    } else if (last_start >= t0 + tile_len) {
        word = HW_VALID; // a read starts exactly where the tile ends: nothing continues past it
    } else { // run heads from the last read start on: nh - rank of that (forced) head
        const uint32_t c = heads_before_raw(S, (uint32_t)(last_start - t0));
        word = HW_VALID | (nh - c);
    }
    if (lane == 0) __hip_atomic_store(W + t, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Dense phase of one tile: hit bitmasks -> validated, ordered minimizer records in the tile's slab.  Returns their number
// and sets `base` (the slab, or a piece of the overflow region).
//  * DESC = true (default path): 8 bytes per minimizer -- the 32-bit hash and {offset of the l-mer's first base inside the
//    tile : 14, span to its last base : 18} -- plus the tile's descriptor word and its list of read segments (TileMeta); read
//    positions and k-min-mers are made from these by s2k_desc.hip.
//  * DESC = false (legacy path, k > 32 or tiles with more than 30 read starts): 16 bytes -- j, jend, hash, read index -- and a
//    per-read minimizer count; s2k_kminmer.hip does the rest with per-read scans.
template <int L, bool HPC, bool DESC, class WL, class IssueNext>
__device__ __forceinline__ uint32_t dense_phase(IssueNext &&issue_next, WL &S, const uint8_t *D, const uint2 *tab,
                                                const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t t,
                                                uint64_t t0, uint32_t tile_len, uint32_t nh, uint32_t halo_n,
                                                uint32_t Tq, uint32_t rcpTq, uint32_t l, uint32_t r0, uint32_t r1, uint64_t bpos0,
                                                uint64_t rs0, int lane, const Records &rec, uint64_t *pool_cursor,
                                                uint32_t *mn_cnt, Counts *counts, uint64_t &base, const Sem &sem,
                                                unsigned long long *__restrict__ d_agg, TileMeta *__restrict__ d_meta, uint32_t K1,
                                                const uint32_t (&caps)[NPC], const uint32_t (&raw)[5], uint32_t lb_early,
                                                bool &lb_dead, ph_ptr_t ph, uint64_t &stamp) {
    // (1) read starts that matter for this tile -> hash-space boundaries HB; an l-mer x is invalid iff
    //     some boundary has HB - w <= x <= HB - 1  (w = l-1 raw positions for Regular: the l-mer must end
    //     before the next read, src/lib.rs:215-230; w = l run heads for Hpc: head x+l must exist in the
    //     same read, src/nthash_hpc.rs:265-267).  The first read start at or after the tile end (or the
    //     end of the stream) is the one external boundary.
    constexpr int LISTCAP = listcap<HPC>();
    const uint32_t wclr = HPC ? (sem.keep_last ? l - 1 : l) : l - 1; // Hpc drops the last l-mer of a read (src/nthash_hpc.rs:265-267)
    // rcpTq = 65536 / (Tq / 16) + 1 for Tq / 16 in {7, 3, 5, 9, 1}: see div_tq (a division costs ~25 instructions per tile); chosen by the kernel together with Tq
    const uint32_t tql = __umul24(Tq, (uint32_t)lane); // first hash position of this lane
    uint32_t vm[5]; // validated hit mask of this lane
    {
        int vc = (int)nh - (int)tql; // hash positions of this lane that exist
#pragma unroll
        for (int d = 0; d < 5; d++) {
            vm[d] = raw[d] & bits_below(vc - 32 * d);
        }
    }
    const uint64_t tile_end = t0 + tile_len;
    uint32_t nb = 0;                  // internal boundaries (reads r0+1 .. r0+nb start inside the tile)
    const bool many = (r1 - r0) > (uint32_t)(NBL - 2); // more read starts than the LDS lists hold: generic per-hit lookups
    bool ext_at_end = false;          // a read starts exactly where the tile ends (or the stream ends there)
    {
        uint64_t bpos = bpos0; // read_off[r0 + 1 + lane], fetched ahead; later chunks are loaded here (rare)
        uint64_t chunk_prev = rs0; // start of the read whose end lane 0 holds
        int32_t hb_carry = 0;      // (HpcSimd) head-space position of the last boundary of the chunk of 64 before
        for (uint32_t c0 = 0;; c0 += 64) { // wave-uniform; one trip unless the tile holds > 63 read starts
            const bool internal = bpos > t0 && bpos < tile_end;
            const bool external = bpos >= tile_end; // entry n_reads (end of stream) always qualifies
            const uint64_t em = __ballot(external);
            const int first_ext = em ? __builtin_ctzll(em) : 64;
            int32_t HB = 0x7FFFFFFF;
            if (internal || (external && lane == first_ext)) {
                if constexpr (HPC) {
                    if (bpos < tile_end) { // rank of the forced run head at raw offset bpos - t0
                        HB = (int32_t)heads_before_raw(S, (uint32_t)(bpos - t0)); // run heads before the forced one
                    } else { // first read start (or stream end) after the tile: count the run heads before it
                        HB = -1; // resolved below by the whole wave
                    }
                } else {
                    const uint64_t rel = bpos - t0;
                    HB = rel > 0x3FFFFFFFull ? 0x3FFFFFFF : (int32_t)rel;
                }
            }
            if constexpr (HPC) {
                if (em) { // first read start (or stream end) at/after the tile end: run heads before it, counted by all lanes
                    const uint64_t eb = ((uint64_t)bcast((uint32_t)(bpos >> 32), first_ext) << 32) | bcast((uint32_t)bpos, first_ext);
                    if (eb >= tile_end) {
                        const uint64_t relb = eb - t0;
                        const bool before = (uint32_t)lane < halo_n && (uint64_t)S.halo_pos[lane] < relb;
                        const int32_t hbx = (int32_t)(nh + (uint32_t)__popcll(__ballot(before)));
                        if (lane == first_ext) HB = hbx;
                    }
                }
            }
            // Lane i holds the END of read r0+c0+i (= start of the next one); its start is the previous lane's value
            // (rs0 / the previous chunk's last value for lane 0).
            uint64_t pstart = 0;
            uint32_t wextra = 0;
            if constexpr (HPC) {
                if (sem.tail_quirk) { // HpcSimd (src/nthash_hpc_simd.rs:35-68): same rule on the run count of the whole read
                    pstart = ((uint64_t)__shfl_up((uint32_t)(bpos >> 32), 1) << 32) | __shfl_up((uint32_t)bpos, 1);
                    if (lane == 0) pstart = chunk_prev;
                    const bool bnd = (internal || (external && lane == first_ext)) && bpos != ~0ull; // lane holds the end of read r0 + c0 + lane
                    uint32_t Rr = 0;
                    if (sem.read_runs) { // fall-back: the table of a pre-pass
                        if (bnd) Rr = sem.read_runs[(uint64_t)r0 + c0 + lane];
                    } else {
                        // The read that ends at boundary i began at boundary i-1 (head-space positions: HB counts the run heads of
                        // the tile before a read start, the start itself being a forced head), so its run count is HB[i] - HB[i-1];
                        // the read that ends at the tile's FIRST boundary began at head 0 -- or in an earlier tile: then the tiles
                        // before this one published how many of its run heads they hold (Sem::tile_heads), and this one looks back.
                        // Only a boundary the rule can reach from inside the tile matters: an end more than l + 16 run heads past
                        // the tile (HB is a lower bound then) clears nothing below nh whatever Rr is.
                        int32_t hprev = (int32_t)__shfl_up((uint32_t)HB, 1);
                        if (lane == 0) hprev = c0 == 0 ? 0 : hb_carry;
                        Rr = (uint32_t)(HB - hprev);
                        const bool began_before = c0 == 0 && !(t0 == 0 || rs0 == t0);
                        const bool reach = bnd && lane == 0 && began_before && HB - (int32_t)(wclr + 16u) < (int32_t)nh;
                        if (__ballot(reach)) { // wave-uniform
                            // ... and only if the tile HAS a hit among the positions the rule would take away beyond the plain clear,
                            // [HB - w - 16, HB - w - 1]: three times in four it has none, and the count does not matter
                            const int32_t hb0 = (int32_t)bcast((uint32_t)HB, 0);
                            const int lo = hb0 - (int)wclr - 16 - (int)tql, hi = hb0 - (int)wclr - 1 - (int)tql; // lane-local, inclusive
                            uint32_t inwin = 0;
                            if (hi >= 0 && lo < (int)Tq) {
#pragma unroll
                                for (int d = 0; d < 5; d++) {
                                    const int a = lo - 32 * d, bnd2 = hi - 32 * d;
                                    if (bnd2 >= 0 && a < 32) {
                                        const uint32_t m_hi = bnd2 >= 31 ? 0xFFFFFFFFu : ((2u << bnd2) - 1u);
                                        const uint32_t m_lo = a <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << a);
                                        inwin |= vm[d] & m_hi & m_lo;
                                    }
                                }
                            }
                            if (__ballot(inwin != 0u)) {
                                const uint32_t carried = lookback_heads(sem.tile_heads, t, lane, counts, lb_early, lb_dead);
                                if (lane == 0) Rr += carried;
                            }
                        }
                        hb_carry = (int32_t)bcast((uint32_t)HB, 63);
                    }
                    if (bnd) {
                        if (bpos - pstart <= (uint64_t)l) {
                            // seq.len() <= l yields nothing (src/lib.rs:97).  Only a read of exactly l distinct bases has a
                            // window that survives the l-1 clear; a shorter one must not reach back into its predecessor.
                            wextra = Rr >= l ? 1u : 0u;
                        } else if (Rr >= l) {
                            const uint32_t sentinel = Rr - l + 1;
                            if (sentinel >= 32 && (sentinel & 15) == 0) wextra = 16;
                        }
                    }
                    chunk_prev = ((uint64_t)bcast((uint32_t)(bpos >> 32), 63) << 32) | bcast((uint32_t)bpos, 63);
                }
            }
            if constexpr (!HPC) {
                pstart = ((uint64_t)__shfl_up((uint32_t)(bpos >> 32), 1) << 32) | __shfl_up((uint32_t)bpos, 1);
                if (lane == 0) pstart = chunk_prev;
                if (sem.tail_quirk && bpos != ~0ull && bpos - pstart > (uint64_t)l) {
                    // Simd semantics: the final block of 16 l-mers is lost when their number is a multiple of 16
                    // (src/nthash_avx512_32.rs:134-138): 16 more positions before this read's end are invalid
                    const uint64_t sentinel = bpos - pstart - l + 1;
                    if (sentinel >= 32 && (sentinel & 15) == 0) wextra = 16;
                }
            }
            // every lane clears [HB - w, HB - 1] for each boundary of this chunk
            uint64_t todo = __ballot(internal) | (em ? (1ull << first_ext) : 0ull);
            while (todo) {
                const int z = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int32_t hbz = (int32_t)bcast((uint32_t)HB, z);
                const int wz = (int)wclr + (int)bcast(wextra, z);
                if (hbz - wz >= (int)nh) continue; // (wave-uniform: the next read starts so far behind the tile that the range holds none of its positions -- most tiles)
                const int lo = hbz - wz - (int)tql, hi = hbz - 1 - (int)tql; // lane-local, inclusive
                // (branch-free: seven instructions per word for every lane; guarded per lane and per word it was a dozen each plus the exec-mask traffic)
#pragma unroll
                for (int d = 0; d < 5; d++) vm[d] &= ~(bits_below(hi + 1 - 32 * d) & ~bits_below(lo - 32 * d));
            }
            if constexpr (!HPC) {
                // A read of exactly l bases has one l-mer that fits, but the reference yields nothing unless
                // seq.len() > l (src/lib.rs:97): clear that position.
                const bool exact = bpos != ~0ull && bpos - pstart == (uint64_t)l && pstart >= t0 && pstart < tile_end;
                uint64_t ex = __ballot(exact);
                while (ex) {
                    const int z = __builtin_ctzll(ex);
                    ex &= ex - 1;
                    const uint32_t hx = bcast((uint32_t)(pstart - t0), z); // tile-local position of that read's only l-mer
                    const int rel = (int)hx - (int)tql;
                    if (rel >= 0 && rel < (int)Tq) {
#pragma unroll
                        for (int d = 0; d < 5; d++) // static indices: vm[] must stay in registers
                            if ((rel >> 5) == d) vm[d] &= ~(1u << (rel & 31));
                    }
                }
                chunk_prev = ((uint64_t)bcast((uint32_t)(bpos >> 32), 63) << 32) | bcast((uint32_t)bpos, 63);
            }
            if (!many && internal && c0 + lane < (uint32_t)(NBL - 1)) { // remembered for the per-hit read lookup
                S.hb[c0 + lane] = (int16_t)HB;
                S.rs16[c0 + lane + 1] = (uint16_t)(bpos - t0);
            }
            nb += (uint32_t)__popcll(__ballot(internal));
            if (em) {
                const uint64_t ebq = ((uint64_t)bcast((uint32_t)(bpos >> 32), first_ext) << 32) | bcast((uint32_t)bpos, first_ext);
                ext_at_end = ebq == tile_end;
                break;
            }
            const uint64_t ri = (uint64_t)r0 + 1 + c0 + 64 + lane;
            bpos = ri <= n_reads ? read_off[ri] : ~0ull;
            __builtin_amdgcn_s_waitcnt(0x0F70); // as in hpc_compact: nothing pending across the back edge
        }
    }
    S2K_STAMP(3); // boundaries
    // (2) per-lane counts -> offsets
    const uint32_t cnt = __popc(vm[0]) + __popc(vm[1]) + __popc(vm[2]) + __popc(vm[3]) + __popc(vm[4]);
    const uint32_t incl = wave_incl_scan(cnt, lane);
    const uint32_t myoff = incl - cnt;
    const uint32_t N = bcast(incl, 63); // valid minimizers of this tile
    base = t * rec.slab_cap;
    bool dropped = false; // no room for the tile's records: the host re-runs with a larger pool
    if (N > rec.slab_cap) { // rare: more hits than the per-tile slab holds
        uint64_t got = 0;
        if (lane == 0) got = atomicAdd(rec.ovf_cursor, (unsigned long long)N);
        got = ((uint64_t)bcast((uint32_t)(got >> 32), 0) << 32) | bcast((uint32_t)got, 0);
        base = rec.ovf_base + got;
        if (base + N > rec.capacity) { // overflow region exhausted: the host re-runs with pool_needed
            if (lane == 0) {
                counts->pool_overflow = 1;
                atomicMax((unsigned long long *)&counts->pool_needed, (unsigned long long)(got + N));
            }
            base = 0;
            dropped = true;
        }
    }
    if constexpr (!DESC) {
        if (N == 0 || dropped) {
            issue_next(0u, base);
            return 0;
        }
    }
    // per-read minimizer counts, once per tile: read r0+i owns the hits in [HB[i-1], HB[i]) -- lane i keeps the count of
    // segment i (`mine`) and the number of hits before it (`segstart`)
    uint32_t mine = 0, segstart = 0;
    if (!many) {
        uint32_t below_prev = 0;
        for (uint32_t i = 0; i <= nb; i++) { // wave-uniform trip count
            uint32_t below = N;
            if (i < nb) {
                const int lim = S.hb[i] - (int)tql; // lane-local bits [0, lim) lie below boundary i
                uint32_t c = 0;
#pragma unroll
                for (int d = 0; d < 5; d++) {
                    c += __popc(vm[d] & bits_below(lim - 32 * d));
                }
                // lanes left of the boundary's lane count everything, lanes right of it nothing: the total is the
                // exclusive offset of that lane plus its own share -- one v_readlane instead of a wave scan
                const int hbi = S.hb[i];
                const int lb = hbi <= 0 ? 0 : ((uint32_t)hbi >= Tq * 64u ? 63 : (int)div_tq((uint32_t)hbi, rcpTq));
                below = bcast(myoff + c, lb);
            }
            if ((uint32_t)lane == i) {
                mine = below - below_prev;
                segstart = below_prev;
            }
            below_prev = below;
        }
        if constexpr (!DESC)
            if ((uint32_t)lane <= nb && mine) atomicAdd(&mn_cnt[r0 + lane], mine);
    }
    if constexpr (DESC) {
        // ---- the tile's descriptor word and its read segments (see s2k_dev.h) ---------------------------------------------
        if (many) { // more read starts than the lists hold (reads shorter than ~300 bases): the whole call takes the legacy path
            if (lane == 0) counts->need_legacy = 1;
            issue_next(0u, base);
            return 0;
        }
        const bool dep = !(t0 == 0 || rs0 == t0); // a read that began before the tile continues into it
        const uint32_t m_first = bcast(mine, 0), m_last = bcast(mine, (int)nb);
        uint32_t wfix = (uint32_t)lane <= nb && mine > K1 ? mine - K1 : 0u; // windows of a segment that starts a read in (or at the start of) the tile
        if (lane == 0 && dep) wfix = 0;                                         // (the first segment's depend on p)
        // (only lanes 0 .. nb hold a window count: one read start per tile as a rule -- two readlanes instead of a scan over the wave)
        uint32_t Cfix;
        if (nb <= 2u) Cfix = bcast(wfix, 0) + (nb >= 1u ? bcast(wfix, 1) : 0u) + (nb >= 2u ? bcast(wfix, 2) : 0u);
        else Cfix = bcast(wave_incl_scan(wfix, lane), 63);
        const bool pass = dep && nb == 0 && !ext_at_end;
        const uint32_t q_out = ext_at_end ? 0u : (m_last < K1 ? m_last : K1);
        if (lane == 0) {
            d_agg[t] = agg_pack(m_first, Cfix, dropped ? 0u : N, q_out, dep, pass);
            TileMeta *m = &d_meta[t];
            m->rec_base = base;
            m->rs0 = rs0;
            m->r0 = r0;
            m->nb = (uint16_t)nb;
            m->nrd = (uint16_t)(r1 - r0);
        }
        if ((uint32_t)lane <= nb) {
            d_meta[t].segstart[lane] = (uint16_t)segstart;
            if (lane) d_meta[t].rs16[lane] = S.rs16[lane];
        }
        if (N == 0 || dropped || (sem.dbg_skip & 128)) { // (KNOBS builds, 128: everything after the tile's word ablated)
            issue_next(0u, base);
            return 0;
        }
    }
    uint32_t njobs = 0;
    // Re-derivation of queued hashes (closed form src/nthash_hpc.rs:144,168): FOUR lanes per job, each takes a quarter of the l
    // bases (two LDS round trips per pass of 16 jobs instead of eight per lane), partial hashes are XOR-ed across the quad by DPP.
    auto flush_jobs = [&]() {
        wave_sync();
        const uint32_t part = (uint32_t)lane & 3u;
        const uint32_t per = (l + 3u) >> 2;          // bases per lane
        const uint32_t i0 = per * part;              // first base of this lane
        for (uint32_t j0 = 0; j0 < njobs; j0 += 16) { // wave-uniform
            const uint32_t job = j0 + ((uint32_t)lane >> 2);
            const bool act = job < njobs;
            const uint8_t *q = D + (act ? S.jobx[job] : 0) + i0;
            uint32_t f = 0, r = 0;
            if constexpr (L > 0) {
                constexpr uint32_t PER = (L + 3) / 4;
                // all of the lane's bytes first, then all of their table entries: two LDS round trips per pass (round 4: four -- groups of
                // four bases with a scheduling barrier between them, from when the kept hashes and hit masks were still live here)
                constexpr uint32_t GRP = S2K_REDERIVE_GROUP;
#pragma unroll
                for (uint32_t h0 = 0; h0 < PER; h0 += GRP) {
                    uint32_t by[GRP];
                    uint2 ti[GRP];
#pragma unroll
                    for (uint32_t ii = 0; ii < GRP; ii++)
                        if (h0 + ii < PER) by[ii] = q[h0 + ii];
#pragma unroll
                    for (uint32_t ii = 0; ii < GRP; ii++)
                        if (h0 + ii < PER) ti[ii] = tab_in(tab, by[ii]); // IN pair = {h[c], rotl(rc[c], l-1)}
                    uint32_t tf[GRP], tr[GRP];
#pragma unroll
                    for (uint32_t ii = 0; ii < GRP; ii++) {
                        tf[ii] = tr[ii] = 0u;
                        if (h0 + ii < PER) {
                            const uint32_t i = i0 + h0 + ii, rot = (uint32_t)L - 1u - i;
                            const bool valid = 3 * PER + h0 + ii < (uint32_t)L || i < (uint32_t)L; // only the last quarter can run past l
                            tf[ii] = valid ? rotl32(ti[ii].x, rot) : 0u;
                            tr[ii] = valid ? rotr32(ti[ii].y, rot) : 0u;
                        }
                    }
#pragma unroll
                    for (uint32_t ii = 0; ii < GRP; ii += 4) {
                        f = xor3(xor3(f, tf[ii], tf[ii + 1]), tf[ii + 2], tf[ii + 3]); // two terms per instruction (v_bitop3_b32)
                        r = xor3(xor3(r, tr[ii], tr[ii + 1]), tr[ii + 2], tr[ii + 3]);
                    }
                    if (GRP < PER) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (uint32_t ii = 0; ii < per; ii++) {
                    const uint32_t i = i0 + ii;
                    if (i < l) {
                        const uint2 ti = tab_in(tab, q[ii]);
                        f ^= rotl32(ti.x, l - 1 - i);
                        r ^= rotr32(ti.y, l - 1 - i);
                    }
                }
            }
            // XOR over the four lanes of the quad: quad_perm [1,0,3,2] then [2,3,0,1]
            f ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0xB1, 0xf, 0xf, false);
            r ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0xB1, 0xf, 0xf, false);
            f ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x4E, 0xf, 0xf, false);
            r ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x4E, 0xf, 0xf, false);
            if (act && part == 0) rec.hash[base + S.jobslot[job]] = f < r ? f : r;
        }
        njobs = 0;
        wave_sync();
    };
    // (3) batches of up to LISTCAP hits: lanes list their own hits (ascending), then every lane takes one hit.
    //     No global LOADS in here: a load would make the compiler drain the previous round's stores.
    // A tile with at most LISTCAP hits (all but low-complexity sequence) takes its own instantiation of the batch: no range test
    // in the listing, and the hit masks and kept hashes (19 registers) are dead once the hits are listed -- inside the loop
    // of the general case they stay live through the rounds for the next batch.
    auto batch = [&](uint32_t b0, auto single_c) {
        constexpr bool SINGLE = decltype(single_c)::value;
        wave_sync();
        // Hpc tiles of at most 112 hash positions per lane (all but incompressible input) whose hits fit one batch: ONE LANE PER HIT (round 6).  The
        // loop below runs one trip per hit of the busiest lane AND word (~14 trips of ~27 instructions for 137 hits); here the lanes leave their hit
        // masks in the part of the tile buffer the compacted tile does not reach (Tq <= 112 -> nh <= 7168: 2 KiB free behind the run heads that
        // follow the tile), and hit k finds its owner (a directory of first hits + a max scan), its word (cumulative counts, as the back-map does),
        // its bit (sel8) and whether a later raw hit of the same 16-position piece took the kept hash; then the kept hashes take the masks' place
        // and every hit that owns one fetches and stores it -- coalesced.
        bool listed = false;
        if constexpr (HPC && SINGLE && S2K_LPH != 0) {
            if (Tq <= 112u) {
                listed = true;
                typedef __attribute__((address_space(3))) uint8_t lds_b;
                typedef __attribute__((address_space(3))) uint32_t lds_w;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2))); // (plain vector types: HIP's uint2 / uint4 classes cannot be assigned through an LDS pointer)
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) u32x2 lds_w2;
                typedef __attribute__((address_space(3))) u32x4 lds_w4;
                constexpr uint32_t XT = TILE_BASES + 128 - 2048; // 7296: behind nh + the run heads that follow the tile (<= 7168 + 80)
                static_assert(XT % 16 == 0 && XT >= 7168 + MAX_L_TILED + 16 && LISTCAP <= 192, "the tables of the listing must not reach into the compacted tile");
                const uint32_t dq = (uint32_t)(uintptr_t)(lds_b *)const_cast<uint8_t *>(D) + XT; // masks: 64 x 16 B; behind them {offset | cum, last hit of every piece}: 64 x 8 B; directory: 192 B
                constexpr uint32_t T2O = 1024, DIRO = 1536;
                if (lane < 12) *reinterpret_cast<lds_w4 *>(dq + DIRO + 16u * (uint32_t)lane) = u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<lds_w4 *>(dq + 16u * (uint32_t)lane) = u32x4{vm[0], vm[1], vm[2], vm[3]};
                const uint32_t c0 = __popc(vm[0]), c1 = c0 + __popc(vm[1]), c2 = c1 + __popc(vm[2]);
                // position (0 .. 15) of the LAST raw hit of each of the lane's seven pieces, a nibble each: the kept hash of a piece is that hit's
                uint32_t lp = 0;
#pragma unroll
                for (int pc = 0; pc < 7; pc++) {
                    const uint32_t hw = (pc & 1) ? raw[pc >> 1] >> 16 : raw[pc >> 1] & 0xFFFFu;
                    lp |= (31u - (uint32_t)__builtin_clz(hw | 1u)) << (4 * pc); // (hw < 2^16: 0 .. 15; a piece without a raw hit has no hit to ask)
                }
                *reinterpret_cast<lds_w2 *>(dq + T2O + 8u * (uint32_t)lane) = u32x2{c0 | (c1 << 8) | (c2 << 16) | (myoff << 24), lp}; // (myoff <= N <= 192)
                if (cnt) *reinterpret_cast<lds_b *>(dq + DIRO + myoff) = (uint8_t)(lane + 1); // (DS operations of a wave execute in order: the zeros above are in place)
                wave_sync();
                uint32_t own[3] = {0, 0, 0}; // per round of 64 hits: owner lane | piece << 8 | "kept hash is not mine" << 15
                uint32_t carry = 0;
#pragma unroll
                for (int rd = 0; rd < 3; rd++) {
                    if (64u * rd < N) { // (wave-uniform)
                        const uint32_t kq = 64u * rd + (uint32_t)lane;
                        const bool act = kq < N;
                        uint32_t dv = act ? (uint32_t)*reinterpret_cast<lds_b *>(dq + DIRO + kq) : 0u;
                        dv = wave_incl_max(dv);
                        dv = dv > carry ? dv : carry;
                        carry = bcast(dv, 63);
                        const uint32_t o = act ? dv - 1u : 0u; // (hit 0 belongs to the first lane that has one: dv >= 1 for every hit)
                        const u32x2 t2 = *reinterpret_cast<lds_w2 *>(dq + T2O + 8u * o);
                        const uint32_t n = kq - (t2.x >> 24); // rank of the hit among its lane's
                        uint32_t g = 0;               // the word that holds it: number of cumulative counts <= n
                        asm("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_0\n\t"
                            "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
                            "v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_1\n\t"
                            "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
                            "v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:BYTE_2\n\t"
                            "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
                            : "+v"(g)
                            : "v"(n), "v"(t2.x)
                            : "vcc");
                        if (!act) g = 0;
                        const uint32_t hbw = heads_before_word(t2.x, g); // hits of the lane before word g
                        const uint32_t wv = *reinterpret_cast<lds_w *>(dq + 16u * o + 4u * g);
                        const uint32_t ps = 32u * g + select_nth_32_lut(wv, act ? n - hbw : 0u); // position inside the lane
                        const uint32_t piece = ps >> 4;
                        const bool later = ((t2.y >> (4u * piece)) & 15u) != (ps & 15u);
                        if (act) S.list[kq] = (uint16_t)((__umul24(Tq, o) + ps) | (later ? 0x8000u : 0u));
                        own[rd] = o | (piece << 8) | (later || !act ? 0x8000u : 0u);
                    }
                }
                wave_sync(); // the masks are dead: the kept hashes take their place, piece-major (lane stride 4 B: conflict-free)
#pragma unroll
                for (int pc = 0; pc < 7; pc++) *reinterpret_cast<lds_w *>(dq + 256u * pc + 4u * (uint32_t)lane) = caps[pc];
                wave_sync();
#pragma unroll
                for (int rd = 0; rd < 3; rd++) {
                    if (64u * rd < N && !(sem.dbg_skip & 16)) {
                        if (!(own[rd] & 0x8000u))
                            rec.hash[base + 64u * rd + (uint32_t)lane] = *reinterpret_cast<lds_w *>(dq + 256u * ((own[rd] >> 8) & 15u) + 4u * (own[rd] & 0xFFu));
                    }
                }
            }
        }
        // every lane lists its own hits and stores the kept hash of those that were the last raw hit of their piece
        if (!listed) {
            uint32_t k = myoff;
#pragma unroll
            for (int d = 0; d < 5; d++) {
                uint32_t wv = vm[d];
                const uint32_t rw = raw[d];
                constexpr int CPW = 32 / CAPP; // kept hashes per 32 positions
                auto capw = [&](int i) { return caps[(CPW * d + i) < NPC ? CPW * d + i : NPC - 1]; }; // static indices: registers
                const uint32_t cap0 = capw(0), cap1 = capw(1), cap2 = capw(CPW > 2 ? 2 : 0), cap3 = capw(CPW > 2 ? 3 : 1);
                while (wv) {
                    const uint32_t bit = __builtin_ctz(wv);
                    wv &= wv - 1;
                    // raw hits after this one inside the same capture piece: the kept hash is not this hit's
                    const uint32_t pm = CAPP == 16 ? (0xFFFFu << (bit & 16)) : (0xFFu << (bit & 24));
                    const uint32_t later = rw & (pm & ~((2u << bit) - 1u));
                    const uint32_t hvk = CAPP == 16 ? ((bit & 16) ? cap1 : cap0) : ((bit & 16) ? ((bit & 8) ? cap3 : cap2) : ((bit & 8) ? cap1 : cap0));
                    if (SINGLE || (k >= b0 && k < b0 + LISTCAP)) {
                        S.list[k - b0] = (uint16_t)((tql + 32 * d + bit) | (later ? 0x8000u : 0u));
                        if (!later && !(sem.dbg_skip & 16)) rec.hash[base + k] = hvk;
                    }
                    k++;
                }
            }
        }
        wave_sync();
        S2K_STAMP(4); // scan + list
        const uint32_t bn = N - b0 < (uint32_t)LISTCAP ? N - b0 : (uint32_t)LISTCAP;
        // hits that were not the last raw hit of their piece (~7 %) have no kept hash: they are queued and re-derived from
        // their l bytes (closed form, src/nthash_hpc.rs:144,168) BEFORE the rounds, so that the tile's bytes are dead from here
        // on and the next tile can be loaded straight into them (global -> LDS, no registers) while the rounds run
        for (uint32_t k0 = 0; k0 < bn; k0 += 64) {
            const uint32_t kq = k0 + lane;
            const uint32_t e = kq < bn ? S.list[kq] : 0u;
            const bool need = (e & 0x8000u) != 0;
            uint64_t jobs = __ballot(need);
#ifdef S2K_PROFILE
            if ((sem.dbg_skip & 8) && lane == 0) ph[7] += (uint64_t)__popcll(jobs);
#endif
            if (sem.dbg_skip & 64) jobs = 0;
            while (jobs) { // wave-uniform: queue up to JOBCAP jobs, flush, queue the rest
                const uint32_t room = (uint32_t)jobcap<HPC>() - njobs;
                const uint32_t rank = (uint32_t)__popcll(jobs & ((1ull << lane) - 1ull));
                const bool minej = need && ((jobs >> lane) & 1ull) && rank < room;
                if (minej) {
                    S.jobx[njobs + rank] = (uint16_t)(e & 0x3FFFu);
                    S.jobslot[njobs + rank] = (uint16_t)(b0 + kq);
                }
                const uint64_t done = __ballot(minej);
                njobs += (uint32_t)__popcll(done);
                jobs &= ~done;
                if (jobs) flush_jobs();
            }
        }
        if (njobs) flush_jobs();
        S2K_STAMP(12); // hash re-derivation
        if (b0 + (uint32_t)LISTCAP >= N) issue_next(N, base);
        // one hit per lane: tile-local hash position -> stream positions of the l-mer's first base and of the last base that
        // belongs to it
        // (tile-relative 32-bit offsets: the descriptor path's record is tile-relative, only the legacy path adds t0)
        auto backmap = [&](uint32_t x, uint32_t &p, uint32_t &e1) {
            if constexpr (HPC) {
                uint32_t rp = 0, re = 0;
                // Hpc: st[p+l] - 1 (src/nthash_hpc.rs:281; head x + l exists: the hit survived validation);
                // HpcSimd: st[p+l-1], the start of the last run (src/nthash_hpc_simd.rs:64)
                const uint32_t back = sem.end_kind == 2 ? 1u : 0u;
                hpc_rawpos2(S, x, l - back, nh, halo_n, Tq, rcpTq, rp, re);
                e1 = re - (1u - back);
                p = rp;
            } else {
                p = x;
                e1 = x + l - 1; // src/lib.rs:226
            }
        };
        auto rounds = [&](auto many_c) {
        constexpr bool MANY = decltype(many_c)::value;
        for (uint32_t k0 = 0; k0 < bn; k0 += 64) {
            // (markers for tools/isa/check_vmcnt.py: the loop body must issue the stores_per_round() vector-memory operations the
            // counted wait at the top of the next tile relies on)
            asm volatile("; S2K_MARK round_begin many=%0" ::"i"(MANY ? 1 : 0));
            const uint32_t kk = k0 + lane;
            const bool act = kk < bn;
            uint32_t x = 0, rid = 0;
            uint32_t prel = 0, erel = 0; // tile-relative: the l-mer's first base; the last base that belongs to it (may lie far behind the tile)
            if (act) x = S.list[kk] & 0x3FFFu;
            S2K_STAMP(8); // round: list read
            if (act) backmap(x, prel, erel);
            S2K_STAMP(9); // round: back-map
            if constexpr (DESC) {
                if (act) {
                    // tile-relative record: where the read starts is looked up from the tile's segment list by the k-min-mer kernel
                    const uint32_t span = erel - prel;
                    if (span > (uint32_t)REC_SPAN_MAX) counts->need_legacy = 1; // (a homopolymer stretch of > 262 kbp inside one l-mer)
                    const uint32_t sp = span > (uint32_t)REC_SPAN_MAX ? (uint32_t)REC_SPAN_MAX : span;
                    if (!(sem.dbg_skip & 16)) rec.j[base + b0 + kk] = prel | (sp << 14);
                }
            } else {
            const uint64_t p = t0 + prel, e1 = t0 + erel;
            if (act) {
                uint64_t rstart;
                if constexpr (!MANY) {
                    uint32_t c = 0;
                    for (uint32_t i = 0; i < nb; i++) c += (S.hb[i] <= (int32_t)x); // wave-uniform trip count, LDS broadcast
                    rid = r0 + c;
                    rstart = c == 0 ? rs0 : t0 + S.rs16[c];
                } else { // > NBL - 2 reads start in this tile: search the read table itself
                    uint32_t lo = r0, hi = r1;
                    while (lo < hi) {
                        uint32_t mid = lo + (hi - lo + 1) / 2;
                        if (read_off[mid] <= p) lo = mid;
                        else hi = mid - 1;
                    }
                    rid = lo;
                    rstart = read_off[lo];
                }
                const uint64_t slot = base + b0 + kk;
                if (!(sem.dbg_skip & 16)) {
                    rec.j[slot] = (uint32_t)(p - rstart);
                    rec.jend[slot] = (uint32_t)(e1 - rstart);
                    rec.rid[slot] = rid;
                }
            }
            S2K_STAMP(10); // round: read lookup + stores
            if constexpr (MANY) { // per-read minimizer counts: one atomic per (round, read); the common case is counted per tile above
                uint64_t remm = __ballot(act);
                while (remm) {
                    const int z = __builtin_ctzll(remm);
                    const uint32_t rz = bcast(rid, z);
                    const uint64_t same = __ballot(act && rid == rz);
                    if (lane == z) atomicAdd(&mn_cnt[rz], (uint32_t)__popcll(same));
                    remm &= ~same;
                }
            }
            }
            asm volatile("; S2K_MARK round_end many=%0" ::"i"(MANY ? 1 : 0));
        }
        };
        // > NBL - 2 reads starting in one tile take the variant that searches the read table itself; keeping it a
        // separate instantiation keeps its global loads out of the common loop
        if constexpr (DESC) {
            rounds(std::false_type{});
        } else {
            if (many) rounds(std::true_type{});
            else rounds(std::false_type{});
        }
    };
    if (N <= (uint32_t)LISTCAP) batch(0u, std::true_type{});
    else
        for (uint32_t b0 = 0; b0 < N; b0 += LISTCAP) batch(b0, std::false_type{});
    return N;
}

// Persistent kernel: a wave takes tiles wave_id, wave_id + n_waves, wave_id + 2 n_waves and then draws from the cursors.
// While the rounds of a tile run, the next tile's 9344 bytes are in flight into the wave's LDS buffer (LDS-DMA), and the
// read-table entries of the next tiles are fetched before they are needed, so no global-load latency sits on the critical
// path except in the first iteration.
// DESC: the descriptor path (8-byte records + one word and a segment list per tile, see dense_phase); else the legacy records.
template <int L, bool HPC, bool DESC>
__global__ __launch_bounds__(64 * tw<HPC>(), waves_per_simd<HPC>()) void tile_minimizer_kernel(
    const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
    uint64_t n_tiles, const uint32_t *__restrict__ tile_read0, Sem sem, Records rec, uint64_t *pool_cursor,
    uint64_t *__restrict__ tile_rec_off, uint32_t *__restrict__ tile_cnt, uint32_t *mn_cnt, Counts *counts,
    unsigned long long *__restrict__ d_agg, TileMeta *__restrict__ d_meta, uint32_t K1, uint64_t tile_begin) {
    // (this launch works on the tiles [tile_begin, n_tiles): the descriptor path cuts a call into a few chunks of tiles so that
    // the k-min-mer kernel of one chunk runs, on a second stream, beside the minimizer kernel of the next)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using WL = WaveLdsT<HPC>;
    uint2 *tab = reinterpret_cast<uint2 *>(smem);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem != 0u) __builtin_trap(); // lut() assumes it
    const int lane0 = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // w in an SGPR: everything per tile is scalar
    int lane = lane0; // re-made opaque at the top of every tile (see the loop): nothing derived from the lane index is loop-invariant
    const uint32_t l = L > 0 ? (uint32_t)L : sem.l;
    constexpr int TW = tw<HPC>();
    for (int c = threadIdx.x; c < 256; c += 64 * TW) {
        uint32_t cc = c;
        // Simd result semantics map bytes by their low nibble (src/nthash_avx512_32.rs:178-193)
        uint32_t h = sem.simd_seeds ? seed_h_simd(cc) : seed_h_scalar(cc), r = sem.simd_seeds ? seed_rc_simd(cc) : seed_rc_scalar(cc);
        tab[c] = make_uint2(h, rotl32(r, l - 1));
        tab[256 + c] = make_uint2(rotl32(h, l), rotr32(r, 1));
    }
    if constexpr (HPC && S2K_WARM2 != 0)
        for (int c = threadIdx.x; c < 256; c += 64 * TW) {
            const uint32_t cc = c;
            const uint32_t h = sem.simd_seeds ? seed_h_simd(cc) : seed_h_scalar(cc), r = sem.simd_seeds ? seed_rc_simd(cc) : seed_rc_scalar(cc);
            reinterpret_cast<uint2 *>(smem + WARM_OFF)[c] = make_uint2(rotl32(h, 1), rotr32(rotl32(r, l - 1), 1));
        }
    if constexpr (HPC && S2K_SEL8 != 0)
        for (int c = threadIdx.x; c < 256 * 8; c += 64 * TW) {
            const uint32_t m = (uint32_t)c >> 3, n = (uint32_t)c & 7u;
            smem[SEL8_OFF + c] = (uint8_t)(n < (uint32_t)__popc(m) ? select_nth_32(m, n) : 0x0Cu); // (0x0C: as a v_perm_b32 selector byte, "zero" -- the table is pass 2's too)
        }
    __syncthreads(); // the only workgroup barrier; waves are independent from here on
#ifdef S2K_DEBUG_KNOBS
    const uint64_t dbg_mt0 = __builtin_amdgcn_s_memtime(), dbg_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    WL &S = *reinterpret_cast<WL *>(smem + table_bytes<HPC>() + (size_t)w * sizeof(WL));
    uint8_t *D = S.buf + HS_OFF;
    const uint64_t n_waves = (uint64_t)gridDim.x * TW;
    uint64_t t = tile_begin + (uint64_t)blockIdx.x * TW + w;
    if (t >= n_tiles) return;
    if (__builtin_amdgcn_readfirstlane((int)counts->bad_input)) return; // malformed read table (validate_read_off_kernel): touch nothing
    uint64_t stamp = __builtin_amdgcn_s_memtime();
#ifdef S2K_PROFILE
    ph_ptr_t ph = reinterpret_cast<ph_ptr_t>((uint32_t)(table_bytes<HPC>() + TW * sizeof(WL) + (size_t)w * 128)); // (the dynamic region starts at LDS address 0)
    if (lane0 < 16) ph[lane0] = 0;
    wave_sync();
#else
    ph_ptr_t ph = nullptr;
#endif

    // a tile is "full" when the tile and its 128 B look-ahead lie inside the stream: staged branch-free
    auto is_full = [&](uint64_t tt) { return (tt + 1) * (uint64_t)TILE_BASES + 128 <= n_bases; };
    // Software pipeline over this wave's tiles (cur = t, then tn, then tnn):
    //   the bases of tn   : loaded straight into this wave's LDS buffer (global_load_lds_dwordx4: no registers, no ds_write)
    //                       as soon as the dense phase of cur has re-derived its queued hashes, i.e. no longer reads the buffer;
    //                       the one-lane-per-hit rounds that follow cover the latency
    //   prevb             : the byte before tn, with them
    //   r0n, r1n          : tile_read0 of tn    (issued one iteration earlier, so their values are usable ...)
    //   bposn, rs0n       : read_off[r0n + 1 + lane], read_off[r0n]   (... to address these, issued with the bases)
    //   r0nn, r1nn        : tile_read0 of tnn
    // so no dependent global load is waited for on the spot after the prologue.
    uint32_t prevb = 0;
    bool have_pre = false;
    typedef __attribute__((address_space(3))) uint8_t lds_u8s;
    // The loads are issued from inline asm on purpose.  As compiler-visible LDS-DMA they made LLVM put a vmcnt(0) in front of the
    // first LDS access of the next tile (it cannot count across the rounds' loop), which also drained the rounds' stores; the
    // counted wait at the top of the loop is the one that covers them.  Operations the compiler does not know about can only
    // make ITS counted waits wait longer (vmcnt is in order), never too short.  M0 (LDS base of an LDS-DMA) is saved and restored
    // inside the statement; the instruction offset moves the global and the LDS address alike (tools/experiments/lds_dma_asm_test.hip).
    auto prefetch = [&](uint64_t tt) { // issue the loads for tile tt; nothing waits here
        static_assert(NPRE == 10, "4 + 4 + 1 wave-wide loads of 1 KiB and one of 128 B");
        const uint8_t *g = bases + tt * (uint64_t)TILE_BASES + 16 * lane;
        const uint32_t lbase = (uint32_t)(uintptr_t)(lds_u8s *)D; // wave-uniform: lane i of a load lands at base + offset + 16 i
        uint32_t save;
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const uint8_t *pg = g + 4096 * grp;
            const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + 4096u * grp));
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(save) : "v"(pg), "s"(m) : "memory");
        }
        const uint8_t *pg = g + 8192;
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + 8192u));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(pg), "s"(m) : "memory");
        if (lane < 8) // the 128 B look-ahead
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(save) : "v"(pg), "s"(m) : "memory");
    };
    auto byte_before = [&](uint64_t tt) -> uint32_t { return tt > 0 ? (uint32_t)bases[tt * (uint64_t)TILE_BASES - 1] : 0u; };
    auto read_entries = [&](uint32_t rr0, uint64_t &bp, uint64_t &rs) {
        const uint64_t bri = (uint64_t)rr0 + 1 + lane;
        bp = bri <= n_reads ? read_off[bri] : ~0ull; // entry n_reads is the end of the stream
        rs = read_off[rr0];
    };
    uint32_t r0 = tile_read0[t], r1 = tile_read0[t + 1];
    uint64_t bpos0, rs0;
    read_entries(r0, bpos0, rs0);
    if (is_full(t)) prevb = byte_before(t);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): no register load is pending when the loop is entered (see stores_after_dma)
    if (is_full(t)) {
        prefetch(t);
        have_pre = true;
    }
    // vmcnt is one in-order counter for loads, LDS-DMA loads and stores on gfx9: the wait for a tile's DMA loads at the top of
    // an iteration would also drain the record stores the rounds issued AFTER them (~2 k cycles per tile, measured as the
    // "staging" phase).  So the iteration keeps count: nothing but those stores may be issued between the DMA loads and
    // that wait -- every other load of the next tile's data is issued before the hash loop and waited for right after it --
    // and the wait leaves exactly that many operations in flight.
    uint32_t stores_after_dma = 0;
    // tn / tnn: the next two tiles of this wave; >= n_tiles: none.  Dynamic tiles are numbered from dyn0 on: cursor g deals
    // dyn0 + g, dyn0 + g + TILE_CURSORS, ... (pool_cursor[16 + 16 g], zeroed by the host before the launch).
    uint64_t tn = t + n_waves, tnn = t + 2 * n_waves;
    const uint64_t dyn0 = tile_begin + 3 * n_waves;
    // Which cursor a wave draws from must not depend on its place in the block alone: the SIMD's arbiter serves its oldest wave
    // first, so waves in the same slot of every block run at the same (un)favourable speed, and a cursor served by slow waves
    // only is still dealing when the others have run dry (TW = 16: 16 b + w mod 64 keeps w -- the last wave finished 2.7 ms
    // after the first).  The second term walks the wave index through the cursors' residues as the blocks go by.
    const uint32_t cur_g = (uint32_t)((blockIdx.x * TW + w + blockIdx.x / (TILE_CURSORS / 4)) % TILE_CURSORS);
    unsigned int *const cursors = (unsigned int *)(pool_cursor + 16); // 32-bit draws (a 64-bit result's dead upper half would be
                                                                       // reused early and pull a vmcnt(0) in front of the compaction); cursor g = word 32 g
    // (A wave stays with its cursor: when that runs dry the wave is done.  Round 2 let it move on to another cursor that was not;
    // measured again in round 3 (same-call A/B, profiles/r03_ab_knobs.txt) that costs 1 % instead of gaining anything: the 64
    // cursors run dry within a tile time of each other, and the extra code sits in the tile loop.)
    uint32_t r0n = 0, r1n = 0;
    if (tn < n_tiles) {
        r0n = tile_read0[tn];
        r1n = tile_read0[tn + 1];
    }

    // (the persistent kernel's waves at a raised issue priority over the k-min-mer kernel's beside them -- s_setprio 1 / 3 for the whole kernel -- changed nothing:
    // profiles/r06_chunks_prio_sweep.txt)
    uint32_t prio_iter = 0;
    // HpcSimd look-back (lookback_heads): a wave that has waited in vain once does not wait again -- nor does any wave once the
    // call is known to be run again (need_runs; e.g. the minimizer kernel of a later chunk)
    bool lb_dead = sem.tile_heads != nullptr && __builtin_amdgcn_readfirstlane((int)counts->need_runs) != 0;
    for (; t < n_tiles;) {
        // A dozen values derived from the lane index (64-bit zero-extensions, 16 x lane offsets, masks) are loop-invariant; LLVM
        // hoists them out of this loop and then holds -- or spills -- them across the hash loop, where the register pressure
        // peaks (the kernels lost 20-55 VGPRs that way).  An opaque copy per tile makes it recompute the two or three
        // instructions each where they are used.
        lane = lane0;
        asm volatile("" : "+v"(lane));
        if (sem.tile_heads) {
            // HpcSimd, look-back for run heads (lookback_heads): the SIMD's arbiter serves its oldest wave first, so the three waves of
            // a SIMD run at lastingly different speeds, and a tile whose predecessor sits on a slower wave waits for that wave's
            // word.  Rotating the issue priority tile by tile evens the speeds out: 8.05 -> 7.60 ms per 10 Gbp (other modes: no
            // gain, not done).
            switch ((prio_iter++ + (uint32_t)(w >> 2)) % 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(2); break;
            }
        }
        const uint64_t t0 = t * (uint64_t)TILE_BASES;
        const uint64_t rem = n_bases - t0;
        const uint32_t avail = rem > (uint64_t)(TILE_BASES + 128) ? (uint32_t)(TILE_BASES + 128) : (uint32_t)rem;
        const uint32_t tile_len = rem > (uint64_t)TILE_BASES ? (uint32_t)TILE_BASES : (uint32_t)rem;
        // ---- the tile (+128 B look-ahead) in LDS ------------------------------------------------------------------
        uint32_t l16_tile = 16 * lane;
        if (__builtin_amdgcn_readfirstlane((int)have_pre)) { // a scalar branch: as a divergent if/else the slow path's loads would precede this wait
            // loaded into the buffer by the previous iteration (prologue: just now): wait for the loads, nothing to move.
            // s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; 0x0F70 = vmcnt(0) only
            constexpr int SPR = stores_per_round<DESC>();
            asm volatile("; S2K_MARK counted_wait per_round=%0" ::"i"(SPR));
            switch (__builtin_amdgcn_readfirstlane((int)stores_after_dma)) { // wave-uniform, and the compiler should know
                case 1 * SPR: __builtin_amdgcn_s_waitcnt(0x0F70 | (1 * SPR)); break;
                case 2 * SPR: __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * SPR)); break;
                case 3 * SPR: __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * SPR)); break;
                case 4 * SPR: __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * SPR)); break;
                default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else { // tile at the end of the stream: guarded loads, zero past the end
            const uint8_t *g = bases + t0;
            uint32_t l16o = l16_tile;
            asm volatile("" : "+v"(l16o)); // opaque: or the ten per-lane addresses of this rare path are hoisted out of the tile loop
                                            // and held (then spilled) across the hash loop
            for (int r = 0; r < NPRE; r++) {
                const uint32_t off = l16o + 1024 * r;
                if (r == NPRE - 1 && lane >= 8) break;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (off + 16 <= avail) {
                    v = *reinterpret_cast<const uint4 *>(g + off);
                } else if (off < avail) {
                    uint32_t tmp[4] = {0, 0, 0, 0};
                    for (uint32_t b = 0; off + b < avail && b < 16; b++) tmp[b >> 2] |= (uint32_t)g[off + b] << (8 * (b & 3));
                    v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
                *reinterpret_cast<uint4 *>(D + off) = v;
            }
            prevb = t0 > 0 ? (uint32_t)bases[t0 - 1] : 0u;
            __builtin_amdgcn_s_waitcnt(0x0F70); // inside this branch: pending here, the load would cost the other branch a vmcnt(0) where they meet
        }
        const uint32_t cr0 = r0, cr1 = r1, cprev = prevb; // this tile's values (the registers get reused below)
        // ---- loads for the next tiles that land in registers: issued now, waited for right after the hash loop ----------
        uint32_t took = 0; // draw for the tile after tnn
        if (lane == 0 && tnn < n_tiles) took = atomicAdd(&cursors[32 * cur_g], 1u);
        // (raw loads at always-valid addresses: the selects that make the final values come after the wait -- a consumer in
        // here would pull a vmcnt(0) in front of the compaction)
        // The wave-uniform ones are VECTOR loads on purpose (index + an opaque zero in a VGPR): as scalar loads they count in
        // lgkmcnt, which every wait for this wave's LDS traffic drains -- the two s_load of this block sat right in front of an
        // s_waitcnt lgkmcnt(0), a full global round trip exposed per tile.
        uint64_t bposn = ~0ull, rs0n = 0;
        uint32_t r0nn = 0, r1nn = 0, prevbn = 0;
        const uint64_t brin = (uint64_t)r0n + 1 + lane;
        uint32_t vzero = 0;
        asm volatile("" : "+v"(vzero));
        if (tn < n_tiles) {
            bposn = read_off[brin <= n_reads ? brin : n_reads];
            rs0n = read_off[(uint64_t)r0n + vzero];
            prevbn = (uint32_t)bases[tn * (uint64_t)TILE_BASES - 1 + vzero]; // tn > 0
            if (tnn < n_tiles) {
                r0nn = tile_read0[tnn + vzero];
                r1nn = tile_read0[tnn + 1 + vzero];
            }
        }
        if (lane == 0) S.buf[HS_OFF - 1] = 0;
        if constexpr (HPC) { // read-start marks are OR-ed in by hpc_compact (a row is 24 bytes: 8-byte stores)
            uint2 *rz = reinterpret_cast<uint2 *>(&S.row[lane][0]);
            rz[0] = make_uint2(0u, 0u);
            rz[1] = make_uint2(0u, 0u);
            S.row[lane][4] = 0;
        }
        wave_sync();
        S2K_STAMP(0); // staging

        uint32_t nh = tile_len; // number of hash positions owned by this tile
        uint32_t lb_early = 0;  // (HpcSimd) early look at the words of the tiles before this one, see below
        uint32_t halo_n = 0;
        int np = 9;
        uint32_t rcpTq = 7282u; // 65536 / 9 + 1
        if constexpr (HPC) {
#ifndef EXP_NOCOMPACT
            if (!(sem.dbg_skip & 4))
                nh = hpc_compact(D, S, bases, read_off, n_reads, n_bases, t0, tile_len, cr0, cr1, l, lane, halo_n, bpos0,
                                 cprev, t0 == 0 || rs0 == t0, sem, ph, stamp);
#endif
            if (sem.tile_heads) { // HpcSimd: tell later tiles how many run heads of the read that continues past this tile it holds
                const uint32_t ns = cr1 - cr0; // read starts in (t0, tile end]; bpos0 of lane i = start of read r0 + 1 + i
                uint64_t last_start = 0;
                if (ns) {
                    if (ns <= 64)
                        last_start = ((uint64_t)bcast((uint32_t)(bpos0 >> 32), (int)ns - 1) << 32) | bcast((uint32_t)bpos0, (int)ns - 1);
                    else
                        last_start = read_off[cr1]; // (hundreds of reads in one tile)
                }
                publish_tile_heads(sem.tile_heads, t, S, nh, t0, tile_len, ns != 0, last_start, t0 == 0 || rs0 == t0, lane);
                // ... and take a first look at the words of the 64 tiles before this one: the tiles next to it are processed at
                // about the same time and have mostly published by now; the load has the hash loop to come back (it is waited
                // for with the other loads right after it), and the dense phase only polls when that look was too early.
                if (!(t0 == 0 || rs0 == t0) && t >= 1 + (uint64_t)lane)
                    lb_early = __hip_atomic_load(sem.tile_heads + (t - 1 - (uint64_t)lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int need = (int)((nh + 1023) >> 10);
            np = need <= 1 ? 1 : need <= 3 ? 3 : need <= 5 ? 5 : need <= 7 ? 7 : 9;
            rcpTq = need <= 1 ? 65537u : need <= 3 ? 21846u : need <= 5 ? 13108u : need <= 7 ? 9363u : 7282u; // 65536 / np + 1 (div_tq)
        }
        S2K_STAMP(1); // hpc compaction
        // ---- the next tile's bases: DMA into this wave's buffer, issued from inside the dense phase (see issue_once) -----
        auto issue_next = [&](uint32_t n_rec, uint64_t rec_base) {
            if constexpr (!DESC) {
                if (lane == 0) { // before the DMA loads: only the rounds' stores may follow them
                    tile_cnt[t] = n_rec;
                    tile_rec_off[t] = rec_base;
                }
            }
            have_pre = false;
            if (tn < n_tiles && is_full(tn)) {
                prefetch(tn);
                have_pre = true;
            }
        };
        const uint32_t Tq = 16 * np;
        uint32_t N = 0;
        uint64_t base = 0;
        uint32_t caps[NPC], raw[5]; // kept hashes / hit bits of this lane's 144 hash positions (hash_stage)
#pragma unroll
        for (int g2 = 0; g2 < NPC; g2++) caps[g2] = 0;
#pragma unroll
        for (int g2 = 0; g2 < 5; g2++) raw[g2] = 0;
        if (nh != 0 && sem.enabled) {
            // ---- the hot loop ------------------------------------------------------------------------------
#ifndef EXP_NOHASH
            if (!(sem.dbg_skip & 1)) hash_stage<L, HPC>(D, tab, sem.bound_le, lane, l, np, caps, raw);
#endif
            wave_sync();
            __builtin_amdgcn_sched_barrier(0);
#ifdef S2K_XCAL // (calibration experiment, tools/ab: what does one more instruction of a kind cost the tile?  256 of them per tile, results unchanged)
            {
                uint32_t dv = (uint32_t)lane, dv2 = (uint32_t)lane + 3u, ds = (uint32_t)t;
                (void)dv; (void)dv2; (void)ds;
#pragma unroll
                for (int i_ = 0; i_ < 64; i_++) {
#if S2K_XCAL == 1
                    asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0" : "+v"(dv), "+v"(dv2));
#elif S2K_XCAL == 2
                    asm volatile("v_alignbit_b32 %0, %0, %1, 7\n\tv_alignbit_b32 %1, %1, %0, 9\n\tv_alignbit_b32 %0, %0, %1, 7\n\tv_alignbit_b32 %1, %1, %0, 9" : "+v"(dv), "+v"(dv2));
#elif S2K_XCAL == 3
                    asm volatile("s_add_u32 %0, %0, 1\n\ts_xor_b32 %0, %0, 5\n\ts_add_u32 %0, %0, 3\n\ts_xor_b32 %0, %0, 9" : "+s"(ds));
#elif S2K_XCAL == 4
                    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\tds_read_b32 %0, %2 offset:8\n\tds_read_b32 %1, %2 offset:12" : "=v"(dv), "=v"(dv2) : "v"(4u * (uint32_t)lane) : "memory");
#elif S2K_XCAL == 6 /* 64 reads, 4-way bank conflict each (lanes 8 apart share a bank): 8 LDS cycles instead of 2 */
                    if (i_ < 16) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\tds_read_b32 %0, %2 offset:8\n\tds_read_b32 %1, %2 offset:12" : "=v"(dv), "=v"(dv2) : "v"(4u * ((uint32_t)lane & 7u) + 128u * ((uint32_t)lane >> 3)) : "memory");
#elif S2K_XCAL == 7 /* 64 reads, conflict-free */
                    if (i_ < 16) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\tds_read_b32 %0, %2 offset:8\n\tds_read_b32 %1, %2 offset:12" : "=v"(dv), "=v"(dv2) : "v"(4u * (uint32_t)lane) : "memory");
#elif S2K_XCAL == 8 /* 64 reads, 16-way bank conflict each: 32 LDS cycles */
                    if (i_ < 16) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\tds_read_b32 %0, %2 offset:8\n\tds_read_b32 %1, %2 offset:12" : "=v"(dv), "=v"(dv2) : "v"(4u * ((uint32_t)lane & 1u) + 128u * ((uint32_t)lane >> 1)) : "memory");
#elif S2K_XCAL == 5
                    asm volatile("v_min_u32 %0, %0, %1\n\tv_max_u32 %1, %1, %0\n\tv_min_u32 %0, %0, %1\n\tv_max_u32 %1, %1, %0" : "+v"(dv), "+v"(dv2));
#endif
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" ::"v"(dv), "v"(dv2), "s"(ds));
            }
#endif
            S2K_STAMP(2); // hash loop
        }
        // the draw made at the top of the iteration is looked at HERE: everything older than it in the vector-memory queue had
        // the whole hash loop to finish, whereas at the end of the iteration a wait for it would also drain this tile's stores
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): everything older had the whole hash loop to finish
        if (brin > n_reads) bposn = ~0ull;  // entry n_reads is the end of the stream, nothing lies beyond it
        rs0n = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rs0n >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rs0n);
        r0nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0nn);
        r1nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1nn);
        prevbn = (uint32_t)__builtin_amdgcn_readfirstlane((int)prevbn);
        uint64_t drawn = ~0ull;
        if (tnn < n_tiles)
            drawn = dyn0 + cur_g + (uint64_t)TILE_CURSORS * (uint32_t)__builtin_amdgcn_readfirstlane((int)took);
        // the next tile's loads go out from inside the dense phase, the moment it no longer reads this tile's bytes
        bool issued = false; // wave-uniform
        auto issue_once = [&](uint32_t n_rec, uint64_t rec_base) {
            if (!issued) issue_next(n_rec, rec_base);
            issued = true;
        };
        // The dense phase gets its OWN copy of the lane index, opaque to the compiler: everything it derives from the lane (a dozen
        // masks, offsets and 64-bit zero-extensions) is otherwise loop-invariant, gets hoisted out of the tile loop and lives --
        // or is spilled -- across the hash loop, which is where the register pressure peaks.
        int lane_d = lane;
        asm volatile("" : "+v"(lane_d));
        if (DESC || (nh != 0 && sem.enabled)) { // (descriptor path: every tile leaves its word and its segment list, hits or not)
#ifndef EXP_NODENSE
            if (!(sem.dbg_skip & 2))
                N = dense_phase<L, HPC, DESC>(issue_once, S, D, tab, read_off, n_reads, t, t0, tile_len, nh, halo_n, Tq, rcpTq, l, cr0, cr1,
                                           bpos0, rs0, lane_d, rec, pool_cursor, mn_cnt, counts, base, sem, d_agg, d_meta, K1,
                                           caps, raw, lb_early, lb_dead, ph, stamp);
#endif
            S2K_STAMP(5); // rounds
        }
        issue_once(0, 0);
        // the rounds of the last batch of hits are all that came after the DMA loads: three stores each (the variant for tiles
        // with very many reads loads and counts in there: it waits for everything)
        {
            const bool many = (cr1 - cr0) > (uint32_t)(NBL - 2) || (sem.dbg_skip & 16) != 0; // (KNOBS builds: stores ablated)
            constexpr uint32_t LISTCAP = (uint32_t)listcap<HPC>();
            const uint32_t last = N == 0 ? 0u : N - ((N - 1) / LISTCAP) * LISTCAP;
            stores_after_dma = many ? 0u : (uint32_t)stores_per_round<DESC>() * ((last + 63u) / 64u);
        }
        wave_sync(); // LDS of this wave is reused by the next tile
        r0 = r0n; r1 = r1n; bpos0 = bposn; rs0 = rs0n; r0n = r0nn; r1n = r1nn; prevb = prevbn; // rotate the pipeline
        t = tn;
        tn = tnn;
        if (tnn < n_tiles) tnn = drawn; // >= n_tiles: this wave's cursor has run dry
        S2K_STAMP(6); // tail
    }
#ifdef S2K_PROFILE
    if ((sem.dbg_skip & 8) && lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd((unsigned long long *)&counts->dbg_cycles[blockIdx.x & 63][i], (unsigned long long)ph[i]);
#endif
#ifdef S2K_DEBUG_KNOBS // S2K_DEBUG_SKIP & 32: when did the first and the last wave run out of tiles (100 MHz clock)?
    if ((sem.dbg_skip & 32) && lane == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        atomicMax((unsigned long long *)&counts->dbg_cycles[0][0], now);
        atomicMax((unsigned long long *)&counts->dbg_cycles[0][1], ~now);
        // shader clock under this kernel's load: s_memtime counts shader cycles, s_memrealtime the constant 100 MHz clock
        atomicAdd((unsigned long long *)&counts->dbg_cycles[1][0], (unsigned long long)(__builtin_amdgcn_s_memtime() - dbg_mt0));
        atomicAdd((unsigned long long *)&counts->dbg_cycles[1][1], (unsigned long long)(now - dbg_rt0));
        const uint32_t wid = blockIdx.x * TW + w;
        if (wid < 4096) {
            counts->dbg_wave[wid][0] = now;
            counts->dbg_wave[wid][1] = ((uint64_t)__builtin_amdgcn_s_getreg((20 | (0 << 6) | (31 << 11))) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((4 | (0 << 6) | (31 << 11)));
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Regular family WITHOUT a tile buffer (round 6; descriptor path, compile-time l, tiles that lie wholly inside the stream).
// The 16-wave block of tile_minimizer_kernel<L, false> owns all 160 KiB of a CU's LDS for its tile buffers: four waves per SIMD is all the LDS
// allows, and nothing -- the k-min-mer kernel least of all -- fits beside it.  A Regular lane hashes 144 + l CONSECUTIVE raw bytes: it can take them
// straight from global memory, 64 bytes (four pieces) at a time so that its accesses to one 128-byte line follow each other -- the hash loop alone
// runs at the LDS-staged loop's rate at every occupancy that way (profiles/r04_hash_loop_without_tile_buffer.txt).  What is left in LDS is the
// dense phase's hit list (< 1 KiB per wave): the registers set the occupancy, and the k-min-mer kernel of the chunk before runs BESIDE this kernel as
// it does beside the Hpc one.  The dense phase is the tiled kernel's own (dense_phase<L, false, true>); the bytes of the few hits whose hash must
// be re-derived come from global memory again (L2).  The stream's last one or two tiles (no 128-byte look-ahead inside the stream) go to the
// tiled kernel.
// MEASURED (profiles/r06_stream.txt, 10 Gbp Regular): bit-exact, and slower than the tiled kernel in every arrangement -- four waves per SIMD, k-min-mer
// stage behind it: 4.11 ms against the tiled kernel's 3.91 (the loop that ran alone at the LDS-staged rate does not inside the kernel); five waves
// (two blocks of ten, 96 VGPRs): 5.3 ms; four waves with the k-min-mer kernel beside it in 4-8 chunks: 5.27-5.45 ms per step against 5.05.  So the
// kernel is compiled only into builds made with -DS2K_STREAM_BUILD=1 (tools/ab/build_variant.sh) and selected there by S2K_STREAM_KERNEL=1; the
// CPU tier keeps it compiling (tests/test_isa_invariants.py).
// ------------------------------------------------------------------------------------------------
#ifndef S2K_STREAM_BUILD
#define S2K_STREAM_BUILD 0
#endif
#if S2K_STREAM_BUILD
#ifndef S2K_STREAM_DEFAULT
#define S2K_STREAM_DEFAULT 0 // 1: the Regular family's descriptor path uses stream_minimizer_kernel unless S2K_STREAM_KERNEL=0
#endif
#ifndef S2K_STREAM_TW
#define S2K_STREAM_TW 16 // waves per block
#endif
#ifndef S2K_STREAM_BPC
#define S2K_STREAM_BPC 1 // blocks per CU (a block has at most 16 waves: five or six waves per SIMD take two blocks)
#endif
#ifndef S2K_STREAM_WPS
#define S2K_STREAM_WPS ((S2K_STREAM_TW * S2K_STREAM_BPC + 3) / 4) // waves per SIMD the registers are budgeted for (more than the kernel's own: room for the k-min-mer kernel's beside them)
#endif
constexpr int STREAM_TW = S2K_STREAM_TW, STREAM_BPC = S2K_STREAM_BPC, STREAM_WPS = S2K_STREAM_WPS;
struct alignas(16) StreamLds { // per wave: what dense_phase keeps in LDS (see WaveLdsT)
    uint16_t list[listcap<false>()];
    uint16_t jobx[jobcap<false>()];
    uint16_t jobslot[jobcap<false>()];
    int16_t hb[NBL];
    uint16_t rs16[NBL];
};
constexpr int stream_lds_bytes() { return SEED_TABLE_BYTES + STREAM_TW * (int)sizeof(StreamLds); }

template <int L>
__global__ __launch_bounds__(64 * STREAM_TW, STREAM_WPS) void stream_minimizer_kernel(
    const uint8_t *__restrict__ bases, const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles /* full tiles only */,
    const uint32_t *__restrict__ tile_read0, Sem sem, Records rec, uint64_t *pool_cursor, Counts *counts, unsigned long long *__restrict__ d_agg,
    TileMeta *__restrict__ d_meta, uint32_t K1, uint64_t tile_begin) {
    static_assert(L > 0, "compile-time l only");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint2 *tab = reinterpret_cast<uint2 *>(smem);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem != 0u) __builtin_trap(); // the seed look-ups assume the tables at LDS address 0
    const int lane0 = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int TW = STREAM_TW;
    constexpr uint32_t l = (uint32_t)L;
    for (int c = threadIdx.x; c < 256; c += 64 * TW) {
        const uint32_t cc = c;
        const uint32_t h = sem.simd_seeds ? seed_h_simd(cc) : seed_h_scalar(cc), r = sem.simd_seeds ? seed_rc_simd(cc) : seed_rc_scalar(cc);
        tab[c] = make_uint2(h, rotl32(r, l - 1));
        tab[256 + c] = make_uint2(rotl32(h, l), rotr32(r, 1));
    }
    __syncthreads(); // the only workgroup barrier; waves are independent from here on
    StreamLds &S = *reinterpret_cast<StreamLds *>(smem + SEED_TABLE_BYTES + (size_t)w * sizeof(StreamLds));
    const uint64_t n_waves = (uint64_t)gridDim.x * TW;
    uint64_t t = tile_begin + (uint64_t)blockIdx.x * TW + w;
    if (t >= n_tiles) return;
    if (__builtin_amdgcn_readfirstlane((int)counts->bad_input)) return; // malformed read table: touch nothing
    uint64_t stamp = 0;
    ph_ptr_t ph = nullptr;
    int lane = lane0;
    // software pipeline over the wave's tiles as in tile_minimizer_kernel: the read-table entries of the next tile and the draw for the one after it
    // are issued before the hash loop and looked at after it
    uint32_t r0 = tile_read0[t], r1 = tile_read0[t + 1];
    uint64_t bpos0, rs0;
    {
        const uint64_t bri = (uint64_t)r0 + 1 + lane;
        bpos0 = bri <= n_reads ? read_off[bri] : ~0ull;
        rs0 = read_off[r0];
    }
    uint64_t tn = t + n_waves, tnn = t + 2 * n_waves;
    const uint64_t dyn0 = tile_begin + 3 * n_waves;
    const uint32_t cur_g = (uint32_t)((blockIdx.x * TW + w + blockIdx.x / (TILE_CURSORS / 4)) % TILE_CURSORS); // (see tile_minimizer_kernel)
    unsigned int *const cursors = (unsigned int *)(pool_cursor + 16);
    uint32_t r0n = 0, r1n = 0;
    if (tn < n_tiles) {
        r0n = tile_read0[tn];
        r1n = tile_read0[tn + 1];
    }
    bool lb_dead = false;
    for (; t < n_tiles;) {
        lane = lane0;
        asm volatile("" : "+v"(lane)); // (nothing derived from the lane index is loop-invariant: see tile_minimizer_kernel)
        const uint64_t t0 = t * (uint64_t)TILE_BASES;
        const uint32_t cr0 = r0, cr1 = r1;
        uint32_t took = 0;
        if (lane == 0 && tnn < n_tiles) took = atomicAdd(&cursors[32 * cur_g], 1u);
        uint64_t bposn = ~0ull, rs0n = 0;
        uint32_t r0nn = 0, r1nn = 0;
        const uint64_t brin = (uint64_t)r0n + 1 + lane;
        uint32_t vzero = 0;
        asm volatile("" : "+v"(vzero));
        if (tn < n_tiles) {
            bposn = read_off[brin <= n_reads ? brin : n_reads];
            rs0n = read_off[(uint64_t)r0n + vzero];
            if (tnn < n_tiles) {
                r0nn = tile_read0[tnn + vzero];
                r1nn = tile_read0[tnn + 1 + vzero];
            }
        }
        uint32_t caps[NPC], raw[5];
#pragma unroll
        for (int g2 = 0; g2 < NPC; g2++) caps[g2] = 0;
#pragma unroll
        for (int g2 = 0; g2 < 5; g2++) raw[g2] = 0;
        const uint8_t *D = bases + t0; // global memory: the tile and its look-ahead lie inside the stream
        if (sem.enabled) hash_loop_static<L, REG_LA, false, true>(D, sem.bound_le, lane, TILE_T / 16, caps, raw);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the loads above had the whole hash loop to finish
        if (brin > n_reads) bposn = ~0ull;
        rs0n = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rs0n >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rs0n);
        r0nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0nn);
        r1nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1nn);
        uint64_t drawn = ~0ull;
        if (tnn < n_tiles) drawn = dyn0 + cur_g + (uint64_t)TILE_CURSORS * (uint32_t)__builtin_amdgcn_readfirstlane((int)took);
        int lane_d = lane;
        asm volatile("" : "+v"(lane_d));
        uint64_t base = 0;
        auto no_issue = [](uint32_t, uint64_t) {};
        (void)dense_phase<L, false, true>(no_issue, S, D, tab, read_off, n_reads, t, t0, (uint32_t)TILE_BASES, (uint32_t)TILE_BASES, 0u, (uint32_t)TILE_T, 7282u, l,
                                          cr0, cr1, bpos0, rs0, lane_d, rec, pool_cursor, (uint32_t *)nullptr, counts, base, sem, d_agg, d_meta, K1, caps, raw, 0u,
                                          lb_dead, ph, stamp);
        wave_sync(); // the wave's LDS is reused by the next tile
        r0 = r0n; r1 = r1n; bpos0 = bposn; rs0 = rs0n; r0n = r0nn; r1n = r1nn;
        t = tn;
        tn = tnn;
        if (tnn < n_tiles) tnn = drawn; // >= n_tiles: this wave's cursor has run dry
    }
}

template <int L>
hipError_t launch_stream_l(hipStream_t st, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles_full,
                           const uint32_t *tile_read0, Sem sem, Records rec, uint64_t *pool_cursor, Counts *counts, const Desc *desc, uint64_t tile_begin) {
    if constexpr (L > 0) {
        auto kern = stream_minimizer_kernel<L>;
        constexpr int lds = stream_lds_bytes();
        constexpr int MAX_DEV = 64;
        static int n_cu_d[MAX_DEV] = {0};
        static std::mutex cache_mu;
        int dev = 0;
        S2K_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(cache_mu);
        if (n_cu_d[dev] == 0) {
            S2K_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipDeviceProp_t prop;
            S2K_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
            n_cu_d[dev] = prop.multiProcessorCount;
        }
        uint64_t blocks = (n_tiles_full - tile_begin + STREAM_TW - 1) / STREAM_TW;
        const uint64_t resident = (uint64_t)n_cu_d[dev] * STREAM_BPC;
        if (blocks > resident) blocks = resident; // persistent: waves loop over the remaining tiles
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * STREAM_TW), lds, st, bases, read_off, n_reads, n_bases, n_tiles_full, tile_read0, sem, rec,
                           pool_cursor, counts, desc->agg, desc->meta, desc->k - 1u, tile_begin);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}
#endif // S2K_STREAM_BUILD

template <int L, bool HPC, bool DESC>
hipError_t launch_tiles_lh(hipStream_t st, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                           uint64_t n_tiles, const uint32_t *tile_read0, Sem sem, Records rec, uint64_t *pool_cursor,
                           uint64_t *tile_rec_off, uint32_t *tile_cnt, uint32_t *mn_cnt, Counts *counts, const Desc *desc, uint64_t tile_begin) {
    auto kern = tile_minimizer_kernel<L, HPC, DESC>;
    constexpr int TW = tw<HPC>();
    const int lds = block_lds_bytes<HPC>();
    // per instantiation AND per device: function attributes and occupancy belong to the device the module is loaded on
    // (contexts on different threads may launch concurrently: the cache is filled under a lock)
    constexpr int MAX_DEV = 64;
    static int n_cu_d[MAX_DEV] = {0}, per_cu_d[MAX_DEV] = {0};
    static std::mutex cache_mu;
    int dev = 0;
    S2K_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(cache_mu);
    if (n_cu_d[dev] == 0) {
        S2K_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipDeviceProp_t prop;
        S2K_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        int occ = 0;
        S2K_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), 64 * TW, lds));
        // The occupancy query ignores the granule LDS is allocated in (round 2: it answered 3 for 3 x 54 400 B of a 163 840 B
        // LDS, two blocks were resident and the third queued behind them).  Trust it only up to what fits with every block rounded
        // up to 2 KiB -- with one 12-wave block per CU that is 1.
        const int fit = (int)((size_t)prop.sharedMemPerMultiprocessor / (((size_t)lds + 2047) & ~(size_t)2047));
        if (fit >= 1 && occ > fit) occ = fit;
        per_cu_d[dev] = occ < 1 ? 1 : occ;
#ifdef S2K_DEBUG_KNOBS // `make KNOBS=1` / `make PROFILE=1` builds only (tools/*.sh): occupancy experiments
        if (const char *e = getenv("S2K_DEBUG_BLOCKS_PER_CU")) per_cu_d[dev] = atoi(e) > 0 ? atoi(e) : per_cu_d[dev];
#endif
        n_cu_d[dev] = prop.multiProcessorCount;
    }
    const int n_cu = n_cu_d[dev], per_cu = per_cu_d[dev];
    uint64_t blocks = (n_tiles - tile_begin + TW - 1) / TW;
    const uint64_t resident = (uint64_t)n_cu * per_cu;
    if (blocks > resident) blocks = resident; // persistent: waves loop over the remaining tiles
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * TW), lds, st, bases, read_off, n_reads, n_bases, n_tiles,
                       tile_read0, sem, rec, pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts, desc ? desc->agg : nullptr,
                       desc ? desc->meta : nullptr, desc ? desc->k - 1u : 0u, tile_begin);
    return hipGetLastError();
}

template <int L>
hipError_t launch_tiles_l(bool hpc, hipStream_t st, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                          uint64_t n_bases, uint64_t n_tiles, const uint32_t *tile_read0, Sem sem, Records rec,
                          uint64_t *pool_cursor, uint64_t *tile_rec_off, uint32_t *tile_cnt, uint32_t *mn_cnt,
                          Counts *counts, const Desc *desc, uint64_t tile_begin) {
#define S2K_GO(H, F)                                                                                                       \
    launch_tiles_lh<L, H, F>(st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec, pool_cursor, tile_rec_off, \
                             tile_cnt, mn_cnt, counts, desc, tile_begin)
#if S2K_STREAM_BUILD
    if constexpr (L > 0) {
        // Regular family, descriptor path: the tiles that lie wholly inside the stream (all but the last one or two) go to the kernel without a
        // tile buffer; S2K_STREAM_KERNEL=0 keeps the tiled kernel for everything (A/B runs)
        static const bool use_stream = S2K_STREAM_DEFAULT ? !(getenv("S2K_STREAM_KERNEL") && atoi(getenv("S2K_STREAM_KERNEL")) == 0)
                                                          : (getenv("S2K_STREAM_KERNEL") && atoi(getenv("S2K_STREAM_KERNEL")) != 0);
        if (desc && !hpc && use_stream) {
            const uint64_t n_full = n_bases >= 128 ? (n_bases - 128) / (uint64_t)TILE_BASES : 0; // tiles t with (t + 1) TILE_BASES + 128 <= n_bases
            const uint64_t hi = n_full < n_tiles ? n_full : n_tiles;
            if (hi > tile_begin) {
                S2K_HIP_CHECK(launch_stream_l<L>(st, bases, read_off, n_reads, n_bases, hi, tile_read0, sem, rec, pool_cursor, counts, desc, tile_begin));
                if (hi >= n_tiles) return hipSuccess;
                const uint64_t tb = hi; // (the one or two tiles at the end of the stream: no draws from the cursors, which the launch above has used)
                return launch_tiles_lh<L, false, true>(st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec, pool_cursor, tile_rec_off, tile_cnt,
                                                       mn_cnt, counts, desc, tb);
            }
        }
    }
#endif
    if (desc) return hpc ? S2K_GO(true, true) : S2K_GO(false, true);
    return hpc ? S2K_GO(true, false) : S2K_GO(false, false);
#undef S2K_GO
}

} // namespace
} // namespace s2k
