// s2k_tile_inst.hip -- one compile-time-l instantiation of the tiled minimizer kernel per object file
// (hipcc -DS2K_TILE_L=<l>; the list is S2K_STATIC_LS in s2k_static_l.h / STATIC_LS in the Makefile).
#ifndef S2K_TILE_L
#error "compile with -DS2K_TILE_L=<l>"
#endif
#include "s2k_tile_impl.h"

namespace s2k {

#define S2K_NAME2(LV) launch_tiles_static_##LV
#define S2K_NAME(LV) S2K_NAME2(LV)
hipError_t S2K_NAME(S2K_TILE_L)(bool hpc, hipStream_t st, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                                uint64_t n_bases, uint64_t n_tiles, const uint32_t *tile_read0, Sem sem, Records rec,
                                uint64_t *pool_cursor, uint64_t *tile_rec_off, uint32_t *tile_cnt, uint32_t *mn_cnt,
                                Counts *counts, const Desc *desc, uint64_t tile_begin) {
    static_assert(S2K_TILE_L >= 1 && S2K_TILE_L <= 32, "the unrolled loop keeps one address register per base of the l-mer");
    return launch_tiles_l<S2K_TILE_L>(hpc, st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec, pool_cursor,
                                      tile_rec_off, tile_cnt, mn_cnt, counts, desc, tile_begin);
}

} // namespace s2k
