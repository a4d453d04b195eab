// s2k_tile.hip -- host side of the tiled minimizer kernel: dispatch on l.  The kernel itself is in
// s2k_tile_impl.h; this unit carries its run-time-l instantiation, s2k_tile_inst.hip the compile-time ones.
#include "s2k_tile_impl.h"

namespace s2k {

#define S2K_DECL(LV)                                                                                                     \
    hipError_t launch_tiles_static_##LV(bool hpc, hipStream_t st, const uint8_t *bases, const uint64_t *read_off,         \
                                        uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles, const uint32_t *tile_read0, \
                                        Sem sem, Records rec, uint64_t *pool_cursor, uint64_t *tile_rec_off,             \
                                        uint32_t *tile_cnt, uint32_t *mn_cnt, Counts *counts, const Desc *desc, uint64_t tile_begin);
S2K_STATIC_LS(S2K_DECL)
#undef S2K_DECL


hipError_t launch_tile_minimizers(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                                  uint64_t n_tiles, const uint32_t *tile_read0, Sem sem, Records rec,
                                  uint64_t *pool_cursor, uint64_t *tile_rec_off, uint32_t *tile_cnt, uint32_t *mn_cnt,
                                  Counts *counts, const Desc *desc, uint64_t tile_begin, hipStream_t st) {
    if (n_tiles <= tile_begin || n_reads == 0) return hipSuccess;
    if (sem.l > (uint32_t)MAX_L_TILED || (sem.hpc && sem.tail_quirk && !sem.read_runs && !sem.tile_heads)) return hipErrorInvalidValue;
    if (desc && (desc->k == 0 || desc->k > 32u || !desc->agg || !desc->meta)) return hipErrorInvalidValue;
    switch (sem.l) {
#define S2K_CASE(LV)                                                                                                   \
    case LV:                                                                                                           \
        return launch_tiles_static_##LV(sem.hpc, st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec, \
                                        pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts, desc, tile_begin);
        S2K_STATIC_LS(S2K_CASE)
#undef S2K_CASE
    default:
        return launch_tiles_l<0>(sem.hpc, st, bases, read_off, n_reads, n_bases, n_tiles, tile_read0, sem, rec,
                                 pool_cursor, tile_rec_off, tile_cnt, mn_cnt, counts, desc, tile_begin);
    }
}

} // namespace s2k
