#!/bin/bash
# per-tile cycle breakdown of the tiled kernel (diagnostic stamps; slows the kernel a little).
# Run on the GPU box: rebuilds the whole library with the stamps compiled in (KNOBS / PROFILE builds change an internal struct: never
# mix their objects with plain ones; the box's copy of the tree is scratch).
rm -f rust-seq2kminmers_amd/csrc/*.o && make -s -C rust-seq2kminmers_amd/csrc PROFILE=1 -j16 libs2k.so > /dev/null 2>&1 || exit 1
S2K_DEBUG_SKIP=${1:-8} timeout -k 10 200 python bench.py --contexts 1 --no-cpu-baseline --no-end-to-end --verify-reads 0 --reads 500000 --steps 1 --warmup 0 2>&1 | grep dbg | sed -n '1p;2p' | awk '{printf "%s staging=%d compact=%d hash=%d bound=%d list=%d rounds=%d tail=%d jobs/tile=%.1f | r.list=%d r.backmap=%d r.stores=%d r.counts=%d rederive=%d c.marks=%d c.flags=%d c.stores=%d\n", (NR==1?"HPC":"REG"), $8/542535, $9/542535, $10/542535, $11/542535, $12/542535, $13/542535, $14/542535, $15/542535, $16/542535, $17/542535, $18/542535, $19/542535, $20/542535, $21/542535, $22/542535, $23/542535}'
