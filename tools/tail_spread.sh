#!/bin/bash
# how long after the first wave does the last wave of the tiled kernel run out of tiles? (KNOBS build, S2K_DEBUG_SKIP=32)
cd $GRAFT_REPO_ROOT
rm -f rust-seq2kminmers_amd/csrc/*.o && make -s -C rust-seq2kminmers_amd/csrc KNOBS=1 -j16 libs2k.so > /dev/null 2>&1 || exit 1
S2K_DEBUG_WAVE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/wave_dump.txt S2K_DEBUG_SKIP=32 python bench.py --contexts 1 --steps 1 --warmup 0 --mode hpc --no-other-mode --no-cpu-baseline --verify-reads 0 2>&1 | grep "s2k dbg"
