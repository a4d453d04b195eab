#!/bin/bash
# per-tile cycle breakdown (PROFILE-build stamps) of prebuilt variant libraries: tools/ab/phases2.sh "<bench args>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
args="$1"; shift
for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f S2K_DESC_CHUNKS=1 S2K_DEBUG_SKIP=8 timeout -k 10 200 python bench.py --contexts 1 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 --reads 500000 --steps 1 --warmup 0 $args 2>&1 | grep dbg | sed -n '1p' | awk -v n=$f '{printf "%s staging=%d compact=%d hash=%d bound=%d list=%d rounds=%d tail=%d jobs/tile=%.1f | r.list=%d r.backmap=%d r.stores=%d r.counts=%d rederive=%d c.marks=%d c.flags=%d c.stores=%d\n", n, $8/542535, $9/542535, $10/542535, $11/542535, $12/542535, $13/542535, $14/542535, $15/542535, $16/542535, $17/542535, $18/542535, $19/542535, $20/542535, $21/542535, $22/542535, $23/542535}'
done
