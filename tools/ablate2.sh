#!/bin/bash
for skip in 8 24 40 72 120; do
  S2K_DEBUG_SKIP=$skip timeout -k 10 200 python bench.py --no-cpu-baseline --verify-reads 0 --reads 500000 --steps 1 --warmup 0 2>&1 | grep dbg | sed -n '2p' | awk -v s=$skip '{printf "skip=%d staging=%d compact=%d hash=%d bound=%d list=%d rounds=%d tail=%d\n", s, $8/542535, $9/542535, $10/542535, $11/542535, $12/542535, $13/542535, $14/542535}'
done
