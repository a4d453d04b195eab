#!/bin/bash
# Hpc kernel at 168 VGPRs with blocks of 2 waves: 8 / 10 / 12 waves per CU (KNOBS build, S2K_DEBUG_BLOCKS_PER_CU)
cd $GRAFT_REPO_ROOT
H=rust-seq2kminmers_amd/csrc/s2k_tile_impl.h
cp $H /tmp/impl.orig.$$
trap 'cp /tmp/impl.orig.$$ $H; rm -f /tmp/impl.orig.$$' EXIT  # the edit below never outlives the script
sed -i 's/__launch_bounds__(64 \* TW, HPC ? 2 : S2K_WAVES_PER_SIMD)/__launch_bounds__(64 * TW, 3)/; s/constexpr int TW = 4; /constexpr int TW = 2; /' $H
rm -f rust-seq2kminmers_amd/csrc/*.o
make -s -C rust-seq2kminmers_amd/csrc KNOBS=1 -j16 libs2k.so > /tmp/make.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/make.log; exit 1; }
for b in 3 4 5 6; do
  S2K_DEBUG_BLOCKS_PER_CU=$b python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocks/CU $b x 2 waves: hpc tile %.3f ms, reg tile %.3f ms, verified=%s' % (d['roofline']['kernel_ms'], d['other_mode']['kernel_ms'], d['verified_vs_oracle']))"
done
